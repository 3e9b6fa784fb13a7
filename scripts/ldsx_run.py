import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "wireframe-3d-prediction_amd"))
from wf3d import ops, telemetry
dev = torch.device("cuda:0"); hw = telemetry.hwmon_dir(0)
M = 131072
for K, N in ((1024, 2048), (2048, 1024)):
    X, W = ops.split_rows(torch.randn(M, K, device=dev)), ops.split_rows(torch.randn(N, K, device=dev))
    out = torch.empty(M, N, device=dev)
    t, w, g = telemetry.run_sampled(lambda: ops.gemm_split(X, W, out=out), hw, 3.0)
    print(f"{os.environ.get('WF3D_LIB', 'default'):50s} {K}->{N}: {t * 1e6:8.1f} us  {w:6.0f} W  {g:5.2f} GHz", flush=True)
