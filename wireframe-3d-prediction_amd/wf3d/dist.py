"""One-process-per-GPU data parallelism for the wireframe path.

The path shards perfectly over clouds (LayerNorm, per-cloud pools, per-sample
attention: nothing crosses samples — SURVEY.md §8e), so the only exchange is the
sum of parameter gradients once per step.  `torch.distributed` backend "nccl"
is RCCL on ROCm; xGMI is point-to-point, so a ring all-reduce is bound by one
link (~153 GB/s): the 124 MB fp32 gradient set is cut into a few large buckets
in BACKWARD order (edge head -> vertex head -> encoder) and each bucket's
all-reduce is launched as soon as its last gradient has been accumulated, which
puts the edge+vertex buckets (63 % of the bytes) under the encoder backward
(~90 % of the step's compute).  The reference has no distributed code at all.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(device_type=None):
    """Initialise the default process group from torchrun's env (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR/PORT).  Returns (rank, world, device)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = (device_type or ("cuda" if torch.cuda.is_available() else "cpu")) == "cuda"
    if use_cuda:
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # "nccl" is RCCL on ROCm.  WF3D_DIST_BACKEND=gloo lets several ranks share one GPU (tests).
        backend = os.environ.get("WF3D_DIST_BACKEND", "nccl" if use_cuda else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def shard_batch(global_batch, rank, world):
    """Contiguous slice [lo, hi) of the global sample indices owned by `rank`."""
    per = global_batch // world
    if per * world != global_batch:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    return rank * per, (rank + 1) * per


@torch.no_grad()
def sync_parameters(module, src=0, group=None):
    """Broadcast every parameter/buffer from `src` (after lazily created layers —
    VertexPredictor.point_pool_proj, SURVEY.md §9 Q1 — exist on every rank)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


def _stage_of(name):
    if name.startswith("edge_predictor."):
        return 0
    if name.startswith("vertex_predictor."):
        return 1
    if name.startswith("encoder.feature_fusion."):
        return 2
    return 3


class GradReducer:
    """Bucketed, overlapped gradient averaging.

    usage:   red = GradReducer(model);  loss.backward();  red.finish()      (one backward per finish)
    Buckets follow backward order; a bucket's all-reduce starts from a
    post-accumulate-grad hook when the last of the parameters that are EXPECTED to receive a
    gradient has received its own, on the communication stream of the process group (async_op),
    and `finish()` waits and scatters the averaged values back into `.grad`.

    Expected = every parameter that has ever received a gradient (learned; parameters that never do —
    EdgePredictor.spatial_proj, SURVEY §9 Q2 — must not hold a bucket back).  A gradient that lands
    AFTER its bucket was launched (a parameter receiving its first gradient, or a second backward before
    finish()) marks the bucket stale: finish() then waits for the in-flight reduce and reduces that
    bucket again from the final .grad values, so nothing stale is ever written back.
    `exposed_ms()` = time the compute stream spent inside finish() (un-overlapped communication)."""

    def __init__(self, module, bucket_mb=48.0, group=None, average=True):
        self.module, self.group, self.average = module, group, average
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self._buckets = None
        self._hooks = []
        self._ever = set()               # parameters that have received a gradient at least once
        self._events = []
        self._build()

    def _build(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        named = [(n, p) for n, p in self.module.named_parameters() if p.requires_grad]
        order = sorted(range(len(named)), key=lambda i: (_stage_of(named[i][0]), -i))
        buckets, cur, cur_bytes, cur_stage = [], [], 0, None
        for i in order:
            n, p = named[i]
            st = _stage_of(n)
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > self.bucket_bytes or st != cur_stage):
                buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
            cur_stage = st
        if cur:
            buckets.append(cur)
        self._buckets = []
        self._owner = {}
        for bi, ps in enumerate(buckets):
            flat = torch.zeros(sum(p.numel() for p in ps), dtype=ps[0].dtype, device=ps[0].device)
            self._buckets.append({"params": ps, "flat": flat, "got": set(), "work": None, "launched": False,
                                  "stale": False, "views": None})
            for p in ps:
                self._owner[p] = bi
        self._n_params = len(named)
        if self.world > 1:
            for _, p in named:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _expected(self, b):
        exp = [p for p in b["params"] if p in self._ever]
        return exp if exp else b["params"]          # first step: nothing learned yet -> wait for everything

    def _on_grad(self, p):
        bi = self._owner.get(p)
        if bi is None:
            return
        b = self._buckets[bi]
        if b["launched"] or p in b["got"]:
            b["stale"] = True                       # a gradient after the launch / a second backward: redo in finish()
        b["got"].add(p)
        if not b["launched"] and all(q in b["got"] for q in self._expected(b)):
            self._launch(b)

    def _launch(self, b):
        flat, off = b["flat"], 0
        views = []
        for p in b["params"]:
            n = p.numel()
            views.append(flat[off:off + n].view_as(p))
            off += n
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b["params"]]
        torch._foreach_copy_(views, grads)
        b["views"] = views
        b["work"] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        b["launched"] = True

    def finish(self):
        """Wait for all buckets (launching any whose hooks did not all fire, e.g.
        parameters without a gradient this step) and write averages into .grad."""
        if self.world == 1:
            return
        timed = torch.cuda.is_available() and self._buckets and self._buckets[0]["flat"].is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        n = sum(1 for p in self.module.parameters() if p.requires_grad)
        if n != self._n_params:                     # a lazy parameter appeared: drain, re-bucket, reduce everything now
            for b in self._buckets:
                if b["work"] is not None:
                    b["work"].wait()
            self._build()
        for b in self._buckets:
            if b["launched"] and b["stale"]:
                b["work"].wait()                    # the early reduce carried stale values: do it again
                b["launched"] = False
            if not b["launched"]:
                self._launch(b)
        scale = 1.0 / self.world if self.average else 1.0
        for b in self._buckets:
            b["work"].wait()
            if scale != 1.0:
                b["flat"].mul_(scale)
            dst = [p.grad for p in b["params"] if p.grad is not None]
            src = [v for p, v in zip(b["params"], b["views"]) if p.grad is not None]
            if dst:
                torch._foreach_copy_(dst, src)
            self._ever.update(p for p in b["params"] if p.grad is not None)
            b["got"], b["work"], b["launched"], b["stale"] = set(), None, False, False
        if timed:
            e1.record()
            self._events.append((e0, e1))

    def exposed_ms(self, reset=True):
        """Per-step times (ms) the compute stream spent in finish(): all-reduce not hidden under backward, plus the
        scale / copy-back kernels.  Synchronises."""
        if not self._events:
            return []
        torch.cuda.synchronize()
        out = [a.elapsed_time(b) for a, b in self._events]
        if reset:
            self._events = []
        return out

    def bucket_summary(self):
        return [(len(b["params"]), b["flat"].numel() * b["flat"].element_size()) for b in self._buckets]
