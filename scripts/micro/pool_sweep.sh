for w in 1024 2048 4096 8192 16384; do echo "WGS=$w"; WF3D_POOL_WGS=$w python bench.py --no-cpu-baseline --steps 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['roofline']['hbm']['pool']['ops']['pool4_fwd'])"; done
