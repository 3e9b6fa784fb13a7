for st in 0 1 2 3 5 -2; do echo "== WF3D_STAGGER=$st"; WF3D_STAGGER=$st ONLY="SPLIT NT" ROUNDS=7 python scripts/bench_gemm.py 2>&1 | grep SPLIT; done
