#!/bin/bash
# Four rocprofv3 passes over `bench.py` on the GPU box (run through gpurun): kernel trace + stats, FETCH_SIZE, WRITE_SIZE
# and the SQ / GRBM counters, each in its own run (counter passes carry no trace flags; program directly after `--`).
#   gpurun --timeout 900 -- 'bash scripts/profile_round.sh r02 cfg2'
# then here:  python scripts/summarize_profiles.py r02        (or: r02_cfg5 cfg5)
set -e
TAG=${1:?tag}; CFG=${2:-cfg2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ "$CFG" = cfg2 ] && T=$TAG || T=${TAG}_${CFG}
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --config $CFG --no-cpu-baseline --no-power-probe"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_trace2 -- $B --steps 3 --warmup 1 > $R/gpurun_out/prof_${T}_trace2.log 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${T}_fetch -- $B --steps 1 --warmup 1 > $R/gpurun_out/prof_${T}_fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${T}_write -- $B --steps 1 --warmup 1 > $R/gpurun_out/prof_${T}_write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_${T}_sq -- $B --steps 1 --warmup 1 > $R/gpurun_out/prof_${T}_sq.log 2>&1
echo "sq done"
tail -1 $R/gpurun_out/prof_${T}_trace2.log | cut -c1-300
