#!/bin/bash
# Kernel-level evidence of one build, on the GPU box:  bash tools/runs/profile.sh <tag>   (e.g. r03_w)
#   1. rocprofv3 --kernel-trace --stats of the bench command          -> gpurun_out/<tag>_kernel_stats.csv (+ _top.txt)
#   2. two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs)         -> gpurun_out/<tag>_pmc_traffic.json
#   3. one SQ counter pass (every kernel alone on the GPU)            -> gpurun_out/<tag>_sq_counters.json
# Counter passes carry --kernel-trace only; the program itself follows `--`.
tag=${1:-prof}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
B="bench.py --no-cpu-baseline --no-extra-modes"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_st -- python3 $B --steps 10 --warmup 3 > gpurun_out/${tag}_prof_bench.json 2> gpurun_out/${tag}_prof_bench.log || exit 1
f=$(find gpurun_out/${tag}_st -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/${tag}_kernel_stats.csv; cut -c1-200 "$f" | head -45 > gpurun_out/${tag}_kernel_stats_top.txt; fi
rm -rf gpurun_out/${tag}_st
echo "[profile] stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pf -- python3 $B --steps 1 --warmup 0 --serial-decode > /dev/null 2> gpurun_out/${tag}_pmc_f.log || exit 1
echo "[profile] fetch pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pw -- python3 $B --steps 1 --warmup 0 --serial-decode > /dev/null 2> gpurun_out/${tag}_pmc_w.log || exit 1
python3 tools/pmc_traffic.py gpurun_out/${tag}_pf gpurun_out/${tag}_pw gpurun_out/${tag}_pmc_traffic.json
rm -rf gpurun_out/${tag}_pf gpurun_out/${tag}_pw
echo "[profile] write pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d gpurun_out/${tag}_sq -- python3 $B --steps 1 --warmup 0 --serial-decode --no-run-ahead > /dev/null 2> gpurun_out/${tag}_sq.log || exit 1
python3 tools/sq_counters.py gpurun_out/${tag}_sq gpurun_out/${tag}_sq_counters.json
rm -rf gpurun_out/${tag}_sq
echo "[profile] sq pass done"
