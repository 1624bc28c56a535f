# Round profile (run on the GPU box through gpurun): kernel stats, HBM traffic (two PMC passes) and SQ counters of the headline bench command with every
# kernel alone on the GPU.  Usage: bash tools/profile_round.sh r02_d     -> gpurun_out/<tag>_*; copy the summaries into profiles/.
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
ARGS="--no-cpu-baseline --no-extra-modes --serial-decode --no-run-ahead"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 $ARGS > $R/gpurun_out/${TAG}_stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pmc_f --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $ARGS > $R/gpurun_out/${TAG}_pmc_f.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_pmc_w --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $ARGS > $R/gpurun_out/${TAG}_pmc_w.log 2>&1
echo write done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_sq --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 $ARGS > $R/gpurun_out/${TAG}_sq.log 2>&1
echo sq done
cd $R && python tools/pmc_traffic.py gpurun_out/${TAG}_pmc_f gpurun_out/${TAG}_pmc_w gpurun_out/${TAG}_pmc_traffic.json > /dev/null
cd $R/tools && python sq_counters.py ../gpurun_out/${TAG}_sq ../gpurun_out/${TAG}_sq_counters.json > ../gpurun_out/${TAG}_sq_summary.txt
cd $R && f=$(ls gpurun_out/${TAG}_stats/*/*kernel_stats.csv | head -1) && cp $f gpurun_out/${TAG}_kernel_stats.csv && head -25 $f > gpurun_out/${TAG}_kernel_stats_top.txt
ls gpurun_out/${TAG}_*
