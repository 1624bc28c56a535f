set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01h_stats --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extra-modes --serial-decode --no-run-ahead > $R/gpurun_out/r01h_stats.log 2>&1
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r01h_pmc_f --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-modes --serial-decode --no-run-ahead > $R/gpurun_out/r01h_pmc_f.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r01h_pmc_w --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-modes --serial-decode --no-run-ahead > $R/gpurun_out/r01h_pmc_w.log 2>&1
echo write done
cd $R && python tools/pmc_traffic.py gpurun_out/r01h_pmc_f gpurun_out/r01h_pmc_w gpurun_out/r01h_pmc_traffic.json > /dev/null
ls gpurun_out/r01h_stats/*/
