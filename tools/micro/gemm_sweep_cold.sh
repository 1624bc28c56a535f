# needs a timing-experiment build of the library: make -C sdvar_amd/csrc clean all EXTRA=-DSDVAR_TIMING_EXPERIMENTS (the product build has no such switches)
# the sweep the f16x2 cost model is fitted to: HBM-cold weights, d12 and d16, single stages and gamma = 2 chunks; the launches whose K-slice sum the
# consumer takes over (qkv, proj, fc2) are timed as the slab launch alone (SDVAR_GEMM_DBG=8), fc1 with its reduce launch
out=${1:-gpurun_out/sweep_cold}
rm -f ${out}_slab.jsonl ${out}_full.jsonl
for d in 16 12; do for c in 1 2; do
  SDVAR_GEMM_DBG=8 python tools/gemm_bench.py --mode f16x2 --depth $d --chunk $c --sweep --cold --dump ${out}_slab.jsonl 2>&1 | grep "one pass"
  python tools/gemm_bench.py --mode f16x2 --depth $d --chunk $c --sweep --cold --dump ${out}_full.jsonl 2>&1 | grep "one pass"
done; done
