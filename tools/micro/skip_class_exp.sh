# needs a timing-experiment build of the library: make -C sdvar_amd/csrc clean all EXTRA=-DSDVAR_TIMING_EXPERIMENTS (the product build has no such switches)
# marginal cost of each kernel class inside the real stage_forward launch sequence (no profiler): stage walls with the class not launched (results wrong)
# bits: 1 ln_modulate, 2 qk_norm_append, 4 attention, 8 fc1, 16 QKV GEMM, 32 proj, 64 fc2
for sk in 0 1 2 4 8 16 32 64; do SDVAR_SKIP_CLASS=$sk python tools/stage_profile.py --depth ${1:-16} 2>&1 | python3 -c "
import re,sys
w=[float(m.group(1)) for m in re.finditer(r'wall\s+([0-9.]+)', sys.stdin.read())]
print('skip=%3d' % $sk, ' '.join('%6.3f' % v for v in w), ' total %.2f' % sum(w))"; done
