#!/bin/bash
# EXPERIMENT: what would dropping the low plane of P in the attention kernels (2 PV products instead of 3, one conversion pass less) buy and cost?
# Builds the library with -DSDVAR_ATTN_P1, times the verify-attention shapes, runs the parity tests that look at logits, then restores the product build.
cd "$GRAFT_REPO_ROOT/sdvar_amd/csrc" || exit 1
make -B attention_f16x2.o EXTRA=-DSDVAR_ATTN_P1 > /dev/null 2>&1 && make > /dev/null 2>&1 || exit 1
cd ../..
echo "== timing (P1 build)"
for shape in "16 16 256 424 3" "16 16 169 255 3" "16 12 256 424 3"; do timeout -k 10 60 python tools/one_attention.py $shape 200 2>&1 | grep "us/launch"; done
echo "== parity tests (P1 build)"
timeout -k 10 500 python -m pytest tests/test_gpu_fullwidth_oracle.py tests/test_gpu_e2e.py tests/test_gpu_ops.py -m gpu -q -k "P1_spec or wide_model or plain_ar_vs_reference or d16_b1 or attention" 2>&1 | tail -15
echo "== logit error (P1 build)"
for n in ar_d6_256_stress ar_d16_256_stress_B1 ar_d4_512_stress; do timeout -k 10 200 python tools/micro/logit_err.py $n 2>&1 | tail -1; done
echo "== bench (P1 build)"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-modes 2>&1 >/dev/null | grep -E "timed region|no-decode|profiled"
cd sdvar_amd/csrc && make -B attention_f16x2.o > /dev/null 2>&1 && make > /dev/null 2>&1; cd ../..
echo "== logit error (product build)"
for n in ar_d6_256_stress ar_d16_256_stress_B1 ar_d4_512_stress; do timeout -k 10 200 python tools/micro/logit_err.py $n 2>&1 | tail -1; done
