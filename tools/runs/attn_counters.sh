#!/bin/bash
# Counter evidence for the verify-attention kernel: how the clock the chip holds, the matrix-pipe occupancy and the vector instruction count of ONE
# launch shape move with the number of busy CUs.   bash tools/runs/attn_counters.sh <tag>      -> gpurun_out/<tag>_attn_counters.json
#   shape: l = 256 queries over K = 680 keys per (row, head) (stage 9 of the 256^2 ladder), f16x2 planes; (row, head) pairs = workgroups = 64 / 128 / 192 / 256
#   pass A: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
#   pass B: SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES    (own run; if a name is unknown to this rocprofv3 the pass is skipped)
# Counter passes carry --kernel-trace only; the program itself follows `--`.
tag=${1:-attn}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for pairs in 64 128 192 256; do
  R=$((pairs / 16))
  python3 tools/one_attention.py $R 16 256 424 3 300 > gpurun_out/${tag}_time_$pairs.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
      --output-format csv -d gpurun_out/${tag}_A_$pairs -- python3 tools/one_attention.py $R 16 256 424 3 100 > /dev/null 2> gpurun_out/${tag}_A_$pairs.log
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES \
      --output-format csv -d gpurun_out/${tag}_B_$pairs -- python3 tools/one_attention.py $R 16 256 424 3 100 > /dev/null 2> gpurun_out/${tag}_B_$pairs.log
  echo "[attn_counters] $pairs pairs done"
done
# config P4's launch (one fp16 plane per operand): R = 16, H = 30, l = 1024, K = 2240
python3 tools/one_attention.py 16 30 1024 1216 4 40 > gpurun_out/${tag}_time_P4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
    --output-format csv -d gpurun_out/${tag}_A_P4 -- python3 tools/one_attention.py 16 30 1024 1216 4 20 > /dev/null 2> gpurun_out/${tag}_A_P4.log
python3 tools/attn_counters.py gpurun_out/${tag} gpurun_out/${tag}_attn_counters.json
for d in gpurun_out/${tag}_A_* gpurun_out/${tag}_B_*; do [ -d "$d" ] && rm -rf "$d"; done
