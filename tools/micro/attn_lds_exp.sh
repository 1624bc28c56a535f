#!/bin/bash
# Where does a tile of attention_f16x2_pp_kernel spend its time?  Builds the library with -DSDVAR_ATT_EXP=<bits> (results WRONG: bit 0 no V fragment reads after tile 0,
# bit 1 no K fragment reads, bit 2 no softmax arithmetic, bit 3 no MFMAs) next to the product library and times the l = 256 / K = 680 launch with each.
#   on this container:  bash tools/micro/attn_lds_exp.sh build      (hipcc cross-compiles)        on the GPU box:  bash tools/micro/attn_lds_exp.sh run
cd "$(dirname "$0")/../.." || exit 1
C=sdvar_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function"
if [ "$1" = build ]; then
  for e in 1 2 3 4 8 12 7; do
    /opt/rocm/bin/hipcc $FL -DSDVAR_ATT_EXP=$e -c $C/attention_f16x2.hip -o /tmp/att_exp_$e.o || exit 1
    objs=$(ls $C/*.o | grep -v attention_f16x2.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/att_exp_$e.o -o tools/micro/libsdvar_att_exp_$e.so || exit 1
  done
  exit 0
fi
for e in 0 1 2 3 4 8 12 7; do
  lib=$C/libsdvar_hip.so; [ $e != 0 ] && lib=tools/micro/libsdvar_att_exp_$e.so
  SDVAR_LIB=$lib python3 - <<PY
import os, sys
sys.path.insert(0, os.getcwd())
from sdvar_amd import engine as E
E.load_library(os.environ["SDVAR_LIB"])
sys.argv = ["one_attention.py", "16", "16", "256", "424", "3", "200"]
import io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    exec(open("tools/one_attention.py").read())
print("exp bits $e:", buf.getvalue().strip().split(":", 1)[1].strip())
PY
done
