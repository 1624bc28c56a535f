#!/usr/bin/env python3
"""Phase split of the bf16x3 verify-attention loop from in-kernel s_memtime stamps.  Needs a diagnostic build:
    make -C sdvar_amd/csrc clean && make -C sdvar_amd/csrc EXTRA=-DSDVAR_ATT_STAMPS
python tools/micro/att_stamps.py R H l prefix"""
import ctypes as C, math, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdvar_amd import engine as E
R, H, l, prefix = (int(v) for v in sys.argv[1:5])
lib = E.load_library(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
Ktot = prefix + l; Lp = (Ktot + 63) // 64 * 64
sm = torch.full((H,), math.log(4.0), device=dev)
kc = torch.zeros(R, H, 3, Lp, 64, device=dev, dtype=torch.int16); vc = torch.zeros(R, H, 3, 64, Lp, device=dev, dtype=torch.int16)
for n, pos0 in ((prefix, 0), (l, prefix)):
    if n:
        qkv = torch.randn(R * n, 3 * 64 * H, device=dev); qo = torch.zeros(R, H, n, 64, device=dev)
        E._check(lib.sdvar_op_qk_norm_append(P(qkv), P(sm), P(qo), P(kc), P(vc), 2, R, n, H, Lp, pos0, st))
out = torch.empty(R, l, H * 64, device=dev)
qb, vs = (C.c_int32 * 1)(0), (C.c_int32 * 1)(Ktot)
for _ in range(3):
    E._check(lib.sdvar_op_attention(P(qo), P(kc), P(vc), 2, P(out), None, 0, 3, R, H, l, Lp, Ktot, 1, qb, vs, st))
torch.cuda.synchronize()
nw = min(4096, R * H * ((l + 127) // 128) * 4)
buf = np.zeros((nw, 8), dtype=np.uint64)
assert lib.sdvar_debug_att_stamps(buf.ctypes.data_as(C.c_void_p), nw) == 0
ntiles = (Ktot + 31) // 32
names = ["wait+barrier+issue", "QK (12 reads, 24 MFMA)", "mask/softmax/split", "PV (24 MFMA)", "loop gap"]
act = buf[buf[:, 1] > 0]
print(f"{len(act)} waves, {ntiles} tiles; s_memtime ticks per tile (mean over waves):")
for i, n in enumerate(names):
    print(f"  {n:28s} {act[:, i].mean() / ntiles:9.1f}")
print(f"  total {act[:, :5].sum(1).mean() / ntiles:9.1f} per tile, {act[:, :5].sum(1).mean():9.0f} per wave")
