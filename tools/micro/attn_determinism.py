#!/usr/bin/env python3
"""Is the verify-attention launch reproducible bit for bit?  Repeats one launch and compares every output with the first (fp32 output, then the planes output).
python tools/micro/attn_determinism.py R H l prefix fmt reps"""
import ctypes as C, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdvar_amd import engine as E
R, H, l, prefix, fmt, reps = (int(v) for v in sys.argv[1:7])
lib = E.load_library(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
Ktot = prefix + l; Lp = (Ktot + 5 + 63) // 64 * 64
sm = torch.full((H,), math.log(4.0), device=dev)
NP = {3: 2, 4: 1}[fmt]
kc = torch.zeros(R, H, NP, Lp, 64, device=dev, dtype=torch.int16); vc = torch.zeros(R, H, NP, 64, Lp, device=dev, dtype=torch.int16)
torch.manual_seed(1)
for n, pos0 in ((prefix, 0), (l, prefix)):
    if n:
        qkv = torch.randn(R * n, 3 * 64 * H, device=dev); qo = torch.zeros(R, H, n, 64, device=dev)
        E._check(lib.sdvar_op_qk_norm_append(P(qkv), P(sm), P(qo), P(kc), P(vc), fmt, R, n, H, Lp, pos0, st))
qb, vs = (C.c_int32 * 1)(0), (C.c_int32 * 1)(Ktot)
M = R * l
first = None; firstp = None; bad = badp = 0
for i in range(reps):
    out = torch.full((R, l, H * 64), float("nan"), device=dev)
    E._check(lib.sdvar_op_attention(P(qo), P(kc), P(vc), fmt, P(out), None, 0, 2, R, H, l, Lp, Ktot, 1, qb, vs, st))
    outp = torch.zeros(2, H * 64 // 32, M, 32, device=dev, dtype=torch.int16)
    E._check(lib.sdvar_op_attention(P(qo), P(kc), P(vc), fmt, None, P(outp), M * H * 64, 2, R, H, l, Lp, Ktot, 1, qb, vs, st))
    if i % 3 == 0: torch.cuda.synchronize()
    if first is None: first, firstp = out.clone(), outp.clone()
    else:
        bad += int(not torch.equal(out, first)); badp += int(not torch.equal(outp, firstp))
print(f"R={R} H={H} l={l} Ktot={Ktot} fmt={fmt}: {reps} launches, fp32 outputs differing from the first: {bad}, planes outputs: {badp}")
