#!/usr/bin/env python3
"""Debug aid: is the f16x2 attention deterministic launch to launch, and do its fp32 and planes outputs agree?  python tools/micro/attn_determinism2.py R H l prefix"""
import ctypes as C, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdvar_amd import engine as E
R, H, l, prefix = (int(v) for v in sys.argv[1:5])
lib = E.load_library(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
Ktot = prefix + l; Lp = (Ktot + 5 + 63) // 64 * 64; fmt = 3
g = torch.Generator(device="cpu").manual_seed(1)
sm = torch.full((H,), math.log(4.0), device=dev)
kc = torch.zeros(R, H, 2, Lp, 64, device=dev, dtype=torch.int16); vc = torch.zeros_like(kc)
for n, pos0 in ((prefix, 0), (l, prefix)):
    if n:
        qkv = torch.randn(R * n, 3 * 64 * H, generator=g).to(dev); qo = torch.zeros(R, H, n, 64, device=dev)
        E._check(lib.sdvar_op_qk_norm_append(P(qkv), P(sm), P(qo), P(kc), P(vc), fmt, R, n, H, Lp, pos0, st))
qb, vs = (C.c_int32 * 1)(0), (C.c_int32 * 1)(Ktot)
outs = []
for i in range(6):
    out = torch.full((R, l, H * 64), float("nan"), device=dev)
    E._check(lib.sdvar_op_attention(P(qo), P(kc), P(vc), fmt, P(out), None, 0, 2, R, H, l, Lp, Ktot, 1, qb, vs, st))
    outs.append(out.cpu())
    if i % 2 == 0:       # something else in between: another shape of the same kernel family
        tmp = torch.empty(R, l, H * 64, device=dev)
        E._check(lib.sdvar_op_attention(P(qo), P(kc), P(vc), fmt, P(tmp), None, 0, 2, R, H, min(l, 40), Lp, min(Ktot, prefix + 40), 1, qb, (C.c_int32 * 1)(min(Ktot, prefix + 40)), st))
print("launch-to-launch max |diff| of the fp32 output:", [float((o - outs[0]).abs().max()) for o in outs[1:]], "nan:", bool(torch.isnan(outs[0]).any()))
# majority vote = the reference; describe where a deviating launch differs
ref = torch.stack(outs).median(0)[0]
for i, o in enumerate(outs):
    d = (o - ref).abs().view(R, l, H, 64)
    nz = d > 0
    if nz.any():
        idx = nz.nonzero()
        rows, qs, hs, cs = (sorted(set(idx[:, k].tolist())) for k in range(4))
        print(f"  launch {i}: {int(nz.sum())} elements differ (max {float(d.max()):.3e}); rows {rows} heads {hs} queries {qs[:12]}{'...' if len(qs) > 12 else ''} ({len(qs)}) channels {cs[:16]}{'...' if len(cs) > 16 else ''} ({len(cs)})")
M = R * l
outp = torch.zeros(2, H * 2, M, 32, device=dev, dtype=torch.int16)
E._check(lib.sdvar_op_attention(P(qo), P(kc), P(vc), fmt, None, P(outp), M * H * 64, 2, R, H, l, Lp, Ktot, 1, qb, vs, st))
v = sum(outp.cpu()[k].view(torch.float16).double() for k in range(2)).permute(1, 0, 2).reshape(M, -1).view(R, l, H * 64)
d = (v - outs[0].double()).abs()
print("planes vs fp32 output: max |diff|", float(d.max()), "at", [int(x) for x in (d == d.max()).nonzero()[0]], "fp32 value there", float(outs[0][tuple((d == d.max()).nonzero()[0])]))
h = outp.cpu()[0].view(torch.float16).double().permute(1, 0, 2).reshape(M, -1).view(R, l, H * 64)
print("  high plane vs fp16(fp32 output): max |diff|", float((h - outs[0].half().double()).abs().max()))
