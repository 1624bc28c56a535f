#!/usr/bin/env python3
"""Slot durations of the 8-wave ping-pong attention kernel (needs a -DSDVAR_ATT_STAMPS build of attention_f16x2.o): s_memtime at the slot boundaries of tile 10
(S, V1, PV, V2) for waves 0 and 4 of workgroup (0, 0, 0), and the kernel's phases.   python tools/micro/attn_pp_stamps.py [R H l prefix]"""
import ctypes as C, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdvar_amd import engine as E
R, H, l, prefix = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (16, 16, 256, 424)
lib = E.load_library(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
Ktot = prefix + l; Lp = (Ktot + 63) // 64 * 64; fmt = 3
sm = torch.full((H,), math.log(4.0), device=dev)
caches = []
for _ in range(4):
    kc = torch.zeros(R, H, 2, Lp, 64, device=dev, dtype=torch.int16); vc = torch.zeros_like(kc)
    for n, pos0 in ((prefix, 0), (l, prefix)):
        qkv = torch.randn(R * n, 3 * 64 * H, device=dev); qo = torch.zeros(R, H, n, 64, device=dev)
        E._check(lib.sdvar_op_qk_norm_append(P(qkv), P(sm), P(qo), P(kc), P(vc), fmt, R, n, H, Lp, pos0, st))
    caches.append((kc, vc))
out = torch.empty(R, l, H * 64, device=dev)
qb, vs = (C.c_int32 * 1)(0), (C.c_int32 * 1)(Ktot)
stamps = torch.zeros(64, dtype=torch.int64, device=dev)
def run(i):
    kc, vc = caches[i % 4]
    E._check(lib.sdvar_op_attention(P(qo), P(kc), P(vc), fmt, P(out), None, 0, 3, R, H, l, Lp, Ktot, 1, qb, vs, st))
for i in range(3000): run(i)
for rep in range(3):
    E._check(lib.sdvar_debug_set_gemm_stamps(P(stamps)))
    run(rep); torch.cuda.synchronize()
    s = stamps.cpu().tolist()
    for w in (0, 1):
        v = s[16 * w:16 * w + 5]
        if v[3] == 0:            # two-slot schedule (SDVAR_ATTN_PP_SCHED=1, the default): stamps 0 / 1 / 2 = start of M(t), of V(t), end of V(t)
            print(f"rep {rep} wave {4 * w}: M {v[1] - v[0]}  V {v[2] - v[1]}   tile total {v[2] - v[0]}")
        else:
            print(f"rep {rep} wave {4 * w}: S {v[1] - v[0]}  V1 {v[2] - v[1]}  PV {v[3] - v[2]}  V2 {v[4] - v[3]}   tile total {v[4] - v[0]}")
    r, c = s[32:36], s[36:40]
    us = [(r[k + 1] - r[k]) / 100.0 for k in range(3)]
    print(f"rep {rep}: prologue {us[0]:.2f} us, loop {us[1]:.2f} us ({(c[2] - c[1]) / max(1e-9, (r[2] - r[1]) * 10.0):.2f} GHz), epilogue {us[2]:.2f} us")
E._check(lib.sdvar_debug_set_gemm_stamps(None))
