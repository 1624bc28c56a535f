#!/usr/bin/env python3
"""Slot durations of the 256 x 256 ping-pong GEMM kernel (needs a -DSDVAR_V4_STAMPS build of gemm_f16x2.o): s_memtime at the 8 slot boundaries of K-step 8,
waves 0 (early half) and 4 (late half) of workgroup 0.   python tools/micro/gemm_v4_stamps.py [M N K]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdvar_amd import engine as E
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 1024)
EPI = int(sys.argv[4]) if len(sys.argv) > 4 else 0
lib = E.load_library(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream); P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.02; b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev)
Xp = torch.empty(2, M, K, dtype=torch.int16, device=dev); Wp = torch.empty(2, N, K, dtype=torch.int16, device=dev); wsc = torch.zeros(4, device=dev)
E._check(lib.sdvar_op_split_planes_f16(P(X), P(Xp), M, K, M * K, None, st)); E._check(lib.sdvar_op_split_planes_f16(P(W), P(Wp), N, K, N * K, P(wsc), st))
stamps = torch.zeros(64, dtype=torch.int64, device=dev)
outp = torch.empty(2, M, N, dtype=torch.int16, device=dev)
E._check(lib.sdvar_debug_set_gemm_cfg(512, 1))
def gemm():
    E._check(lib.sdvar_op_gemm_f16x2(P(Xp), M * K, P(Wp), N * K, P(wsc), P(b), P(out), N, P(outp), M * N, M, N, K, EPI, None, N, None, 1, 0, st))
for _ in range(6000): gemm()            # ~0.6 s of back-to-back launches: the clock the chip holds under this load
for rep in range(4):
    E._check(lib.sdvar_debug_set_gemm_stamps(P(stamps)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gemm()
    e1.record(); torch.cuda.synchronize()
    print(f"rep {rep}: event time {e0.elapsed_time(e1) * 1e3:.1f} us")
    s = stamps.cpu().tolist()
    for w in (0, 1):
        v = s[16 * w:16 * w + 9]
        print(f"rep {rep} wave {4 * w}: " + " ".join(f"{v[k + 1] - v[k]:5d}" for k in range(8)) + f"   K-step total {v[8] - v[0]}")
    for name, o in (("first", 32), ("last", 40)):
        r, c = s[o:o + 4], s[o + 4:o + 8]
        if r[3] > r[0]:
            us = [(r[k + 1] - r[k]) / 100.0 for k in range(3)]
            ghz = [(c[k + 1] - c[k]) / max(1e-9, (r[k + 1] - r[k]) * 10.0) for k in range(3)]
            print(f"rep {rep} workgroup {name}: prologue {us[0]:.2f} us, loop {us[1]:.2f} us ({ghz[1]:.2f} GHz), epilogue {us[2]:.2f} us; entry offset vs first {(r[0] - s[32]) / 100.0:.2f} us")
E._check(lib.sdvar_debug_set_gemm_stamps(None)); E._check(lib.sdvar_debug_set_gemm_cfg(0, 0))
