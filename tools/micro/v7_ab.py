#!/usr/bin/env python3
"""A/B of the large-tile f16x2 GEMM kernels on the shapes N = 3 C / 4 C of d12 and d16 (cold weights, forced tile): 128 x 128 (bm 128), 256 x 128 (256), 256 x 256 (512),
256 x 192 (768), and the cost model's own choice (0).   python tools/micro/v7_ab.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdvar_amd import engine as E
lib = E.load_library(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
def bench(M, N, K, epi, bm, iters=30):
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.02; b = torch.randn(N, device=dev)
    Xp = torch.empty(2, M, K, dtype=torch.int16, device=dev); wsc = torch.zeros(4, device=dev)
    E._check(lib.sdvar_op_split_planes_f16(P(X), P(Xp), M, K, M * K, None, st))
    Wps = [torch.empty(2, N, K, dtype=torch.int16, device=dev) for _ in range(max(2, int(600e6 / (N * K * 4))))]
    for Wp in Wps: E._check(lib.sdvar_op_split_planes_f16(P(W), P(Wp), N, K, N * K, P(wsc), st))
    out = torch.empty(M, N, device=dev); outp = torch.empty(2, M, N, dtype=torch.int16, device=dev)
    E._check(lib.sdvar_debug_set_gemm_cfg(bm, 1 if bm else 0))
    i = [0]
    def run():
        i[0] += 1
        E._check(lib.sdvar_op_gemm_f16x2(P(Xp), M * K, P(Wps[i[0] % len(Wps)]), N * K, P(wsc), P(b), P(out), N, P(outp), M * N, M, N, K, epi, None, N, None, 1, 0, st))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    E._check(lib.sdvar_debug_set_gemm_cfg(0, 0))
    return e0.elapsed_time(e1) * 1e3 / iters
for Cw in (1024, 768):
    for name, N, epi in (("qkv", 3 * Cw, 0), ("fc1", 4 * Cw, 1)):
        for M in (1024, 1600, 2704, 4096, 6800):
            row = []
            for bm in (0, 128, 256, 512, 768):
                us = bench(M, N, Cw, epi, bm)
                row.append(f"bm{bm:>3d} {us:6.1f}us {2.0 * M * N * Cw / us / 1e6:5.0f}TF")
            print(f"C={Cw} {name} M={M:5d} N={N:5d}: " + " | ".join(row), flush=True)
