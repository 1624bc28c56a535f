#!/usr/bin/env python3
"""fc2 / proj (N = C) at large M: the 128 x 128 kernel the cost model picks against the 256 x 256 and 256 x 192 tiles with a K split (slab launch + reduce launch here; inside
the model the LayerNorm consumer would sum the slabs).   python tools/micro/fc2_split_ab.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdvar_amd import engine as E
lib = E.load_library(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def bench(M, N, K, bm, split, iters=30):
    X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.02; b = torch.randn(N, device=dev)
    Xp = torch.empty(2, M, K, dtype=torch.int16, device=dev); wsc = torch.zeros(4, device=dev)
    E._check(lib.sdvar_op_split_planes_f16(P(X), P(Xp), M, K, M * K, None, st))
    Wps = [torch.empty(2, N, K, dtype=torch.int16, device=dev) for _ in range(max(2, int(600e6 / (N * K * 4))))]
    for Wp in Wps: E._check(lib.sdvar_op_split_planes_f16(P(W), P(Wp), N, K, N * K, P(wsc), st))
    out = torch.randn(M, N, device=dev); gate = torch.randn(16, 6 * N, device=dev)
    E._check(lib.sdvar_debug_set_gemm_cfg(bm, split))
    i = [0]
    def run():
        i[0] += 1
        E._check(lib.sdvar_op_gemm_f16x2(P(Xp), M * K, P(Wps[i[0] % len(Wps)]), N * K, P(wsc), P(b), P(out), N, None, 0, M, N, K, 2, P(out), N, P(gate), M // 16, 6 * N, st))
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    E._check(lib.sdvar_debug_set_gemm_cfg(0, 0))
    return e0.elapsed_time(e1) * 1e3 / iters
for Cw in (1024, 768):
    for name, K in (("fc2", 4 * Cw), ("proj", Cw)):
        for M in (2704, 4096):
            row = [f"auto {bench(M, Cw, K, 0, 0):6.1f}"]
            for bm, sp in ((128, 1), (512, 2), (512, 4), (768, 2), (768, 4), (256, 2)):
                if K // 32 < 2 * sp: continue
                row.append(f"bm{bm}/s{sp} {bench(M, Cw, K, bm, sp):6.1f}")
            print(f"C={Cw} {name} M={M} N={Cw} K={K}: " + " | ".join(row) + "  (us; split launches include their reduce launch)", flush=True)
