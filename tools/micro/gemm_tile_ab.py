#!/usr/bin/env python3
"""A/B of the f16x2 GEMM tiles on large-M shapes with HBM-cold weights, interleaved rounds in ONE process (cdna_hip_programming.md rule 24).
python tools/micro/gemm_tile_ab.py [--depth 16] [--rounds 7]   ->  per shape: median us and TF/s of auto / 256x256 (bm 512) / 256x128 / 128x128"""
import argparse, ctypes as C, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdvar_amd import engine as E
ap = argparse.ArgumentParser(); ap.add_argument("--depth", type=int, default=16); ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--ms", default="1024,1600,2704,4096,6800"); ap.add_argument("--cfgs", default="0:0,512:1,256:1,128:1")
a = ap.parse_args()
lib = E.load_library(); dev = torch.device("cuda:0"); Cw = 64 * a.depth
st = C.c_void_p(torch.cuda.current_stream().cuda_stream); P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
cfgs = [tuple(int(v) for v in c.split(":")) for c in a.cfgs.split(",")]
for M in [int(m) for m in a.ms.split(",")]:
    for name, N, K, epi in (("qkv", 3 * Cw, Cw, 0), ("fc1", 4 * Cw, Cw, 1), ("head", 4096, Cw, 0), ("proj", Cw, Cw, 2), ("fc2", Cw, 4 * Cw, 2)):
        X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.02; b = torch.randn(N, device=dev)
        out = torch.randn(M, N, device=dev); gate = torch.randn(16, 6 * Cw, device=dev)
        Xp = torch.empty(2, M, K, dtype=torch.int16, device=dev); wsc = torch.zeros(4, device=dev)
        E._check(lib.sdvar_op_split_planes_f16(P(X), P(Xp), M, K, M * K, None, st))
        Wps = [torch.empty(2, N, K, dtype=torch.int16, device=dev) for _ in range(max(2, int(600e6 / (N * K * 4))))]
        for Wp in Wps: E._check(lib.sdvar_op_split_planes_f16(P(W), P(Wp), N, K, N * K, P(wsc), st))
        outp = torch.empty(2, M, N, dtype=torch.int16, device=dev)
        k = [0]
        def run():
            k[0] += 1; Wp = Wps[k[0] % len(Wps)]
            E._check(lib.sdvar_op_gemm_f16x2(P(Xp), M * K, P(Wp), N * K, P(wsc), P(b), P(out), N, P(outp), M * N, M, N, K, epi, P(out) if epi == 2 else None, N,
                                             P(gate) if epi == 2 else None, M // 16, 6 * Cw, st))
        def timeit(n=12):
            for _ in range(2): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): run()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / n
        res = {c: [] for c in cfgs}
        for r in range(a.rounds):
            for c in cfgs:
                E._check(lib.sdvar_debug_set_gemm_cfg(*c)); res[c].append(timeit())
        E._check(lib.sdvar_debug_set_gemm_cfg(0, 0))
        fl = 2.0 * M * N * K
        print(f"{name:5s} M={M:5d} N={N:5d} K={K:5d} " + "  ".join(f"[{c[0]}:{c[1]}] {statistics.median(v):7.1f}us {fl / statistics.median(v) / 1e6:6.1f}TF" for c, v in res.items()), flush=True)
