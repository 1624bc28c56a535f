#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM over the exact (M, N, K, epilogue) set one sampling step launches.
Usage (GPU box):  python tools/gemm_bench.py [--depth 16] [--batch 8] [--iters 20]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdvar_amd import engine as E  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=16)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--chunk", type=int, default=1)
    ap.add_argument("--mode", default="f32", choices=["f32", "bf16x3", "f16x2"])
    ap.add_argument("--dump", default="", help="append the full sweep table (JSON lines) to this file")
    ap.add_argument("--sweep", action="store_true", help="time every (row tile, K slices) candidate per shape")
    ap.add_argument("--cold", action="store_true", help="f16x2: rotate through ~600 MB of weight tensors per shape so that every launch streams its weights from HBM, "
                    "as inside a model pass (the default re-reads one tensor, which stays in the 256 MB Infinity Cache)")
    a = ap.parse_args()
    lib = E.load_library()
    dev = torch.device("cuda:0")
    Cw, R = 64 * a.depth, 2 * a.batch
    lens = [p * p for p in (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)]
    if a.chunk > 1:
        lens = [sum(lens[i:i + a.chunk]) for i in range(0, len(lens), a.chunk)]
    shapes = [("qkv", 3 * Cw, Cw, 0), ("proj", Cw, Cw, 2), ("fc1", 4 * Cw, Cw, 1), ("fc2", Cw, 4 * Cw, 2)]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot_t, tot_f = 0.0, 0.0
    print(f"{'op':5s} {'M':>6s} {'N':>5s} {'K':>5s} {'us':>9s} {'TFLOP/s':>8s}")
    for l in lens:
        M = R * l
        for name, N, K, epi in shapes:
            X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.02; b = torch.randn(N, device=dev)
            out = torch.randn(M, N, device=dev); gate = torch.randn(R, 6 * Cw, device=dev)
            if a.mode == "bf16x3":
                Xp = torch.empty(3, M, K, dtype=torch.int16, device=dev); Wp = torch.empty(3, N, K, dtype=torch.int16, device=dev)
                E._check(lib.sdvar_op_split_planes(C.c_void_p(X.data_ptr()), C.c_void_p(Xp.data_ptr()), M, K, M * K, st))
                E._check(lib.sdvar_op_split_planes(C.c_void_p(W.data_ptr()), C.c_void_p(Wp.data_ptr()), N, K, N * K, st))
                outp = torch.empty(3, M, N, dtype=torch.int16, device=dev)
            if a.mode == "f16x2":
                Xp = torch.empty(2, M, K, dtype=torch.int16, device=dev); wsc = torch.zeros(4, device=dev)
                E._check(lib.sdvar_op_split_planes_f16(C.c_void_p(X.data_ptr()), C.c_void_p(Xp.data_ptr()), M, K, M * K, None, st))
                Wps = [torch.empty(2, N, K, dtype=torch.int16, device=dev) for _ in range(max(2, int(600e6 / (N * K * 4))) if a.cold else 1)]
                for Wp in Wps: E._check(lib.sdvar_op_split_planes_f16(C.c_void_p(W.data_ptr()), C.c_void_p(Wp.data_ptr()), N, K, N * K, C.c_void_p(wsc.data_ptr()), st))
                outp = torch.empty(2, M, N, dtype=torch.int16, device=dev)
            calls = [0]
            def run():
                if a.mode == "f16x2":
                    calls[0] += 1; Wp = Wps[calls[0] % len(Wps)]
                    E._check(lib.sdvar_op_gemm_f16x2(C.c_void_p(Xp.data_ptr()), M * K, C.c_void_p(Wp.data_ptr()), N * K, C.c_void_p(wsc.data_ptr()), C.c_void_p(b.data_ptr()),
                                                     C.c_void_p(out.data_ptr()), N, C.c_void_p(outp.data_ptr()), M * N, M, N, K, epi,
                                                     C.c_void_p(out.data_ptr()) if epi == 2 else None, N, C.c_void_p(gate.data_ptr()) if epi == 2 else None, l, 6 * Cw, st))
                    return
                if a.mode == "bf16x3":
                    E._check(lib.sdvar_op_gemm_bf16x3(C.c_void_p(Xp.data_ptr()), M * K, C.c_void_p(Wp.data_ptr()), N * K, C.c_void_p(b.data_ptr()),
                                                      C.c_void_p(out.data_ptr()), N, C.c_void_p(outp.data_ptr()), M * N, M, N, K, epi,
                                                      C.c_void_p(out.data_ptr()) if epi == 2 else None, N, C.c_void_p(gate.data_ptr()) if epi == 2 else None, l, 6 * Cw, st))
                    return
                E._check(lib.sdvar_op_gemm(C.c_void_p(X.data_ptr()), K, C.c_void_p(W.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()), N, M, N, K, epi,
                                           C.c_void_p(out.data_ptr()) if epi == 2 else None, N, C.c_void_p(gate.data_ptr()) if epi == 2 else None, l, 6 * Cw, st))
            def timeit():
                for _ in range(3): run()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                n = max(a.iters, 2 * len(Wps)) if a.mode == "f16x2" and a.cold else a.iters
                for _ in range(n): run()
                e1.record(); torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / n
            us = timeit()
            if a.sweep:
                res = []
                for bm in ((32, 64, 128, 256) if a.mode != "f32" else (32, 64, 128)):
                    for split in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32):
                        if split > K // 64: continue
                        E._check(lib.sdvar_debug_set_gemm_cfg(bm, split))
                        res.append((timeit(), bm, split))
                E._check(lib.sdvar_debug_set_gemm_cfg(0, 0))
                if a.dump:
                    import json
                    with open(a.dump, "a") as f:
                        f.write(json.dumps(dict(mode=a.mode, op=name, M=M, N=N, K=K, auto_us=us, cold=bool(a.cold), slab_only=bool(int(os.environ.get("SDVAR_GEMM_DBG", "0")) & 8),
                                                cands=[(t, bm, sp) for t, bm, sp in res])) + "\n")
                res.sort()
                print(f"   sweep {name} M={M}: auto {us:.1f}us; best " + ", ".join(f"{t:.1f}us(bm{bm},s{sp})" for t, bm, sp in res[:4]))
            fl = 2.0 * M * N * K
            tot_t += us * a.depth; tot_f += fl * a.depth
            print(f"{name:5s} {M:6d} {N:5d} {K:5d} {us:9.1f} {fl / us / 1e6:8.1f}")
    print(f"one pass over all stages, {a.depth} layers: {tot_t / 1e3:.2f} ms, {tot_f / tot_t / 1e6:.1f} TFLOP/s average")


if __name__ == "__main__":
    main()
