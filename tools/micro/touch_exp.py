#!/usr/bin/env python3
"""Upper bound of an in-kernel weight prefetch: a GEMM on HBM-cold weights against the same GEMM whose weight tensor was just read once by another kernel on the same stream
(so it sits in the Infinity Cache and in the L2 of the XCDs that touched it - not necessarily the L2 of the XCD that will use it), against warm (the same tensor every time).
HIP events around the GEMM launch only.   python tools/micro/touch_exp.py [C]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdvar_amd import engine as E
Cw = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = E.load_library(); dev = torch.device("cuda:0"); P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def bench(M, N, K, epi, mode, iters=40):
    X = torch.randn(M, K, device=dev); Xp = torch.empty(2, M, K, dtype=torch.int16, device=dev); wsc = torch.zeros(4, device=dev)
    E._check(lib.sdvar_op_split_planes_f16(P(X), P(Xp), M, K, M * K, None, st()))
    W = torch.randn(N, K, device=dev) * 0.02
    Wps = [torch.empty(2, N, K, dtype=torch.int16, device=dev) for _ in range(1 if mode == "warm" else max(3, int(700e6 / (N * K * 4))))]
    for Wp in Wps: E._check(lib.sdvar_op_split_planes_f16(P(W), P(Wp), N, K, N * K, P(wsc), st()))
    b = torch.randn(N, device=dev); out = torch.randn(M, N, device=dev); outp = torch.empty(2, M, N, dtype=torch.int16, device=dev); gate = torch.randn(16, 6 * Cw, device=dev)
    tot = 0.0
    dummies = [torch.empty_like(Wps[0]) for _ in range(max(3, int(700e6 / (N * K * 4))))]          # the control: the same kernel over an unrelated tensor of the same size (cold too)
    for i in range(iters + 3):
        Wp = Wps[i % len(Wps)]
        if mode == "touched":
            Wp.view(torch.int32).sum()                                  # one pass over the planes on the same stream
        else:
            dummies[i % len(dummies)].view(torch.int32).sum()           # keeps the queue equally busy in front of the timed launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        E._check(lib.sdvar_op_gemm_f16x2(P(Xp), M * K, P(Wp), N * K, P(wsc), P(b), P(out), N, P(outp), M * N, M, N, K, epi, P(out) if epi == 2 else None, N,
                                         P(gate) if epi == 2 else None, max(M // 16, 1), 6 * Cw, st()))
        e1.record(); torch.cuda.synchronize()
        if i >= 3: tot += e0.elapsed_time(e1) * 1e3
    return tot / iters
for M in (144, 576, 1024, 2704, 4096):
    for name, N, K, epi in (("qkv", 3 * Cw, Cw, 0), ("proj", Cw, Cw, 2), ("fc1", 4 * Cw, Cw, 1), ("fc2", Cw, 4 * Cw, 2)):
        t = [bench(M, N, K, epi, m) for m in ("cold", "touched", "warm")]
        print(f"C={Cw} M={M:5d} {name:4s}: cold {t[0]:7.1f} us | touched {t[1]:7.1f} us | warm {t[2]:7.1f} us", flush=True)
