#!/usr/bin/env python3
"""Run ONE GEMM shape repeatedly (for rocprofv3 --pmc).  python tools/one_gemm.py M N K mode iters"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdvar_amd import engine as E
M, N, K, mode, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
lib = E.load_library(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
X = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.02; b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev)
P = lambda t: C.c_void_p(t.data_ptr())
if mode == "bf16x3":
    Xp = torch.empty(3, M, K, dtype=torch.int16, device=dev); Wp = torch.empty(3, N, K, dtype=torch.int16, device=dev)
    E._check(lib.sdvar_op_split_planes(P(X), P(Xp), M, K, M * K, st)); E._check(lib.sdvar_op_split_planes(P(W), P(Wp), N, K, N * K, st))
for _ in range(iters):
    if mode == "bf16x3":
        E._check(lib.sdvar_op_gemm_bf16x3(P(Xp), M * K, P(Wp), N * K, P(b), P(out), N, None, 0, M, N, K, 0, None, 0, None, 1, 0, st))
    else:
        E._check(lib.sdvar_op_gemm(P(X), K, P(W), P(b), P(out), N, M, N, K, 0, None, 0, None, 1, 0, st))
torch.cuda.synchronize()
