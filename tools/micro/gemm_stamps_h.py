#!/usr/bin/env python3
"""Where a workgroup of the f16x2 small-M / 128 x 128 GEMM kernels spends its time (in-kernel stamps, HBM-cold weights).
python tools/micro/gemm_stamps_h.py M N K bm split [epi]"""
import sys, os, torch, ctypes as C, numpy as np
sys.path.insert(0, os.getcwd())
from sdvar_amd import engine as E
lib = E.load_library(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
M, N, K, bm, split = [int(v) for v in sys.argv[1:6]]; epi = int(sys.argv[6]) if len(sys.argv) > 6 else 0
nW = max(2, int(600e6 / (N * K * 4)))
X = torch.randn(M, K, device=dev); b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev); gate = torch.randn(16, 6 * 1024, device=dev)
Xp = torch.empty(2, M, K, dtype=torch.int16, device=dev); wsc = torch.zeros(4, device=dev); outp = torch.empty(2, M, N, dtype=torch.int16, device=dev)
E._check(lib.sdvar_op_split_planes_f16(P(X), P(Xp), M, K, M * K, None, st))
W = torch.randn(N, K, device=dev) * 0.02
Wps = [torch.empty(2, N, K, dtype=torch.int16, device=dev) for _ in range(nW)]
for w in Wps: E._check(lib.sdvar_op_split_planes_f16(P(W), P(w), N, K, N * K, P(wsc), st))
E._check(lib.sdvar_debug_set_gemm_cfg(bm, split))
def run(i): E._check(lib.sdvar_op_gemm_f16x2(P(Xp), M * K, P(Wps[i % nW]), N * K, P(wsc), P(b), P(out), N, P(outp), M * N, M, N, K, epi, P(out) if epi == 2 else None, N, P(gate) if epi == 2 else None, max(M // 16, 1), 6 * 1024, st))
for i in range(nW): run(i)
stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
E._check(lib.sdvar_debug_set_gemm_stamps(P(stamps)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(nW + 1); e1.record(); torch.cuda.synchronize()
E._check(lib.sdvar_debug_set_gemm_stamps(None))
s = stamps.cpu().view(4096, 8).numpy().astype(np.float64); s = s[s[:, 0] > 0]
t0 = s[:, 0].min(); r = (s[:, :4] - t0) / 100.0      # us
clk = (s[:, 7] - s[:, 4]) / np.maximum(s[:, 3] - s[:, 0], 1) * 100.0   # MHz
print(f"M={M} N={N} K={K} bm={bm} split={split} epi={epi}: {len(s)} workgroups, launch (events, incl. reduce if any) {e0.elapsed_time(e1) * 1e3:.1f} us")
q = lambda a: "min %.2f med %.2f max %.2f" % (a.min(), np.median(a), a.max())
print("  entry (us after the first workgroup's entry): " + q(r[:, 0]))
print("  entry -> first K-step landed:                 " + q(r[:, 1] - r[:, 0]))
print("  K loop:                                       " + q(r[:, 2] - r[:, 1]))
print("  epilogue (stores retired):                    " + q(r[:, 3] - r[:, 2]))
print("  exit (us after the first entry):              " + q(r[:, 3]) + f"   in-kernel clock {np.median(clk):.0f} MHz")
