#!/usr/bin/env python3
"""Run ONE verify-attention shape repeatedly (timing with HIP events, or under rocprofv3 --pmc).
python tools/one_attention.py R H l prefix fmt iters [rot]     fmt: 0 fp32 cache, 1 fp16, 2 bf16x3 planes, 3 f16x2 planes, 4 one fp16 plane
rot (default 4): number of distinct caches the launches rotate through - inside a model pass every block has its own cache, so the keys come from HBM, not from
the 256 MB Infinity Cache a single re-read cache would sit in."""
import ctypes as C, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdvar_amd import engine as E
R, H, l, prefix, fmt, iters = (int(v) for v in sys.argv[1:7])
ROT = int(sys.argv[7]) if len(sys.argv) > 7 else 4
lib = E.load_library(); dev = torch.device("cuda:0"); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
Ktot = prefix + l; Lp = (Ktot + 63) // 64 * 64
sm = torch.full((H,), math.log(4.0), device=dev)
caches = []
for _ in range(ROT):
    if fmt >= 2:
        NP = {2: 3, 3: 2, 4: 1}[fmt]
        kc = torch.zeros(R, H, NP, Lp, 64, device=dev, dtype=torch.int16); vc = torch.zeros(R, H, NP, 64, Lp, device=dev, dtype=torch.int16)
    else:
        dt = torch.float16 if fmt == 1 else torch.float32
        kc = torch.zeros(R, H, Lp, 64, device=dev, dtype=dt); vc = torch.zeros_like(kc)
    for n, pos0 in ((prefix, 0), (l, prefix)):
        if n:
            qkv = torch.randn(R * n, 3 * 64 * H, device=dev); qo = torch.zeros(R, H, n, 64, device=dev)
            E._check(lib.sdvar_op_qk_norm_append(P(qkv), P(sm), P(qo), P(kc), P(vc), fmt, R, n, H, Lp, pos0, st))
    caches.append((kc, vc))
out = torch.empty(R, l, H * 64, device=dev)
qb, vs = (C.c_int32 * 1)(0), (C.c_int32 * 1)(Ktot)
cnt = [0]
def run():
    cnt[0] += 1; kc, vc = caches[cnt[0] % ROT]
    E._check(lib.sdvar_op_attention(P(qo), P(kc), P(vc), fmt, P(out), None, 0, 3, R, H, l, Lp, Ktot, 1, qb, vs, st))
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
fl = 4.0 * R * H * l * Ktot * 64
print(f"fmt={fmt} R={R} H={H} l={l} Ktot={Ktot}: {us:.1f} us/launch, {fl / us * 1e-6:.1f} TFLOP/s (algorithmic), "
      f"{R * H * 64 * 4 * (2 * Ktot + 2 * l) / us * 1e-3:.0f} GB/s (algorithmic fp32 bytes)")
