#!/usr/bin/env python3
"""Feasibility of partitioning the chip between the sampler's small stages and the VQVAE decode (hipExtStreamCreateWithCUMask).

The sampler's stages 0-5 are latency bound and leave most of the 256 CUs idle; the decode of the previous batch is matrix-pipe bound.  Run together on two
ordinary streams they do not overlap usefully: a decoder convolution occupies every CU's LDS (156 KB per workgroup), so each of the ~600 dependent small
launches of the sampler waits for convolution workgroups to retire.  Question asked here: with the decode confined to one set of CUs and the small stages to
the rest, do the two run at the same time at (nearly) their stand-alone speeds?

  python tools/micro/cu_mask_exp.py [--small-cus-per-xcd 8]
Prints: a large GEMM on masked streams (does the mask bite, and which bit order spreads over the XCDs), decode alone (all CUs / its partition), stages 0-5
alone (all CUs / its partition), and both concurrently."""
import argparse, ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdvar_amd import engine as E
from sdvar_amd.ladder import LADDER_256, as_ladder
from sdvar_amd.weights import var_state_dict_device, vae_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--small-cus-per-xcd", type=int, default=8)
ap.add_argument("--layout", default="interleaved", choices=["interleaved", "blocked"], help="bit i -> XCD i % 8 (interleaved) or XCD i // 32 (blocked)")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.init(); torch.zeros(1, device=dev)
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int


def masked_stream(cus):
    """cus: iterable of CU bit indices (0..255)"""
    words = [0] * 8
    for c in cus:
        words[c // 32] |= 1 << (c % 32)
    arr = (C.c_uint32 * 8)(*words)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


def bits(per_xcd_lo, per_xcd_hi):
    """CU bit indices of CUs [lo, hi) of every XCD under the assumed layout"""
    if a.layout == "interleaved":
        return [8 * k + x for x in range(8) for k in range(per_xcd_lo, per_xcd_hi)]
    return [32 * x + k for x in range(8) for k in range(per_xcd_lo, per_xcd_hi)]


def timed(fn, stream, n=5):
    with torch.cuda.stream(stream):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n): fn()
        e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


# 1. does the mask bite?  a 4096 x 4096 x 4096 fp32 torch matmul on streams with 256 / 128 / 64 / 32 CUs
x = torch.randn(4096, 4096, device=dev); y = torch.randn(4096, 4096, device=dev)
full = torch.cuda.Stream(device=dev)
print(f"matmul 4096^3 fp32: all CUs {timed(lambda: x @ y, full):.3f} ms")
for n in (16, 8, 4):
    print(f"  {n:2d} CUs per XCD (assumed {a.layout}): {timed(lambda: x @ y, masked_stream(bits(0, n))):.3f} ms;  first {8 * n} bits: {timed(lambda: x @ y, masked_stream(range(8 * n))):.3f} ms")

# 2. the two workloads
lad = as_ladder(LADDER_256); B = 8
ctx = E.ModelCtx(var_state_dict_device(16, LADDER_256, dev), 16, LADDER_256, B, 1, dev)
vc = E.VaeCtx(vae_state_dict(LADDER_256, "perf", with_encoder=False), B, dev)
labels = torch.arange(B, device=dev) % 1000
xs = torch.randn(2 * B * lad.lens[-1] * ctx.Cw, device=dev); lg = torch.empty(2 * B * lad.lens[-1] * ctx.V, device=dev)
f_hat = torch.randn(B, 32, 16, 16, device=dev); img = torch.empty(B, 3, 256, 256, device=dev)


def small_stages():
    ctx.begin(labels)
    for s in range(6): ctx.forward(xs, s, 1, lg)
    ctx.kv_set_len(0)


def decode():
    vc.decode(f_hat, out=img)


ns = a.small_cus_per_xcd
s_small, s_dec = masked_stream(bits(0, ns)), masked_stream(bits(ns, 32))
s_full2 = torch.cuda.Stream(device=dev)
print(f"stages 0-5 (d16, B=8): all CUs {timed(small_stages, full):.3f} ms; {8 * ns} CUs {timed(small_stages, s_small):.3f} ms")
print(f"decode (B=8):          all CUs {timed(decode, full):.3f} ms; {256 - 8 * ns} CUs {timed(decode, s_dec):.3f} ms")


def both(sa, sb, n=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(sa): small_stages()
        with torch.cuda.stream(sb): decode()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


both(full, s_full2, 2); both(s_small, s_dec, 2); both(full, s_dec, 2)
print(f"both at once: two ordinary streams {both(full, s_full2):.3f} ms; partitioned {both(s_small, s_dec):.3f} ms; only the decode confined {both(full, s_dec):.3f} ms per (stages 0-5 + decode)")
