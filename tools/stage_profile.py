#!/usr/bin/env python3
"""Per-stage time of one model's stage_forward (HIP events per kernel class).  python tools/stage_profile.py [--depth 16] [--batch 8]"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdvar_amd import engine as E
from sdvar_amd.ladder import LADDER_256, as_ladder
from sdvar_amd.weights import var_state_dict_device
ap = argparse.ArgumentParser(); ap.add_argument("--depth", type=int, default=16); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--gemm-mode", default=None)
a = ap.parse_args()
dev = torch.device("cuda:0"); lad = as_ladder(LADDER_256); B = a.batch
ctx = E.ModelCtx(var_state_dict_device(a.depth, LADDER_256, dev), a.depth, LADDER_256, B, 1, dev, gemm_mode=a.gemm_mode)
labels = (torch.arange(B, device=dev) % 1000)
x = torch.randn(2 * B * lad.lens[-1] * ctx.Cw, device=dev); lg = torch.empty(2 * B * lad.lens[-1] * ctx.V, device=dev)
def one_pass(profile):
    ctx.begin(labels); out = []
    for s in range(lad.S):
        if profile: E.prof_enable(True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.forward(x, s, 1, lg); e1.record(); torch.cuda.synchronize()
        cls = E.prof_collect() if profile else None
        if profile: E.prof_enable(False)
        out.append((e0.elapsed_time(e1), cls))
    ctx.kv_set_len(0); return out
for _ in range(2): one_pass(False)
wall = one_pass(False); prof = one_pass(True)
print(f"d{a.depth} B={B} mode={ctx.gemm_mode}: stage, tokens, wall ms (unprofiled), per-class ms (event-timed)")
tot = 0
for s in range(lad.S):
    c = prof[s][1]; tot += wall[s][0]
    print(f"  s{s} l={lad.lens[s]:3d} M={2*B*lad.lens[s]:5d}  wall {wall[s][0]:6.3f}  gemm {c['gemm']['ms'] + c['gemm_small']['ms']:6.3f} ({c['gemm']['launches'] + c['gemm_small']['launches']} launches) attn {c['attention']['ms'] + c['attention_small']['ms']:5.3f} ln {c['ln_modulate']['ms']:5.3f} qk {c['qk_norm_append']['ms']:5.3f}")
print(f"  total {tot:.2f} ms")
