#!/usr/bin/env python3
"""Run stage_forward of one model for the given stages a few times (for `rocprofv3 --kernel-trace`); tools/trace_timeline.py prints the per-kernel timeline.
python tools/stage_trace.py [--depth 16] [--batch 8] [--stages 0,1,2,3,4,5] [--reps 3]"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdvar_amd import engine as E
from sdvar_amd.ladder import LADDER_256, as_ladder
from sdvar_amd.weights import var_state_dict_device
ap = argparse.ArgumentParser(); ap.add_argument("--depth", type=int, default=16); ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--stages", default="0,1,2,3,4,5"); ap.add_argument("--reps", type=int, default=3); ap.add_argument("--gemm-mode", default=None)
a = ap.parse_args()
dev = torch.device("cuda:0"); lad = as_ladder(LADDER_256); B = a.batch
ctx = E.ModelCtx(var_state_dict_device(a.depth, LADDER_256, dev), a.depth, LADDER_256, B, 1, dev, gemm_mode=a.gemm_mode)
labels = (torch.arange(B, device=dev) % 1000)
x = torch.randn(2 * B * lad.lens[-1] * ctx.Cw, device=dev); lg = torch.empty(2 * B * lad.lens[-1] * ctx.V, device=dev)
want = [int(s) for s in a.stages.split(",")]
for rep in range(a.reps):
    ctx.begin(labels)
    for s in range(max(want) + 1):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.forward(x, s, 1, lg); e1.record(); torch.cuda.synchronize()
        if rep == a.reps - 1 and s in want: print(f"stage {s} M={2 * B * lad.lens[s]} wall {e0.elapsed_time(e1):.3f} ms", flush=True)
    ctx.kv_set_len(0)
