#!/usr/bin/env python3
"""Is a small stage's forward CPU-launch bound?  Times ctx.forward per stage as launched from the host against a replay of the same launches
captured in a HIP graph (torch.cuda.CUDAGraph on the stream the library launches on).  python tools/micro/graph_exp.py [--depth 16]"""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sdvar_amd import engine as E
from sdvar_amd.ladder import LADDER_256, as_ladder
from sdvar_amd.weights import var_state_dict_device
ap = argparse.ArgumentParser(); ap.add_argument("--depth", type=int, default=16); ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0"); lad = as_ladder(LADDER_256); B = a.batch
ctx = E.ModelCtx(var_state_dict_device(a.depth, LADDER_256, dev), a.depth, LADDER_256, B, 1, dev)
labels = (torch.arange(B, device=dev) % 1000)
x = torch.randn(2 * B * lad.lens[-1] * ctx.Cw, device=dev); lg = torch.empty(2 * B * lad.lens[-1] * ctx.V, device=dev)
for _ in range(2):
    ctx.begin(labels)
    for s in range(lad.S): ctx.forward(x, s, 1, lg)
    ctx.kv_set_len(0)
torch.cuda.synchronize()
side = torch.cuda.Stream()
def timed(fn, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    ts.sort(key=lambda p: p[1]); return ts[len(ts) // 2]
ctx.begin(labels)
tot_h = tot_g = 0.0
for s in range(lad.S):
    # eager (host launches); the KV length is rewound so every repeat is the same launch sequence
    def eager(): ctx.kv_set_len(lad.begin(s)); ctx.forward(x, s, 1, lg)
    he, we = timed(eager)
    g = torch.cuda.CUDAGraph()
    ctx.kv_set_len(lad.begin(s))
    with torch.cuda.graph(g, stream=side):          # the library launches on torch's current stream = the capturing one
        ctx.forward(x, s, 1, lg)
    def replay(): g.replay()
    hg, wg = timed(replay)
    tot_h += we; tot_g += wg
    print(f"s{s} M={2 * B * lad.lens[s]:5d}: eager host-enqueue {he:6.3f} ms, done {we:6.3f} ms | graph replay enqueue {hg:6.3f} ms, done {wg:6.3f} ms", flush=True)
print(f"total eager {tot_h:.2f} ms, graph {tot_g:.2f} ms")
