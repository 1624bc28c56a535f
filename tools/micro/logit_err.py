#!/usr/bin/env python3
"""Per-stage max |logit error| of the HIP plain sampler against the CPU oracle on a reference fixture's configuration (the oracle is fed the HIP path's own
sampled ids stage by stage through the shared noise stream, so the numbers are the arithmetic error of one forward, not drift).  python tools/micro/logit_err.py [name]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import var_oracle as orc
from sdvar_amd import engine as E
from sdvar_amd.noise import exponential_noise
from sdvar_amd.weights import var_state_dict, vae_state_dict
name = sys.argv[1] if len(sys.argv) > 1 else "ar_d6_256_stress"
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
depth, pns, B = int(g["depth"]), tuple(int(p) for p in g["patch_nums"]), int(g["B"])
sd, sdv = var_state_dict(depth, pns, str(g["mode"]), int(g["wseed"])), vae_state_dict(pns, str(g["mode"]), int(g["wseed"]), with_encoder=False)
dev = torch.device("cuda:0"); torch.set_grad_enabled(False)
ctx, qc = E.ModelCtx(sd, depth, pns, B, 1, dev), E.QuantCtx(sdv, pns, B, dev)
labels = torch.from_numpy(g["labels"]).long()
seed = int(g["g_seed"])
res = E.Sampler(ctx, qc).plain_ar(labels.to(dev), float(g["cfg"]), int(g["top_k"]), float(g["top_p"]), E.Noise("host", seed), trace=True)
tr = orc.plain_ar(orc.OracleVAR(sd, depth, pns), orc.OracleQuant(sdv, pns), labels, float(g["cfg"]), int(g["top_k"]), float(g["top_p"]),
                  orc.array_noise(lambda d, B_, l, V: exponential_noise(seed, d, B_, l, V)), keep=True)
ids_ok = np.array_equal(res.ids.cpu().numpy(), g["ids"].astype(np.int64))
errs = [float((res.trace["logits"][s].cpu() - tr.logits[s]).abs().max()) for s in range(len(pns))]
scale = max(float(t.abs().max()) for t in tr.logits)
print(f"{name}: ids == fixture: {ids_ok}; max |logit| {scale:.2f}; per-stage max |dlogit| " + " ".join(f"{e:.1e}" for e in errs))
