#!/usr/bin/env python3
"""Does the 1024^2 ladder (utils/arg_util.py:249: 14 stages, L = 9451, a 64 x 64 final map) run through the HIP sampler?  d2 model, B = 1, plain AR against the oracle."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import var_oracle as orc
from sdvar_amd import engine as E
from sdvar_amd.ladder import LADDER_1024, as_ladder
from sdvar_amd.noise import exponential_noise
from sdvar_amd.weights import var_state_dict, vae_state_dict
torch.set_grad_enabled(False)
pns, depth, B, seed = LADDER_1024, 2, 1, 3
lad = as_ladder(pns)
sd, sdv = var_state_dict(depth, pns, "stress", 1234), vae_state_dict(pns, "stress", 1234, ch=32, with_encoder=False)
dev = torch.device("cuda:0")
ctx, qc = E.ModelCtx(sd, depth, pns, B, 1, dev), E.QuantCtx(sdv, pns, B, dev)
labels = torch.tensor([417])
t0 = time.time()
res = E.Sampler(ctx, qc).plain_ar(labels.to(dev), 1.5, 900, 0.96, E.Noise("host", seed), trace=True)
torch.cuda.synchronize(); print(f"HIP plain_ar: {time.time() - t0:.2f}s, L = {lad.L}")
t0 = time.time()
tr = orc.plain_ar(orc.OracleVAR(sd, depth, pns), orc.OracleQuant(sdv, pns), labels, 1.5, 900, 0.96,
                  orc.array_noise(lambda d, B_, l, V: exponential_noise(seed, d, B_, l, V)), keep=True)
print(f"oracle: {time.time() - t0:.1f}s")
ids, want = res.ids.cpu().numpy(), torch.cat(tr.ids, 1).numpy()
nd = int((ids != want).sum())
errs = [float((res.trace["logits"][s].cpu() - tr.logits[s]).abs().max()) for s in range(lad.S)]
print("ids differing:", nd, "of", want.size, "; per-stage max |dlogit|", " ".join(f"{e:.1e}" for e in errs))
print("f_hat max diff", float((res.f_hat.cpu() - tr.f_hat).abs().max()), "min margin", min(tr.margins))
# the decoder at a 64 x 64 latent (1024^2 image), small width
try:
    vc = E.VaeCtx(sdv, 1, dev, latent_hw=64)
    g = torch.Generator().manual_seed(5)
    f = torch.randn(1, 32, 64, 64, generator=g) * 1.5
    t0 = time.time(); img = vc.decode(f.to(dev)).clamp(-1, 1).add(1).mul(0.5).cpu(); print(f"HIP decode 1024^2: {time.time() - t0:.2f}s", tuple(img.shape))
    t0 = time.time(); want = orc.decode_image(sdv, f.clone()); print(f"oracle decode: {time.time() - t0:.1f}s; max diff {float((img - want).abs().max()):.2e}")
except Exception as e:
    print("decoder at latent 64 failed:", repr(e)[:300])
