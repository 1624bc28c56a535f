#!/usr/bin/env python3
"""VQVAE decode (f_hat -> image) timing: HIP decoder (csrc/conv.hip, csrc/vae.hip) against the PyTorch / MIOpen decoder on the same weights.
python tools/decode_bench.py [--batch 8] [--iters 10] [--no-torch]"""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from sdvar_amd import engine as E
from torch_ref import fhat_to_img_torch
from sdvar_amd.vqvae import VQVAE
from sdvar_amd.weights import vae_state_dict
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--no-torch", action="store_true"); ap.add_argument("--latent", type=int, default=16)
a = ap.parse_args()
dev = torch.device("cuda:0")
pns = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
sd = vae_state_dict(pns, "stress", 3, with_encoder=False)
ctx = E.VaeCtx(sd, a.batch, dev, latent_hw=a.latent)
f_hat = torch.randn(a.batch, 32, a.latent, a.latent, device=dev)
def timed(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / a.iters * 1e3
ms = timed(lambda: ctx.decode(f_hat))
print(f"HIP decoder     B={a.batch}: {ms:.2f} ms / batch, {a.batch / ms * 1e3:.1f} images/s")
if not a.no_torch:
    vae = VQVAE(vocab_size=4096, z_channels=32, ch=160, v_patch_nums=pns, with_encoder=False); vae.load_state_dict(sd); vae = vae.to(dev)
    ms2 = timed(lambda: fhat_to_img_torch(vae, f_hat.clone()))
    print(f"PyTorch/MIOpen  B={a.batch}: {ms2:.2f} ms / batch, {a.batch / ms2 * 1e3:.1f} images/s")
    print("max |diff| =", float((ctx.decode(f_hat) - fhat_to_img_torch(vae, f_hat.clone())).abs().max()))
