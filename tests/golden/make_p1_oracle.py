#!/usr/bin/env python3
"""Golden vectors of the CPU oracle at config P1's FULL size (d12 draft + d16 verify, B = 8, gamma = 2, 256^2), so that the GPU suite can hold the HIP path to
several seeds without spending minutes of host time per seed inside the GPU test budget (one oracle run of this size takes 40 - 120 s of CPU).

The oracle is the pinned restatement (oracle/var_oracle.py, checked against /root/reference by tests/golden/make_golden.py); the weights are the portable host
streams of sdvar_amd.weights.var_state_dict ('stress' init), the noise the portable Philox stream - the GPU test rebuilds both bit for bit.
Per (mode, seed): token ids, f_hat, every counter, the per-round acceptance record, the sampler's top-2 margins (tie detector) and a SAMPLE of the per-round CFG
logits of the target (what acceptance reads): per token the 8 largest entries and 8 fixed pseudo-random columns (index + value).  tests/test_gpu_fullwidth_oracle.py
still runs ONE seed against the live oracle with the full logit tensors.

    python tests/golden/make_p1_oracle.py          # writes tests/golden/p1_oracle.npz (about 15 minutes on 8 cores)
"""
import os, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import var_oracle as orc                      # noqa: E402
from sdvar_amd.ladder import LADDER_256                   # noqa: E402
from sdvar_amd.noise import exponential_noise             # noqa: E402
from sdvar_amd.weights import var_state_dict, vae_state_dict   # noqa: E402

torch.set_grad_enabled(False)
MODES = {"accept_all": 0.0, "natural": 0.5}
SEEDS = (5, 6, 7, 8)
B, CFG, GAMMA, TOP_K, TOP_P, NTOP, NRND = 8, 1.5, 2, 900, 0.96, 8, 8


def sample_columns(V, ntok):
    """8 fixed pseudo-random columns per token (numpy Philox, keyed by the token count of the round)."""
    g = np.random.Generator(np.random.Philox(key=[ntok, 777]))
    return g.integers(0, V, size=(ntok, NRND), dtype=np.int64)


def main():
    pns = LADDER_256
    t0 = time.time()
    sd_d, sd_t = var_state_dict(12, pns, "stress"), var_state_dict(16, pns, "stress")
    sd_v = vae_state_dict(pns, "stress", with_encoder=False)
    od, ot, oq = orc.OracleVAR(sd_d, 12, pns), orc.OracleVAR(sd_t, 16, pns), orc.OracleQuant(sd_v, pns)
    labels = (torch.arange(B) * 113 + 5) % 1000
    out = dict(labels=labels.numpy(), B=B, cfg=CFG, gamma=GAMMA, top_k=TOP_K, top_p=TOP_P, seeds=np.array(SEEDS), modes=np.array(list(MODES)))
    for mode, thr in MODES.items():
        for seed in SEEDS:
            noise = orc.array_noise(lambda d, B_, l, V: exponential_noise(seed, d, B_, l, V))
            tr = orc.spec_decode(od, ot, oq, labels, CFG, GAMMA, TOP_K, TOP_P, noise, thr=thr, keep=True)
            k = f"{mode}_{seed}_"
            out[k + "ids"] = torch.cat(tr.ids, 1).numpy().astype(np.int16)
            out[k + "f_hat"] = tr.f_hat.numpy()
            out[k + "margins"] = np.array(tr.margins, dtype=np.float64)
            st = tr.stats
            out[k + "counters"] = np.array([st["target_calls"], st["draft_stage_calls"], st["forced_accepts"], st["accepted_tokens"], st["gamma_final"]], dtype=np.int64)
            out[k + "rounds"] = np.array([[r["stage"], r["g"], r["n_accept"], int(r["forced"])] + list(r["matched"]) + [0] * (GAMMA - len(r["matched"])) for r in st["rounds"]], dtype=np.int64)
            for ri, cls in enumerate(tr.cfg_logits):
                cl = torch.cat(cls, 1)                                      # (B, tokens of the round, V)
                ntok = cl.shape[1]
                top = cl.topk(NTOP, dim=-1)
                rnd_idx = torch.from_numpy(sample_columns(cl.shape[-1], ntok)).unsqueeze(0).expand(B, -1, -1)
                idx = torch.cat([top.indices, rnd_idx], -1)
                out[k + f"r{ri}_idx"] = idx.numpy().astype(np.int16)
                out[k + f"r{ri}_val"] = torch.gather(cl, -1, idx).numpy()
            print(f"{mode} seed {seed}: {len(st['rounds'])} rounds, min margin {min(tr.margins):.2e}, {time.time() - t0:.0f}s", flush=True)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "p1_oracle.npz"), **out)
    print("written", os.path.getsize(os.path.join(ROOT, "tests", "golden", "p1_oracle.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
