"""GPU: the other BASELINE.json configurations at FULL size, checked through size-independent properties (the CPU oracle
cannot run these in test time): determinism, I1 (identical draft/target => plain-AR ids, ceil(S/gamma) verify calls),
chunk-verify == stage-wise logits, rollback, and for P4 the fp16 KV cache against the fp32 one.
Weights: the stress init drawn on the device (sdvar_amd.weights.var_state_dict_device)."""
import numpy as np
import pytest
import torch

from sdvar_amd import engine as E
from torch_ref import fhat_to_img_torch
from sdvar_amd.ladder import LADDER_256, LADDER_512, as_ladder
from sdvar_amd.weights import vae_state_dict, var_state_dict_device

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _pair(dev, d_draft, d_target, pns, B, gamma, kv_fp16=False, same=False):
    sd_t = var_state_dict_device(d_target, pns, dev, mode="stress")
    sd_d = sd_t if same else var_state_dict_device(d_draft, pns, dev, mode="stress")
    sd_v = vae_state_dict(pns, "stress", with_encoder=False)
    dc = E.ModelCtx(sd_d, d_target if same else d_draft, pns, B, 1, dev, kv_fp16=kv_fp16)
    tc = E.ModelCtx(sd_t, d_target, pns, B, gamma, dev, kv_fp16=kv_fp16)
    qc = E.QuantCtx(sd_v, pns, B, dev)
    return E.Sampler(tc, qc, dc), (dc, tc, qc)


def _close(objs):
    for o in objs:
        o.close()
    torch.cuda.empty_cache()


def _chunk_vs_stagewise(smp, labels, s0, n, tol=1e-3):
    """HIP chunk forward over stages [s0, s0+n) == HIP stage-by-stage forwards on the same inputs."""
    t, lad, B, V = smp.t, smp.lad, labels.shape[0], smp.t.V
    dev = labels.device
    g = torch.Generator(device=dev); g.manual_seed(7)
    xs = [torch.randn(2 * B, lad.lens[s], t.Cw, generator=g, device=dev) for s in range(s0 + n)]
    t.begin(labels)
    outs = []
    for s in range(s0 + n):
        x = xs[s].clone()
        t.forward(x, s, 1, smp.logits_t)
        if s >= s0:
            outs.append(smp.logits_t[:2 * B * lad.lens[s] * V].view(2 * B, lad.lens[s], V).clone())
    t.kv_set_len(lad.begin(s0))                                               # rollback, then ONE pass over the chunk
    x = torch.cat(xs[s0:s0 + n], 1).contiguous()
    lsum = x.shape[1]
    t.forward(x, s0, n, smp.logits_t)
    lg = smp.logits_t[:2 * B * lsum * V].view(2 * B, lsum, V)
    err = (lg - torch.cat(outs, 1)).abs().max().item()
    t.kv_set_len(0)
    assert err <= tol, err
    return err


def test_P2_d16_draft_d24_verify_B16(dev):
    smp, objs = _pair(dev, 16, 24, LADDER_256, 16, 2)
    labels = (torch.arange(16, device=dev) * 61) % 1000
    a = smp.spec_decode(labels, 1.5, 2, 900, 0.96, E.Noise("device", 3), thr=0.0)
    ids_a, st_a = a.ids.clone(), dict(a.stats)
    b = smp.spec_decode(labels, 1.5, 2, 900, 0.96, E.Noise("device", 3), thr=0.0)
    assert torch.equal(ids_a, b.ids) and st_a["target_calls"] == 5 and st_a["accepted_tokens"] == 680      # deterministic, accept_all
    # accept_all ids are the plain draft AR ids on the same draws (every stage comes from the draft)
    ref = E.Sampler(objs[0], objs[2]).plain_ar(labels, 1.5, 900, 0.96, E.Noise("device", 3)).ids
    assert torch.equal(ids_a, ref)
    _chunk_vs_stagewise(smp, labels, 7, 2)
    _close(objs)


def test_P3_shard_d16_draft_d30_verify_B8(dev):
    """One rank's share of config P3 (B=64 over 8 GPUs): natural / reject_all policy bookkeeping at full size."""
    smp, objs = _pair(dev, 16, 30, LADDER_256, 8, 3)
    labels = (torch.arange(8, device=dev) + 8 * 3) % 1000                      # the shard of rank 3
    res = smp.spec_decode(labels, 1.5, 3, 900, 0.96, E.Noise("device", 11, image_offset=24), thr=2.0)
    st = res.stats
    assert st["forced_accepts"] == 10 and st["gamma_final"] == 1 and st["target_calls"] == 12 and st["draft_stage_calls"] == 3 + 2 + 10
    # rollback leaves no trace: the committed ids are the plain draft AR ids on the draws of the committed stages
    draws, d = [], 0
    for r in st["rounds"]:
        if r["n_accept"]:
            draws.append(d)
        d += r["g"]
    # replay: plain AR of the draft consuming draw indices `draws`
    m, qz, lad = objs[0], objs[2], smp.lad
    s2 = E.Sampler(m, qz)
    ids = torch.zeros_like(res.ids)
    m.begin(labels); f = torch.zeros(8, 32, 16, 16, device=dev); m.place_first(s2.x_t, 1)
    for si in range(10):
        l = lad.lens[si]
        m.forward(s2.x_t, si, 1, s2.logits_t)
        E.cfg_sample(s2.logits_t, 8, l, 4096, lad.cfg_t(1.5, si), 900, 0.96, None, 11, draws[si], 24, ids, lad.begin(si), lad.L)
        qz.next(si, ids[:, lad.begin(si):], lad.L, f, None if si == 9 else s2.nxt[0], 8)
        if si < 9:
            m.embed_next(s2.nxt[0], si + 1, s2.x_t, lad.lens[si + 1], 0)
    m.kv_set_len(0)
    assert torch.equal(ids, res.ids)
    assert (f - res.f_hat).abs().max().item() == 0.0
    _chunk_vs_stagewise(smp, labels, 6, 3)
    _close(objs)


def test_P4_d30_512_fp16_kv_cfg3(dev):
    """d30, 512^2 ladder (L = 2240), B = 8, cfg 3.0, fp16 KV cache."""
    B = 8
    labels = (torch.arange(B, device=dev) * 97) % 1000
    smp16, objs16 = _pair(dev, 30, 30, LADDER_512, B, 2, kv_fp16=True, same=True)
    # I1 at full size: identical models, greedy (top_k = 1) => all accepted, plain-AR ids, ceil(10/2) verify calls
    ref = E.Sampler(objs16[1], objs16[2]).plain_ar(labels, 3.0, 1, 0.0, E.Noise("device", 5), trace=True)
    ids_ref, logits16 = ref.ids.clone(), [t.clone() for t in ref.trace["logits"][:4]]
    res = smp16.spec_decode(labels, 3.0, 2, 1, 0.0, E.Noise("device", 5))
    assert torch.equal(res.ids, ids_ref), f"{(res.ids != ids_ref).sum().item()} ids differ"
    assert res.stats["target_calls"] == 5 and res.stats["forced_accepts"] == 0 and res.stats["accepted_tokens"] == as_ladder(LADDER_512).L
    err = _chunk_vs_stagewise(smp16, labels, 8, 2, tol=2e-3)
    _close(objs16)
    # fp16 cache vs fp32 cache on the same weights: the first stages' logits move by the fp16 rounding of k, v only
    smp32, objs32 = _pair(dev, 30, 30, LADDER_512, B, 1, kv_fp16=False, same=True)
    ref32 = E.Sampler(objs32[1], objs32[2]).plain_ar(labels, 3.0, 1, 0.0, E.Noise("device", 5), trace=True)
    d0 = (ref32.trace["logits"][0] - logits16[0]).abs().max().item()
    scale = ref32.trace["logits"][0].abs().max().item()
    assert 0 < d0 <= 5e-2 * scale, (d0, scale)
    _close(objs32)


def test_P1_d12_draft_d16_verify_B8(dev):
    """BASELINE.json configs[1], the configuration bench.py is quoted on: d12 draft + d16 verify, 256^2, B = 8 - determinism, reject_all ==
    plain draft AR on the consumed draws (I3/I4), run-ahead == lock-step for every gamma x threshold, chunk-verify == stage-wise logits at
    the first and last chunk, and the decoded images (HIP decoder finite and within 1e-4 of the PyTorch decoder)."""
    from sdvar_amd.vqvae import VQVAE
    B, pns = 8, LADDER_256
    smp, objs = _pair(dev, 12, 16, pns, B, 3)
    lad = smp.lad
    labels = torch.arange(B, device=dev) % 1000
    run = lambda gamma, thr, ra, seed=7: smp.spec_decode(labels, 1.5, gamma, 900, 0.96, E.Noise("device", seed), thr=thr, run_ahead=ra)
    # determinism of the benched call (gamma 2, natural threshold, run-ahead)
    a = run(2, 0.5, True); ids_a, f_a, st_a = a.ids.clone(), a.f_hat.clone(), {k: v for k, v in a.stats.items()}
    b = run(2, 0.5, True)
    assert torch.equal(b.ids, ids_a) and torch.equal(b.f_hat, f_a) and b.stats["rounds"] == st_a["rounds"]
    assert st_a["target_calls"] == 11 and st_a["draft_stage_calls"] == 12 and st_a["forced_accepts"] == 10       # what the bench's `natural` mode runs
    # run-ahead == lock-step: ids, f_hat, every counter
    for gamma in (1, 2, 3):
        for thr in (0.0, 0.5, 2.0):
            x = run(gamma, thr, False); ids_x, f_x, st_x = x.ids.clone(), x.f_hat.clone(), {k: v for k, v in x.stats.items()}
            y = run(gamma, thr, True)
            assert torch.equal(y.ids, ids_x) and torch.equal(y.f_hat, f_x), (gamma, thr)
            for k in ("target_calls", "draft_stage_calls", "forced_accepts", "accepted_tokens", "gamma_final"):
                assert y.stats[k] == st_x[k], (gamma, thr, k)
            assert y.stats["rounds"] == st_x["rounds"], (gamma, thr)
            if thr == 0.0:
                assert st_x["target_calls"] == -(-10 // gamma) and st_x["accepted_tokens"] == 680
    # reject_all: the committed ids are the plain draft AR ids on the draws of the committed stages (rollback leaves no trace)
    res = run(2, 2.0, True, seed=11)
    draws, d = [], 0
    for r in res.stats["rounds"]:
        if r["n_accept"]:
            draws.append(d)
        d += r["g"]
    m, qz = objs[0], objs[2]
    s2 = E.Sampler(m, qz)
    ids = torch.zeros_like(res.ids)
    m.begin(labels); f = torch.zeros(B, 32, 16, 16, device=dev); m.place_first(s2.x_t, 1)
    for si in range(10):
        m.forward(s2.x_t, si, 1, s2.logits_t)
        E.cfg_sample(s2.logits_t, B, lad.lens[si], 4096, lad.cfg_t(1.5, si), 900, 0.96, None, 11, draws[si], 0, ids, lad.begin(si), lad.L)
        qz.next(si, ids[:, lad.begin(si):], lad.L, f, None if si == 9 else s2.nxt[0], B)
        if si < 9:
            m.embed_next(s2.nxt[0], si + 1, s2.x_t, lad.lens[si + 1], 0)
    m.kv_set_len(0)
    assert torch.equal(ids, res.ids) and (f - res.f_hat).abs().max().item() == 0.0
    # chunk verify == stage-wise at the first and the last gamma = 2 chunk
    _chunk_vs_stagewise(smp, labels, 0, 2)
    _chunk_vs_stagewise(smp, labels, 8, 2)
    # decode: HIP decoder vs the PyTorch decoder on the sampled f_hat
    sd_v = vae_state_dict(pns, "stress", with_encoder=False)
    vae = VQVAE(vocab_size=4096, z_channels=32, ch=160, v_patch_nums=pns, with_encoder=False)
    vae.load_state_dict(sd_v); vae = vae.to(dev)
    img = vae.fhat_to_img(f_a.clone())
    ref = fhat_to_img_torch(vae, f_a.clone())
    assert torch.isfinite(img).all() and img.shape == (B, 3, 256, 256)
    assert (img - ref).abs().max().item() <= 1e-4
    _close(objs)


def test_d36_shared_aln_width(dev):
    """VAR-d36-s, the upstream shared_aln checkpoint (var.py:16-19, 81; README model zoo): C = 2304 is wider than every other model.  Shape /
    property test at B = 1 on random weights: two stages run, logits finite, deterministic, chunk == stage-wise."""
    from sdvar_amd.weights import var_state_dict
    pns, B, depth = LADDER_256, 1, 36
    sd = {k: v.to(dev) for k, v in var_state_dict(depth, pns, "stress", 5, shared_aln=True).items()}
    sd_v = vae_state_dict(pns, "stress", with_encoder=False)
    tc, qc = E.ModelCtx(sd, depth, pns, B, 2, dev), E.QuantCtx(sd_v, pns, B, dev)
    smp = E.Sampler(tc, qc)
    labels = torch.tensor([207], device=dev)
    a = smp.plain_ar(labels, 1.5, 900, 0.96, E.Noise("device", 1), trace=True)
    ids_a, lg_a = a.ids.clone(), a.trace["logits"][3].clone()
    b = smp.plain_ar(labels, 1.5, 900, 0.96, E.Noise("device", 1), trace=True)
    assert torch.equal(ids_a, b.ids) and torch.equal(lg_a, b.trace["logits"][3]) and torch.isfinite(lg_a).all() and lg_a.std().item() > 0.1
    _chunk_vs_stagewise(smp, labels, 2, 2)
    tc.close(); qc.close(); torch.cuda.empty_cache()


@pytest.mark.parametrize("kv_fp16", [False, True])
def test_qkv_epilogue_equals_qk_norm_append(dev, kv_fp16):
    """An unsplit QKV launch of a block finishes q, k AND v in its epilogue (gemm_f16x2.hip HEPI_QKV: bias, head L2 norm, q scale, k / v rows appended to the
    cache planes) and qk_norm_append does not run at all.  Same stages with the epilogue switched off (the separate qk_norm_append_rows pass): logits within 2e-5 of the logits' scale, both cache formats (two fp16 planes / one)."""
    pns, B, depth = LADDER_256, 8, 16
    lad = as_ladder(pns)
    sd = var_state_dict_device(depth, pns, dev, mode="stress")
    tc = E.ModelCtx(sd, depth, pns, B, 2, dev, kv_fp16=kv_fp16)
    assert tc.gemm_mode == "f16x2"
    labels = torch.arange(B, device=dev) % 1000
    g = torch.Generator(device="cpu").manual_seed(3)
    xs = [torch.randn(2 * B * lad.lens[s] * tc.Cw, generator=g).to(dev) for s in range(lad.S)]
    lg = torch.empty(2 * B * (lad.lens[-1] + lad.lens[-2]) * tc.V, device=dev)

    def run(fuse, chunked):
        E._check(tc.lib.sdvar_debug_set_qkv_fuse(1 if fuse else 0))
        try:
            tc.begin(labels); out = []
            for s in range(lad.S - 2):
                tc.forward(xs[s].clone(), s, 1, lg)           # the residual stream is updated in place
            if chunked:                                 # the last two stages as one gamma = 2 chunk (6800 rows)
                tc.forward(torch.cat([xs[-2].view(2 * B, -1), xs[-1].view(2 * B, -1)], 1).contiguous().view(-1), lad.S - 2, 2, lg)
                out.append(lg[:2 * B * (lad.lens[-1] + lad.lens[-2]) * tc.V].clone())
            else:
                for s in (lad.S - 2, lad.S - 1):
                    tc.forward(xs[s].clone(), s, 1, lg); out.append(lg[:2 * B * lad.lens[s] * tc.V].clone())
            return out
        finally:
            E._check(tc.lib.sdvar_debug_set_qkv_fuse(1))
    for chunked in (False, True):
        a, b = run(True, chunked), run(False, chunked)
        for x, y in zip(a, b):
            assert torch.isfinite(x).all() and (x - y).abs().max().item() <= 2e-5 * max(1.0, y.abs().max().item())
    tc.close(); torch.cuda.empty_cache()
