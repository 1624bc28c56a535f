"""GPU: the production WIDTHS against the CPU oracle (not against the HIP path itself).

tests/test_gpu_fullsize.py checks the BASELINE configurations through size-independent properties (HIP vs HIP); the reference fixtures stop
at d16 B = 1.  This file closes that gap with the pinned oracle (oracle/var_oracle.py, checked against /root/reference by
tests/golden/make_golden.py) at the widths and row counts the large-tile kernels really run at:

  * config P1 (BASELINE.json configs[1], the benched one): d12 draft + d16 verify, B = 8, gamma = 2, the speculative loop of
    models/var.py:1284-1383 - token ids bit-exact, per-round target logits <= 1e-3, f_hat <= 1e-4, every counter;
  * d24 (C = 1536) and d30 (C = 1920): stage forwards 0..4 and one gamma = 2 chunk against OracleVAR.forward (basic_var.py:90-159);
  * the VQVAE decoder at the reference width ch = 160 against oracle.decode_image (basic_vae.py:163-226).
Weights: the stress init drawn on the device, copied to the host for the oracle.  Noise: the portable Philox stream (Noise("host"))."""
import numpy as np
import pytest
import torch

from conftest import oracle_memo, rnd
from oracle import var_oracle as orc
from sdvar_amd import engine as E
from sdvar_amd.ladder import LADDER_256, as_ladder
from sdvar_amd.noise import exponential_noise
from sdvar_amd.weights import vae_state_dict, var_state_dict_device

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
LOGIT_TOL = 1e-3          # BASELINE.json north_star: "within 1e-3 on logits"
FHAT_TOL = 1e-4


def _noise_o(seed):
    return orc.array_noise(lambda d, B, l, V: exponential_noise(seed, d, B, l, V))


@pytest.fixture(scope="module")
def p1(dev):
    pns, B = LADDER_256, 8
    sd_d = var_state_dict_device(12, pns, dev, mode="stress")
    sd_t = var_state_dict_device(16, pns, dev, mode="stress")
    sd_v = vae_state_dict(pns, "stress", with_encoder=False)
    dc, tc, qc = E.ModelCtx(sd_d, 12, pns, B, 1, dev), E.ModelCtx(sd_t, 16, pns, B, 2, dev), E.QuantCtx(sd_v, pns, B, dev)
    assert tc.gemm_mode == "f16x2"                                            # the benched arithmetic
    cpu = lambda sd: {k: v.cpu() for k, v in sd.items()}
    od, ot, oq = orc.OracleVAR(cpu(sd_d), 12, pns), orc.OracleVAR(cpu(sd_t), 16, pns), orc.OracleQuant(sd_v, pns)
    yield E.Sampler(tc, qc, dc), (od, ot, oq)
    dc.close(); tc.close(); qc.close(); torch.cuda.empty_cache()


def _first_flip(ids, want, lad):
    b, t = sorted(map(tuple, np.argwhere(ids != want)), key=lambda x: x[1])[0]
    return next(i for i in range(lad.S) if t < lad.cum[i])


@pytest.mark.parametrize("mode,thr,seeds", [("accept_all", 0.0, (6,)), ("natural", 0.5, ())])
def test_P1_spec_decode_vs_oracle(dev, p1, mode, thr, seeds):
    """d12 -> d16, B = 8 (16 CFG rows: the 256-row tiles, the unsplit-QKV fused epilogue and the hybrid tail split all run), gamma = 2, against the LIVE oracle with
    the FULL per-round logit tensors (device-drawn weights).  One seed of one mode: a P1-size oracle run costs 40 - 120 s of host time, and round 3 spent 200 s of
    the suite here; the seeds x modes matrix is test_P1_spec_decode_vs_oracle_fixture below (same oracle, run ahead of time: tests/golden/make_p1_oracle.py).
    The seed passes when ids, counters, per-round logits and f_hat agree; if its ids differ they must differ FIRST at a draw the oracle itself had within 1e-3 of a
    tie (the rule of test_unselected_seeds_flip_only_on_sub_margin_ties), with the logits still in tolerance up to there."""
    if not seeds:
        pytest.skip("natural acceptance at P1 size: covered by the fixture matrix (4 seeds); the live oracle runs accept_all only")
    smp, (od, ot, oq) = p1
    lad, B, V = smp.lad, 8, 4096
    labels = (torch.arange(B) * 113 + 5) % 1000
    clean = 0
    for seed in seeds:
        tr = oracle_memo(("P1", mode, seed), lambda: orc.spec_decode(od, ot, oq, labels, 1.5, 2, 900, 0.96, _noise_o(seed), thr=thr, keep=True))
        want = torch.cat(tr.ids, 1).numpy()
        res = smp.spec_decode(labels.to(dev), 1.5, 2, 900, 0.96, E.Noise("host", seed), thr=thr, run_ahead=True)         # the benched loop
        ids_ra, f_ra, st_ra = res.ids.cpu().numpy().copy(), res.f_hat.cpu().clone(), {k: v for k, v in res.stats.items()}
        res = smp.spec_decode(labels.to(dev), 1.5, 2, 900, 0.96, E.Noise("host", seed), thr=thr, trace=True)              # lock-step, logits kept
        ids = res.ids.cpu().numpy()
        assert np.array_equal(ids, ids_ra) and torch.equal(res.f_hat.cpu(), f_ra)                                       # run-ahead == lock-step
        rounds_h, rounds_o = res.stats["rounds"], tr.stats["rounds"]
        flip_stage = None if np.array_equal(ids, want) else _first_flip(ids, want, lad)
        # per-round target logits (CFG-combined, what acceptance reads) while both paths have seen the same inputs
        for ri, (cur, g, lg) in enumerate(res.trace["target_logits"]):
            if ri >= len(rounds_o) or rounds_o[ri]["stage"] != cur or (flip_stage is not None and cur + g > flip_stage):
                break
            off, lg = 0, lg.cpu()
            for j in range(g):
                n = lad.lens[cur + j]
                cl = orc.cfg_combine(lg[:, off:off + n], B, 1.5 * ((cur + j) / (lad.S - 1))); off += n
                err = (cl - tr.cfg_logits[ri][j]).abs().max().item()
                assert err <= LOGIT_TOL, (seed, ri, cur + j, err)
        if flip_stage is None:
            clean += 1
            assert (res.f_hat.cpu() - tr.f_hat).abs().max().item() <= FHAT_TOL
            for k in ("target_calls", "draft_stage_calls", "forced_accepts", "accepted_tokens", "gamma_final"):
                assert res.stats[k] == tr.stats[k] == st_ra[k], (seed, k)
            assert [(r["stage"], r["g"], r["n_accept"], r["matched"]) for r in rounds_h] == [(r["stage"], r["g"], r["n_accept"], r["matched"]) for r in rounds_o]
            if mode == "accept_all":
                assert res.stats["target_calls"] == 5 and res.stats["accepted_tokens"] == lad.L
            continue
        # a flip: it must sit on a near-tie of the oracle's own draw.  tr.margins holds the smallest relative top-2 gap of every sampler call in draw order
        # and draw d of this loop samples the stage it was drafted for: find the draws of `flip_stage` and require one below 1e-3
        draws, d = [], 0
        for r in rounds_o:
            for j in range(r["g"]):
                if r["stage"] + j == flip_stage:
                    draws.append(d)
                d += 1
        assert draws and min(tr.margins[x] for x in draws) < 1e-3, f"seed {seed}: ids differ from stage {flip_stage} on although no draw of that stage was within 1e-3 of a tie"
    print(f"\n[P1 live oracle, {mode}] clean seeds {clean} of {len(seeds)}")


@pytest.fixture(scope="module")
def p1_host(dev):
    """The P1 pair on the PORTABLE host-stream weights tests/golden/make_p1_oracle.py used (sdvar_amd.weights.var_state_dict, 'stress')."""
    from sdvar_amd.weights import var_state_dict
    pns, B = LADDER_256, 8
    sd_d, sd_t = var_state_dict(12, pns, "stress"), var_state_dict(16, pns, "stress")
    sd_v = vae_state_dict(pns, "stress", with_encoder=False)
    dc, tc, qc = E.ModelCtx(sd_d, 12, pns, B, 1, dev), E.ModelCtx(sd_t, 16, pns, B, 2, dev), E.QuantCtx(sd_v, pns, B, dev)
    assert tc.gemm_mode == "f16x2"
    del sd_d, sd_t
    yield E.Sampler(tc, qc, dc)
    dc.close(); tc.close(); qc.close(); torch.cuda.empty_cache()


@pytest.mark.parametrize("mode,thr", [("accept_all", 0.0), ("natural", 0.5)])
def test_P1_spec_decode_vs_oracle_fixture(dev, p1_host, mode, thr):
    """Config P1 at full size in the BENCHED arithmetic against the oracle's golden vectors (tests/golden/p1_oracle.npz: 4 seeds x 2 acceptance modes): per seed the token
    ids (bit-exact, or a first flip on a draw the oracle had within 1e-3 of a tie), the per-round CFG logits of the target at the sampled columns (the 8 largest and 8
    pseudo-random entries of every token: <= 1e-3), f_hat <= 1e-4, every counter and the per-round acceptance record; run-ahead == lock-step on every seed.
    The NUMBER of fully clean seeds is asserted (>= 3 of 4) and the per-seed record is printed, so a regression that turns a clean seed into a 'tie flip' shows."""
    from conftest import golden
    g = golden("p1_oracle")
    smp = p1_host
    lad, B = smp.lad, int(g["B"])
    labels = torch.from_numpy(g["labels"]).long()
    cfg, gamma, top_k, top_p = float(g["cfg"]), int(g["gamma"]), int(g["top_k"]), float(g["top_p"])
    record, clean = [], 0
    for seed in (int(x) for x in g["seeds"]):
        k = f"{mode}_{seed}_"
        want, margins, rounds_o = g[k + "ids"].astype(np.int64), g[k + "margins"], g[k + "rounds"]
        res = smp.spec_decode(labels.to(dev), cfg, gamma, top_k, top_p, E.Noise("host", seed), thr=thr, run_ahead=True)            # the benched loop
        ids_ra, f_ra, st_ra = res.ids.cpu().numpy().copy(), res.f_hat.cpu().clone(), dict(res.stats)
        res = smp.spec_decode(labels.to(dev), cfg, gamma, top_k, top_p, E.Noise("host", seed), thr=thr, trace=True)                 # lock-step, logits kept
        ids = res.ids.cpu().numpy()
        assert np.array_equal(ids, ids_ra) and torch.equal(res.f_hat.cpu(), f_ra)
        flip_stage = None if np.array_equal(ids, want) else _first_flip(ids, want, lad)
        max_err = 0.0
        for ri, (cur, gg, lg) in enumerate(res.trace["target_logits"]):
            if ri >= len(rounds_o) or rounds_o[ri][0] != cur or (flip_stage is not None and cur + gg > flip_stage):
                break
            lg, off, cls = lg.cpu(), 0, []
            for j in range(gg):
                n = lad.lens[cur + j]
                cls.append(orc.cfg_combine(lg[:, off:off + n], B, cfg * ((cur + j) / (lad.S - 1)))); off += n
            got = torch.gather(torch.cat(cls, 1), -1, torch.from_numpy(g[k + f"r{ri}_idx"].astype(np.int64)))
            err = float((got - torch.from_numpy(g[k + f"r{ri}_val"])).abs().max())
            max_err = max(max_err, err)
            assert err <= LOGIT_TOL, (seed, ri, cur, err)
        flip_margin = None
        if flip_stage is None:
            clean += 1
            assert (res.f_hat.cpu() - torch.from_numpy(g[k + "f_hat"])).abs().max().item() <= FHAT_TOL
            cnt = [res.stats[x] for x in ("target_calls", "draft_stage_calls", "forced_accepts", "accepted_tokens", "gamma_final")]
            assert cnt == list(g[k + "counters"]) == [st_ra[x] for x in ("target_calls", "draft_stage_calls", "forced_accepts", "accepted_tokens", "gamma_final")], (seed, cnt)
            got_r = [[r["stage"], r["g"], r["n_accept"], int(r["forced"])] + list(r["matched"]) + [0] * (gamma - len(r["matched"])) for r in res.stats["rounds"]]
            assert got_r == rounds_o.tolist(), seed
        else:
            draws, d = [], 0
            for r in rounds_o:
                for j in range(int(r[1])):
                    if int(r[0]) + j == flip_stage:
                        draws.append(d)
                    d += 1
            flip_margin = float(min(margins[x] for x in draws)) if draws else None
            assert flip_margin is not None and flip_margin < 1e-3, f"seed {seed}: ids differ from stage {flip_stage} on although no draw of that stage was within 1e-3 of a tie"
        record.append(dict(seed=seed, clean=flip_stage is None, first_flip_stage=flip_stage, flip_margin=flip_margin, oracle_min_margin=float(margins.min()), max_logit_err=max_err))
    print(f"\n[P1 fixture, {mode}] clean seeds {clean} of {len(record)}: " + "; ".join(
        f"seed {r['seed']}: {'clean' if r['clean'] else 'flip at stage %d (margin %.1e)' % (r['first_flip_stage'], r['flip_margin'])}, oracle min margin {r['oracle_min_margin']:.1e}, max|dlogit| {r['max_logit_err']:.1e}"
        for r in record))
    assert clean >= 3, record


@pytest.mark.parametrize("depth", [24, 30])
def test_wide_model_stage_forward_vs_oracle(dev, depth):
    """d24 (C = 1536, 24 heads) and d30 (C = 1920, 30 heads) - widths no fixture reaches: stages 0..4 one at a time, then stages 5-6 as ONE gamma = 2 chunk
    under the block-causal rows, against OracleVAR.forward on the same inputs (B = 1: two CFG rows)."""
    pns, B = LADDER_256, 1
    lad = as_ladder(pns)
    sd = var_state_dict_device(depth, pns, dev, mode="stress")
    tc = E.ModelCtx(sd, depth, pns, B, 2, dev)
    om = orc.OracleVAR({k: v.cpu() for k, v in sd.items()}, depth, pns)
    labels = torch.tensor([371])
    cond, _, _ = om.prologue(labels)
    om.kv_reset(); tc.begin(labels.to(dev))
    C, V = 64 * depth, 4096
    lg = torch.empty(2 * B * (lad.lens[5] + lad.lens[6]) * V, device=dev)
    errs = []
    for s0, n in [(0, 1), (1, 1), (2, 1), (3, 1), (4, 1), (5, 2)]:
        lsum = sum(lad.lens[s0:s0 + n])
        x = rnd(100 * depth + s0, (2 * B, lsum, C))
        want = om.forward(x, cond, s0, n)
        tc.forward(x.to(dev).contiguous(), s0, n, lg)
        got = lg[:2 * B * lsum * V].view(2 * B, lsum, V).cpu()
        errs.append((got - want).abs().max().item())
        assert torch.isfinite(got).all()
    tc.kv_set_len(0); om.kv_reset()
    tc.close(); torch.cuda.empty_cache()
    assert max(errs) <= LOGIT_TOL, errs


def test_ladder_1024_sampler_vs_oracle(dev):
    """The 1024^2 ladder of the reference (utils/arg_util.py:249: 14 stages, L = 9451, a 64 x 64 final map - no BASELINE configuration uses it): plain AR and
    the speculative loop (gamma = 2, every stage accepted: 7 chunk verifies over up to 6400 tokens) of small models against the oracle, then the 1024^2 decode of
    the sampled f_hat (4096-token attention blocks: probabilities through the decoder's workspace) against orc.decode_image."""
    from sdvar_amd.ladder import LADDER_1024
    from sdvar_amd.weights import var_state_dict
    pns, B, seed = LADDER_1024, 1, 3
    lad = as_ladder(pns)
    assert lad.L == 9451 and lad.S == 14
    sd_d, sd_t = var_state_dict(2, pns, "stress", 1234), var_state_dict(2, pns, "stress", 4321)
    sd_v = vae_state_dict(pns, "stress", 1234, ch=32, with_encoder=False)
    dc, tc, qc = E.ModelCtx(sd_d, 2, pns, B, 1, dev), E.ModelCtx(sd_t, 2, pns, B, 2, dev), E.QuantCtx(sd_v, pns, B, dev)
    od, ot, oq = orc.OracleVAR(sd_d, 2, pns), orc.OracleVAR(sd_t, 2, pns), orc.OracleQuant(sd_v, pns)
    labels = torch.tensor([417])
    smp = E.Sampler(tc, qc, dc)
    # plain AR of the target
    res = smp.plain_ar(labels.to(dev), 1.5, 900, 0.96, E.Noise("host", seed), trace=True)
    tr = orc.plain_ar(ot, oq, labels, 1.5, 900, 0.96, _noise_o(seed), keep=True)
    ids, want = res.ids.cpu().numpy().copy(), torch.cat(tr.ids, 1).numpy()
    stop = lad.S if np.array_equal(ids, want) else _first_flip(ids, want, lad)
    if stop < lad.S:
        assert tr.margins[stop] < 1e-3, f"ids differ from stage {stop} on although its draw was not within 1e-3 of a tie"
    errs = [float((res.trace["logits"][s].cpu() - tr.logits[s]).abs().max()) for s in range(min(stop + 1, lad.S))]
    assert max(errs) <= LOGIT_TOL, errs
    assert stop >= 10, f"first flip already at stage {stop}"
    if stop == lad.S:
        assert (res.f_hat.cpu() - tr.f_hat).abs().max().item() <= FHAT_TOL
    vc = E.VaeCtx(sd_v, B, dev, latent_hw=64)
    img = vc.decode(res.f_hat.clone()).clamp(-1, 1).add(1).mul(0.5).cpu()
    vc.close()
    assert img.shape == (B, 3, 1024, 1024) and (img - orc.decode_image(sd_v, res.f_hat.cpu().clone())).abs().max().item() <= 1e-4
    # the speculative loop, every stage accepted
    res = smp.spec_decode(labels.to(dev), 1.5, 2, 900, 0.96, E.Noise("host", seed), thr=0.0, run_ahead=True)
    trs = orc.spec_decode(od, ot, oq, labels, 1.5, 2, 900, 0.96, _noise_o(seed), thr=0.0, keep=False)
    ids, want = res.ids.cpu().numpy(), torch.cat(trs.ids, 1).numpy()
    assert res.stats["target_calls"] == trs.stats["target_calls"] == 7 and res.stats["accepted_tokens"] == lad.L
    if not np.array_equal(ids, want):
        flip = _first_flip(ids, want, lad)
        assert min(trs.margins[flip:flip + 1]) < 1e-3, f"speculative ids differ from stage {flip} on without a near-tie"
    dc.close(); tc.close(); qc.close(); torch.cuda.empty_cache()


def test_decoder_reference_width_vs_oracle(dev):
    """The HIP VQVAE decoder at the reference width (ch = 160, 256^2) against the oracle's decode_image (vqvae.py:62-63, basic_vae.py:163-226, var.py:215) -
    not against MIOpen on the same GPU (tests/test_gpu_vae.py does that)."""
    pns, B = LADDER_256, 1
    sd_v = vae_state_dict(pns, "stress", with_encoder=False)
    f_hat = rnd(77, (B, 32, 16, 16), 1.5)
    want = orc.decode_image(sd_v, f_hat.clone())
    vc = E.VaeCtx(sd_v, B, dev)
    img = vc.decode(f_hat.to(dev)).clamp(-1, 1).add(1).mul(0.5).cpu()
    vc.close(); torch.cuda.empty_cache()
    assert img.shape == (B, 3, 256, 256) and torch.isfinite(img).all()
    assert (img - want).abs().max().item() <= 1e-4
    assert want.std().item() > 0.02                                              # not a saturated / constant image


def test_f16x2_guard_reports_saturation_and_nan(dev):
    """ADVICE r2 / VERDICT r2 item 9: the f16x2 operand format saturates finite activations at +-65504; with the guard on, a run that leaves that range is
    REPORTED (SampleResult.stats["f16x2_guard"]), and a NaN stays a NaN all the way to the logits instead of turning into -65504."""
    from conftest import state_dicts
    pns, B, depth = LADDER_256, 2, 2
    lad = as_ladder(pns)
    sd, sd_v = state_dicts(depth, pns)
    C = 64 * depth
    labels = torch.tensor([3, 977], device=dev)
    E.f16x2_guard(True)
    try:
        ctx, qc = E.ModelCtx(sd, depth, pns, B, 1, dev), E.QuantCtx(sd_v, pns, B, dev)
        res = E.Sampler(ctx, qc).plain_ar(labels, 1.5, 900, 0.96, E.Noise("device", 1))
        g = res.stats["f16x2_guard"]
        assert g["elements"] > 0 and g["saturated"] == 0 and g["non_finite"] == 0, g                # a healthy run is clean
        ctx.close()
        # a block whose adaLN shift is 1e5: LN(x) * (1 + scale) + shift leaves the fp16 range
        bad = {k: v.clone() for k, v in sd.items()}
        bad["blocks.0.ada_lin.1.bias"][4 * C:5 * C] = 1e5
        ctx = E.ModelCtx(bad, depth, pns, B, 1, dev)
        res = E.Sampler(ctx, qc).plain_ar(labels, 1.5, 900, 0.96, E.Noise("device", 1))
        g = res.stats["f16x2_guard"]
        assert g["saturated"] >= 2 * B * C and g["non_finite"] == 0, g
        # NaN in the residual stream: reported, and the logits are NaN (not finite garbage)
        x = torch.zeros(2 * B * lad.lens[0] * C, device=dev); x[5] = float("nan")
        lg = torch.empty(2 * B * lad.lens[0] * 4096, device=dev)
        ctx.begin(labels); ctx.forward(x, 0, 1, lg); ctx.kv_set_len(0)
        g = E.f16x2_guard_collect()
        assert g["non_finite"] > 0, g
        assert torch.isnan(lg.view(2 * B, -1)[0]).any() and torch.isfinite(lg.view(2 * B, -1)[1]).all()      # row 0 carried the NaN, row 1 did not
        ctx.close(); qc.close()
    finally:
        E.f16x2_guard(False)
    torch.cuda.empty_cache()
