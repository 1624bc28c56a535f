"""GPU: the samplers around the core loop - the draft->target hand-off (sd_test3, var.py:604-865), more_smooth (var.py:206-208), the
richer acceptance rules (var.py:1229-1243), and the constructed partial-acceptance cases of the speculative loop (rollback paths of the
run-ahead scheduler).  Reference-generated fixtures where the reference runs; the CPU oracle elsewhere."""
import numpy as np
import pytest
import torch

from conftest import golden, oracle_memo, state_dicts
from oracle import var_oracle as orc
from sdvar_amd import engine as E
from sdvar_amd.ladder import LADDER_256, as_ladder
from sdvar_amd.noise import exponential_noise

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
GEMM_MODES = ["f32", "bf16x3", "f16x2"]          # fp32 MFMA and the two split-operand GEMMs: the same parity bar for all


def _noise_o(seed):
    return orc.array_noise(lambda d, B, l, V: exponential_noise(seed, d, B, l, V))


@pytest.fixture(scope="module", params=GEMM_MODES)
def pair6(dev, request):
    """d4 draft + d6 target, B = 2; the target context can take the whole pyramid in one pass (hand-off prefill)."""
    pns = LADDER_256
    sd_d, sd_v = state_dicts(4, pns); sd_t, _ = state_dicts(6, pns)
    dc, tc, qc = E.ModelCtx(sd_d, 4, pns, 2, 1, dev, gemm_mode=request.param), E.ModelCtx(sd_t, 6, pns, 2, 10, dev, gemm_mode=request.param), E.QuantCtx(sd_v, pns, 2, dev)
    yield E.Sampler(tc, qc, dc), (orc.OracleVAR(sd_d, 4, pns), orc.OracleVAR(sd_t, 6, pns), orc.OracleQuant(sd_v, pns))
    dc.close(); tc.close(); qc.close()


# ------------------------------------------------------------------------------------------------ f4: hand-off sampler
def test_handoff_masks_match_the_reference_tensors():
    """engine.handoff_mask == the oracle's restatement, which make_golden.py checked against the reference's own attn_bias_for_sdmasking /
    attn_bias_for_block tensors (var.py:557-578)."""
    lad = as_ladder(LADDER_256)
    o = orc.OracleVAR(state_dicts(2, LADDER_256)[0], 2, LADDER_256)
    for entry in (0, 3, 5, 9):
        for mk in (1, 2, 4, 5):
            assert torch.equal(E.handoff_mask(lad, entry, mk), orc.handoff_mask(o, entry, mk)[0, 0]), (entry, mk)


def test_handoff_vs_reference_fixture(dev, pair6):
    smp, _ = pair6
    g, gc = golden("sd_handoff"), golden("sd_components")
    labels = torch.from_numpy(g["labels"]).long().to(dev)
    SEED = int(g["seed"])
    for entry, mask in list(g["cases"]) + list(g["mask_cases"]):          # mask_cases: sd_mask 1, 2, 4, 5 through the explicit-mask attention
        res = smp.handoff(labels, 1.5, 900, 0.96, E.Noise("host", SEED), int(entry), int(mask))
        assert np.array_equal(res.ids.cpu().numpy(), g[f"e{entry}_m{mask}_ids"].astype(np.int64)), (entry, mask)
        np.testing.assert_allclose(res.f_hat.cpu().numpy(), g[f"e{entry}_m{mask}_f_hat"], atol=1e-4)
        assert res.stats["draft_stage_calls"] == entry and res.stats["target_calls"] == 10 - entry
    # I6 against the reference's own runs (sd_components): entry_num = 0 is the plain target AR, entry_num = S the plain draft AR
    S0 = int(gc["seed"])
    for entry, key in ((0, "handoff_target_ids"), (10, "handoff_draft_ids")):
        for mask in (0, 3):
            res = smp.handoff(labels, 1.5, 900, 0.96, E.Noise("host", S0), entry, mask)
            if entry == 0 and mask == 3:
                continue        # the prefill variant takes stage 0's logits from the input map: not the plain target AR (var.py:809-811)
            assert np.array_equal(res.ids.cpu().numpy(), gc[key].astype(np.int64)), (entry, mask)


def test_handoff_api_and_more_smooth(dev):
    """SDVAR.sdvar_autoregressive_infer_cfg_sd_test3 with the reference's signature; more_smooth through both models (early stages: the
    soft mix is ill-conditioned later on, see tests/golden/make_golden.py: smooth_fixture)."""
    import sdvar_amd
    g = golden("sd_handoff")
    vae, draft, target, sd = sdvar_amd.build_vae_var_speculative_decoding(device=dev, depth_draft=4, depth_target=6)
    for m in (draft, target):
        sdm, _ = state_dicts(m.depth, LADDER_256)
        m.load_state_dict({k: v.to(dev) for k, v in sdm.items()})
    _, sd_v = state_dicts(4, LADDER_256)
    vae.load_state_dict({k: v.to(dev) for k, v in sd_v.items()}, strict=False)
    for m in (draft, target):
        m.invalidate_engine()
    sd.noise_kind = "host"
    labels, SEED = torch.from_numpy(g["labels"]).long().to(dev), int(g["seed"])
    img = sd.sdvar_autoregressive_infer_cfg_sd_test3(B=2, label_B=labels, g_seed=SEED, cfg=1.5, top_k=900, top_p=0.96, entry_num=5, sd_mask=3)
    assert img.shape == (2, 3, 256, 256) and torch.isfinite(img).all()
    assert np.array_equal(sd.last_result.ids.cpu().numpy(), g["e5_m3_ids"].astype(np.int64))
    sd.sdvar_autoregressive_infer_cfg_sd_test3(B=2, label_B=labels, g_seed=SEED, cfg=1.5, top_k=900, top_p=0.96, entry_num=5, sd_mask=0, more_smooth=True)
    ids = sd.last_result.ids.cpu().numpy()
    assert np.array_equal(ids[:, :91], g["e5_m0_smooth_ids"].astype(np.int64)[:, :91])
    assert np.abs(sd.last_result.f_hat.cpu().numpy() - g["e5_m0_smooth_f_hat"]).max() <= 0.5
    sd.sdvar_autoregressive_infer_cfg_sd_test3(B=2, label_B=labels, g_seed=SEED, cfg=1.5, top_k=900, top_p=0.96, entry_num=5, sd_mask=4)
    assert np.array_equal(sd.last_result.ids.cpu().numpy(), g["e5_m4_ids"].astype(np.int64))
    with pytest.raises(E.SdvarError):
        sd.sdvar_autoregressive_infer_cfg_sd_test3(B=2, label_B=labels, g_seed=SEED, entry_num=5, sd_mask=7)


def test_draft_round1_reference_fixture_on_gpu(dev, pair6):
    """Round 1 of SDVAR.draft_generate_batch (var.py:949-1024), gamma = 1..3: the ids the REFERENCE drafted."""
    smp, _ = pair6
    g = golden("sd_components")
    labels, SEED = torch.from_numpy(g["labels"]).long().to(dev), int(g["seed"])
    for gamma in (1, 2, 3):
        st = smp.spec_begin(labels, 1.5, gamma, 900, 0.96, E.Noise("host", SEED))
        assert smp.spec_draft(st) == gamma
        ref = g[f"draft_round1_g{gamma}"].astype(np.int64)
        assert np.array_equal(smp.ids[:2, :ref.shape[1]].cpu().numpy(), ref), gamma
        smp.spec_commit(st, 0); smp.spec_end(st)


# ------------------------------------------------------------------------------------------------ a19: more_smooth
def test_gumbel_mix_reference_known_answers(dev):
    """sdvar_gumbel_mix against what the reference's gumbel_softmax_with_rng returned on the reference's own logits."""
    g = golden("ar_d4_256_smooth")
    pns = tuple(int(p) for p in g["patch_nums"])
    _, sd_v = state_dicts(4, pns)
    qc = E.QuantCtx(sd_v, pns, 2, dev)
    B, V, SEED = 2, 4096, int(g["g_seed"])
    for s, ntok in ((0, 1), (2, 9), (9, 4)):
        cl = torch.from_numpy(g[f"gum_s{s}_cfg"])                                  # CFG logits as the sampler received them (B, ntok, V)
        lg2 = torch.cat([cl, torch.zeros_like(cl)], 0).to(dev).contiguous()         # t = 0: the kernel's CFG is the identity
        q = torch.from_numpy(exponential_noise(SEED, s, B, pns[s] ** 2, V)[:, :ntok]).reshape(-1, V).to(dev).contiguous()
        e = torch.from_numpy(exponential_noise(SEED, s | E.GUMBEL_DRAW, B, pns[s] ** 2, V)[:, :ntok]).to(dev).contiguous()
        ids = torch.zeros(B, ntok, dtype=torch.int64, device=dev)
        masked = torch.empty(B, ntok, V, device=dev)
        E.cfg_sample(lg2, B, ntok, V, 0.0, 900, 0.96, q, SEED, s, 0, ids, 0, ntok, masked)
        h = torch.empty(B, ntok, 32, device=dev)
        ratio = s / 9
        qc.gumbel_mix(masked, B, ntok, ratio, max(0.27 * (1 - ratio * 0.95), 0.005), e, SEED, s | E.GUMBEL_DRAW, 0, h)
        np.testing.assert_allclose(h.cpu().numpy(), g[f"gum_s{s}_h"], atol=2e-5)
    # device noise: the in-kernel Philox stream at draw | GUMBEL_DRAW is the portable stream
    s, ntok = 2, 9
    cl = torch.from_numpy(g["gum_s2_cfg"])
    _, m_o = orc.sample_topk_topp(cl, 900, 0.96, torch.from_numpy(exponential_noise(SEED, s, B, 9, V)).view(-1, V))
    h_dev = torch.empty(B, ntok, 32, device=dev)
    qc.gumbel_mix(m_o.to(dev).contiguous(), B, ntok, s / 9, max(0.27 * (1 - s / 9 * 0.95), 0.005), None, SEED, s | E.GUMBEL_DRAW, 0, h_dev)
    np.testing.assert_allclose(h_dev.cpu().numpy(), g["gum_s2_h"], atol=2e-5)
    qc.close()


@pytest.mark.parametrize("gm", GEMM_MODES)
def test_more_smooth_plain_ar(dev, gm):
    """VAR.autoregressive_infer_cfg(more_smooth=True): stage by stage against the oracle fed with the HIP path's own logits (the soft mix
    feeds logit differences back x 1/tau, so a free-running comparison is only meaningful on the early stages), then the free run."""
    g = golden("ar_d4_256_smooth")
    pns = tuple(int(p) for p in g["patch_nums"])
    lad = as_ladder(pns)
    sd_var, sd_vae = state_dicts(4, pns)
    B, V, SEED = 2, 4096, int(g["g_seed"])
    ctx, qc = E.ModelCtx(sd_var, 4, pns, B, 1, dev, gemm_mode=gm), E.QuantCtx(sd_vae, pns, B, dev)
    smp = E.Sampler(ctx, qc)
    labels = torch.from_numpy(g["labels"]).long()
    res = smp.plain_ar(labels.to(dev), 1.5, 900, 0.96, E.Noise("host", SEED), trace=True, more_smooth=True)
    ids = res.ids.cpu().numpy()
    assert np.array_equal(ids[:, :91], g["ids"].astype(np.int64)[:, :91])                        # stages 0-5 of the reference's run
    assert np.abs(res.f_hat.cpu().numpy() - g["f_hat"]).max() <= 0.5 and torch.isfinite(res.f_hat).all()
    # teacher-forced: every stage's sampling + soft mix + f_hat update from the HIP logits of that stage
    oq, nfn = orc.OracleQuant(sd_vae, pns), _noise_o(SEED)
    f = torch.zeros(B, 32, 16, 16)
    for s, pn in enumerate(pns):
        cl = orc.cfg_combine(res.trace["logits"][s].cpu(), B, lad.cfg_t(1.5, s))
        ids_o, h = orc._stage_features(oq, cl, pn, 900, 0.96, nfn, s, True, s / 9, None)
        assert np.array_equal(ids_o.numpy(), ids[:, lad.begin(s):lad.cum[s]]), s
        f, _ = oq.next_input(s, f, h)
    # f accumulated from HIP logits by the oracle == the HIP f_hat (no feedback through the transformer in this comparison)
    assert (f - res.f_hat.cpu()).abs().max().item() <= 2e-3 * max(1.0, f.abs().max().item())
    ctx.close(); qc.close()


def test_more_smooth_api_flag(dev):
    import sdvar_amd
    vae, var = sdvar_amd.build_vae_var(device=dev, depth=2)
    img = var.autoregressive_infer_cfg(B=2, label_B=5, g_seed=1, cfg=1.5, top_k=900, top_p=0.96, more_smooth=True)
    ids_a = var.last_result.ids.clone()
    img2 = var.autoregressive_infer_cfg(B=2, label_B=5, g_seed=1, cfg=1.5, top_k=900, top_p=0.96, more_smooth=False)
    assert img.shape == (2, 3, 256, 256) and torch.isfinite(img).all() and not torch.equal(img, img2)
    assert torch.equal(ids_a[:, :1], var.last_result.ids[:, :1])                  # stage 0 draws are the same; later ones see a different f_hat


# ------------------------------------------------------------------------------------------------ f3: richer acceptance
@pytest.mark.parametrize("rule", [E.MatchRule(), E.MatchRule("topk", top_k=5), E.MatchRule("topk", top_k=900), E.MatchRule("kl", kl_thr=0.5)])
def test_verify_rules_vs_oracle(dev, rule):
    """sdvar_verify_accept_ex: per-token verdicts, corrected ids, match counts and n_accept of every rule against the oracle."""
    B, V, lens, ts = 2, 4096, [4, 9, 16], [0.3, 0.45, 0.6]
    rng = np.random.Generator(np.random.Philox(key=[3, rule.code + rule.top_k]))
    tl = [torch.from_numpy(rng.standard_normal(size=(2 * B, n, V), dtype=np.float32) * 2) for n in lens]
    dlg = [t + torch.from_numpy(rng.standard_normal(size=(2 * B, n, V), dtype=np.float32) * 0.6) for t, n in zip(tl, lens)]      # a "draft" near the target
    cls_t = [orc.cfg_combine(t, B, tt) for t, tt in zip(tl, ts)]
    cls_d = [orc.cfg_combine(t, B, tt) for t, tt in zip(dlg, ts)]
    ids = [c.argsort(-1, descending=True)[..., 0] for c in cls_d]                                  # the draft's own argmax
    ids[1][0, :3] = cls_t[1].argsort(-1, descending=True)[0, :3, 7]                                # a few tokens of known target rank
    orule = orc.MatchRule(rule.rule, rule.top_k, rule.kl_thr)
    n_o, matched_o, total_o, masks_o, corr_o = orc.accept_scan_ex(ids, cls_t, 0.5, orule, cls_d)
    lsum = sum(lens)
    lg = torch.cat(tl, 1).to(dev).contiguous()
    dl = torch.cat([d.reshape(-1) for d in dlg]).to(dev).contiguous()
    ids_d = torch.cat(ids, 1).to(dev).contiguous()
    counts = torch.zeros(40, dtype=torch.int32, device=dev)
    match = torch.zeros(B, lsum, dtype=torch.uint8, device=dev); corr = torch.zeros(B, lsum, dtype=torch.int64, device=dev); am = torch.zeros_like(corr)
    E.verify_accept(lg, B, lens, V, ts, ids_d, 0, lsum, 0.5, counts, argmax_out=am, rule=rule, draft_logits=dl, match_out=match, corrected_out=corr)
    c = counts.cpu().tolist()
    assert c[:3] == matched_o and c[16] == n_o and c[17:20] == total_o
    assert torch.equal(match.cpu().bool(), torch.cat(masks_o, 1))
    assert torch.equal(corr.cpu(), torch.cat(corr_o, 1))
    assert torch.equal(am.cpu(), torch.cat([c_.argmax(-1) for c_ in cls_t], 1))
    if rule.rule == "topk":
        assert 0 < sum(matched_o) < B * lsum or rule.top_k == 900


@pytest.mark.parametrize("rule,thr", [(E.MatchRule(token_level=True), 2.0), (E.MatchRule(token_level=True), 0.5), (E.MatchRule("topk", top_k=2000), 0.5),
                                      (E.MatchRule("topk", top_k=2000, token_level=True), 0.6), (E.MatchRule("kl", kl_thr=30.0), 0.5)])
@pytest.mark.parametrize("gamma", [2, 3])
def test_spec_decode_with_rules_vs_oracle(dev, pair6, rule, thr, gamma):
    smp, (od, ot, oq) = pair6
    g = golden("sd_components")
    labels, SEED = torch.from_numpy(g["labels"]).long(), int(g["seed"])
    res = smp.spec_decode(labels.to(dev), 1.5, gamma, 900, 0.96, E.Noise("host", SEED), thr=thr, match=rule)
    tr = oracle_memo(("rules", SEED, gamma, thr, rule.rule, rule.top_k, rule.kl_thr, rule.token_level), lambda: orc.spec_decode(
        od, ot, oq, labels, 1.5, gamma, 900, 0.96, _noise_o(SEED), thr=thr, match=orc.MatchRule(rule.rule, rule.top_k, rule.kl_thr, rule.token_level)))
    assert np.array_equal(res.ids.cpu().numpy(), torch.cat(tr.ids, 1).numpy())
    assert (res.f_hat.cpu() - tr.f_hat).abs().max().item() <= 1e-4
    for k in ("target_calls", "draft_stage_calls", "forced_accepts", "accepted_tokens", "gamma_final"):
        assert res.stats[k] == tr.stats[k], k
    assert res.stats.get("corrected_tokens", 0) == tr.stats.get("corrected_tokens", 0)
    assert [r["matched"] for r in res.stats["rounds"]] == [r["matched"] for r in tr.stats["rounds"]]
    if rule.token_level and thr > 1:      # I7: nothing reaches the threshold -> the target's greedy tokens, one stage per round
        sref = E.Sampler(smp.t, smp.q).plain_ar(labels.to(dev), 1.5, 1, 0.0, E.Noise("host", SEED)).ids
        assert torch.equal(res.ids, sref) and res.stats["target_calls"] == 10 and res.stats["forced_accepts"] == 0


def test_advanced_token_matching_api(dev):
    """SDVAR.advanced_token_matching: the reference's stub behaviour by default (== basic), the sketched rules when match_rule is set."""
    import sdvar_amd
    vae, draft, target, sd = sdvar_amd.build_vae_var_speculative_decoding(device=dev, depth_draft=2, depth_target=2)
    B, V = 2, 4096
    lg = [torch.randn(B, n, V, device=dev) for n in (4, 9)]
    order = [l.argsort(-1, descending=True) for l in lg]
    tk = [o[..., 0].clone() for o in order]
    tk[1] = order[1][..., 3].clone()                                   # every token of stage 1 is the target's 4th choice
    assert sd.advanced_token_matching(tk, lg, None, B) == sd.basic_token_matching(tk, lg, None, B) == 1
    sd.match_rule = E.MatchRule("topk", top_k=4)
    assert sd.advanced_token_matching(tk, lg, None, B) == 2 and bool(sd.last_match.all())
    sd.match_rule = E.MatchRule("topk", top_k=3)
    assert sd.advanced_token_matching(tk, lg, None, B) == 1
    sd.match_rule = E.MatchRule("kl", kl_thr=1e-6)
    assert sd.advanced_token_matching(tk, lg, None, B, draft_logits=[l.clone() for l in lg]) == 2       # identical distributions: KL = 0
    assert sd.advanced_token_matching(tk, lg, None, B, draft_logits=[l * 3 for l in lg]) == 0
    with pytest.raises(ValueError):
        sd.advanced_token_matching(tk, lg, None, B)


# ------------------------------------------------------------------------------------------------ partial acceptance, rollback paths
def _perturbed_target(sd, stage, lad, scale=3.0):
    """The same weights with the level embedding of one stage moved: the model agrees with the original on every stage before `stage`
    (inputs, caches and logits identical) and disagrees from there on."""
    out = {k: v.clone() for k, v in sd.items()}
    g = np.random.Generator(np.random.Philox(key=[77, stage]))
    out["lvl_embed.weight"][stage] += torch.from_numpy(g.standard_normal(size=out["lvl_embed.weight"].shape[1], dtype=np.float32) * np.float32(scale))
    return out


@pytest.mark.parametrize("gamma,stage", [(2, 3), (2, 4), (3, 3), (3, 4), (3, 7)])
def test_partial_acceptance_and_optimistic_rollback(dev, gamma, stage):
    """target = draft weights with ONE stage's level embedding perturbed, greedy sampling: rounds before that stage are accepted in full
    (so the optimistic run-ahead starts), the round containing it accepts a strict prefix or nothing - the verdict arrives after the next
    round was drafted speculatively, and Sampler._resolve has to restore the lock-step state (accepted prefix, gamma policy, draw counter,
    both KV cursors).  run_ahead == lock-step == oracle, and the rollback really happened."""
    pns = LADDER_256
    lad = as_ladder(pns)
    sd, sd_v = state_dicts(4, pns)
    sd_t = _perturbed_target(sd, stage, lad)
    B = 2
    labels = torch.tensor([11, 470])
    od, ot, oq = orc.OracleVAR(sd, 4, pns), orc.OracleVAR(sd_t, 4, pns), orc.OracleQuant(sd_v, pns)
    tr = orc.spec_decode(od, ot, oq, labels, 1.5, gamma, 1, 0.0, _noise_o(5), thr=0.5, keep=True)
    n_accs = [r["n_accept"] for r in tr.stats["rounds"]]
    gs = [r["g"] for r in tr.stats["rounds"]]
    first_bad = next(i for i, (n, g_) in enumerate(zip(n_accs, gs)) if n < g_ or tr.stats["rounds"][i]["forced"])
    assert first_bad >= 1 and all(n == g_ for n, g_ in zip(n_accs[:first_bad], gs[:first_bad])), "the construction needs fully accepted rounds first"
    # the decisions must not hinge on a near-tie of the target's top-2 logits (HIP and CPU logits differ by ~1e-5)
    for cls in tr.cfg_logits:
        for c in cls:
            top2 = c.topk(2, dim=-1)[0]
            assert float((top2[..., 0] - top2[..., 1]).min()) > 2e-4, "pick other labels: near-tie in the oracle's argmax"
    dc, tc, qc = E.ModelCtx(sd, 4, pns, B, 1, dev), E.ModelCtx(sd_t, 4, pns, B, gamma, dev), E.QuantCtx(sd_v, pns, B, dev)
    smp = E.Sampler(tc, qc, dc)
    a = smp.spec_decode(labels.to(dev), 1.5, gamma, 1, 0.0, E.Noise("host", 5), thr=0.5, run_ahead=False)
    ids_a, f_a, st_a = a.ids.cpu().clone(), a.f_hat.cpu().clone(), {k: v for k, v in a.stats.items()}
    b = smp.spec_decode(labels.to(dev), 1.5, gamma, 1, 0.0, E.Noise("host", 5), thr=0.5, run_ahead=True)
    assert np.array_equal(ids_a.numpy(), torch.cat(tr.ids, 1).numpy())
    assert torch.equal(b.ids.cpu(), ids_a) and torch.equal(b.f_hat.cpu(), f_a)
    assert (f_a - tr.f_hat).abs().max().item() <= 1e-4
    for k in ("target_calls", "draft_stage_calls", "forced_accepts", "accepted_tokens", "gamma_final"):
        assert b.stats[k] == st_a[k] == tr.stats[k], (k, b.stats[k], st_a[k], tr.stats[k])
    key = lambda rs: [(r["stage"], r["g"], r["n_accept"], r["forced"], tuple(r["matched"])) for r in rs]
    assert key(b.stats["rounds"]) == key(st_a["rounds"]) == key(tr.stats["rounds"])
    assert b.stats.get("discarded_speculative_rounds", 0) >= 1, "the optimistic path never had to roll back: the case does not test _resolve"
    partial = any(0 < r["n_accept"] < r["g"] for r in st_a["rounds"])
    assert partial == (stage % gamma != 0)        # a strict prefix is accepted exactly when the perturbed stage is not the first of its round
    dc.close(); tc.close(); qc.close()
