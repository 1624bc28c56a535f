"""GPU: the whole sampling path through the C ABI against (1) fixtures produced by the REFERENCE, (2) the CPU oracle.
Token ids are compared bit-exactly, logits within 1e-3 (BASELINE.json north_star); f_hat within 1e-4."""
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
LOGIT_TOL = 1e-3
GEMM_MODES = ["f32", "bf16x3", "f16x2"]          # fp32 MFMA and the two split-operand GEMMs: the same parity bar for all


def _noise_o(seed):
    return orc.array_noise(lambda d, B, l, V: exponential_noise(seed, d, B, l, V))


def _flip_report(ids_hip, ids_ref, lad):
    bad = np.argwhere(ids_hip != ids_ref)
    if len(bad) == 0:
        return ""
    b, t = bad[0]
    s = next(i for i in range(lad.S) if t < lad.cum[i])
    return f"{len(bad)} ids differ; first at image {b}, token {t} (stage {s})"


@pytest.mark.parametrize("gm", GEMM_MODES)
@pytest.mark.parametrize("name", ["ar_d4_256_stress", "ar_d6_256_stress", "ar_d4_512_stress", "ar_d4_256_notopkp", "ar_d4_256_sharedaln", "ar_d4_256_nol2"])
def test_plain_ar_vs_reference_fixture(dev, name, gm):
    g = golden(name)
    depth, pns = int(g["depth"]), tuple(int(p) for p in g["patch_nums"])
    lad = as_ladder(pns)
    sd_var, sd_vae = state_dicts(depth, pns, str(g["mode"]), int(g["wseed"]), shared_aln=bool(int(g["shared_aln"])) if "shared_aln" in g else False,
                                    attn_l2_norm=bool(int(g["attn_l2_norm"])) if "attn_l2_norm" in g else True)
    B = int(g["B"])
    ctx = E.ModelCtx(sd_var, depth, pns, B, 1, dev, gemm_mode=gm); qc = E.QuantCtx(sd_vae, pns, B, dev)
    smp = E.Sampler(ctx, qc)
    labels = torch.from_numpy(g["labels"]).long().to(dev)
    res = smp.plain_ar(labels, float(g["cfg"]), int(g["top_k"]), float(g["top_p"]), E.Noise("host", int(g["g_seed"])), trace=True)
    ids = res.ids.cpu().numpy()
    # diagnostic: per-stage logits error of the HIP path against the oracle fed with the HIP path's own inputs
    tr = oracle_memo(("plain", name), lambda: orc.plain_ar(orc.OracleVAR(sd_var, depth, pns), orc.OracleQuant(sd_vae, pns), labels.cpu(), float(g["cfg"]), int(g["top_k"]),
                                                           float(g["top_p"]), _noise_o(int(g["g_seed"])), keep=True))
    errs = [float((res.trace["logits"][s].cpu() - tr.logits[s]).abs().max()) for s in range(lad.S)]
    msg = _flip_report(ids, g["ids"].astype(np.int64), lad) + f" (fixture min margin {g['min_rel_margin'].min():.1e}); per-stage max|dlogit| vs oracle {['%.1e' % e for e in errs]}"
    assert np.array_equal(ids, g["ids"].astype(np.int64)), msg
    assert max(errs) <= LOGIT_TOL, msg
    # logits of token 0 / image 0 at every stage, after CFG, vs the reference's
    for s in range(lad.S):
        lg = res.trace["logits"][s].cpu()
        cl = orc.cfg_combine(lg, B, lad.cfg_t(float(g["cfg"]), s))
        assert (cl[0, 0].numpy() - g["cfg_logits_rows"][s][0]).__abs__().max() <= LOGIT_TOL, s
    np.testing.assert_allclose(res.f_hat.cpu().numpy(), g["f_hat"], atol=1e-4)
    ctx.close(); qc.close()


def test_resumed_sampler_helper1_vs_reference_fixture(dev):
    """VAR.autoregressive_infer_cfg_sd_helper1 through the module mirror (sdvar_amd.var.VAR) -> engine.Sampler.resume_ar -> sdvar_model_begin_cond /
    sdvar_kv_set_origin / sdvar_embed_next_at, against the reference's own three chained calls (tests/golden/make_golden.py helper1): ids bit-exact,
    CFG logits <= 1e-3, the next map and f_hat handed from call to call, the history shapes of var.py:436-443 and the in-place f_hat aliasing."""
    from sdvar_amd.var import build_vae_var
    g = golden("ar_d4_256_helper1")
    pns = tuple(int(p) for p in g["patch_nums"])
    lad = as_ladder(pns)
    sd_var, sd_vae = state_dicts(4, pns, "stress", int(g["wseed"]))
    vae, var = build_vae_var(dev, patch_nums=pns, depth=4)
    sd_full = dict(vae.state_dict()); sd_full.update(sd_vae)
    vae.load_state_dict(sd_full, strict=True); var.load_state_dict(sd_var, strict=True)
    B = int(g["B"])
    om = orc.OracleVAR(sd_var, 4, pns)
    cond, lvl_pos, _ = om.prologue(torch.from_numpy(g["labels"]).long())
    f = torch.zeros(B, 32, pns[-1], pns[-1], device=dev); nm = None
    for ci, (cs, st) in enumerate(g["plan"].tolist()):
        inp, fh, lgs, ids = var.autoregressive_infer_cfg_sd_helper1(B, cs, st, nm, f, E.Noise("host", int(g["g_seed"])), cond.to(dev), lvl_pos.to(dev),
                                                                    cfg=float(g["cfg"]), top_k=int(g["top_k"]), top_p=float(g["top_p"]))
        assert len(fh) == st + 1 and all(x is f for x in fh)                                  # one aliased tensor, updated in place
        assert len(lgs) == len(ids) == st and len(inp) == len(g[f"c{ci}_inputs_digest"])
        got = torch.cat(ids, 1).cpu().numpy()
        assert np.array_equal(got, g[f"c{ci}_ids"].astype(np.int64)), (ci, _flip_report(got, g[f"c{ci}_ids"].astype(np.int64), lad) if cs == 0 else "")
        for k, x in enumerate(lgs):
            assert x.shape == (B, lad.lens[cs + k], 4096)
            row, want = x[0, 0].cpu().numpy(), g[f"c{ci}_logits_row0"][k]                   # masked in place by the sampler, as helpers.py:10,15 do
            fin = np.isfinite(want)
            assert (np.isfinite(row) != fin).sum() <= 2 and np.abs(row[fin & np.isfinite(row)] - want[fin & np.isfinite(row)]).max() <= LOGIT_TOL, (ci, k)
            assert abs(int(torch.isfinite(x).sum()) - int(g[f"c{ci}_n_keep"][k])) <= 2 * x.shape[0] * x.shape[1]      # a near-tie at the top-p edge may move one entry per token
        for k, x in enumerate(inp[:-1]):
            si = cs + k + (1 if cs == 0 else 0)
            assert x.shape == (B, lad.lens[si], 32)
        np.testing.assert_allclose(inp[-1].cpu().numpy(), g[f"c{ci}_next_map"], atol=1e-4)
        np.testing.assert_allclose(f.cpu().numpy(), g[f"c{ci}_f_hat"], atol=1e-4)
        nm = inp[-1]
    var.invalidate_engine()


@pytest.mark.parametrize("gm", GEMM_MODES)
def test_d16_b1_vs_reference_fixture(dev, gm):
    g = golden("ar_d16_256_stress_B1")
    pns = tuple(int(p) for p in g["patch_nums"])
    sd_var, sd_vae = state_dicts(16, pns, "stress", int(g["wseed"]))
    ctx = E.ModelCtx(sd_var, 16, pns, 1, 1, dev, gemm_mode=gm); qc = E.QuantCtx(sd_vae, pns, 1, dev)
    res = E.Sampler(ctx, qc).plain_ar(torch.from_numpy(g["labels"]).long().to(dev), 1.5, 900, 0.96, E.Noise("host", int(g["g_seed"])), trace=True)
    ids = res.ids.cpu().numpy()
    assert np.array_equal(ids, g["ids"].astype(np.int64)), _flip_report(ids, g["ids"].astype(np.int64), as_ladder(pns))
    lg = res.trace["logits"][-1].cpu()
    cl = orc.cfg_combine(lg, 1, 1.5)
    assert np.abs(cl[:, :2].numpy() - g["cfg_logits_last_rows"]).max() <= LOGIT_TOL
    np.testing.assert_allclose(res.f_hat.cpu().numpy(), g["f_hat"], atol=1e-4)
    ctx.close(); qc.close()


@pytest.fixture(scope="module", params=GEMM_MODES)
def pair(dev, request):
    pns = LADDER_256
    sd_d, sd_v = state_dicts(4, pns); sd_t, _ = state_dicts(6, pns)
    B = 2
    gm = request.param
    dc, tc, qc = E.ModelCtx(sd_d, 4, pns, B, 1, dev, gemm_mode=gm), E.ModelCtx(sd_t, 6, pns, B, 3, dev, gemm_mode=gm), E.QuantCtx(sd_v, pns, B, dev)
    od, ot, oq = orc.OracleVAR(sd_d, 4, pns), orc.OracleVAR(sd_t, 6, pns), orc.OracleQuant(sd_v, pns)
    yield E.Sampler(tc, qc, dc), (od, ot, oq)
    dc.close(); tc.close(); qc.close()


def test_chunk_verify_logits_vs_reference_fixture(dev, pair):
    """One target forward over a gamma-chunk under the block-causal rows == the reference modules' own chunk forward
    (fixture rows from make_golden.sd_fixture, I2)."""
    smp, (od, ot, oq) = pair
    g = golden("sd_components")
    labels = torch.from_numpy(g["labels"]).long()
    tr = orc.plain_ar(ot, oq, labels, 1.5, 900, 0.96, _noise_o(int(g["seed"])), keep=True)
    t, lad, B, V = smp.t, smp.lad, 2, 4096
    for (s0, n) in ((3, 2), (5, 3), (0, 2), (8, 2)):
        t.begin(labels.to(dev))
        for s in range(s0):
            x = tr.x_in[s].to(dev).contiguous()
            t.forward(x, s, 1, smp.logits_t)
        x = torch.cat(tr.x_in[s0:s0 + n], 1).to(dev).contiguous()
        lsum = x.shape[1]
        t.forward(x, s0, n, smp.logits_t)
        lg = smp.logits_t[:2 * B * lsum * V].view(2 * B, lsum, V).cpu()
        assert np.abs(lg[0, -1].numpy() - g[f"chunk_{s0}_{n}_row"]).max() <= LOGIT_TOL
        want = torch.cat(tr.logits[s0:s0 + n], 1)
        assert (lg - want).abs().max().item() <= LOGIT_TOL
        t.kv_set_len(0)


@pytest.mark.parametrize("mode,thr", [("natural", 0.5), ("accept_all", 0.0), ("reject_all", 2.0)])
@pytest.mark.parametrize("gamma", [1, 2, 3])
def test_spec_decode_vs_oracle(dev, pair, mode, thr, gamma):
    smp, (od, ot, oq) = pair
    g = golden("sd_components")
    labels = torch.from_numpy(g["labels"]).long()
    SEED = int(g["seed"])
    res = smp.spec_decode(labels.to(dev), 1.5, gamma, 900, 0.96, E.Noise("host", SEED), thr=thr)
    want_ids = g[f"spec_{mode}_g{gamma}_ids"].astype(np.int64)
    ids = res.ids.cpu().numpy()
    assert np.array_equal(ids, want_ids), _flip_report(ids, want_ids, smp.lad)
    st = res.stats
    assert [st["target_calls"], st["draft_stage_calls"], st["forced_accepts"], st["accepted_tokens"]] == list(g[f"spec_{mode}_g{gamma}_stats"])
    tr = oracle_memo(("spec", SEED, gamma, thr), lambda: orc.spec_decode(od, ot, oq, labels, 1.5, gamma, 900, 0.96, _noise_o(SEED), thr=thr))
    assert (res.f_hat.cpu() - tr.f_hat).abs().max().item() <= 1e-4
    assert [r["n_accept"] for r in st["rounds"]] == [r["n_accept"] for r in tr.stats["rounds"]]
    assert [r["matched"] for r in st["rounds"]] == [r["matched"] for r in tr.stats["rounds"]]


def test_spec_invariants_identical_models_and_rollback(dev):
    """I1: draft == target weights, top_k = 1 -> every stage accepted, ids == plain AR, ceil(S/gamma) target calls.
    I4: a rejected round leaves no trace: reject_all ids == plain draft AR with the same draw sequence (I3)."""
    pns = LADDER_256
    sd, sd_v = state_dicts(4, pns)
    B = 2
    a, b, qc = E.ModelCtx(sd, 4, pns, B, 1, dev), E.ModelCtx(sd, 4, pns, B, 3, dev), E.QuantCtx(sd_v, pns, B, dev)
    smp = E.Sampler(b, qc, a)
    labels = torch.tensor([1, 2], device=dev)
    ref = smp.plain_ar(labels, 1.5, 1, 0.0, E.Noise("host", 5)).ids.clone()
    for gamma in (1, 2, 3):
        res = smp.spec_decode(labels, 1.5, gamma, 1, 0.0, E.Noise("host", 5))
        assert torch.equal(res.ids, ref)
        assert res.stats["target_calls"] == -(-10 // gamma) and res.stats["forced_accepts"] == 0 and res.stats["accepted_tokens"] == 680
    # reject_all with sampling noise: forced accepts only; the draws consumed are 0..n in order
    res = smp.spec_decode(labels, 1.5, 2, 900, 0.96, E.Noise("host", 5), thr=2.0)
    st = res.stats
    assert st["forced_accepts"] == 10 and st["accepted_tokens"] == 0 and st["gamma_final"] == 1
    draws = []                                                                  # draw index used by each committed stage
    d = 0
    for r in st["rounds"]:
        if r["n_accept"]:
            draws.append(d)
        d += r["g"]
    od, oq = orc.OracleVAR(sd, 4, pns), orc.OracleQuant(sd_v, pns)
    seq = iter(draws)
    tr = orc.plain_ar(od, oq, labels.cpu(), 1.5, 900, 0.96, orc.array_noise(lambda dd, B_, l, V: exponential_noise(5, draws[dd], B_, l, V)), keep=False)
    assert np.array_equal(res.ids.cpu().numpy(), torch.cat(tr.ids, 1).numpy())
    a.close(); b.close(); qc.close()


def test_device_noise_run_is_reproduced_by_oracle_with_dumped_noise(dev, pair):
    """Fast path: Philox generated inside the sampler kernel.  The oracle is fed the same stream dumped by
    sdvar_op_noise_fill, so the in-kernel generator is tied to the portable stream definition."""
    import ctypes as C
    smp, (od, ot, oq) = pair
    lib = E.load_library()
    labels = torch.tensor([7, 500])
    res = smp.spec_decode(labels.to(dev), 1.5, 2, 900, 0.96, E.Noise("device", 1234), thr=0.0)
    def dumped(draw, B, l, V):
        q = torch.empty(B * l, V, device=dev)
        E._check(lib.sdvar_op_noise_fill(C.c_void_p(q.data_ptr()), B, l, V, 1234, draw, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return q.cpu()
    tr = orc.spec_decode(od, ot, oq, labels, 1.5, 2, 900, 0.96, dumped, thr=0.0)
    assert np.array_equal(res.ids.cpu().numpy(), torch.cat(tr.ids, 1).numpy())


def test_api_surface_drop_in(dev):
    """models.build_vae_var_speculative_decoding / VAR.autoregressive_infer_cfg / SDVAR...parallel_v1 signatures."""
    import sdvar_amd
    vae, draft, target, sd = sdvar_amd.build_vae_var_speculative_decoding(device=dev, depth_draft=2, depth_target=4)
    assert sd.draft_model is draft and sd.target_model is target and target.vae_proxy[0] is vae
    assert target.patch_nums == LADDER_256 and target.L == 680 and target.num_stages_minus_1 == 9 and target.begin_ends[1] == (1, 5)
    img = target.autoregressive_infer_cfg(B=2, label_B=torch.tensor([1, 2], device=dev), g_seed=0, cfg=1.5, top_k=900, top_p=0.96)
    assert img.shape == (2, 3, 256, 256) and img.min().item() >= 0 and img.max().item() <= 1 and torch.isfinite(img).all()
    ids1 = target.last_result.ids.clone()
    img2 = target.autoregressive_infer_cfg(B=2, label_B=torch.tensor([1, 2], device=dev), g_seed=0, cfg=1.5, top_k=900, top_p=0.96)
    assert torch.equal(ids1, target.last_result.ids)                                                 # same seed -> same tokens
    assert (img - img2).abs().max().item() <= 1e-4                                                   # MIOpen conv algorithms are not bitwise repeatable
    img3 = sd.sdvar_autoregressive_infer_cfg_parallel_v1(B=2, label_B=3, g_seed=1, cfg=1.5, gamma=2, top_k=900, top_p=0.96)
    assert img3.shape == (2, 3, 256, 256) and torch.isfinite(img3).all()
    assert sd.last_result.stats["target_calls"] >= 5
    for blk in target.blocks: blk.attn.kv_caching(False)
    img4 = draft.autoregressive_infer_cfg(B=1, label_B=None, g_seed=3)
    assert img4.shape == (1, 3, 256, 256)


def test_engine_objects_grow_with_the_batch(dev):
    """One VAR / SDVAR object called with a small batch first and a larger one later: the model context AND the quantizer context are
    rebuilt (the quantizer used to keep its first max_batch and fail with 'quant_next: stage 0 B ...'); ids equal a fresh object's."""
    import sdvar_amd
    vae, draft, target, sd = sdvar_amd.build_vae_var_speculative_decoding(device=dev, depth_draft=2, depth_target=4)
    lab = lambda B: torch.arange(B, device=dev) + 3
    target.autoregressive_infer_cfg(B=1, label_B=lab(1), g_seed=0, cfg=1.5, top_k=900, top_p=0.96)
    target.autoregressive_infer_cfg(B=4, label_B=lab(4), g_seed=0, cfg=1.5, top_k=900, top_p=0.96)
    ids_grown = target.last_result.ids.clone()
    sd.sdvar_autoregressive_infer_cfg_parallel_v1(B=2, label_B=lab(2), g_seed=1, cfg=1.5, gamma=2, top_k=900, top_p=0.96)
    sd.sdvar_autoregressive_infer_cfg_parallel_v1(B=6, label_B=lab(6), g_seed=1, cfg=1.5, gamma=2, top_k=900, top_p=0.96)
    ids_sd = sd.last_result.ids.clone()
    target.autoregressive_infer_cfg(B=3, label_B=lab(3), g_seed=0, cfg=1.5, top_k=900, top_p=0.96)      # back to a smaller batch on the grown objects
    vae2, draft2, target2, sd2 = sdvar_amd.build_vae_var_speculative_decoding(device=dev, depth_draft=2, depth_target=4)
    target2.autoregressive_infer_cfg(B=4, label_B=lab(4), g_seed=0, cfg=1.5, top_k=900, top_p=0.96)
    assert torch.equal(ids_grown, target2.last_result.ids)
    sd2.sdvar_autoregressive_infer_cfg_parallel_v1(B=6, label_B=lab(6), g_seed=1, cfg=1.5, gamma=2, top_k=900, top_p=0.96)
    assert torch.equal(ids_sd, sd2.last_result.ids)


def test_public_api_runs_the_benched_loop(dev):
    """SDVAR.sdvar_autoregressive_infer_cfg_parallel_v1 IS Sampler.spec_decode(run_ahead=True) (what bench.py times): its ids, f_hat and
    counters equal the lock-step loop's and the step-wise helper methods', for an accepting and a rejecting configuration."""
    import sdvar_amd
    vae, draft, target, sd = sdvar_amd.build_vae_var_speculative_decoding(device=dev, depth_draft=2, depth_target=4)
    for m in (draft, target):
        sdm, _ = state_dicts(m.depth, LADDER_256)
        m.load_state_dict({k: v.to(dev) for k, v in sdm.items()})
    B, labels = 2, torch.tensor([11, 700], device=dev)
    for thr, gamma in ((0.5, 2), (0.0, 3), (2.0, 3)):
        sd.match_threshold = thr
        sd.sdvar_autoregressive_infer_cfg_parallel_v1(B=B, label_B=labels, g_seed=9, cfg=1.5, gamma=gamma, top_k=900, top_p=0.96, more_smooth=True)   # flag is inert here, as in the reference
        res = sd.last_result
        ids, f, st = res.ids.clone(), res.f_hat.clone(), {k: v for k, v in res.stats.items()}
        state = sd._initialize_inference_state(B, labels, 9, 1.5, gamma)
        state.top_k, state.top_p = 900, 0.96
        while state.current_stage < state.total_stages:                            # the reference's loop body (var.py:1318-1367)
            toks = sd.draft_generate_batch(state, B)
            logits, g = sd.target_verify_batch(toks, state, B)
            n = sd.basic_token_matching(toks, logits, state, B)
            if n == 0:
                if state.gamma > 1:
                    state.gamma -= 1
                else:
                    n = 1
            sd.update_state_with_accepted_tokens(toks, n, state, B)
        state.sampler.spec_end(state)
        assert torch.equal(state.sampler.ids[:B], ids) and torch.equal(state.draft_f_hat, f), (thr, gamma)
        assert state.target_calls == st["target_calls"]


def test_fp16_kv_cache_vs_oracle(dev):
    """BASELINE config P4's cache format on a small model: HIP with kv_fp16 == the oracle that rounds k, v to fp16 at the
    append (ids bit-exact, logits 1e-3)."""
    from sdvar_amd.ladder import LADDER_512
    pns, depth, B = LADDER_512, 4, 1
    lad = as_ladder(pns)
    sd, sd_v = state_dicts(depth, pns)
    ctx = E.ModelCtx(sd, depth, pns, B, 1, dev, kv_fp16=True); qc = E.QuantCtx(sd_v, pns, B, dev)
    labels = torch.tensor([417])
    res = E.Sampler(ctx, qc).plain_ar(labels.to(dev), 3.0, 900, 0.96, E.Noise("host", 1), trace=True)
    tr = orc.plain_ar(orc.OracleVAR(sd, depth, pns, kv_fp16=True), orc.OracleQuant(sd_v, pns), labels, 3.0, 900, 0.96, _noise_o(1), keep=True)
    errs = [float((res.trace["logits"][s].cpu() - tr.logits[s]).abs().max()) for s in range(lad.S)]
    assert np.array_equal(res.ids.cpu().numpy(), torch.cat(tr.ids, 1).numpy()), errs
    assert max(errs) <= LOGIT_TOL, errs
    ctx.close(); qc.close()


def test_sdvar_helper_methods_follow_the_reference_call_sequence(dev):
    """The reference's own step functions (var.py:871-1282), driven exactly like parallel_v1 drives them, reproduce the
    one-call sampler; init_param returns the prologue tensors of var.py:580-601."""
    import sdvar_amd
    vae, draft, target, sd = sdvar_amd.build_vae_var_speculative_decoding(device=dev, depth_draft=2, depth_target=4)
    for m in (draft, target):                                                   # stress weights so the blocks matter
        sdm, _ = state_dicts(m.depth, LADDER_256)
        m.load_state_dict({k: v.to(dev) for k, v in sdm.items()})
    B, labels = 2, torch.tensor([11, 700], device=dev)
    sd.noise_kind = "host"
    sd.match_threshold = 0.0                                                    # accept everything: exercises multi-stage commits
    img = sd.sdvar_autoregressive_infer_cfg_parallel_v1(B=B, label_B=labels, g_seed=9, cfg=1.5, gamma=3, top_k=900, top_p=0.96)
    ids_ref, st_ref = sd.last_result.ids.clone(), dict(sd.last_result.stats)
    # the same run, step by step
    state = sd._initialize_inference_state(B, labels, 9, 1.5, 3)
    state.top_k, state.top_p = 900, 0.96
    calls = 0
    while state.current_stage < state.total_stages:
        toks = sd.draft_generate_batch(state, B)
        logits, g = sd.target_verify_batch(toks, state, B)
        assert g == len(toks) and logits[0].shape == (B, toks[0].shape[1], 4096)
        n = sd.basic_token_matching(toks, logits, state, B)
        assert n == g                                                           # thr = 0
        sd.update_state_with_accepted_tokens(toks, n, state, B)
        calls += 1
    state.sampler.spec_end(state)
    assert calls == st_ref["target_calls"] == 4 and state.accept_count == 10 and state.target_calls == 4
    assert torch.equal(state.sampler.ids[:B], ids_ref)
    assert torch.equal(state.draft_f_hat, sd.last_result.f_hat)
    # known-answer behaviour of basic_token_matching on explicit tensors: 100 % / 50 % / 49 % match -> 2 stages
    V = 4096
    lg = [torch.randn(B, n, V, device=dev) for n in (4, 9, 16)]
    tk = [l.argmax(-1) for l in lg]
    tk[1].view(-1)[9:] = (tk[1].view(-1)[9:] + 1) % V                           # 9 of 18 match = 0.5 -> accepted
    tk[2].view(-1)[15:] = (tk[2].view(-1)[15:] + 1) % V                         # 15 of 32 < 0.5 -> rejected
    sd.match_threshold = 0.5
    assert sd.basic_token_matching(tk, lg, None, B) == 2
    # init_param (var.py:580-601)
    sos, cond, cond2, lvl_pos, first, f0 = sd.init_param(target, B, labels)
    o = orc.OracleVAR({k: v.cpu() for k, v in target.state_dict().items()}, target.depth, LADDER_256)
    c_o, lp_o, f_o = o.prologue(labels.cpu())
    assert torch.equal(cond.cpu(), c_o) and (lvl_pos.cpu() - lp_o).abs().max() == 0 and (first.cpu() - f_o).abs().max() <= 1e-6
    assert f0.shape == (B, 32, 16, 16) and f0.abs().sum() == 0


def test_two_samplers_on_two_host_threads_and_streams(dev):
    """Thread contract of include/sdvar_hip.h: different host threads may drive different model objects on different streams at the
    same time (per-thread split-K workspaces).  Two speculative samplers + decoders run concurrently, repeatedly; each must reproduce
    the ids and images it produces when run alone."""
    import threading
    pns = LADDER_256
    sd_d, sd_v = state_dicts(2, pns, "stress", 7)
    sd_t, _ = state_dicts(4, pns, "stress", 7)
    sd_v = {k: v for k, v in sd_v.items()}
    from sdvar_amd.weights import vae_state_dict
    sd_dec = vae_state_dict(pns, "stress", 7, ch=32, with_encoder=False)
    B = 2
    jobs = []
    for j in range(2):
        dc, tc, qc = E.ModelCtx(sd_d, 2, pns, B, 1, dev, gemm_mode="bf16x3"), E.ModelCtx(sd_t, 4, pns, B, 2, dev, gemm_mode="bf16x3"), E.QuantCtx(sd_v, pns, B, dev)
        jobs.append(dict(smp=E.Sampler(tc, qc, dc), dec=E.VaeCtx(sd_dec, B, dev), labels=torch.tensor([5 + j, 700 + j]).to(dev), seed=11 + j,
                         stream=torch.cuda.Stream(device=dev), out=[]))

    def run(job, reps):
        with torch.cuda.stream(job["stream"]):
            for _ in range(reps):
                res = job["smp"].spec_decode(job["labels"], 1.5, 2, 900, 0.96, E.Noise("device", job["seed"]), thr=0.5)
                img = job["dec"].decode(res.f_hat)
                job["stream"].synchronize()
                job["out"].append((res.ids.cpu().clone(), img.cpu().clone()))

    for job in jobs:                       # alone
        run(job, 1)
    alone = [job["out"].pop() for job in jobs]
    ts = [threading.Thread(target=run, args=(job, 4)) for job in jobs]
    for t in ts: t.start()
    for t in ts: t.join()
    for job, (ids0, img0) in zip(jobs, alone):
        assert len(job["out"]) == 4
        for ids, img in job["out"]:
            assert torch.equal(ids, ids0)
            assert torch.equal(img, img0)


@pytest.mark.parametrize("gamma,thr", [(1, 0.5), (2, 0.5), (3, 2.0), (2, 0.0)])
def test_run_ahead_equals_lock_step(dev, pair, gamma, thr):
    """The verifier may run one round behind the draft once gamma == 1 (engine.Sampler._spec_run_ahead): ids, f_hat and every counter
    equal the lock-step loop's."""
    smp, _ = pair
    labels = torch.tensor([11, 470]).to(dev)
    a = smp.spec_decode(labels, 1.5, gamma, 900, 0.96, E.Noise("device", 5), thr=thr, run_ahead=False)
    ids_a, f_a, st_a = a.ids.cpu().clone(), a.f_hat.cpu().clone(), dict(a.stats)
    b = smp.spec_decode(labels, 1.5, gamma, 900, 0.96, E.Noise("device", 5), thr=thr, run_ahead=True)
    assert torch.equal(b.ids.cpu(), ids_a) and torch.equal(b.f_hat.cpu(), f_a)
    for k in ("target_calls", "draft_stage_calls", "forced_accepts", "accepted_tokens", "gamma_final"):
        assert b.stats[k] == st_a[k], k
    assert b.stats["rounds"] == st_a["rounds"]


def test_shared_aln_module_api(dev):
    """VAR(shared_aln=True) (SharedAdaLin checkpoints such as VAR-d36-s, var.py:16-19, 81): the module container binds shared_ada_lin +
    blocks.i.ada_gss and samples the ids of the reference fixture."""
    from sdvar_amd.var import VAR
    from sdvar_amd.vqvae import VQVAE
    g = golden("ar_d4_256_sharedaln")
    pns = tuple(int(p) for p in g["patch_nums"])
    sd_var, sd_vae = state_dicts(4, pns, str(g["mode"]), int(g["wseed"]), shared_aln=True)
    vae = VQVAE(vocab_size=4096, z_channels=32, ch=160, v_patch_nums=pns, with_encoder=False)
    vae.load_state_dict(sd_vae)
    var = VAR(vae_local=vae, depth=4, embed_dim=256, num_heads=4, shared_aln=True, attn_l2_norm=True, patch_nums=pns)
    var.load_state_dict(sd_var)
    vae, var = vae.to(dev), var.to(dev)
    var.noise_kind = "host"
    img = var.autoregressive_infer_cfg(B=int(g["B"]), label_B=torch.from_numpy(g["labels"]).long().to(dev), g_seed=int(g["g_seed"]), cfg=float(g["cfg"]),
                                       top_k=int(g["top_k"]), top_p=float(g["top_p"]))
    assert np.array_equal(var.last_result.ids.cpu().numpy(), g["ids"].astype(np.int64))
    assert img.shape == (int(g["B"]), 3, 256, 256) and float(img.min()) >= 0.0 and float(img.max()) <= 1.0


def test_run_ahead_partial_batch_and_repeated_calls(dev, pair):
    """Fewer images than max_batch, alternating gammas and thresholds on ONE sampler object: the run-ahead / optimistic paths leave no state
    behind (slots, counter rows, second stream) - every call equals the lock-step result of the same call."""
    smp, _ = pair
    labels = torch.tensor([901]).to(dev)                                         # B = 1 on a max_batch = 2 sampler
    for gamma, thr, seed in [(2, 0.0, 3), (1, 0.5, 4), (3, 0.0, 5), (2, 2.0, 6), (2, 0.0, 3)]:
        a = smp.spec_decode(labels, 1.5, gamma, 900, 0.96, E.Noise("device", seed), thr=thr, run_ahead=False)
        ids_a, st_a = a.ids.cpu().clone(), dict(a.stats)
        b = smp.spec_decode(labels, 1.5, gamma, 900, 0.96, E.Noise("device", seed), thr=thr, run_ahead=True)
        assert torch.equal(b.ids.cpu(), ids_a), (gamma, thr)
        assert b.stats["rounds"] == st_a["rounds"] and b.stats["target_calls"] == st_a["target_calls"] and b.stats["draft_stage_calls"] == st_a["draft_stage_calls"]


def test_attn_l2_norm_false_module_api(dev):
    """build_vae_var(attn_l2_norm=False) (basic_var.py:66-72: no scale_mul parameter, softmax scale 0.25 / sqrt(64)) samples the ids of the
    reference fixture through the module container."""
    import sdvar_amd
    g = golden("ar_d4_256_nol2")
    pns = tuple(int(p) for p in g["patch_nums"])
    sd_var, sd_vae = state_dicts(4, pns, str(g["mode"]), int(g["wseed"]), attn_l2_norm=False)
    vae, var = sdvar_amd.build_vae_var(device=dev, depth=4, attn_l2_norm=False)
    assert not any("scale_mul" in k for k in var.state_dict())
    var.load_state_dict({k: v.to(dev) for k, v in sd_var.items()})
    vae.load_state_dict({k: v.to(dev) for k, v in sd_vae.items()}, strict=False)
    var.noise_kind = "host"
    var.autoregressive_infer_cfg(B=int(g["B"]), label_B=torch.from_numpy(g["labels"]).long().to(dev), g_seed=int(g["g_seed"]), cfg=float(g["cfg"]),
                                 top_k=int(g["top_k"]), top_p=float(g["top_p"]))
    assert np.array_equal(var.last_result.ids.cpu().numpy(), g["ids"].astype(np.int64))


@pytest.mark.parametrize("gm", GEMM_MODES)
def test_unselected_seeds_flip_only_on_sub_margin_ties(dev, gm):
    """The fixtures are tie-free by seed search (make_golden.py MIN_MARGIN).  On seeds nobody selected the HIP path may legitimately
    differ from the CPU oracle where a draw is a near-tie - and ONLY there: for each of a few arbitrary seeds, either all ids agree, or the
    FIRST differing token (everything after it sees different inputs) is one whose two best p/q ratios were within 1e-3 of each other
    in the oracle, and the logits up to that stage still agree to the 1e-3 tolerance."""
    pns = LADDER_256
    lad = as_ladder(pns)
    sd_var, sd_vae = state_dicts(4, pns)
    B, V = 2, 4096
    ctx, qc = E.ModelCtx(sd_var, 4, pns, B, 1, dev, gemm_mode=gm), E.QuantCtx(sd_vae, pns, B, dev)
    smp = E.Sampler(ctx, qc)
    od, oq = orc.OracleVAR(sd_var, 4, pns), orc.OracleQuant(sd_vae, pns)
    labels = torch.tensor([17, 402])
    flips = 0
    for seed in (1001, 1002, 1003, 1004):
        res = smp.plain_ar(labels.to(dev), 1.5, 900, 0.96, E.Noise("host", seed), trace=True)
        tr = oracle_memo(("unselected", seed), lambda: orc.plain_ar(od, oq, labels, 1.5, 900, 0.96, _noise_o(seed), keep=True))
        ids, want = res.ids.cpu().numpy(), torch.cat(tr.ids, 1).numpy()
        if np.array_equal(ids, want):
            continue
        flips += 1
        b, t = sorted(map(tuple, np.argwhere(ids != want)), key=lambda x: x[1])[0]          # the earliest differing token position
        s = next(i for i in range(lad.S) if t < lad.cum[i])
        for s2 in range(s + 1):                                    # up to and including that stage both paths saw the same inputs
            assert float((res.trace["logits"][s2].cpu() - tr.logits[s2]).abs().max()) <= LOGIT_TOL, (seed, s2)
        # every differing token OF THAT STAGE must be a near-tie of the oracle's draw
        q = _noise_o(seed)(s, B, lad.lens[s], V).view(B, lad.lens[s], V)
        _, masked = orc.sample_topk_topp(tr.cfg_logits[s], 900, 0.96, q.reshape(-1, V))
        top2 = (masked.softmax(-1) / q).topk(2, dim=-1)[0]
        margin = (top2[..., 0] - top2[..., 1]) / top2[..., 0]
        l0 = lad.begin(s)
        for bb, tt in map(tuple, np.argwhere(ids[:, l0:lad.cum[s]] != want[:, l0:lad.cum[s]])):
            assert float(margin[bb, tt]) < 1e-3, f"seed {seed}: token ({bb}, {l0 + tt}) of stage {s} differs although its draw margin is {float(margin[bb, tt]):.2e}"
    ctx.close(); qc.close()
