"""GPU: every HIP kernel, called through the C ABI, against the oracle / torch fp32 on the same seeded inputs."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import golden, rnd, state_dicts
from oracle import var_oracle as orc
from sdvar_amd import engine as E
from sdvar_amd.ladder import LADDER_256, LADDER_512, as_ladder, bicubic_up_matrix, area_down_matrix
from sdvar_amd.noise import exponential_noise

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("M,N,K", [(16, 768, 256), (1, 128, 32), (33, 384, 1024), (64, 3072, 1024), (100, 1024, 4096), (400, 1152, 384),
                                   (576, 4096, 1024), (1600, 1024, 1024), (4096, 256, 1024), (130, 192, 64)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_epilogues(dev, M, N, K, epi):
    lib = E.load_library()
    X, W, b = rnd(1, (M, K)).to(dev), rnd(2, (N, K), 1 / math.sqrt(K)).to(dev), rnd(3, (N,)).to(dev)
    rows_per_gate = 7 if M > 7 else 1
    R = (M + rows_per_gate - 1) // rows_per_gate
    res, gate = rnd(4, (M, N)).to(dev), rnd(5, (R, 2 * N)).to(dev)
    out = res.clone() if epi == 2 else torch.empty(M, N, device=dev)
    E._check(lib.sdvar_op_gemm(_p(X), K, _p(W), _p(b), _p(out), N, M, N, K, epi, _p(out) if epi == 2 else None, N, _p(gate) if epi == 2 else None,
                               rows_per_gate, 2 * N, _st()))
    ref = X.double() @ W.double().t() + b.double()
    if epi == 1:
        ref = F.gelu(ref, approximate="tanh")
    if epi == 2:
        g = gate[:, :N].double().repeat_interleave(rows_per_gate, 0)[:M]
        ref = res.double() + ref * g
    err = (out.double() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
    # against torch fp32 on the same device as a second opinion on scale
    assert torch.isfinite(out).all()


def test_gemm_is_deterministic(dev):
    lib = E.load_library()
    X, W = rnd(1, (576, 1024)).to(dev), rnd(2, (3072, 1024), 0.03).to(dev)
    outs = []
    for _ in range(2):
        o = torch.empty(576, 3072, device=dev)
        E._check(lib.sdvar_op_gemm(_p(X), 1024, _p(W), None, _p(o), 3072, 576, 3072, 1024, 0, None, 0, None, 1, 0, _st()))
        outs.append(o)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("rows,Cw,rpi", [(5, 256, 5), (64, 1024, 16), (41, 384, 41), (18, 1920, 9), (7, 768, 1)])
def test_ln_modulate(dev, rows, Cw, rpi):
    lib = E.load_library()
    x = rnd(1, (rows, Cw), 2.0).to(dev) + 0.3
    R = (rows + rpi - 1) // rpi
    mod = rnd(2, (R, 6 * Cw)).to(dev)
    out = torch.empty_like(x)
    E._check(lib.sdvar_op_ln_modulate(_p(x), C_void(mod, 2 * Cw), C_void(mod, 4 * Cw), _p(out), None, 0, 3, rows, Cw, rpi, 6 * Cw, _st()))
    sc = mod[:, 2 * Cw:3 * Cw].repeat_interleave(rpi, 0)[:rows]; sh = mod[:, 4 * Cw:5 * Cw].repeat_interleave(rpi, 0)[:rows]
    ref = F.layer_norm(x.cpu(), (Cw,), eps=1e-6).mul(sc.cpu().add(1)).add_(sh.cpu())
    assert (out.cpu() - ref).abs().max().item() <= 2e-5


def C_void(t, off_elems):
    return C.c_void_p(t.data_ptr() + 4 * off_elems)


@pytest.mark.parametrize("R,l,H,pos0", [(2, 1, 4, 0), (4, 9, 6, 5), (2, 36, 16, 55), (16, 4, 12, 1)])
def test_qk_norm_append(dev, R, l, H, pos0):
    lib = E.load_library()
    Cw, Lmax = 64 * H, pos0 + l + 3
    qkv = rnd(1, (R * l, 3 * Cw)).to(dev)
    sm = (rnd(2, (H,), 0.5) + math.log(4.0)); sm[0] = 6.0                       # one head above the ln(100) clamp
    sm = sm.to(dev)
    qo = torch.zeros(R, H, l, 64, device=dev); kc = torch.zeros(R, H, Lmax, 64, device=dev); vc = torch.zeros_like(kc)
    E._check(lib.sdvar_op_qk_norm_append(_p(qkv), _p(sm), _p(qo), _p(kc), _p(vc), 0, R, l, H, Lmax, pos0, _st()))
    q, k, v = qkv.cpu().view(R, l, 3, H, 64).permute(2, 0, 3, 1, 4).unbind(0)
    scale = sm.cpu().view(1, H, 1, 1).clamp_max(math.log(100.0)).exp()
    assert (qo.cpu() - F.normalize(q, dim=-1).mul(scale)).abs().max().item() <= 2e-5
    assert (kc.cpu()[:, :, pos0:pos0 + l] - F.normalize(k, dim=-1)).abs().max().item() <= 1e-6
    assert torch.equal(vc.cpu()[:, :, pos0:pos0 + l], v)
    assert kc[:, :, :pos0].abs().max().item() == 0 if pos0 else True           # nothing outside the appended window
    assert kc[:, :, pos0 + l:].abs().max().item() == 0


def _attn_ref(q, k, v, qbeg, vis):
    l, K = q.shape[2], k.shape[2]
    mask = torch.zeros(l, K)
    for j in range(len(qbeg)):
        e = qbeg[j + 1] if j + 1 < len(qbeg) else l
        mask[qbeg[j]:e, vis[j]:] = -torch.inf
    return F.scaled_dot_product_attention(q.double(), k.double(), v.double(), attn_mask=mask.double().view(1, 1, l, K), scale=1.0)


@pytest.mark.parametrize("R,H,lens,prefix", [(2, 4, [1], 0), (2, 2, [4], 1), (4, 3, [25], 30), (2, 4, [100], 155), (2, 16, [256], 424),
                                             (2, 4, [9, 16], 5), (2, 3, [64, 100, 169], 91), (2, 2, [1, 4], 0), (1, 2, [169, 256], 255), (2, 2, [324], 640)])
def test_attention_matches_sdpa_with_block_causal_rows(dev, R, H, lens, prefix):
    lib = E.load_library()
    l = sum(lens); Ktot = prefix + l; Lmax = Ktot + 5
    q = (F.normalize(rnd(1, (R, H, l, 64)), dim=-1) * 4.0).to(dev)
    kc = torch.full((R, H, Lmax, 64), float("nan"), device=dev); vc = torch.full_like(kc, float("nan"))   # unread tail must not leak
    kc[:, :, :Ktot] = F.normalize(rnd(2, (R, H, Ktot, 64)), dim=-1).to(dev); vc[:, :, :Ktot] = rnd(3, (R, H, Ktot, 64)).to(dev)
    qbeg = [int(sum(lens[:j])) for j in range(len(lens))]
    vis = [prefix + int(sum(lens[:j + 1])) for j in range(len(lens))]
    out = torch.empty(R, l, H * 64, device=dev)
    n = len(lens)
    E._check(lib.sdvar_op_attention(_p(q), _p(kc), _p(vc), 0, _p(out), None, 0, 3, R, H, l, Lmax, Ktot, n, (C.c_int32 * n)(*qbeg), (C.c_int32 * n)(*vis), _st()))
    ref = _attn_ref(q.cpu(), kc.cpu()[:, :, :Ktot], vc.cpu()[:, :, :Ktot], qbeg, vis).transpose(1, 2).reshape(R, l, H * 64)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5, err


def test_attention_forced_online_rescale(dev):
    """A key far above the rest late in the cache forces the running max to jump (rule 26 of the CDNA guide)."""
    lib = E.load_library()
    R, H, l, Ktot = 1, 1, 40, 300
    q = F.normalize(rnd(1, (R, H, l, 64)), dim=-1) * 50.0
    k = F.normalize(rnd(2, (R, H, Ktot, 64)), dim=-1)
    k[0, 0, 200] = q[0, 0, 3] / 50.0
    k[0, 0, 299] = q[0, 0, 17] / 50.0
    v = rnd(3, (R, H, Ktot, 64))
    out = torch.empty(R, l, 64, device=dev)
    qd, kd, vd = q.to(dev), k.to(dev).contiguous(), v.to(dev)                  # keep the device copies alive across the call
    E._check(lib.sdvar_op_attention(_p(qd), _p(kd), _p(vd), 0, _p(out), None, 0, 3, R, H, l, Ktot, Ktot, 1, (C.c_int32 * 1)(0), (C.c_int32 * 1)(Ktot), _st()))
    ref = _attn_ref(q, k, v, [0], [Ktot]).transpose(1, 2).reshape(R, l, 64)
    assert (out.cpu().double() - ref).abs().max().item() <= 2e-5


def test_noise_fill_matches_host_stream(dev):
    lib = E.load_library()
    B, l, V = 3, 7, 4096
    q = torch.empty(B * l, V, device=dev)
    E._check(lib.sdvar_op_noise_fill(_p(q), B, l, V, 0x1234567890ABCDEF, 5, 11, _st()))
    host = torch.from_numpy(exponential_noise(0x1234567890ABCDEF, 5, B, l, V, image_offset=11)).view(B * l, V)
    rel = ((q.cpu() - host).abs() / host).max().item()
    assert rel <= 5e-6, rel                                                    # same uniforms; logf vs float64 log


def _sampler_inputs(ci, scale):
    rng = np.random.Generator(np.random.Philox(key=[11, 22]))
    lg = None
    for c in range(ci + 1):                                                    # same draw sequence as make_golden.sampler_fixture
        lg = torch.from_numpy(rng.standard_normal(size=(2, 5, 4096), dtype=np.float32) * np.float32(golden("sampler_cases")["scale"][c]))
    if ci == 2:
        srt = lg.sort(-1, descending=True)[0]
        lg[lg == srt[..., 899:900]] = 0.0
        lg[..., :7] = srt[..., 899:900]
    return lg


def test_cfg_sample_reference_known_answers(dev):
    """helpers.py:6-19 fixtures (ties at the k-th value, top-k/top-p on and off): ids and the surviving set are exact."""
    g = golden("sampler_cases")
    B, l, V = 2, 5, 4096
    for ci in range(len(g["top_k"])):
        lg = _sampler_inputs(ci, g["scale"][ci])
        q = torch.from_numpy(exponential_noise(77, ci, B, l, V)).view(-1, V).to(dev)
        logits2 = torch.cat([lg, torch.zeros_like(lg)], 0).to(dev).contiguous()         # t = 0 -> CFG is the identity on the cond rows
        ids = torch.zeros(B, l, dtype=torch.int64, device=dev); dbg = torch.empty(B, l, V, device=dev)
        E.cfg_sample(logits2, B, l, V, 0.0, int(g["top_k"][ci]), float(g["top_p"][ci]), q, 0, 0, 0, ids, 0, l, dbg)
        assert np.array_equal(ids.cpu().numpy(), g["ids"][ci]), f"case {ci}"
        assert np.array_equal((~torch.isinf(dbg)).sum(-1).cpu().numpy(), g["n_keep"][ci]), f"case {ci} surviving set"


@pytest.mark.parametrize("B,l,t,tk,tp", [(2, 16, 0.75, 900, 0.96), (8, 64, 1.5, 0, 0.9), (3, 5, 0.0, 40, 0.0), (1, 256, 1.3333, 900, 0.96), (2, 9, 0.5, 0, 0.0)])
def test_cfg_sample_vs_oracle(dev, B, l, t, tk, tp):
    V = 4096
    lg = rnd(7, (2 * B, l, V), 2.5)
    q = torch.from_numpy(exponential_noise(3, 1, B, l, V)).view(-1, V)
    ids_o, masked_o = orc.sample_topk_topp(orc.cfg_combine(lg, B, t), tk, tp, q)
    ids = torch.zeros(B, 1000, dtype=torch.int64, device=dev); dbg = torch.empty(B, l, V, device=dev)
    E.cfg_sample(lg.to(dev), B, l, V, t, tk, tp, q.to(dev), 0, 0, 0, ids, 100, 1000, dbg)
    assert torch.equal(dbg.cpu(), masked_o), "masked CFG logits must be bit-identical (same roundings as torch)"
    assert torch.equal(ids.cpu()[:, 100:100 + l], ids_o)
    assert ids[:, :100].abs().sum().item() == 0 and ids[:, 100 + l:].abs().sum().item() == 0


def test_cfg_sample_device_noise_equals_explicit_noise(dev):
    lib = E.load_library()
    B, l, V = 4, 25, 4096
    lg = rnd(9, (2 * B, l, V), 2.0).to(dev)
    q = torch.empty(B * l, V, device=dev)
    E._check(lib.sdvar_op_noise_fill(_p(q), B, l, V, 99, 3, 8, _st()))
    a = torch.zeros(B, l, dtype=torch.int64, device=dev); b = torch.zeros_like(a)
    E.cfg_sample(lg, B, l, V, 0.9, 900, 0.96, None, 99, 3, 8, a, 0, l)
    E.cfg_sample(lg, B, l, V, 0.9, 900, 0.96, q, 0, 0, 0, b, 0, l)
    assert torch.equal(a, b)


def test_verify_accept_known_answers(dev):
    """basic_token_matching fixtures (var.py:1160-1227): accepted stages and match counts are exact."""
    g = golden("sd_components")
    kat = g["accept_kat"]
    pns = LADDER_256
    rng = np.random.Generator(np.random.Philox(key=[9, 9]))
    B, V = 2, 4096
    fr = [[1.0, 1.0, 1.0], [1.0, 0.52, 0.2], [0.5, 0.5, 0.49], [0.49, 1.0, 1.0], [1.0, 1.0, 0.0], [0.75, 0.5, 0.5]]
    for case in range(6):
        toks, lgs = [], []
        for j in range(3):
            n = pns[4 + j] ** 2
            lg = torch.from_numpy(rng.standard_normal(size=(B, n, V), dtype=np.float32))
            flat = lg.argmax(-1).reshape(-1).clone()
            k = int(round(fr[case][j] * B * n))
            wrong = (flat + 1) % V
            flat[k:] = wrong[k:]
            toks.append(flat.view(B, n)); lgs.append(lg)
        lens = [t.shape[1] for t in toks]
        cond = torch.cat(lgs, 1)
        logits2 = torch.cat([cond, torch.zeros_like(cond)], 0).to(dev).contiguous()
        ids = torch.cat(toks, 1).to(dev).contiguous()
        counts = torch.zeros(40, dtype=torch.int32, device=dev)
        E.verify_accept(logits2, B, lens, V, [0.0, 0.0, 0.0], ids, 0, ids.shape[1], 0.5, counts)
        c = counts.cpu().tolist()
        assert [c[16]] + c[:3] + c[17:20] == list(kat[case]), case


def test_verify_accept_cfg_per_stage_vs_oracle(dev):
    B, V, lens, ts = 3, 4096, [4, 9, 16], [0.3, 0.45, 0.6]
    lg = rnd(21, (2 * B, sum(lens), V), 2.0)
    cls, off = [], 0
    for n, t in zip(lens, ts):
        cls.append(orc.cfg_combine(lg[:, off:off + n], B, t)); off += n
    am = torch.cat([c.argmax(-1) for c in cls], 1)
    ids = am.clone(); ids[0, 5:] = (ids[0, 5:] + 3) % V
    n_o, matched, total = orc.accept_scan([ids[:, :4], ids[:, 4:13], ids[:, 13:]], cls, 0.5)
    counts = torch.zeros(40, dtype=torch.int32, device=dev); amx = torch.zeros(B, sum(lens), dtype=torch.int64, device=dev)
    E.verify_accept(lg.to(dev), B, lens, V, ts, ids.to(dev), 0, ids.shape[1], 0.5, counts, amx)
    c = counts.cpu().tolist()
    assert torch.equal(amx.cpu(), am)
    assert c[16] == n_o and c[:3] == matched and c[17:20] == total


@pytest.mark.parametrize("lname,pns", [("256", LADDER_256), ("512", LADDER_512)])
def test_quant_next_vs_reference_fixture_and_oracle(dev, lname, pns):
    g = golden("quant_" + lname)
    _, sd_vae = state_dicts(2, pns, "stress", 1234)
    oq = orc.OracleQuant(sd_vae, pns)
    qc = E.QuantCtx(sd_vae, pns, 2, dev)
    B, L = 2, sum(p * p for p in pns)
    ids_all = torch.from_numpy(g["ids"].astype(np.int64)).to(dev).contiguous()
    f = torch.zeros(B, 32, pns[-1], pns[-1], device=dev); f_o = torch.zeros(B, 32, pns[-1], pns[-1])
    off = 0
    for si, pn in enumerate(pns):
        last = si == len(pns) - 1
        nxt = None if last else torch.empty(B, pns[si + 1] ** 2, 32, device=dev)
        qc.next(si, ids_all[:, off:], L, f, nxt, B)
        f_o, nxt_o = oq.next_input(si, f_o, oq.embed_ids(ids_all[:, off:off + pn * pn].cpu(), pn))
        off += pn * pn
        assert (f.cpu() - f_o).abs().max().item() <= 2e-5, si
        if not last:
            want = nxt_o.view(B, 32, -1).transpose(1, 2)
            assert (nxt.cpu() - want).abs().max().item() <= 2e-5, si
    np.testing.assert_allclose(f.cpu().numpy(), g["f_hat"], atol=3e-5)         # the reference's own f_hat


@pytest.mark.parametrize("share", [0, 1])
def test_quant_phi_layouts_vs_reference_fixture(dev, share):
    """VectorQuantizer2 built with share_quant_resi = 0 (PhiNonShared: quant_resi.<k>, one Phi per scale) or 1 (PhiShared: quant_resi.qresi), quant.py:27-32,
    209-216, 232-243: QuantCtx reads either layout from the state_dict; f_hat against the reference's own (tests/golden/make_golden.py quant_layouts)."""
    from sdvar_amd.weights import vae_state_dict
    g = golden("quant_layouts_256")
    pns = tuple(int(p) for p in g["patch_nums"])
    sd_vae = vae_state_dict(pns, "stress", 1234, ch=32, share_quant_resi=share, with_encoder=False)
    oq = orc.OracleQuant(sd_vae, pns)
    qc = E.QuantCtx(sd_vae, pns, 2, dev)
    assert len(qc.pw) == (len(pns) if share == 0 else 1)
    B, L = 2, sum(p * p for p in pns)
    ids_all = torch.from_numpy(g[f"s{share}_ids"].astype(np.int64)).to(dev).contiguous()
    f = torch.zeros(B, 32, pns[-1], pns[-1], device=dev); f_o = torch.zeros(B, 32, pns[-1], pns[-1])
    off = 0
    for si, pn in enumerate(pns):
        last = si == len(pns) - 1
        nxt = None if last else torch.empty(B, pns[si + 1] ** 2, 32, device=dev)
        qc.next(si, ids_all[:, off:], L, f, nxt, B)
        f_o, nxt_o = oq.next_input(si, f_o, oq.embed_ids(ids_all[:, off:off + pn * pn].cpu(), pn))
        off += pn * pn
        assert (f.cpu() - f_o).abs().max().item() <= 2e-5, si
        if not last:
            assert (nxt.cpu() - nxt_o.view(B, 32, -1).transpose(1, 2)).abs().max().item() <= 2e-5, si
    np.testing.assert_allclose(f.cpu().numpy(), g[f"s{share}_f_hat"], atol=3e-5)         # the reference's own f_hat
    qc.close()


def test_fp16_kv_cache_append_and_attention(dev):
    """BASELINE config P4: the cache holds fp16 (round-to-nearest-even of the fp32 values); attention widens while staging."""
    lib = E.load_library()
    R, l, H, pos0 = 2, 36, 4, 55
    Cw, Lmax = 64 * H, pos0 + l
    qkv = rnd(1, (R * l, 3 * Cw)).to(dev)
    sm = torch.full((H,), math.log(4.0), device=dev)
    qo = torch.zeros(R, H, l, 64, device=dev)
    kc = torch.zeros(R, H, Lmax, 64, device=dev, dtype=torch.float16); vc = torch.zeros_like(kc)
    kc[:, :, :pos0] = F.normalize(rnd(2, (R, H, pos0, 64)), dim=-1).half().to(dev); vc[:, :, :pos0] = rnd(3, (R, H, pos0, 64)).half().to(dev)
    E._check(lib.sdvar_op_qk_norm_append(_p(qkv), _p(sm), _p(qo), _p(kc), _p(vc), 1, R, l, H, Lmax, pos0, _st()))
    q, k, v = qkv.cpu().view(R, l, 3, H, 64).permute(2, 0, 3, 1, 4).unbind(0)
    k16 = F.normalize(k, dim=-1).half()
    diff = (kc.cpu()[:, :, pos0:].float() - k16.float()).abs()
    assert (diff > 0).float().mean().item() < 1e-3 and diff.max().item() <= 1e-3       # same rounding up to 1-ulp fp32 differences before it
    assert torch.equal(vc.cpu()[:, :, pos0:], v.half())
    out = torch.empty(R, l, H * 64, device=dev)
    E._check(lib.sdvar_op_attention(_p(qo), _p(kc), _p(vc), 1, _p(out), None, 0, 3, R, H, l, Lmax, Lmax, 1, (C.c_int32 * 1)(0), (C.c_int32 * 1)(Lmax), _st()))
    ref = _attn_ref(qo.cpu(), kc.cpu().float(), vc.cpu().float(), [0], [Lmax]).transpose(1, 2).reshape(R, l, H * 64)
    assert (out.cpu().double() - ref).abs().max().item() <= 2e-5


# ------------------------------------------------------------------------------------------------ bf16x3 split-operand path
@pytest.mark.parametrize("R,H,lens,prefix", [(2, 4, [1], 0), (2, 2, [4], 1), (4, 3, [25], 30), (2, 4, [100], 155), (2, 16, [256], 424),
                                             (2, 4, [9, 16], 5), (2, 3, [64, 100, 169], 91), (2, 2, [1, 4], 0), (1, 2, [169, 256], 255), (2, 2, [324], 640)])
def test_planes_kv_cache_append_and_attention(dev, R, H, lens, prefix):
    """Cache format 2 (bf16x3 planes, attention on the bf16 matrix cores): append in two calls, then the block-causal attention;
    K and V are held exactly (three planes = the fp32 value), so the bar is the fp32 kernel's."""
    lib = E.load_library()
    l = sum(lens); Ktot = prefix + l; Lp = (Ktot + 5 + 63) // 64 * 64
    Cw = 64 * H
    sm = torch.full((H,), math.log(4.0), device=dev)
    kc = torch.zeros(R, H, 3, Lp, 64, device=dev, dtype=torch.int16); vc = torch.zeros(R, H, 3, 64, Lp, device=dev, dtype=torch.int16)
    parts = []
    for i, (n, pos0) in enumerate([(prefix, 0), (l, prefix)]):
        if n == 0:
            continue
        qkv = rnd(10 + i, (R * n, 3 * Cw)).to(dev)
        qo = torch.zeros(R, H, n, 64, device=dev)
        E._check(lib.sdvar_op_qk_norm_append(_p(qkv), _p(sm), _p(qo), _p(kc), _p(vc), 2, R, n, H, Lp, pos0, _st()))
        parts.append(qkv.cpu().view(R, n, 3, H, 64).permute(2, 0, 3, 1, 4))
    k = F.normalize(torch.cat([p[1] for p in parts], dim=2), dim=-1); v = torch.cat([p[2] for p in parts], dim=2)
    # the planes reproduce K exactly: sum of the three bf16 planes == fp32 value written by the fp32-format kernel
    kpl = kc.cpu().view(torch.bfloat16).float().sum(2)[:, :, :Ktot]
    assert (kpl - k).abs().max().item() <= 1e-6
    perm = [(p & ~12) | ((p & 4) << 1) | ((p & 8) >> 1) for p in range(Ktot)]
    vpl = vc.cpu().view(torch.bfloat16).float().sum(2)[:, :, :, perm].transpose(2, 3)
    assert torch.equal(vpl, v)
    qbeg = [int(sum(lens[:j])) for j in range(len(lens))]
    vis = [prefix + int(sum(lens[:j + 1])) for j in range(len(lens))]
    out = torch.empty(R, l, H * 64, device=dev)
    n = len(lens)
    E._check(lib.sdvar_op_attention(_p(qo), _p(kc), _p(vc), 2, _p(out), None, 0, 3, R, H, l, Lp, Ktot, n, (C.c_int32 * n)(*qbeg), (C.c_int32 * n)(*vis), _st()))
    ref = _attn_ref(qo.cpu(), k, v, qbeg, vis).transpose(1, 2).reshape(R, l, H * 64)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5, err
    # plane output == fp32 output (the projection GEMM's operand)
    M = R * l
    outp = torch.zeros(3, H * 64 // 32, M, 32, device=dev, dtype=torch.int16)
    E._check(lib.sdvar_op_attention(_p(qo), _p(kc), _p(vc), 2, None, _p(outp), M * H * 64, 3, R, H, l, Lp, Ktot, n, (C.c_int32 * n)(*qbeg), (C.c_int32 * n)(*vis), _st()))
    assert torch.equal(_unplanes(outp.cpu()).float().view(R, l, H * 64), out.cpu())


def test_planes_attention_forced_online_rescale(dev):
    lib = E.load_library()
    R, H, l, Ktot, Lp = 1, 1, 40, 300, 320
    q = F.normalize(rnd(1, (R, H, l, 64)), dim=-1) * 50.0
    k = F.normalize(rnd(2, (R, H, Ktot, 64)), dim=-1)
    k[0, 0, 200] = q[0, 0, 3] / 50.0
    k[0, 0, 299] = q[0, 0, 17] / 50.0
    v = rnd(3, (R, H, Ktot, 64))

    def planes3(t):                                                            # exact 3-way truncation split, like common.h split3
        out, rest = [], t.clone()
        for _ in range(3):
            hi = (rest.view(torch.int32) & -65536).view(torch.float32)
            out.append((hi.view(torch.int32) >> 16).to(torch.int16)); rest = rest - hi
        return torch.stack(out)
    kc = torch.zeros(R, H, 3, Lp, 64, dtype=torch.int16); vc = torch.zeros(R, H, 3, 64, Lp, dtype=torch.int16)
    kc[:, :, :, :Ktot] = planes3(k).permute(1, 2, 0, 3, 4)
    perm = torch.tensor([(p & ~12) | ((p & 4) << 1) | ((p & 8) >> 1) for p in range(Ktot)])
    vc[:, :, :, :, perm] = planes3(v).permute(1, 2, 0, 4, 3)
    out = torch.empty(R, l, 64, device=dev)
    qd, kd, vd = q.to(dev), kc.to(dev).contiguous(), vc.to(dev).contiguous()
    E._check(lib.sdvar_op_attention(_p(qd), _p(kd), _p(vd), 2, _p(out), None, 0, 3, R, H, l, Lp, Ktot, 1, (C.c_int32 * 1)(0), (C.c_int32 * 1)(Ktot), _st()))
    ref = _attn_ref(q, k, v, [0], [Ktot]).transpose(1, 2).reshape(R, l, 64)
    assert (out.cpu().double() - ref).abs().max().item() <= 2e-5


def _planes(t, dev):
    """fp32 (rows, K) -> device K-blocked planes (3, K/32, rows, 32) int16 through the library's splitter."""
    lib = E.load_library()
    x = t.to(dev).contiguous()
    rows, K = x.shape
    p = torch.empty(3, K // 32, rows, 32, dtype=torch.int16, device=dev)
    E._check(lib.sdvar_op_split_planes(_p(x), _p(p), rows, K, rows * K, _st()))
    return p


def _unplanes(p):
    """K-blocked planes (3, K/32, rows, 32) -> fp64 (rows, K)."""
    v = sum((p[k].to(torch.int32) << 16).view(torch.float32).double() for k in range(3))
    return v.permute(1, 0, 2).reshape(v.shape[1], -1)


def test_split_planes_is_exact(dev):
    x = rnd(1, (257, 64), 3.0)
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 1e-30, 3.4e38, 1.0000001, -123456.789])
    p = _planes(x, dev)
    assert torch.equal(_unplanes(p).cpu(), x.double())                          # x == p0 + p1 + p2 exactly


@pytest.mark.parametrize("M,N,K", [(16, 768, 256), (1, 128, 32), (33, 384, 1024), (64, 3072, 1024), (100, 1024, 4096), (400, 1152, 384),
                                   (576, 4096, 1024), (1600, 1024, 1024), (4096, 256, 1024), (130, 192, 64), (2704, 1024, 4096)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_bf16x3_epilogues(dev, M, N, K, epi):
    """Split-operand GEMM is fp32-accurate: error vs fp64 within the band of the fp32 MFMA kernel."""
    lib = E.load_library()
    X, W, b = rnd(1, (M, K)), rnd(2, (N, K), 1 / math.sqrt(K)), rnd(3, (N,)).to(dev)
    Xp, Wp = _planes(X, dev), _planes(W, dev)
    rows_per_gate = 7 if M > 7 else 1
    R = (M + rows_per_gate - 1) // rows_per_gate
    res, gate = rnd(4, (M, N)).to(dev), rnd(5, (R, 2 * N)).to(dev)
    out = res.clone() if epi == 2 else torch.empty(M, N, device=dev)
    outp = torch.empty(3, N // 32, M, 32, dtype=torch.int16, device=dev) if epi == 1 else None
    E._check(lib.sdvar_op_gemm_bf16x3(_p(Xp), M * K, _p(Wp), N * K, _p(b), _p(out), N, _p(outp), M * N, M, N, K, epi, _p(out) if epi == 2 else None, N,
                                      _p(gate) if epi == 2 else None, rows_per_gate, 2 * N, _st()))
    ref = X.double() @ W.double().t() + b.cpu().double()
    if epi == 1:
        ref = F.gelu(ref, approximate="tanh"); got = _unplanes(outp).cpu()
    elif epi == 2:
        g = gate.cpu()[:, :N].double().repeat_interleave(rows_per_gate, 0)[:M]
        ref = res.cpu().double() + ref * g; got = out.cpu().double()
    else:
        got = out.cpu().double()
    err = (got - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


def test_ln_and_attention_plane_outputs_equal_fp32_outputs(dev):
    lib = E.load_library()
    rows, Cw = 41, 384
    x = rnd(1, (rows, Cw), 2.0).to(dev); mod = rnd(2, (1, 6 * Cw)).to(dev)
    o32 = torch.empty_like(x); op = torch.empty(3, Cw // 32, rows, 32, dtype=torch.int16, device=dev)
    E._check(lib.sdvar_op_ln_modulate(_p(x), C_void(mod, 2 * Cw), C_void(mod, 4 * Cw), _p(o32), None, 0, 3, rows, Cw, rows, 6 * Cw, _st()))
    E._check(lib.sdvar_op_ln_modulate(_p(x), C_void(mod, 2 * Cw), C_void(mod, 4 * Cw), None, _p(op), rows * Cw, 3, rows, Cw, rows, 6 * Cw, _st()))
    assert torch.equal(_unplanes(op).float(), o32)
    R, H, l, K = 2, 3, 100, 255
    q = (F.normalize(rnd(1, (R, H, l, 64)), dim=-1) * 4).to(dev); kc = F.normalize(rnd(2, (R, H, K, 64)), dim=-1).to(dev); vc = rnd(3, (R, H, K, 64)).to(dev)
    a32 = torch.empty(R, l, H * 64, device=dev); ap = torch.empty(3, H * 2, R * l, 32, dtype=torch.int16, device=dev)
    one = (C.c_int32 * 1)
    E._check(lib.sdvar_op_attention(_p(q), _p(kc), _p(vc), 0, _p(a32), None, 0, 3, R, H, l, K, K, 1, one(0), one(K), _st()))
    E._check(lib.sdvar_op_attention(_p(q), _p(kc), _p(vc), 0, None, _p(ap), R * l * H * 64, 3, R, H, l, K, K, 1, one(0), one(K), _st()))
    assert torch.equal(_unplanes(ap).float().view(R, l, H * 64), a32)


@pytest.mark.parametrize("bm", [128, 256])
@pytest.mark.parametrize("M,N,K,split", [(130, 192, 64, 1), (1, 128, 32, 1), (257, 384, 1024, 3), (4096, 256, 1024, 1), (2704, 1024, 4096, 5), (100, 4096, 1024, 2)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_bf16x3_lds_dma_kernels_forced_tile(dev, M, N, K, split, epi, bm):
    """The LDS-DMA pipelined 128x128 / 256x128 kernels on ragged edges (clamped source rows), short K loops and split-K."""
    lib = E.load_library()
    X, W, b = rnd(1, (M, K)), rnd(2, (N, K), 1 / math.sqrt(K)), rnd(3, (N,)).to(dev)
    Xp, Wp = _planes(X, dev), _planes(W, dev)
    res, gate = rnd(4, (M, N)).to(dev), rnd(5, (M, 2 * N)).to(dev)
    out = res.clone() if epi == 2 else torch.full((M, N), float("nan"), device=dev)
    outp = torch.empty(3, N // 32, M, 32, dtype=torch.int16, device=dev) if epi == 1 else None
    E._check(lib.sdvar_debug_set_gemm_cfg(bm, split))
    try:
        E._check(lib.sdvar_op_gemm_bf16x3(_p(Xp), M * K, _p(Wp), N * K, _p(b), _p(out), N, _p(outp), M * N, M, N, K, epi, _p(out) if epi == 2 else None, N,
                                          _p(gate) if epi == 2 else None, 1, 2 * N, _st()))
    finally:
        E._check(lib.sdvar_debug_set_gemm_cfg(0, 0))
    ref = X.double() @ W.double().t() + b.cpu().double()
    if epi == 1:
        ref = F.gelu(ref, approximate="tanh"); got = _unplanes(outp).cpu()
    elif epi == 2:
        ref = res.cpu().double() + ref * gate.cpu()[:, :N].double(); got = out.cpu().double()
    else:
        got = out.cpu().double()
    err = (got - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,N,K,epi", [(2704, 3072, 1024, 0), (4096, 2304, 768, 0), (2704, 4096, 1024, 1), (4096, 3072, 768, 2), (2500, 3072, 256, 0)])
def test_gemm_bf16x3_hybrid_tail_split(dev, M, N, K, epi):
    """Shapes whose 256 x 128 tiles do not fill whole rounds of the 256 CUs: the full rounds run unsplit, the partial last round is split along K
    (compact slabs) and reduced per tile.  The result must equal the plain kernels' (the 128-row kernel forced) to fp32 rounding."""
    lib = E.load_library()
    x = rnd(1, (M, K)).to(dev); w = (rnd(2, (N, K)) / K ** 0.5).to(dev); b = rnd(3, (N,), 0.1).to(dev)
    res = rnd(4, (M, N)).to(dev); gate = rnd(5, (4, N)).to(dev)
    xp, wp = _planes(x, dev), _planes(w, dev)

    def run(force):
        E._check(lib.sdvar_debug_set_gemm_cfg(*force))
        out = res.clone() if epi == 2 else torch.zeros(M, N, device=dev)
        outp = torch.zeros(3, N // 32, M, 32, device=dev, dtype=torch.int16)
        E._check(lib.sdvar_op_gemm_bf16x3(_p(xp), M * K, _p(wp), N * K, _p(b), _p(out), N, _p(outp), M * N, M, N, K, epi, _p(out) if epi == 2 else None, N,
                                          _p(gate) if epi == 2 else None, (M + 3) // 4, N, _st()))
        E._check(lib.sdvar_debug_set_gemm_cfg(0, 0))
        return _unplanes(outp.cpu()) if epi == 1 else out.cpu().double()
    auto, ref = run((0, 0)), run((128, 1))
    err = (auto - ref).abs().max().item()
    assert err <= 3e-6 * max(1.0, ref.abs().max().item()), err


# ------------------------------------------------------------------------------------------------ f16x2 split-operand path
def _planes_h(t, dev, scaled=False):
    """fp32 (rows, K) -> (K-blocked fp16 planes (2, K/32, rows, 32) int16, scale tensor or None) through the library's splitter."""
    lib = E.load_library()
    x = t.to(dev).contiguous()
    rows, K = x.shape
    p = torch.empty(2, K // 32, rows, 32, dtype=torch.int16, device=dev)
    sc = torch.zeros(4, device=dev) if scaled else None
    E._check(lib.sdvar_op_split_planes_f16(_p(x), _p(p), rows, K, rows * K, _p(sc) if scaled else None, _st()))
    return p, sc


def _unplanes_h(p):
    """K-blocked fp16 planes (2, K/32, rows, 32) -> fp64 (rows, K)."""
    v = sum(p[k].view(torch.float16).double() for k in range(2))
    return v.permute(1, 0, 2).reshape(v.shape[1], -1)


def test_split_planes_f16_accuracy_and_scale(dev):
    """x ~ h + l to 2^-22 relative while l is a normal fp16 number (|x| >= 2^-3), absolute 2^-25 below; values beyond the fp16 range saturate;
    the weight scale is the power of two that brings max|w| into (2^12, 2^13]."""
    x = rnd(1, (257, 64), 3.0)
    x[0, :6] = torch.tensor([0.0, 1.0, -1.0, 1e-9, 7e4, -1e9])
    p, _ = _planes_h(x, dev)
    got = _unplanes_h(p).cpu()
    xs = x.double().clamp(-65504, 65504)
    err = (got - xs).abs()
    assert bool((err <= torch.maximum(xs.abs() * 2.0 ** -21.9, torch.tensor(2.0 ** -24.9, dtype=torch.float64))).all())
    w = rnd(2, (128, 96), 0.02)
    p, sc = _planes_h(w, dev, scaled=True)
    S = sc.cpu()
    assert S[0] * S[1] == 1.0 and 2 ** 12 < float(w.abs().max()) * float(S[0]) <= 2 ** 13 and math.log2(float(S[0])).is_integer()
    rel = ((_unplanes_h(p).cpu() * float(S[1]) - w.double()).abs() / w.double().abs().clamp_min(1e-30))
    assert float(rel[w.abs() > 1e-4].max()) <= 2.0 ** -21.9


@pytest.mark.parametrize("M,N,K", [(16, 768, 256), (1, 128, 32), (33, 384, 1024), (64, 3072, 1024), (100, 1024, 4096), (400, 1152, 384),
                                   (576, 4096, 1024), (1600, 1024, 1024), (4096, 256, 1024), (130, 192, 64), (2704, 1024, 4096)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_f16x2_epilogues(dev, M, N, K, epi):
    """The three-product fp16 scheme with scaled weights: error vs fp64 within the band of the fp32 MFMA kernel (same bound as bf16x3)."""
    lib = E.load_library()
    X, W, b = rnd(1, (M, K)), rnd(2, (N, K), 1 / math.sqrt(K)), rnd(3, (N,)).to(dev)
    (Xp, _), (Wp, sc) = _planes_h(X, dev), _planes_h(W, dev, scaled=True)
    rows_per_gate = 7 if M > 7 else 1
    R = (M + rows_per_gate - 1) // rows_per_gate
    res, gate = rnd(4, (M, N)).to(dev), rnd(5, (R, 2 * N)).to(dev)
    out = res.clone() if epi == 2 else torch.empty(M, N, device=dev)
    outp = torch.empty(2, N // 32, M, 32, dtype=torch.int16, device=dev) if epi == 1 else None
    E._check(lib.sdvar_op_gemm_f16x2(_p(Xp), M * K, _p(Wp), N * K, _p(sc), _p(b), _p(out), N, _p(outp), M * N, M, N, K, epi, _p(out) if epi == 2 else None, N,
                                     _p(gate) if epi == 2 else None, rows_per_gate, 2 * N, _st()))
    ref = X.double() @ W.double().t() + b.cpu().double()
    if epi == 1:
        ref = F.gelu(ref, approximate="tanh"); got = _unplanes_h(outp).cpu()
    elif epi == 2:
        g = gate.cpu()[:, :N].double().repeat_interleave(rows_per_gate, 0)[:M]
        ref = res.cpu().double() + ref * g; got = out.cpu().double()
    else:
        got = out.cpu().double()
    err = (got - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("bm", [32, 64, 128, 256, 512, 768])
@pytest.mark.parametrize("M,N,K,split", [(130, 192, 64, 1), (1, 128, 32, 1), (257, 384, 1024, 3), (4096, 256, 1024, 1), (2704, 1024, 4096, 5), (100, 4096, 1024, 2),
                                         (300, 256, 96, 1), (300, 256, 96, 3), (700, 768, 160, 1)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_f16x2_kernels_forced_tile(dev, M, N, K, split, epi, bm):
    """Every f16x2 kernel (LDS-DMA 32/64-row tiles, 128x128 and 256x128 with 3-stage rings, the 256x256 ping-pong kernel = bm 512) on ragged edges, K loops
    of 1 .. 128 steps (ring fill / drain paths, odd and even step counts) and split-K."""
    lib = E.load_library()
    X, W, b = rnd(1, (M, K)), rnd(2, (N, K), 1 / math.sqrt(K)), rnd(3, (N,)).to(dev)
    (Xp, _), (Wp, sc) = _planes_h(X, dev), _planes_h(W, dev, scaled=True)
    res, gate = rnd(4, (M, N)).to(dev), rnd(5, (M, 2 * N)).to(dev)
    out = res.clone() if epi == 2 else torch.full((M, N), float("nan"), device=dev)
    outp = torch.empty(2, N // 32, M, 32, dtype=torch.int16, device=dev) if epi == 1 else None
    E._check(lib.sdvar_debug_set_gemm_cfg(bm, split))
    try:
        E._check(lib.sdvar_op_gemm_f16x2(_p(Xp), M * K, _p(Wp), N * K, _p(sc), _p(b), _p(out), N, _p(outp), M * N, M, N, K, epi, _p(out) if epi == 2 else None, N,
                                         _p(gate) if epi == 2 else None, 1, 2 * N, _st()))
    finally:
        E._check(lib.sdvar_debug_set_gemm_cfg(0, 0))
    ref = X.double() @ W.double().t() + b.cpu().double()
    if epi == 1:
        ref = F.gelu(ref, approximate="tanh"); got = _unplanes_h(outp).cpu()
    elif epi == 2:
        ref = res.cpu().double() + ref * gate.cpu()[:, :N].double(); got = out.cpu().double()
    else:
        got = out.cpu().double()
    err = (got - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


VARIANTS = [("gemm_h4_var", 0, 512), ("gemm_h4_var", 1, 512), ("gemm_h4_var", 2, 512),          # the 256 x 256 kernel on 32x32x16 MFMAs, three DMA placements
            ("gemm_h2_stages", 4, 128), ("gemm_h2_stages", 3, 128), ("gemm_h2_stages", 2, 128), ("gemm_h2_stages", 5, 128),      # the 128 x 128 kernel without the ping-pong schedule
            ("gemm_small_pp", 0, 64),                                                                    # 64-row tiles on the 4-wave ring kernel
            ("gemm_small_pp", 1, 32), ("gemm_small_pp", 0, 32)]                                          # 32-row tiles on the 4-wave ring kernel


@pytest.mark.parametrize("name,value,bm", VARIANTS)
def test_gemm_f16x2_non_default_variants(dev, name, value, bm):
    """The kernel variants that only an environment switch selects in production (kept for A/B runs) still compute the same thing: the forced-tile shapes of
    test_gemm_f16x2_kernels_forced_tile with the variant selected through sdvar_debug_set_variant, all three epilogues, one QKV-epilogue model run."""
    lib = E.load_library()
    E._check(lib.sdvar_debug_set_variant(name.encode(), value))
    try:
        for (M, N, K, split) in [(257, 384, 1024, 3), (4096, 256, 1024, 1), (700, 768, 160, 1), (300, 256, 96, 1)]:
            for epi in (0, 1, 2):
                test_gemm_f16x2_kernels_forced_tile(dev, M, N, K, split if bm != 512 else 1, epi, bm)
        test_qkv_epilogue_forced_tiles(dev, bm, False)
    finally:
        E._check(lib.sdvar_debug_set_variant(name.encode(), -1))
    assert lib.sdvar_debug_set_variant(b"no_such_variant", 0) != 0


@pytest.mark.parametrize("sched", [0, 2, 3])
def test_attention_pp_schedule_variants(dev, sched):
    """The 8-wave attention kernel's other schedules (SDVAR_ATTN_PP_SCHED: 0 = four slots per tile, 2 / 3 = the two-slot schedule with 6 / 8 ring stages) against
    fp64 SDPA on the shapes that reach it (more than 128 queries per (row, head)), both cache formats, single stage and a two-stage chunk."""
    lib = E.load_library()
    E._check(lib.sdvar_debug_set_variant(b"attn_pp_sched", sched))
    try:
        for fmt in (3, 4):
            for (R, H, lens, prefix) in [(2, 16, [256], 424), (1, 2, [169, 256], 255), (2, 2, [324], 640), (2, 3, [64, 100, 169], 91)]:
                test_f16_planes_kv_cache_append_and_attention(dev, R, H, lens, prefix, fmt)
    finally:
        E._check(lib.sdvar_debug_set_variant(b"attn_pp_sched", -1))


@pytest.mark.parametrize("M,N,K,split", [(16, 3072, 1024, 0), (1, 16, 32, 0), (16, 768, 4096, 0), (17, 1024, 4096, 0), (33, 2304, 768, 0), (64, 4096, 1024, 0), (80, 1024, 1024, 0),
                                         (80, 768, 3072, 0), (48, 1024, 96, 0), (64, 1024, 4096, 8), (16, 528, 160, 3)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_f16x2_skinny_kernel(dev, M, N, K, split, epi):
    """The skinny kernel of stages 0 - 1 (M <= 80 rows: 16 output columns x all rows per workgroup, weights straight to registers, waves split K, LDS reduce):
    every row-tile count 1..5, ragged M, K loops of 1 .. 128 steps (1 .. 4 workgroups along K, forced more), all three epilogues, against fp64; and it IS what the
    automatic choice launches for these shapes."""
    lib = E.load_library()
    X, W, b = rnd(1, (M, K)), rnd(2, (N, K), 1 / math.sqrt(K)), rnd(3, (N,)).to(dev)
    (Xp, _), (Wp, sc) = _planes_h(X, dev), _planes_h(W, dev, scaled=True)
    rows_per_gate = 5 if M > 5 else 1
    R = (M + rows_per_gate - 1) // rows_per_gate
    res, gate = rnd(4, (M, N)).to(dev), rnd(5, (R, 2 * N)).to(dev)

    def run(force):
        out = res.clone() if epi == 2 else torch.full((M, N), float("nan"), device=dev)
        outp = torch.zeros(2, (N + 31) // 32, M, 32, dtype=torch.int16, device=dev) if epi == 1 and N % 32 == 0 else None
        if force:
            E._check(lib.sdvar_debug_set_gemm_cfg(16, split))
        try:
            E.last_gemm_cfg()
            E._check(lib.sdvar_op_gemm_f16x2(_p(Xp), M * K, _p(Wp), N * K, _p(sc), _p(b), _p(out), N, _p(outp), M * N, M, N, K, epi, _p(out) if epi == 2 else None, N,
                                             _p(gate) if epi == 2 else None, rows_per_gate, 2 * N, _st()))
            assert E.last_gemm_cfg()["bm"] == 16 or (not force and K > 1024), "not the skinny kernel"          # the automatic choice takes it up to K = 1024 (one workgroup per column panel)
        finally:
            E._check(lib.sdvar_debug_set_gemm_cfg(0, 0))
        return _unplanes_h(outp).cpu() if epi == 1 else out.cpu().double()
    if epi == 1 and N % 32:
        pytest.skip("plane outputs are K-blocked by 32 columns")
    ref = X.double() @ W.double().t() + b.cpu().double()
    if epi == 1:
        ref = F.gelu(ref, approximate="tanh")
    elif epi == 2:
        ref = res.cpu().double() + ref * gate.cpu()[:, :N].double().repeat_interleave(rows_per_gate, 0)[:M]
    for force in (True, False):
        got = run(force)
        err = (got - ref).abs().max().item()
        assert err <= 2e-5 * max(1.0, ref.abs().max().item()), (force, err)


@pytest.mark.parametrize("bm", [32, 64, 128, 256, 512, 768])
@pytest.mark.parametrize("epi", [0, 2])
def test_gemm_f16x2_unaligned_epilogue(dev, bm, epi):
    """The epilogues move 16 bytes per access when every pointer and leading dimension allows it; odd leading dimensions and a bias / gate / res pointer off
    the 16-byte grid take the element-wise path of the same kernels."""
    lib = E.load_library()
    M, N, K = 150, 260, 96
    X, W = rnd(1, (M, K)), rnd(2, (N, K), 1 / math.sqrt(K))
    (Xp, _), (Wp, sc) = _planes_h(X, dev), _planes_h(W, dev, scaled=True)
    bbuf, resb, gbuf = rnd(3, (N + 1,)).to(dev), rnd(4, (M, N + 3)).to(dev), rnd(5, (M, 2 * N + 1)).to(dev)
    b = bbuf[1:]                                       # 4 bytes off the 16-byte grid
    out = resb.clone() if epi == 2 else torch.full((M, N + 3), float("nan"), device=dev)
    E._check(lib.sdvar_debug_set_gemm_cfg(bm, 1))
    try:
        E._check(lib.sdvar_op_gemm_f16x2(_p(Xp), M * K, _p(Wp), N * K, _p(sc), _p(b), _p(out), N + 3, None, 0, M, N, K, epi, _p(out) if epi == 2 else None, N + 3,
                                         _p(gbuf) if epi == 2 else None, 1, 2 * N + 1, _st()))
    finally:
        E._check(lib.sdvar_debug_set_gemm_cfg(0, 0))
    ref = X.double() @ W.double().t() + b.cpu().double()
    if epi == 2:
        ref = resb.cpu()[:, :N].double() + ref * gbuf.cpu()[:, :N].double()
    got = out.cpu()[:, :N].double()
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    if epi == 0: assert torch.isnan(out[:, N:]).all()           # the padding columns of the rows are not touched


@pytest.mark.parametrize("M,N,K,epi", [(2704, 3072, 1024, 0), (4096, 2304, 768, 0), (2704, 4096, 1024, 1), (4096, 3072, 768, 2), (2700, 3200, 256, 0)])
def test_gemm_f16x2_hybrid_tail_split(dev, M, N, K, epi):
    lib = E.load_library()
    x = rnd(1, (M, K)).to(dev); w = (rnd(2, (N, K)) / K ** 0.5).to(dev); b = rnd(3, (N,), 0.1).to(dev)
    res = rnd(4, (M, N)).to(dev); gate = rnd(5, (4, N)).to(dev)
    (xp, _), (wp, sc) = _planes_h(x, dev), _planes_h(w, dev, scaled=True)

    def run(force):
        E._check(lib.sdvar_debug_set_gemm_cfg(*force))
        out = res.clone() if epi == 2 else torch.zeros(M, N, device=dev)
        outp = torch.zeros(2, N // 32, M, 32, device=dev, dtype=torch.int16)
        E._check(lib.sdvar_op_gemm_f16x2(_p(xp), M * K, _p(wp), N * K, _p(sc), _p(b), _p(out), N, _p(outp), M * N, M, N, K, epi, _p(out) if epi == 2 else None, N,
                                         _p(gate) if epi == 2 else None, (M + 3) // 4, N, _st()))
        E._check(lib.sdvar_debug_set_gemm_cfg(0, 0))
        return _unplanes_h(outp.cpu()) if epi == 1 else out.cpu().double()
    # the cost model takes the hybrid tail only for some of these shapes (and that moves with every refit): FORCE it (bm 256, split -4 = tail split 4 ways) and let
    # the launch counter prove that launch_h3_hybrid + splitk_reduce_tiles_h_kernel ran; the automatic choice is compared too, whatever it is
    E.last_gemm_cfg()                                   # reset the launch counters
    hyb = run((256, -4))
    cfg = E.last_gemm_cfg()
    assert cfg["bm"] == 256 and cfg["tail_launches"] == 1, cfg
    auto, ref = run((0, 0)), run((128, 1))
    for got in (hyb, auto):
        err = (got - ref).abs().max().item()
        assert err <= 3e-6 * max(1.0, ref.abs().max().item()), err


def test_split_planes_f16_keeps_nan(dev):
    """A NaN stays a NaN in both planes (a bare min/max clamp would store -65504: common.h clamp_f16_range); +-inf saturates like any out-of-range value."""
    x = rnd(1, (8, 32)); x[0, 0] = float("nan"); x[1, 3] = float("inf"); x[2, 5] = -float("inf")
    p, _ = _planes_h(x, dev)
    h = p[0].view(torch.float16).permute(1, 0, 2).reshape(8, 32).cpu()
    got = _unplanes_h(p).cpu()
    assert torch.isnan(h[0, 0]) and torch.isnan(got[0, 0]) and got[1, 3] == 65504.0 and got[2, 5] == -65504.0
    assert torch.isfinite(got.flatten()[1:]).all()


@pytest.mark.parametrize("mode", ["f16x2", "bf16x3"])
def test_split_gemm_outlier_and_tiny_activations(dev, mode):
    """every other test feeds O(1) activations.  Here the X operand has a wide dynamic range - a few columns scaled by 1e3 .. 3e4 (massive
    activations), a few whole rows at 1e-4 - and the result is held against fp64 PER ROW:
      * rows with O(1) or larger data: |err| <= 2e-5 x the row's largest output, both modes (the bar of every GEMM test in this file);
      * the 1e-4 rows: bf16x3 keeps the 2e-5 relative bar (its split is exact); f16x2 is held to its DOCUMENTED absolute floor - fp16's subnormal spacing,
        2^-25 per element, i.e. 5 x 2^-25 x max ||w_n||_2 on the dot product (gemm_f16x2.hip header) - which is a relative error of
        ~1e-3 on such a row.  bf16x3 is the mode without a range limit; this test pins the difference."""
    lib = E.load_library()
    M, N, K = 320, 384, 1024
    X, W, b = rnd(11, (M, K)), rnd(12, (N, K), 1 / math.sqrt(K)), torch.zeros(N, device=dev)
    X[:, [3, 97, 511, 640]] = X[:, [3, 97, 511, 640]].clamp(-4, 4) * torch.tensor([1e3, 3e3, 1e4, 1.6e4])     # up to 6.4e4: inside the fp16 range (65504)
    tiny = torch.arange(M) % 16 == 5
    X[tiny] = rnd(13, (int(tiny.sum()), K), 1e-4)
    X[7, 200] = 2.0e5                                  # ONE value beyond the fp16 range: f16x2 saturates it at 65504 (documented), bf16x3 keeps it
    out = torch.empty(M, N, device=dev)
    if mode == "f16x2":
        (Xp, _), (Wp, sc) = _planes_h(X, dev), _planes_h(W, dev, scaled=True)
        E._check(lib.sdvar_op_gemm_f16x2(_p(Xp), M * K, _p(Wp), N * K, _p(sc), _p(b), _p(out), N, None, 0, M, N, K, 0, None, N, None, 1, 0, _st()))
    else:
        Xp, Wp = _planes(X, dev), _planes(W, dev)
        E._check(lib.sdvar_op_gemm_bf16x3(_p(Xp), M * K, _p(Wp), N * K, _p(b), _p(out), N, None, 0, M, N, K, 0, None, N, None, 1, 0, _st()))
    ref = (X.clamp(-65504, 65504) if mode == "f16x2" else X).double() @ W.double().t()
    err = (out.cpu().double() - ref).abs().amax(dim=1)
    rowmax = ref.abs().amax(dim=1)
    assert torch.isfinite(out).all()
    assert bool((err[~tiny] <= 2e-5 * rowmax[~tiny]).all()), (err[~tiny] / rowmax[~tiny]).max().item()
    if mode == "bf16x3":
        assert bool((err[tiny] <= 2e-5 * rowmax[tiny]).all()), (err[tiny] / rowmax[tiny]).max().item()
    else:
        floor = 5 * 2.0 ** -25 * W.double().norm(dim=1).max().item()
        assert bool((err[tiny] <= floor).all()), (err[tiny].max().item(), floor)
        assert (err[tiny] / rowmax[tiny]).max().item() <= 5e-3             # the documented loss of relative precision, bounded


@pytest.mark.parametrize("bm", [32, 64, 128, 256, 512, 768])
@pytest.mark.parametrize("kv_fp16", [False, True])
def test_qkv_epilogue_forced_tiles(dev, bm, kv_fp16):
    """The HEPI_QKV epilogue of EVERY f16x2 kernel (32- / 64-row ring kernel, 128 x 128, 256 x 128), forced through sdvar_debug_set_gemm_cfg(bm, 1), against
    the unfused path (QKV GEMM -> fp32 buffer -> qk_norm_append) on the same model: logits within 2e-5 of their scale; the launch counter proves the fused
    epilogue ran (and did not run with the switch off).  test_qkv_epilogue_equals_qk_norm_append covers only the tiles the cost model happens to choose."""
    pns, B, depth = LADDER_256, 2, 4
    lad = as_ladder(pns)
    sd, _ = state_dicts(depth, pns)
    tc = E.ModelCtx(sd, depth, pns, B, 2, dev, kv_fp16=kv_fp16)
    assert tc.gemm_mode == "f16x2"
    labels = torch.tensor([3, 977], device=dev)
    xs = [rnd(20 + s, (2 * B * lad.lens[s] * tc.Cw,)).to(dev) for s in range(7)]
    lg = torch.empty(2 * B * (lad.lens[5] + lad.lens[6]) * tc.V, device=dev)

    def run(fuse):
        E._check(tc.lib.sdvar_debug_set_qkv_fuse(1 if fuse else 0)); E._check(tc.lib.sdvar_debug_set_gemm_cfg(bm, 1))
        try:
            E.last_gemm_cfg()
            tc.begin(labels); out = []
            for s in range(5):
                tc.forward(xs[s].clone(), s, 1, lg); out.append(lg[:2 * B * lad.lens[s] * tc.V].clone())
            x = torch.cat([xs[5].view(2 * B, -1), xs[6].view(2 * B, -1)], 1).contiguous().view(-1)        # stages 5-6 as one chunk: ragged row tiles
            tc.forward(x, 5, 2, lg); out.append(lg[:2 * B * (lad.lens[5] + lad.lens[6]) * tc.V].clone())
            tc.kv_set_len(0)
            return out, E.last_gemm_cfg()["fused_qkv_launches"]
        finally:
            E._check(tc.lib.sdvar_debug_set_qkv_fuse(1)); E._check(tc.lib.sdvar_debug_set_gemm_cfg(0, 0))
    a, na = run(True)
    b, nb = run(False)
    assert na == 6 * depth and nb == 0, (na, nb)
    for x, y in zip(a, b):
        assert torch.isfinite(x).all() and (x - y).abs().max().item() <= 2e-5 * max(1.0, y.abs().max().item())
    tc.close(); torch.cuda.empty_cache()


def test_ln_and_attention_f16x2_plane_outputs(dev):
    """The producers' f16x2 planes hold the fp32 outputs to 2^-22 relative / 2^-25 absolute."""
    lib = E.load_library()
    rows, Cw = 41, 384
    x = rnd(1, (rows, Cw), 2.0).to(dev); mod = rnd(2, (1, 6 * Cw)).to(dev)
    o32 = torch.empty_like(x); op = torch.empty(2, Cw // 32, rows, 32, dtype=torch.int16, device=dev)
    E._check(lib.sdvar_op_ln_modulate(_p(x), C_void(mod, 2 * Cw), C_void(mod, 4 * Cw), _p(o32), None, 0, 3, rows, Cw, rows, 6 * Cw, _st()))
    E._check(lib.sdvar_op_ln_modulate(_p(x), C_void(mod, 2 * Cw), C_void(mod, 4 * Cw), None, _p(op), rows * Cw, 2, rows, Cw, rows, 6 * Cw, _st()))
    tol = lambda ref: torch.maximum(ref.abs() * 2.0 ** -21.9, torch.tensor(2.0 ** -24.9, dtype=torch.float64, device=ref.device))
    assert bool(((_unplanes_h(op) - o32.double()).abs() <= tol(o32.double())).all())
    R, H, l, K = 2, 3, 100, 255
    q = (F.normalize(rnd(1, (R, H, l, 64)), dim=-1) * 4).to(dev); kc = F.normalize(rnd(2, (R, H, K, 64)), dim=-1).to(dev); vc = rnd(3, (R, H, K, 64)).to(dev)
    a32 = torch.empty(R, l, H * 64, device=dev); ap = torch.empty(2, H * 2, R * l, 32, dtype=torch.int16, device=dev)
    one = (C.c_int32 * 1)
    E._check(lib.sdvar_op_attention(_p(q), _p(kc), _p(vc), 0, _p(a32), None, 0, 3, R, H, l, K, K, 1, one(0), one(K), _st()))
    E._check(lib.sdvar_op_attention(_p(q), _p(kc), _p(vc), 0, None, _p(ap), R * l * H * 64, 2, R, H, l, K, K, 1, one(0), one(K), _st()))
    ref = a32.double().view(R * l, H * 64)
    assert bool(((_unplanes_h(ap) - ref).abs() <= tol(ref)).all())


@pytest.mark.parametrize("fmt", [3, 4])
@pytest.mark.parametrize("R,H,lens,prefix", [(2, 4, [1], 0), (2, 2, [4], 1), (4, 3, [25], 30), (2, 4, [100], 155), (2, 16, [256], 424),
                                             (2, 4, [9, 16], 5), (2, 3, [64, 100, 169], 91), (2, 2, [1, 4], 0), (1, 2, [169, 256], 255), (2, 2, [324], 640)])
def test_f16_planes_kv_cache_append_and_attention(dev, R, H, lens, prefix, fmt):
    """Cache formats 3 (two fp16 planes per value: the fp32 cache to 2^-22, attention on the f16 matrix cores with 3 + 3 products) and
    4 (ONE fp16 plane: the fp16 KV cache of config P4, 2 + 2 products): append in two calls, then the block-causal attention.
    Format 3 is held to the fp32 kernel's bar against the fp32 K / V; format 4 against the fp16-rounded K / V it stores."""
    lib = E.load_library()
    NP = 2 if fmt == 3 else 1
    l = sum(lens); Ktot = prefix + l; Lp = (Ktot + 5 + 63) // 64 * 64
    Cw = 64 * H
    sm = torch.full((H,), math.log(4.0), device=dev)
    kc = torch.zeros(R, H, NP, Lp, 64, device=dev, dtype=torch.int16); vc = torch.zeros(R, H, NP, Lp, 64, device=dev, dtype=torch.int16)     # K and V planes: the same row-major layout
    parts = []
    for i, (n, pos0) in enumerate([(prefix, 0), (l, prefix)]):
        if n == 0:
            continue
        qkv = rnd(10 + i, (R * n, 3 * Cw)).to(dev)
        qo = torch.zeros(R, H, n, 64, device=dev)
        E._check(lib.sdvar_op_qk_norm_append(_p(qkv), _p(sm), _p(qo), _p(kc), _p(vc), fmt, R, n, H, Lp, pos0, _st()))
        parts.append(qkv.cpu().view(R, n, 3, H, 64).permute(2, 0, 3, 1, 4))
    k = F.normalize(torch.cat([p[1] for p in parts], dim=2), dim=-1); v = torch.cat([p[2] for p in parts], dim=2)
    kpl = kc.cpu().view(torch.float16).double().sum(2)[:, :, :Ktot]
    vpl = vc.cpu().view(torch.float16).double().sum(2)[:, :, :Ktot]
    assert int(kc[:, :, :, Ktot:].abs().max()) == 0 and int(vc[:, :, :, Ktot:].abs().max()) == 0               # nothing written past the appended positions
    if fmt == 3:
        tol = lambda ref: torch.maximum(ref.abs() * 2.0 ** -21.9, torch.tensor(2.0 ** -24.9, dtype=torch.float64))
        assert bool(((kpl - k.double()).abs() <= tol(k.double()) + 1e-7).all())            # + the 1-ulp fp32 differences of the normalisation
        assert bool(((vpl - v.double()).abs() <= tol(v.double())).all())
        kr, vr = k, v
    else:
        assert (kpl.float() - k.half().float()).abs().max().item() <= 1e-3 and torch.equal(vpl.float(), v.half().float())
        kr, vr = kpl.float(), vpl.float()                                                    # what the cache holds
    qbeg = [int(sum(lens[:j])) for j in range(len(lens))]
    vis = [prefix + int(sum(lens[:j + 1])) for j in range(len(lens))]
    out = torch.empty(R, l, H * 64, device=dev)
    n = len(lens)
    E._check(lib.sdvar_op_attention(_p(qo), _p(kc), _p(vc), fmt, _p(out), None, 0, 2, R, H, l, Lp, Ktot, n, (C.c_int32 * n)(*qbeg), (C.c_int32 * n)(*vis), _st()))
    ref = _attn_ref(qo.cpu(), kr, vr, qbeg, vis).transpose(1, 2).reshape(R, l, H * 64)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5, err
    M = R * l
    outp = torch.zeros(2, H * 64 // 32, M, 32, device=dev, dtype=torch.int16)
    E._check(lib.sdvar_op_attention(_p(qo), _p(kc), _p(vc), fmt, None, _p(outp), M * H * 64, 2, R, H, l, Lp, Ktot, n, (C.c_int32 * n)(*qbeg), (C.c_int32 * n)(*vis), _st()))
    o = out.cpu().double()
    assert bool(((_unplanes_h(outp.cpu()).view(R, l, H * 64) - o).abs() <= torch.maximum(o.abs() * 2.0 ** -21.9, torch.tensor(2.0 ** -24.9, dtype=torch.float64))).all())


@pytest.mark.parametrize("fmt", [3, 4])
def test_f16_planes_attention_forced_online_rescale(dev, fmt):
    """Scores of +-50 with the row maximum arriving late: the running-maximum rescale and the two-plane split of P."""
    lib = E.load_library()
    R, H, l, Ktot, Lp = 1, 1, 40, 300, 320
    q = F.normalize(rnd(1, (R, H, l, 64)), dim=-1) * 50.0
    k = F.normalize(rnd(2, (R, H, Ktot, 64)), dim=-1)
    k[0, 0, 200] = q[0, 0, 3] / 50.0
    k[0, 0, 299] = q[0, 0, 17] / 50.0
    v = rnd(3, (R, H, Ktot, 64))
    NP = 2 if fmt == 3 else 1

    def planes(t):
        h = t.half(); lo = (t - h.float()).half()
        return torch.stack([h, lo][:NP]).view(torch.int16)
    kc = torch.zeros(R, H, NP, Lp, 64, dtype=torch.int16); vc = torch.zeros(R, H, NP, Lp, 64, dtype=torch.int16)
    kc[:, :, :, :Ktot] = planes(k).permute(1, 2, 0, 3, 4)
    vc[:, :, :, :Ktot] = planes(v).permute(1, 2, 0, 3, 4)
    out = torch.empty(R, l, 64, device=dev)
    qd, kd, vd = q.to(dev), kc.to(dev).contiguous(), vc.to(dev).contiguous()
    E._check(lib.sdvar_op_attention(_p(qd), _p(kd), _p(vd), fmt, _p(out), None, 0, 2, R, H, l, Lp, Ktot, 1, (C.c_int32 * 1)(0), (C.c_int32 * 1)(Ktot), _st()))
    kr, vr = (k, v) if fmt == 3 else (k.half().float(), v.half().float())
    ref = _attn_ref(q, kr, vr, [0], [Ktot]).transpose(1, 2).reshape(R, l, 64)
    assert (out.cpu().double() - ref).abs().max().item() <= (2e-5 if fmt == 4 else 6e-5)      # scores of 50: the 2^-22 of K moves the exponent by 50 * 2^-22


# ------------------------------------------------------------------------------------------------ row-block launches (M <= 80, round 4)
def _ln_mod_ref(x, mod, K, rpi):
    """fp64 LayerNorm(eps 1e-6) * (1 + scale[g]) + shift[g], g = row / rpi; mod rows hold (.., scale at 2K, .., shift at 4K, ..) as the adaLN table does."""
    xd = x.double()
    n = (xd - xd.mean(-1, keepdim=True)) / torch.sqrt(xd.var(-1, unbiased=False, keepdim=True) + 1e-6)
    g = torch.arange(x.shape[0]) // rpi
    return n * (1 + mod[g, 2 * K:3 * K].double()) + mod[g, 4 * K:5 * K].double()


@pytest.mark.parametrize("M,N,K,rpi", [(16, 3072, 1024, 1), (64, 4096, 1024, 4), (80, 2304, 768, 5), (2, 768, 256, 1), (50, 4096, 1024, 25), (72, 1024, 256, 36), (33, 4096, 1024, 33)])
@pytest.mark.parametrize("epi", [0, 1])
def test_gemm_rowblk_layernorm_operand(dev, M, N, K, rpi, epi):
    """gemm_f16x2_rowblk_kernel with the LayerNorm + modulation prologue (basic_var.py:157-158 / 172-174 in front of the QKV / fc1 / head GEMM): against fp64 at the bar
    of the stand-alone GEMM tests, with an outlier channel in every row (the statistics are the two-pass form) and ragged last row blocks."""
    lib = E.load_library()
    x = rnd(1, (M, K), 2.0); x[:, 5] *= 300.0
    R = (M + rpi - 1) // rpi
    mod, W, b = rnd(2, (R, 6 * K)), rnd(3, (N, K), 1 / math.sqrt(K)), rnd(4, (N,)).to(dev)
    Wp, sc = _planes_h(W, dev, scaled=True)
    xd, md = x.to(dev), mod.to(dev)
    out = torch.empty(M, N, device=dev) if epi == 0 else None
    outp = torch.empty(2, N // 32, M, 32, dtype=torch.int16, device=dev) if epi == 1 else None
    E._check(lib.sdvar_op_gemm_rowblk(_p(xd), K, C_void(md, 2 * K), C_void(md, 4 * K), rpi, 6 * K, None, 0, _p(Wp), N * K, _p(sc), _p(b), _p(out), N, _p(outp), M * N, M, N, K, epi,
                                      None, 0, None, 1, 0, None, None, None, None, 0, 0, 0, 0, 0, _st()))
    ref = _ln_mod_ref(x, mod, K, rpi) @ W.double().t() + b.cpu().double()
    if epi == 1:
        ref = F.gelu(ref, approximate="tanh"); got = _unplanes_h(outp).cpu()
    else:
        got = out.cpu().double()
    err = (got - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("M,N,K", [(16, 1024, 4096), (80, 768, 3072), (33, 1024, 1024), (64, 256, 1024), (1, 16, 32)])
@pytest.mark.parametrize("epi", [0, 2])
def test_gemm_rowblk_plane_operand(dev, M, N, K, epi):
    """The same kernel on operand planes with K up to 4096 streamed UNSPLIT by one workgroup per 16 x 16 output tile (fc2 of stages 0-1: no slabs, no pending
    residual), bias and gated-residual epilogues, against fp64."""
    lib = E.load_library()
    X, W, b = rnd(1, (M, K)), rnd(2, (N, K), 1 / math.sqrt(K)), rnd(3, (N,)).to(dev)
    (Xp, _), (Wp, sc) = _planes_h(X, dev), _planes_h(W, dev, scaled=True)
    rows_per_gate = 7 if M > 7 else 1
    R = (M + rows_per_gate - 1) // rows_per_gate
    res, gate = rnd(4, (M, N)).to(dev), rnd(5, (R, 2 * N)).to(dev)
    out = res.clone() if epi == 2 else torch.empty(M, N, device=dev)
    E._check(lib.sdvar_op_gemm_rowblk(None, 0, None, None, 1, 0, _p(Xp), M * K, _p(Wp), N * K, _p(sc), _p(b), _p(out), N, None, 0, M, N, K, epi, _p(out) if epi == 2 else None, N,
                                      _p(gate) if epi == 2 else None, rows_per_gate, 2 * N, None, None, None, None, 0, 0, 0, 0, 0, _st()))
    ref = X.double() @ W.double().t() + b.cpu().double()
    if epi == 2:
        ref = res.cpu().double() + ref * gate.cpu()[:, :N].double().repeat_interleave(rows_per_gate, 0)[:M]
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("fmt", [3, 4])
@pytest.mark.parametrize("R,l,H,pos0,l2", [(16, 1, 16, 0, True), (16, 4, 12, 1, True), (16, 5, 4, 0, True), (2, 36, 6, 55, True), (4, 9, 4, 5, False)])
def test_gemm_rowblk_qkv_finish(dev, R, l, H, pos0, l2, fmt):
    """LayerNorm -> QKV -> bias, per-head L2 norm of q and k, q scale, k / v rows into the cache planes (basic_var.py:93-109) in ONE launch: q, and the cache rows read back
    from the planes, against fp64; rows of the cache outside [pos0, pos0 + l) stay untouched."""
    lib = E.load_library()
    Cw, M, NP, Lp = 64 * H, R * l, (2 if fmt == 3 else 1), 128
    x, mod = rnd(1, (M, Cw), 1.5), rnd(2, (R, 6 * Cw))
    W, b, sm = rnd(3, (3 * Cw, Cw), 1 / math.sqrt(Cw)), rnd(4, (3 * Cw,), 0.1), (rnd(5, (H,), 0.3) + math.log(4.0))
    b[Cw:2 * Cw] = 0.0                                                     # zero_k_bias (basic_var.py:93)
    Wp, sc = _planes_h(W, dev, scaled=True)
    xd, md, bd, smd = x.to(dev), mod.to(dev), b.to(dev), sm.to(dev)
    q = torch.zeros(R, H, l, 64, device=dev)
    kc = torch.full((R, H, NP, Lp, 64), 0x3c00, dtype=torch.int16, device=dev); vc = kc.clone()          # fp16 1.0 everywhere: untouched rows must keep it
    E._check(lib.sdvar_op_gemm_rowblk(_p(xd), Cw, C_void(md, 2 * Cw), C_void(md, 4 * Cw), l, 6 * Cw, None, 0, _p(Wp), 3 * Cw * Cw, _p(sc), _p(bd), None, 0, None, 0, M, 3 * Cw, Cw, 0,
                                      None, 0, None, 1, 0, _p(smd) if l2 else None, _p(q), _p(kc), _p(vc), l, H, Lp, pos0, fmt, _st()))
    qkv = (_ln_mod_ref(x, mod, Cw, l) @ W.double().t() + b.double()).view(R, l, 3, H, 64).permute(2, 0, 3, 1, 4)
    qr, kr, vr = qkv[0], qkv[1], qkv[2]
    if l2:
        qr = F.normalize(qr, dim=-1) * sm.double().clamp_max(math.log(100.0)).exp().view(1, H, 1, 1); kr = F.normalize(kr, dim=-1)
    else:
        qr = qr * 0.03125
    unp = lambda c: c.view(torch.float16).double().sum(2).cpu()             # (R, H, Lp, 64)
    gk, gv = unp(kc), unp(vc)
    tol = 2e-5 if fmt == 3 else 1.1e-3                                     # one fp16 plane (config P4's cache) holds 11 bits
    assert (q.cpu().double() - qr).abs().max().item() <= 2e-5 * max(1.0, qr.abs().max().item())
    assert (gk[:, :, pos0:pos0 + l] - kr).abs().max().item() <= tol * max(1.0, kr.abs().max().item())
    assert (gv[:, :, pos0:pos0 + l] - vr).abs().max().item() <= tol * max(1.0, vr.abs().max().item())
    keep = torch.ones(Lp, dtype=torch.bool); keep[pos0:pos0 + l] = False
    assert bool((gk[:, :, keep] == NP).all()) and bool((gv[:, :, keep] == NP).all())


@pytest.mark.parametrize("kv_fp16", [False, True])
def test_rowblk_stage_forward_equals_the_unfused_sequence(dev, kv_fp16):
    """stage_forward at M <= 80 (five launches per block: LayerNorm in the GEMM prologues, QKV finish, unsplit fc2) against the eight-launch sequence every other row count
    runs, on one model: stages 0-3 one at a time and stages 0-1 as one chunk, then stage 4 / the chunk 2-3 (M > 80: unfused in both runs) on the caches the fused launches
    wrote.  The kernel id of the last GEMM proves which path ran."""
    pns, B, depth = LADDER_256, 2, 4
    lad = as_ladder(pns)
    sd, _ = state_dicts(depth, pns)
    tc = E.ModelCtx(sd, depth, pns, B, 2, dev, kv_fp16=kv_fp16)
    assert tc.gemm_mode == "f16x2"
    labels = torch.tensor([3, 977], device=dev)
    xs = [rnd(40 + s, (2 * B * lad.lens[s] * tc.Cw,)).to(dev) for s in range(5)]
    lg = torch.empty(2 * B * (lad.lens[2] + lad.lens[3]) * tc.V, device=dev)
    cat = lambda a, b: torch.cat([xs[a].view(2 * B, -1), xs[b].view(2 * B, -1)], 1).contiguous().view(-1)

    def run(on):
        E._check(tc.lib.sdvar_debug_set_rowblk(on if isinstance(on, int) and not isinstance(on, bool) else (2 if on else 0)))      # 2: the row-block sequence at EVERY call of at most 80 rows (default 1: from 32 rows)
        try:
            out, ids = [], []
            tc.begin(labels)
            for s in range(5):                                                 # M = 4, 16, 36, 64 fused; 100 not
                tc.forward(xs[s].clone(), s, 1, lg); out.append(lg[:2 * B * lad.lens[s] * tc.V].clone()); ids.append(E.last_gemm_cfg()["bm"])
            tc.kv_set_len(0); tc.begin(labels)
            tc.forward(cat(0, 1), 0, 2, lg); out.append(lg[:2 * B * 5 * tc.V].clone()); ids.append(E.last_gemm_cfg()["bm"])       # M = 20
            tc.forward(cat(2, 3), 2, 2, lg); out.append(lg[:2 * B * 25 * tc.V].clone()); ids.append(E.last_gemm_cfg()["bm"])      # M = 100
            tc.kv_set_len(0)
            return out, ids
        finally:
            E._check(tc.lib.sdvar_debug_set_rowblk(1))
    a, ia = run(True)
    b, ib = run(False)
    _, idf = run(1)                                                            # the default also has a width floor (C >= 1024): this C = 256 model stays on the old sequence
    assert 17 not in idf, idf
    assert ia[:4] == [17] * 4 and ia[5] == 17 and ia[4] != 17 and ia[6] != 17 and 17 not in ib, (ia, ib)
    # the two sequences sum the LayerNorm statistics in different orders: k and v differ by an ulp of fp32 before the cache rounds them - invisible in the two-plane cache,
    # but the ONE-plane fp16 cache of config P4 may round such a pair to different fp16 neighbours (2^-11 relative on one element): the fp16-cache bar of the other tests
    tol = 2e-3 if kv_fp16 else 2e-5
    for x, y in zip(a, b):
        assert torch.isfinite(x).all() and (x - y).abs().max().item() <= tol * max(1.0, y.abs().max().item())
    tc.close(); torch.cuda.empty_cache()


def test_rowblk_default_gate_at_d16_width(dev):
    """The default gate of the row-block sequence (measured): C >= 1024 and 32 <= M <= 80 - at d16 / B = 8 that is stage 1 (M = 64) and the
    first verify chunk (M = 80), not stage 0 (M = 16, where the eight-launch sequence is faster) and not stage 2 (M = 144)."""
    from sdvar_amd.weights import var_state_dict_device
    pns, B = LADDER_256, 8
    lad = as_ladder(pns)
    tc = E.ModelCtx(var_state_dict_device(16, pns, dev, mode="stress"), 16, pns, B, 2, dev)
    assert tc.gemm_mode == "f16x2"
    labels = (torch.arange(B, device=dev) * 7) % 1000
    lg = torch.empty(2 * B * (lad.lens[0] + lad.lens[1] + lad.lens[2]) * tc.V, device=dev)
    got = []
    tc.begin(labels)
    for s in range(3):
        tc.forward(rnd(60 + s, (2 * B * lad.lens[s] * tc.Cw,)).to(dev), s, 1, lg); got.append(E.last_gemm_cfg()["bm"] == 17)
    tc.kv_set_len(0); tc.begin(labels)
    tc.forward(rnd(70, (2 * B * 5 * tc.Cw,)).to(dev), 0, 2, lg); got.append(E.last_gemm_cfg()["bm"] == 17)
    tc.kv_set_len(0)
    assert torch.isfinite(lg[:2 * B * 5 * tc.V]).all()
    assert got == [False, True, False, True], got
    tc.close(); torch.cuda.empty_cache()
