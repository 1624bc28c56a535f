"""VQVAE decoder on HIP (csrc/conv.hip, csrc/vae.hip) against PyTorch fp32 on the same weights.

The decoder is the caller side of the sampler (vae.fhat_to_img, /root/reference/models/vqvae.py:62-63,
models/basic_vae.py:163-226).  Floating point: the bar is 1e-4 on the [-1, 1] image (the same bar the MIOpen decode is held to in
test_gpu_e2e.py) and 2e-5 on single convolutions; the split-operand arithmetic itself is exact to ~1e-6 (test_gpu_ops.py)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rnd
from sdvar_amd import engine as E
from torch_ref import fhat_to_img_torch
from sdvar_amd.vqvae import VQVAE
from sdvar_amd.weights import vae_state_dict

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _rows(x):
    """(B, C, H, W) -> channel-last rows (B H W, C)"""
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def _unrows(r, B, H, W):
    return r.view(B, H, W, -1).permute(0, 3, 1, 2)


def _guard(W):
    return (W + 3 + 15) // 16 * 16


def _unplanes_any(xp, pf):
    """operand planes (3 bf16 | 2 fp16, ...) -> fp64 values"""
    if pf == 3:
        return sum((xp[k].to(torch.int32) << 16).view(torch.float32).double() for k in range(3))
    return sum(xp[k].view(torch.float16).double() for k in range(2))


PFMTS = [3, 2]          # bf16x3 and f16x2 operand planes


@pytest.mark.parametrize("pf", PFMTS)
@pytest.mark.parametrize("B,Cin,Cout,H,taps,res,split", [(2, 64, 96, 8, 9, False, 0), (1, 32, 32, 16, 9, False, 0), (2, 160, 160, 12, 9, True, 0),
                                                         (2, 320, 160, 6, 1, False, 0), (2, 128, 320, 5, 9, True, 3), (1, 640, 1920, 4, 1, False, 4),
                                                         (3, 96, 40, 7, 9, False, 2), (2, 32, 64, 9, 1, False, 0), (2, 64, 64, 9, 1, False, 0)])
def test_conv_matches_torch(dev, B, Cin, Cout, H, taps, res, split, pf, monkeypatch):
    lib = E.load_library()
    if pf == 2 and (B + H) % 2:          # half of the f16x2 cases on the 32x32x16 ping-pong loop (the default is the 16x16x32 one)
        monkeypatch.setenv("SDVAR_CONV_PP", "1")
    W = H + 1                                                                  # non-square on purpose
    x = rnd(1, (B, Cin, H, W)).to(dev)
    k = 3 if taps == 9 else 1
    w = (rnd(2, (Cout, Cin, k, k)) / (Cin * taps) ** 0.5).to(dev); b = rnd(3, (Cout,), 0.1).to(dev)
    r = rnd(4, (B, Cout, H, W)).to(dev) if res else None
    G = _guard(W); M = B * H * W; R = B * (H + 2) * (W + 2) + 2 * G
    xp = torch.full((pf, Cin // 32, R, 32), 0x7FC0 if pf == 3 else 0x7E00, dtype=torch.int16, device=dev)          # NaN patterns: every row must be written
    E._check(lib.sdvar_op_vae_prep(_p(_rows(x)), None, None, None, _p(xp), (Cin // 32) * R * 32, pf, B, Cin, H, W, 0, 0, G, _st()))
    wp = torch.zeros(pf, taps * Cin // 32, Cout, 32, dtype=torch.int16, device=dev)
    wsc = torch.zeros(4, device=dev)
    E._check(lib.sdvar_op_conv_weight_planes(_p(w), _p(wp), Cout, Cin, taps, taps * Cin * Cout, pf, _p(wsc), _st()))
    out = torch.empty(M, Cout, device=dev)
    ws = torch.empty(max(split, 1) * M * Cout, device=dev)
    rr = _rows(r) if res else None
    E._check(lib.sdvar_op_conv_planes(_p(xp), (Cin // 32) * R * 32, R, G, _p(wp), taps * Cin * Cout, pf, _p(wsc) if pf == 2 else None, _p(b), _p(rr), _p(out), B, H, W,
                                      Cout, Cin, taps, _p(ws), ws.numel(), split, _st()))
    want = F.conv2d(x.double(), w.double(), b.double(), padding=k // 2) + (r.double() if res else 0)
    got = _unrows(out, B, H, W).double()
    assert torch.isfinite(out).all()
    err = (got - want).abs().max().item()
    assert err <= 2e-5, err


@pytest.mark.parametrize("pf", PFMTS)
@pytest.mark.parametrize("up,mode", [(0, 0), (0, 3), (1, 0), (0, 1)])
def test_prep_groupnorm_silu_upsample(dev, up, mode, pf):
    lib = E.load_library()
    B, Cc, H, W = 2, 64, 6, 5
    x = rnd(5, (B, Cc, H, W), 2.0).to(dev)
    gamma, beta = (1 + rnd(6, (Cc,), 0.1)).to(dev), rnd(7, (Cc,), 0.1).to(dev)
    xg = x.view(B, 32, -1).double()
    mean, var = xg.mean(-1), xg.var(-1, unbiased=False)
    stats = torch.stack([mean, 1 / torch.sqrt(var + 1e-6)], -1).float().contiguous()
    Ho, Wo = H << up, W << up
    G = _guard(Wo); M = B * (Ho + 2) * (Wo + 2); R = M + 2 * G
    xp = torch.full((pf, Cc // 32, R, 32), 0x7FC0 if pf == 3 else 0x7E00, dtype=torch.int16, device=dev)
    E._check(lib.sdvar_op_vae_prep(_p(_rows(x)), _p(stats), _p(gamma), _p(beta), _p(xp), (Cc // 32) * R * 32, pf, B, Cc, H, W, up, mode, G, _st()))
    v = _unplanes_any(xp, pf)   # (C/32, R, 32)
    v = v.permute(1, 0, 2).reshape(R, Cc)
    assert v[:G].abs().max().item() == 0 and v[G + M:].abs().max().item() == 0              # guards
    full = v[G:G + M].view(B, Ho + 2, Wo + 2, Cc)
    assert full[:, 0].abs().max().item() == 0 and full[:, :, 0].abs().max().item() == 0 and full[:, -1].abs().max().item() == 0 and full[:, :, -1].abs().max().item() == 0
    want = x
    if mode & 1:
        want = F.group_norm(want, 32, gamma, beta, eps=1e-6)
    if mode & 2:
        want = F.silu(want)
    if up:
        want = F.interpolate(want, scale_factor=2, mode="nearest")
    got = full[:, 1:-1, 1:-1].permute(0, 3, 1, 2)
    assert (got - want.double()).abs().max().item() <= (2e-6 if (mode or pf == 2) else 0.0)       # bf16x3 planes are exact, f16x2 planes hold 2^-22


def _decode_pair(dev, ch, B, latent=16, seed=11, conv_mode=None):
    pns = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
    sd = vae_state_dict(pns, "stress", seed, V=64, Cvae=32, ch=ch, with_encoder=False)
    vae = VQVAE(vocab_size=64, z_channels=32, ch=ch, v_patch_nums=pns, with_encoder=False)
    vae.load_state_dict(sd)
    vae = vae.to(dev)
    ctx = E.VaeCtx(sd, B, dev, latent_hw=latent, conv_mode=conv_mode)
    f_hat = rnd(seed + 1, (B, 32, latent, latent), 1.5).to(dev)
    return vae, ctx, f_hat


@pytest.mark.parametrize("cm", ["bf16x3", "f16x2"])
def test_decoder_small_width_matches_pytorch(dev, cm):
    vae, ctx, f_hat = _decode_pair(dev, 32, 3, conv_mode=cm)
    got = ctx.decode(f_hat)
    want = fhat_to_img_torch(vae, f_hat.clone())
    assert got.shape == want.shape == (3, 3, 256, 256)
    err = (got - want).abs().max().item()
    assert err <= 1e-4, err
    assert (got - ctx.decode(f_hat)).abs().max().item() == 0                    # deterministic
    sub = ctx.decode(f_hat[:1])                                                # other batch size: other split-K choices, same image
    assert (sub - got[:1]).abs().max().item() <= 2e-5


@pytest.mark.parametrize("cm", ["bf16x3", "f16x2", "f16x2:pp1", "f16x2:pp0"])
def test_decoder_reference_width_matches_pytorch(dev, cm):
    """vae_ch160v4096z32 geometry (the checkpoint the reference loads): ch = 160, widths 640/320/160, 256^2 output.  The f16x2 conv kernel has three K-loops
    (SDVAR_CONV_PP: 2 = ping-pong on v_mfma_f32_16x16x32_f16 with the 16-byte epilogue, the default; 1 = ping-pong on 32x32x16; 0 = the round-2 loop):
    all three against PyTorch, fused GroupNorm statistics and the up-sampling phase scatter included."""
    lib = E.load_library()
    if ":pp" in cm:
        E._check(lib.sdvar_debug_set_variant(b"conv_pp", int(cm[-1]))); cm = "f16x2"
    try:
        vae, ctx, f_hat = _decode_pair(dev, 160, 2, conv_mode=cm)
        got = ctx.decode(f_hat)
    finally:
        E._check(lib.sdvar_debug_set_variant(b"conv_pp", -1))
    want = fhat_to_img_torch(vae, f_hat.clone())
    err = (got - want).abs().max().item()
    assert err <= 1e-4, err
    assert float(got.min()) >= -1.0 and float(got.max()) <= 1.0


def test_decoder_512_geometry(dev):
    """BASELINE config P4 decodes 32x32 latents to 512^2: 1024-token attention blocks, 4x the rows everywhere."""
    vae, ctx, f_hat = _decode_pair(dev, 32, 1, latent=32)
    got = ctx.decode(f_hat)
    want = fhat_to_img_torch(vae, f_hat.clone())
    assert got.shape == want.shape == (1, 3, 512, 512)
    err = (got - want).abs().max().item()
    assert err <= 1e-4, err


def test_decoder_1024_geometry(dev):
    """The 1024^2 ladder's decode: 64 x 64 latents, 4096-token attention blocks - beyond what the attention kernel can keep in LDS, so its probabilities go
    through the decoder's split-K workspace (vae_attn_kernel<true>)."""
    vae, ctx, f_hat = _decode_pair(dev, 32, 1, latent=64)
    got = ctx.decode(f_hat)
    want = fhat_to_img_torch(vae, f_hat.clone())
    assert got.shape == want.shape == (1, 3, 1024, 1024)
    err = (got - want).abs().max().item()
    assert err <= 1e-4, err


def test_decoder_rejects_bad_shapes(dev):
    _, ctx, f_hat = _decode_pair(dev, 32, 1)
    with pytest.raises(E.SdvarError):
        ctx.decode(torch.zeros(2, 32, 16, 16, device=dev))                      # exceeds max_batch
    with pytest.raises(E.SdvarError):
        ctx.decode(torch.zeros(1, 32, 8, 8, device=dev))


def test_module_entry_point_uses_the_hip_decoder(dev):
    """VQVAE.fhat_to_img (the call VAR.autoregressive_infer_cfg makes, var.py:215) == engine.VaeCtx.decode; CPU tensors are refused."""
    vae, ctx, f_hat = _decode_pair(dev, 32, 2)
    got = vae.fhat_to_img(f_hat)
    assert torch.equal(got, ctx.decode(f_hat))
    assert (got - fhat_to_img_torch(vae, f_hat.clone())).abs().max().item() <= 1e-4
    with pytest.raises(E.SdvarError):
        vae.fhat_to_img(f_hat.cpu())


def test_bind_rejects_wrong_tensor_list(dev):
    """sdvar_vae_bind checks the tensor count against the descriptor (host logic of csrc/vae.hip) and leaves the object unbound."""
    lib = E.load_library()
    d = E._VaeDesc()
    d.ch, d.z_channels, d.n_mult, d.num_res_blocks, d.max_batch, d.latent_hw = 32, 32, 5, 2, 1, 16
    for i, m in enumerate((1, 1, 2, 2, 4)):
        d.ch_mult[i] = m
    h = C.c_void_p()
    E._check(lib.sdvar_vae_create(C.byref(d), C.byref(h)))
    t = torch.zeros(16, device=dev)
    arr = (C.c_void_p * 3)(t.data_ptr(), t.data_ptr(), t.data_ptr())
    assert lib.sdvar_vae_bind(h, arr, 3, _st()) != 0 and b"expected" in lib.sdvar_last_error()
    img = torch.zeros(1, 3, 256, 256, device=dev); f = torch.zeros(1, 32, 16, 16, device=dev)
    assert lib.sdvar_vae_decode(h, _p(f), 1, _p(img), _st()) != 0 and b"not bound" in lib.sdvar_last_error()
    lib.sdvar_vae_destroy(h)
