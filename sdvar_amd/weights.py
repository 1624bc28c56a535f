"""state_dict contract and deterministic random inits for the VAR / VQVAE pair.

Key names and shapes are the reference's (/root/reference/models/var.py:52-117,
models/basic_var.py:58-174, models/quant.py:15-43, models/vqvae.py:16-50) so that upstream checkpoints load
unchanged.  The two init recipes are SURVEY.md App. C.3:
  * perf  - the reference factory's distributions (models/var.py:261-311, models/__init__.py:24) drawn from one
            seeded generator, plus a deterministic init of the VQVAE which the reference leaves uninitialised;
  * stress- an init under which attention/FFN/adaLN all matter to the logits (needed for meaningful parity).
No tensor here is copied from the reference; the values come from `torch.Generator(seed)` in the order below.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Sequence

import numpy as np
import torch

from .ladder import as_ladder


def _gen(seed: int, name: str) -> np.random.Generator:
    """One counter-based generator per tensor, keyed by (seed, tensor name): values do not depend on creation order
    and come from numpy's Philox bit stream (portable across hosts, unlike torch.randn's vectorised paths)."""
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]))


def _n(g: np.random.Generator, shape, std) -> torch.Tensor:
    return torch.from_numpy(g.standard_normal(size=tuple(shape), dtype=np.float32) * np.float32(std))


def _tn(g: np.random.Generator, shape, std) -> torch.Tensor:
    """N(0, std) truncated to +-2 std by redrawing outliers (the law nn.init.trunc_normal_(std=std, a=-2*std...) has
    for the reference's std values is the same up to the truncation point convention; only used for perf-init)."""
    x = g.standard_normal(size=tuple(shape), dtype=np.float32).reshape(-1)
    idx = np.flatnonzero(np.abs(x) > 2.0)
    while idx.size:
        r = g.standard_normal(size=idx.size, dtype=np.float32)
        x[idx] = r
        idx = idx[np.abs(r) > 2.0]
    x = x.reshape(tuple(shape))
    return torch.from_numpy(x * np.float32(std))


def var_state_dict(depth: int, patch_nums: Sequence[int], mode: str = "perf", seed: int = 1234,
                   V: int = 4096, Cvae: int = 32, num_classes: int = 1000, shared_aln: bool = False, attn_l2_norm: bool = True, init_adaln: float = 0.5,
                   init_adaln_gamma: float = 1e-5, init_head: float = 0.02, init_std: float = -1) -> "OrderedDict[str, torch.Tensor]":
    """init_* are VAR.init_weights' arguments (var.py:261-311; factory defaults models/__init__.py:24) and shape the 'perf' init only."""
    lad = as_ladder(patch_nums)
    C, H, L, S = 64 * depth, depth, lad.L, lad.S
    if mode not in ("perf", "stress"):
        raise ValueError(mode)
    stress = mode == "stress"
    sseed = seed * 31 + depth
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    std0 = math.sqrt(1.0 / C / 3.0) if init_std < 0 else float(init_std)      # var.py:262

    def w(name, shape, s_std):      # weight-like: stress -> N(0, s_std), perf -> trunc_normal(std0)
        g = _gen(sseed, name)
        sd[name] = _n(g, shape, s_std) if stress else _tn(g, shape, std0)
        return sd[name]

    def b(name, shape, s_std):      # bias-like: stress -> N(0, s_std), perf -> 0
        sd[name] = _n(_gen(sseed, name), shape, s_std) if stress else torch.zeros(shape)
        return sd[name]

    w("pos_start", (1, 1, C), 0.5)
    w("pos_1LC", (1, L, C), 0.5)
    w("word_embed.weight", (C, Cvae), 1 / math.sqrt(Cvae)); b("word_embed.bias", (C,), 0.1)
    w("class_emb.weight", (num_classes + 1, C), 1.0)
    w("lvl_embed.weight", (S, C), 0.5)
    for i in range(depth):
        p = f"blocks.{i}."
        if attn_l2_norm:                 # basic_var.py:66-72: the parameter only exists with q/k L2 normalisation
            sd[p + "attn.scale_mul_1H11"] = torch.full((1, H, 1, 1), math.log(4.0)) + (
                _n(_gen(sseed, p + "attn.scale_mul_1H11"), (1, H, 1, 1), 0.3) if stress else 0.0)
        b(p + "attn.q_bias", (C,), 0.1); b(p + "attn.v_bias", (C,), 0.1)
        sd[p + "attn.zero_k_bias"] = torch.zeros(C)
        w(p + "attn.mat_qkv.weight", (3 * C, C), 1 / math.sqrt(C))
        w(p + "attn.proj.weight", (C, C), 1 / math.sqrt(C)).div_(math.sqrt(2 * depth)); b(p + "attn.proj.bias", (C,), 0.02)
        w(p + "ffn.fc1.weight", (4 * C, C), 1 / math.sqrt(C)); b(p + "ffn.fc1.bias", (4 * C,), 0.1)
        w(p + "ffn.fc2.weight", (C, 4 * C), 1 / math.sqrt(4 * C)).div_(math.sqrt(2 * depth)); b(p + "ffn.fc2.bias", (C,), 0.02)
        if shared_aln:                              # SharedAdaLin (var.py:16-19, 81): one Linear for all blocks + a per-block offset (basic_var.py:143-144)
            gss = _n(_gen(sseed, p + "ada_gss"), (1, 1, 6, C), 0.3 if stress else 1 / math.sqrt(C))
            if stress:
                gss[:, :, :2] += 1.0                # gamma1, gamma2 channels open
            else:
                gss[:, :, 2:] *= init_adaln; gss[:, :, :2] *= init_adaln_gamma   # var.py:309-311
            sd[p + "ada_gss"] = gss
            continue
        aw = w(p + "ada_lin.1.weight", (6 * C, C), 0.5 / math.sqrt(C))
        ab = torch.zeros(6 * C)
        if stress:
            ab[: 2 * C] = 1.0                       # gamma1, gamma2 channels open
        else:
            aw[2 * C:] *= init_adaln; aw[: 2 * C] *= init_adaln_gamma   # var.py:305-306
        sd[p + "ada_lin.1.bias"] = ab
    if shared_aln:
        sw = w("shared_ada_lin.1.weight", (6 * C, C), 0.5 / math.sqrt(C)); b("shared_ada_lin.1.bias", (6 * C,), 0.05)
        if not stress:
            sw[2 * C:] *= init_adaln; sw[: 2 * C] *= init_adaln_gamma   # var.py:299-302
    hw = w("head_nm.ada_lin.1.weight", (2 * C, C), 0.5 / math.sqrt(C)); b("head_nm.ada_lin.1.bias", (2 * C,), 0.1)
    hd = w("head.weight", (V, C), 2 / math.sqrt(C)); b("head.bias", (V,), 0.1)
    if not stress:
        hw *= init_adaln                             # var.py:290-293
        if init_head >= 0:
            hd *= init_head                          # var.py:283-289
    # buffers (models/var.py:108-113)
    lvl = torch.cat([torch.full((n,), i, dtype=torch.int64) for i, n in enumerate(lad.lens)]).view(1, L)
    sd["lvl_1L"] = lvl
    d = lvl.view(1, L, 1)
    sd["attn_bias_for_masking"] = torch.where(d >= d.transpose(1, 2), 0.0, -torch.inf).reshape(1, 1, L, L).contiguous()
    return sd


# ---- VQVAE ---------------------------------------------------------------------------------------------------------
def phi_names(share_quant_resi: int, n_scales: int, prefix: str = "quantize.quant_resi."):
    """Module names of the Phi convolutions for the three layouts of VectorQuantizer2 (quant.py:27-32): 0 = PhiNonShared (an nn.ModuleList: one Phi per scale,
    "<k>"), 1 = PhiShared (one Phi, "qresi"), >= 2 = PhiPartiallyShared ("qresi_ls.<k>", the released checkpoints use 4)."""
    if share_quant_resi == 0:
        return [f"{prefix}{k}" for k in range(n_scales)]
    if share_quant_resi == 1:
        return [f"{prefix}qresi"]
    return [f"{prefix}qresi_ls.{k}" for k in range(share_quant_resi)]


def phi_names_in(sd, prefix: str = "quantize.quant_resi."):
    """The Phi module names a VQVAE state_dict holds, whichever of the three layouts it was saved with ([] if none)."""
    if f"{prefix}qresi.weight" in sd:
        return [f"{prefix}qresi"]
    for stem in (f"{prefix}qresi_ls.", prefix):
        n = 0
        while f"{stem}{n}.weight" in sd:
            n += 1
        if n:
            return [f"{stem}{k}" for k in range(n)]
    return []


def _vae_shapes(V: int, Cvae: int, ch: int, patch_nums: Sequence[int], n_phi: int = 4, with_encoder: bool = True):
    """(name, shape, kind) for every VQVAE tensor; kind in conv_w|conv_b|gn_w|gn_b|emb|buf.
    Structure follows models/basic_vae.py:99-226 with ch_mult=(1,1,2,2,4), 2 res blocks, attention at the lowest
    resolution and in the middle (models/vqvae.py:32-38)."""
    ch_mult, nrb = (1, 1, 2, 2, 4), 2
    out = []

    def conv(name, cin, cout, k): out.append((name + ".weight", (cout, cin, k, k), "conv_w")); out.append((name + ".bias", (cout,), "conv_b"))
    def gn(name, c): out.append((name + ".weight", (c,), "gn_w")); out.append((name + ".bias", (c,), "gn_b"))

    def res(name, cin, cout):
        gn(name + ".norm1", cin); conv(name + ".conv1", cin, cout, 3); gn(name + ".norm2", cout); conv(name + ".conv2", cout, cout, 3)
        if cin != cout: conv(name + ".nin_shortcut", cin, cout, 1)

    def attn(name, c): gn(name + ".norm", c); conv(name + ".qkv", c, 3 * c, 1); conv(name + ".proj_out", c, c, 1)

    nres = len(ch_mult)
    if with_encoder:
        conv("encoder.conv_in", 3, ch, 3)
        in_mult = (1,) + ch_mult
        bi = ch
        for lv in range(nres):
            bi, bo = ch * in_mult[lv], ch * ch_mult[lv]
            for ib in range(nrb):
                res(f"encoder.down.{lv}.block.{ib}", bi, bo); bi = bo
                if lv == nres - 1: attn(f"encoder.down.{lv}.attn.{ib}", bi)
            if lv != nres - 1: conv(f"encoder.down.{lv}.downsample.conv", bi, bi, 3)
        res("encoder.mid.block_1", bi, bi); attn("encoder.mid.attn_1", bi); res("encoder.mid.block_2", bi, bi)
        gn("encoder.norm_out", bi); conv("encoder.conv_out", bi, Cvae, 3)
    bi = ch * ch_mult[-1]
    conv("decoder.conv_in", Cvae, bi, 3)
    res("decoder.mid.block_1", bi, bi); attn("decoder.mid.attn_1", bi); res("decoder.mid.block_2", bi, bi)
    for lv in reversed(range(nres)):
        bo = ch * ch_mult[lv]
        for ib in range(nrb + 1):
            res(f"decoder.up.{lv}.block.{ib}", bi, bo); bi = bo
            if lv == nres - 1: attn(f"decoder.up.{lv}.attn.{ib}", bi)
        if lv != 0: conv(f"decoder.up.{lv}.upsample.conv", bi, bi, 3)
    gn("decoder.norm_out", bi); conv("decoder.conv_out", bi, 3, 3)
    for name in phi_names(n_phi, len(patch_nums)): conv(name, Cvae, Cvae, 3)          # n_phi = share_quant_resi of the reference constructor
    out.append(("quantize.ema_vocab_hit_SV", (len(patch_nums), V), "buf"))
    out.append(("quantize.embedding.weight", (V, Cvae), "emb"))
    conv("quant_conv", Cvae, Cvae, 3); conv("post_quant_conv", Cvae, Cvae, 3)
    return out


def vae_state_dict(patch_nums: Sequence[int], mode: str = "perf", seed: int = 1234, V: int = 4096, Cvae: int = 32,
                   ch: int = 160, with_encoder: bool = True, share_quant_resi: int = 4) -> "OrderedDict[str, torch.Tensor]":
    stress = mode == "stress"
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape, kind in _vae_shapes(V, Cvae, ch, patch_nums, n_phi=share_quant_resi, with_encoder=with_encoder):
        g = _gen(seed * 31 + 17, name)
        if kind == "conv_w":
            fan_in = shape[1] * shape[2] * shape[3]
            sd[name] = _n(g, shape, 1 / math.sqrt(fan_in)) if stress else _tn(g, shape, 0.02)
        elif kind == "conv_b":
            sd[name] = torch.full(shape, 0.02) if stress else torch.zeros(shape)
        elif kind == "gn_w": sd[name] = torch.ones(shape)
        elif kind == "gn_b": sd[name] = torch.zeros(shape)
        elif kind == "emb": sd[name] = _n(g, shape, 1.0)
        elif kind == "buf": sd[name] = torch.zeros(shape)
    return sd


def var_state_dict_device(depth: int, patch_nums: Sequence[int], device, seed: int = 1234, V: int = 4096, Cvae: int = 32,
                          num_classes: int = 1000, mode: str = "perf") -> "OrderedDict[str, torch.Tensor]":
    """The 'perf' / 'stress' inits generated directly on the GPU (torch's device generator): same shapes and scales as
    var_state_dict, different values.  For benchmarks and full-size property tests only - fixture parity uses the
    portable host streams above."""
    lad = as_ladder(patch_nums)
    C, H, L, S = 64 * depth, depth, lad.L, lad.S
    g = torch.Generator(device=device); g.manual_seed(seed * 31 + depth)
    std0 = math.sqrt(1.0 / C / 3.0)
    stress = mode == "stress"
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def w(shape, s_std, scale=1.0):     # weight-like
        t = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        return t.mul_(s_std * scale) if stress else t.clamp_(-2.0, 2.0).mul_(std0 * scale)

    def b(shape, s_std):                # bias-like
        return torch.randn(shape, generator=g, device=device, dtype=torch.float32).mul_(s_std) if stress else torch.zeros(shape, device=device)

    sd["pos_start"], sd["pos_1LC"] = w((1, 1, C), 0.5), w((1, L, C), 0.5)
    sd["word_embed.weight"], sd["word_embed.bias"] = w((C, Cvae), 1 / math.sqrt(Cvae)), b((C,), 0.1)
    sd["class_emb.weight"], sd["lvl_embed.weight"] = w((num_classes + 1, C), 1.0), w((S, C), 0.5)
    for i in range(depth):
        p = f"blocks.{i}."
        sm = torch.full((1, H, 1, 1), math.log(4.0), device=device)
        sd[p + "attn.scale_mul_1H11"] = sm + (torch.randn((1, H, 1, 1), generator=g, device=device) * 0.3 if stress else 0.0)
        sd[p + "attn.q_bias"], sd[p + "attn.v_bias"] = b((C,), 0.1), b((C,), 0.1)
        sd[p + "attn.zero_k_bias"] = torch.zeros(C, device=device)
        sd[p + "attn.mat_qkv.weight"] = w((3 * C, C), 1 / math.sqrt(C))
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = w((C, C), 1 / math.sqrt(C), 1 / math.sqrt(2 * depth)), b((C,), 0.02)
        sd[p + "ffn.fc1.weight"], sd[p + "ffn.fc1.bias"] = w((4 * C, C), 1 / math.sqrt(C)), b((4 * C,), 0.1)
        sd[p + "ffn.fc2.weight"], sd[p + "ffn.fc2.bias"] = w((C, 4 * C), 1 / math.sqrt(4 * C), 1 / math.sqrt(2 * depth)), b((C,), 0.02)
        aw = w((6 * C, C), 0.5 / math.sqrt(C)); ab = torch.zeros(6 * C, device=device)
        if stress:
            ab[: 2 * C] = 1.0
        else:
            aw[2 * C:] *= 0.5; aw[: 2 * C] *= 1e-5
        sd[p + "ada_lin.1.weight"], sd[p + "ada_lin.1.bias"] = aw, ab
    sd["head_nm.ada_lin.1.weight"] = w((2 * C, C), 0.5 / math.sqrt(C), 1.0 if stress else 0.5)
    sd["head_nm.ada_lin.1.bias"] = b((2 * C,), 0.1)
    sd["head.weight"], sd["head.bias"] = w((V, C), 2 / math.sqrt(C), 1.0 if stress else 0.02), b((V,), 0.1)
    return sd
