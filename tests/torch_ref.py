"""PyTorch restatements used ONLY by tests and tools as a second, independent checker next to oracle/ (test infrastructure, never imported by sdvar_amd/).

`fhat_to_img_torch(vae, f_hat)`: the VQVAE decoder (/root/reference/models/vqvae.py:62-63, models/basic_vae.py:163-226) on torch ops, reading the parameters of
an `sdvar_amd.vqvae.VQVAE` (whose modules are parameter containers without forward()).  On a GPU tensor it runs on MIOpen - the decoder A/B of tools/decode_bench.py.
"""
import torch
import torch.nn.functional as F


def _gn(m, x):
    return F.group_norm(x, m.num_groups, m.weight, m.bias, m.eps)


def _conv(m, x):
    return F.conv2d(x, m.weight, m.bias, m.stride, m.padding)


def _res(b, x):                                        # basic_vae.py:42-64
    h = _conv(b.conv1, F.silu(_gn(b.norm1, x)))
    h = _conv(b.conv2, F.silu(_gn(b.norm2, h)))
    sc = x if isinstance(b.nin_shortcut, torch.nn.Identity) else _conv(b.nin_shortcut, x)
    return sc + h


def _attn(a, x):                                       # basic_vae.py:67-91
    B, C, H, W = x.shape
    q, k, v = _conv(a.qkv, _gn(a.norm, x)).reshape(B, 3, C, H * W).unbind(1)
    w = torch.bmm(q.transpose(1, 2), k).mul_(C ** -0.5).softmax(dim=2)
    h = torch.bmm(v, w.transpose(1, 2)).view(B, C, H, W)
    return x + _conv(a.proj_out, h)


@torch.no_grad()
def decoder_torch(dec, z):                             # basic_vae.py:163-226
    h = _res(dec.mid.block_2, _attn(dec.mid.attn_1, _res(dec.mid.block_1, _conv(dec.conv_in, z))))
    for lv in reversed(range(len(dec.up))):
        up = dec.up[lv]
        for ib, blk in enumerate(up.block):
            h = _res(blk, h)
            if len(up.attn):
                h = _attn(up.attn[ib], h)
        if lv != 0:
            h = _conv(up.upsample.conv, F.interpolate(h, scale_factor=2, mode="nearest"))
    return _conv(dec.conv_out, F.silu(_gn(dec.norm_out, h)))


@torch.no_grad()
def fhat_to_img_torch(vae, f_hat):                     # vqvae.py:62-63
    return decoder_torch(vae.decoder, _conv(vae.post_quant_conv, f_hat)).clamp_(-1, 1)
