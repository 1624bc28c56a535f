"""VQVAE parameter container with the reference's state_dict layout.

The quantizer tensors (codebook, shared Phi convs) feed sdvar_amd/csrc/quant.hip; the conv decoder `fhat_to_img`
(/root/reference/models/vqvae.py:62-63, models/basic_vae.py:163-226) runs as hand-written HIP
(csrc/conv.hip, csrc/vae.hip) through engine.VaeCtx.  The nn.Module tree below exists for the parameter names only - they follow
the upstream checkpoint `vae_ch160v4096z32.pth` so it loads unchanged.  No module here has a forward(): there is no torch math in this package
(the PyTorch decoder the GPU parity tests compare against lives in tests/torch_ref.py).  The encoder is parameters only (sampling never encodes).
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn


def _gn(c): return nn.GroupNorm(32, c, eps=1e-6, affine=True)


class _Res(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1, self.conv1 = _gn(cin), nn.Conv2d(cin, cout, 3, 1, 1)
        self.norm2, self.conv2 = _gn(cout), nn.Conv2d(cout, cout, 3, 1, 1)
        self.nin_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else nn.Identity()


class _Attn(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.C = c
        self.norm, self.qkv, self.proj_out = _gn(c), nn.Conv2d(c, 3 * c, 1), nn.Conv2d(c, c, 1)


class _Up(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 1, 1)


class _Down(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 2, 0)


class Decoder(nn.Module):
    def __init__(self, ch, ch_mult, nrb, z):
        super().__init__()
        nres = len(ch_mult)
        bi = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z, bi, 3, 1, 1)
        self.mid = nn.Module()
        self.mid.block_1, self.mid.attn_1, self.mid.block_2 = _Res(bi, bi), _Attn(bi), _Res(bi, bi)
        self.up = nn.ModuleList()
        for lv in reversed(range(nres)):
            bo = ch * ch_mult[lv]
            up = nn.Module()
            up.block, up.attn = nn.ModuleList(), nn.ModuleList()
            for _ in range(nrb + 1):
                up.block.append(_Res(bi, bo)); bi = bo
                if lv == nres - 1:
                    up.attn.append(_Attn(bi))
            if lv != 0:
                up.upsample = _Up(bi)
            self.up.insert(0, up)
        self.norm_out, self.conv_out = _gn(bi), nn.Conv2d(bi, 3, 3, 1, 1)


class Encoder(nn.Module):
    """Parameters only (state_dict compatibility); sampling never encodes."""
    def __init__(self, ch, ch_mult, nrb, z):
        super().__init__()
        nres = len(ch_mult)
        self.conv_in = nn.Conv2d(3, ch, 3, 1, 1)
        in_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        bi = ch
        for lv in range(nres):
            bi, bo = ch * in_mult[lv], ch * ch_mult[lv]
            dn = nn.Module()
            dn.block, dn.attn = nn.ModuleList(), nn.ModuleList()
            for _ in range(nrb):
                dn.block.append(_Res(bi, bo)); bi = bo
                if lv == nres - 1:
                    dn.attn.append(_Attn(bi))
            if lv != nres - 1:
                dn.downsample = _Down(bi)
            self.down.append(dn)
        self.mid = nn.Module()
        self.mid.block_1, self.mid.attn_1, self.mid.block_2 = _Res(bi, bi), _Attn(bi), _Res(bi, bi)
        self.norm_out, self.conv_out = _gn(bi), nn.Conv2d(bi, z, 3, 1, 1)


class _PhiList(nn.Module):          # PhiPartiallyShared (quant.py:219-229): quant_resi.qresi_ls.<k>
    def __init__(self, n, c):
        super().__init__()
        self.qresi_ls = nn.ModuleList([nn.Conv2d(c, c, 3, 1, 1) for _ in range(n)])


class _PhiOne(nn.Module):           # PhiShared (quant.py:209-216): quant_resi.qresi
    def __init__(self, c):
        super().__init__()
        self.qresi = nn.Conv2d(c, c, 3, 1, 1)


class Quantizer(nn.Module):
    """quantize.* tensors of the checkpoint (models/quant.py:15-43) in the layout `share_quant_resi` selects there (quant.py:27-32): 0 = one Phi per scale
    (PhiNonShared, an nn.ModuleList: quant_resi.<k>), 1 = one Phi for all (PhiShared), >= 2 = partially shared.  The arithmetic lives in csrc/quant.hip."""
    def __init__(self, vocab_size, Cvae, v_patch_nums, share_quant_resi=4):
        super().__init__()
        self.vocab_size, self.Cvae, self.v_patch_nums = vocab_size, Cvae, tuple(v_patch_nums)
        if share_quant_resi == 0:
            self.quant_resi = nn.ModuleList([nn.Conv2d(Cvae, Cvae, 3, 1, 1) for _ in range(len(v_patch_nums))])
        elif share_quant_resi == 1:
            self.quant_resi = _PhiOne(Cvae)
        else:
            self.quant_resi = _PhiList(share_quant_resi, Cvae)
        self.register_buffer("ema_vocab_hit_SV", torch.zeros(len(v_patch_nums), vocab_size))
        self.embedding = nn.Embedding(vocab_size, Cvae)


class VQVAE(nn.Module):
    def __init__(self, vocab_size=4096, z_channels=32, ch=128, share_quant_resi=4, v_patch_nums: Sequence[int] = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16),
                 test_mode=True, with_encoder=True, **_unused):
        super().__init__()
        self.V = self.vocab_size = vocab_size
        self.Cvae = z_channels
        ch_mult, nrb = (1, 1, 2, 2, 4), 2
        self._ch_mult, self._nrb, self._hip_ctx = ch_mult, nrb, None
        if with_encoder:
            self.encoder = Encoder(ch, ch_mult, nrb, z_channels)
        self.decoder = Decoder(ch, ch_mult, nrb, z_channels)
        self.downsample = 2 ** (len(ch_mult) - 1)
        self.quantize = Quantizer(vocab_size, z_channels, v_patch_nums, share_quant_resi)
        self.quant_conv = nn.Conv2d(z_channels, z_channels, 3, 1, 1)
        self.post_quant_conv = nn.Conv2d(z_channels, z_channels, 3, 1, 1)
        if test_mode:
            self.eval()
            for p in self.parameters():
                p.requires_grad_(False)

    @torch.no_grad()
    def fhat_to_img(self, f_hat: torch.Tensor) -> torch.Tensor:      # vqvae.py:62-63
        """(B, Cvae, h, w) -> (B, 3, 16h, 16w) in [-1, 1] on the HIP decoder.  GPU tensors only: there is no CPU path."""
        from . import engine as E
        if not f_hat.is_cuda:
            raise E.SdvarError("VQVAE.fhat_to_img runs on the HIP decoder and needs a GPU tensor (tests/torch_ref.py holds the PyTorch test reference)")
        B, hw = f_hat.shape[0], f_hat.shape[-1]
        ctx = self._hip_ctx
        if ctx is None or ctx.device != f_hat.device or ctx.max_batch < B or ctx.latent_hw != hw:
            if ctx is not None:
                ctx.close()
            sd = {k: v for k, v in self.state_dict().items() if k.startswith(("decoder.", "post_quant_conv."))}
            ctx = self._hip_ctx = E.VaeCtx(sd, B, f_hat.device, latent_hw=hw, ch_mult=self._ch_mult, num_res_blocks=self._nrb)
        return ctx.decode(f_hat)

    def refresh_hip(self):
        """Drop the HIP decoder's copy of the weights (call after changing decoder parameters in place)."""
        if self._hip_ctx is not None:
            self._hip_ctx.close()
        self._hip_ctx = None

    def load_state_dict(self, state_dict, strict=True, assign=False):  # vqvae.py:92-95
        key = "quantize.ema_vocab_hit_SV"
        if key in state_dict and state_dict[key].shape[0] != self.quantize.ema_vocab_hit_SV.shape[0]:
            state_dict[key] = self.quantize.ema_vocab_hit_SV
        self.refresh_hip()
        return super().load_state_dict(state_dict, strict=strict, assign=assign)
