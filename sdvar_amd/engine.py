"""ctypes binding of libsdvar_hip.so (include/sdvar_hip.h) and the host-side sampling loops built on it.

PyTorch is plumbing here: it owns the weight tensors and a few staging buffers (device memory + streams); every
operation of the draft -> verify loop is a call into the C ABI.  There is NO fallback: if the shared library is missing
or a call fails this module raises - the product path never routes through torch ops or the CPU oracle.

Reference loops restated on top of the ABI:
  * `Sampler.plain_ar`  - VAR.autoregressive_infer_cfg         (/root/reference/models/var.py:127-215)
  * `Sampler.spec_decode` - SDVAR.sdvar_autoregressive_infer_cfg_parallel_v1 with the resolved semantics of
    SURVEY.md App. C.1 (models/var.py:949-1070 draft/verify, 1160-1227 acceptance, 1318-1372 loop policy).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from .ladder import Ladder, as_ladder
from .noise import exponential_noise

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libsdvar_hip.so")
MAX_STAGES = 16
ABI_VERSION = 4
# GEMM arithmetic of the transformer blocks: 'f32' = fp32-in/fp32-accumulate MFMA; 'bf16x3' = exact 3-way bf16 split of both
# operands, 6 bf16 MFMA products, fp32 accumulate (fp32-accurate, 2.67x the matrix-pipe throughput).  See DESIGN.md section 4.
#   'f16x2' = two fp16 planes per operand (x ~ xh + xl to 2^-22), 3 fp16 MFMA products, fp32 accumulate: half the matrix work of bf16x3 at the same
#             measured accuracy while activations stay inside the fp16 range (csrc/gemm_f16x2.hip).
GEMM_MODES = ("f32", "bf16x3", "f16x2")
DEFAULT_GEMM_MODE = "f16x2"
PROF_CLASSES = ("gemm", "attention", "ln_modulate", "qk_norm_append", "sampler", "verify", "quant", "embed_misc", "attention_small", "gemm_small")


class SdvarError(RuntimeError):
    pass


class _ModelDesc(C.Structure):
    _fields_ = [("depth", C.c_int32), ("n_stages", C.c_int32), ("patch_nums", C.c_int32 * MAX_STAGES), ("vocab", C.c_int32),
                ("cvae", C.c_int32), ("num_classes", C.c_int32), ("max_batch", C.c_int32), ("max_chunk_stages", C.c_int32), ("kv_dtype", C.c_int32), ("gemm_mode", C.c_int32)]


class _VaeDesc(C.Structure):
    _fields_ = [("ch", C.c_int32), ("z_channels", C.c_int32), ("n_mult", C.c_int32), ("ch_mult", C.c_int32 * 8), ("num_res_blocks", C.c_int32),
                ("max_batch", C.c_int32), ("latent_hw", C.c_int32), ("plane_format", C.c_int32)]


_P, _I, _D, _U64, _U32 = C.c_void_p, C.c_int32, C.c_double, C.c_uint64, C.c_uint32
# name -> (restype, argtypes); must list every symbol declared in include/sdvar_hip.h (tests/test_abi.py checks it)
_SIGNATURES = {
    "sdvar_abi_version": (_I, []),
    "sdvar_last_error": (C.c_char_p, []),
    "sdvar_model_create": (_I, [C.POINTER(_ModelDesc), C.POINTER(_P)]),
    "sdvar_model_destroy": (_I, [_P]),
    "sdvar_model_bind_embed": (_I, [_P] * 8),
    "sdvar_model_bind_block": (_I, [_P, _I] + [_P] * 13),
    "sdvar_model_bind_shared_aln": (_I, [_P, _P, _P]),
    "sdvar_model_bind_head": (_I, [_P] * 6),
    "sdvar_model_begin": (_I, [_P, _I, _P, _P]),
    "sdvar_model_begin_cond": (_I, [_P, _I, _P, _P]),
    "sdvar_model_export_prologue": (_I, [_P, _P, _P, _P, _P]),
    "sdvar_model_place_first": (_I, [_P, _P, _I, _P]),
    "sdvar_kv_len": (_I, [_P]),
    "sdvar_kv_set_len": (_I, [_P, _I]),
    "sdvar_kv_set_origin": (_I, [_P, _I]),
    "sdvar_head_forward": (_I, [_P, _P, _I, _P, _P]),
    "sdvar_embed_next": (_I, [_P, _P, _I, _P, _I, _I, _P]),
    "sdvar_embed_next_at": (_I, [_P, _P, _I, _I, _P, _I, _I, _P]),
    "sdvar_stage_forward": (_I, [_P, _P, _I, _I, _P, _P]),
    "sdvar_stage_forward_masked": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "sdvar_quant_create": (_I, [_I, C.POINTER(_I), _I, _I, _I, _I, C.POINTER(_P)]),
    "sdvar_quant_destroy": (_I, [_P]),
    "sdvar_quant_bind": (_I, [_P, _P, C.POINTER(_P), C.POINTER(_P)]),
    "sdvar_quant_next": (_I, [_P, _I, _P, _I, _P, _P, _I, _P]),
    "sdvar_quant_next_from": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _P]),
    "sdvar_quant_next_h": (_I, [_P, _I, _P, _P, _P, _I, _P]),
    "sdvar_gumbel_mix": (_I, [_P, _P, _I, _I, _D, _D, _P, _U64, _U32, _U32, _P, _P]),
    "sdvar_vae_create": (_I, [C.POINTER(_VaeDesc), C.POINTER(_P)]),
    "sdvar_vae_destroy": (_I, [_P]),
    "sdvar_vae_tensor_count": (_I, [C.POINTER(_VaeDesc)]),
    "sdvar_vae_bind": (_I, [_P, C.POINTER(_P), _I, _P]),
    "sdvar_vae_decode": (_I, [_P, _P, _I, _P, _P]),
    "sdvar_cfg_sample": (_I, [_P, _I, _I, _I, _D, _I, _D, _P, _U64, _U32, _U32, _P, _I, _P, _P]),
    "sdvar_verify_accept": (_I, [_P, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_D), _P, _I, _D, _P, _P, _P]),
    "sdvar_cfg_combine": (_I, [_P, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_D), _P, _P]),
    "sdvar_verify_accept_ex": (_I, [_P, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_D), _P, _I, _D, _I, _I, _D, _P, _P, _P, _P, _P, _P]),
    "sdvar_op_gemm": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P]),
    "sdvar_op_ln_modulate": (_I, [_P, _P, _P, _P, _P, _U64, _I, _I, _I, _I, _I, _P]),
    "sdvar_op_split_planes_f16": (_I, [_P, _P, _I, _I, _U64, _P, _P]),
    "sdvar_op_gemm_f16x2": (_I, [_P, _U64, _P, _U64, _P, _P, _P, _I, _P, _U64, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P]),
    "sdvar_op_split_planes": (_I, [_P, _P, _I, _I, _U64, _P]),
    "sdvar_op_gemm_bf16x3": (_I, [_P, _U64, _P, _U64, _P, _P, _I, _P, _U64, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P]),
    "sdvar_op_qk_norm_append": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "sdvar_op_attention": (_I, [_P, _P, _P, _I, _P, _P, _U64, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), _P]),
    "sdvar_op_conv_weight_planes": (_I, [_P, _P, _I, _I, _I, _U64, _I, _P, _P]),
    "sdvar_op_vae_prep": (_I, [_P, _P, _P, _P, _P, _U64, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sdvar_op_conv_planes": (_I, [_P, _U64, _U64, _I, _P, _U64, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _U64, _I, _P]),
    "sdvar_op_noise_fill": (_I, [_P, _I, _I, _I, _U64, _U32, _U32, _P]),
    "sdvar_debug_set_gemm_cfg": (_I, [_I, _I]),
    "sdvar_debug_set_gemm_stamps": (_I, [_P]),
    "sdvar_debug_set_qkv_fuse": (_I, [_I]),
    "sdvar_debug_set_rowblk": (_I, [_I]),
    "sdvar_op_gemm_rowblk": (_I, [_P, _I, _P, _P, _I, _I, _P, _U64, _P, _U64, _P, _P, _P, _I, _P, _U64, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sdvar_debug_set_variant": (_I, [C.c_char_p, _I]),
    "sdvar_debug_get_gemm_cfg": (_I, [C.POINTER(_I)]),
    "sdvar_debug_set_f16x2_guard": (_I, [_I]),
    "sdvar_debug_get_f16x2_guard": (_I, [C.POINTER(_U64), _I]),
    "sdvar_prof_enable": (_I, [_I]),
    "sdvar_prof_collect": (_I, [C.POINTER(_D), C.POINTER(C.c_int64), C.POINTER(_D), C.POINTER(_D)]),
}

_lib = None


def load_library(path: str = LIB_PATH):
    """dlopen the HIP library and attach prototypes.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise SdvarError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         f"(make -C sdvar_amd/csrc); there is no CPU fallback for the sampler")
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.sdvar_abi_version() != ABI_VERSION:
        raise SdvarError("libsdvar_hip.so ABI version mismatch")
    _lib = lib
    return lib


def _check(rc: int):
    if rc != 0:
        raise SdvarError(f"libsdvar_hip error {rc}: {_lib.sdvar_last_error().decode()}")


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device tensors must be contiguous CUDA(HIP) tensors"
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t: torch.Tensor, dev) -> torch.Tensor:
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


# ------------------------------------------------------------------------------------------------------- objects
class ModelCtx:
    """sdvar_model_t for one VAR transformer given its state_dict (reference key names)."""

    def __init__(self, sd: Dict[str, torch.Tensor], depth: int, patch_nums: Sequence[int], max_batch: int, max_chunk: int,
                 device, num_classes: int = 1000, kv_fp16: bool = False, gemm_mode: Optional[str] = None):
        self.lib = load_library()
        self.lad = as_ladder(patch_nums)
        self.depth, self.Cw, self.H = depth, 64 * depth, depth
        self.V = sd["head.weight"].shape[0]
        self.device = torch.device(device)
        self.max_batch, self.max_chunk = max_batch, max_chunk
        d = _ModelDesc()
        d.depth, d.n_stages, d.vocab, d.cvae, d.num_classes = depth, self.lad.S, self.V, sd["word_embed.weight"].shape[1], num_classes
        d.max_batch, d.max_chunk_stages, d.kv_dtype = max_batch, max_chunk, (1 if kv_fp16 else 0)
        self.kv_fp16 = bool(kv_fp16)
        self.gemm_mode = gemm_mode or os.environ.get("SDVAR_GEMM_MODE", DEFAULT_GEMM_MODE)
        if self.gemm_mode not in GEMM_MODES:
            raise SdvarError(f"gemm_mode {self.gemm_mode!r}: expected one of {GEMM_MODES}")
        d.gemm_mode = GEMM_MODES.index(self.gemm_mode)
        for i, p in enumerate(self.lad.patch_nums):
            d.patch_nums[i] = p
        self.h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib.sdvar_model_create(C.byref(d), C.byref(self.h)))
            self._keep: List[torch.Tensor] = []
            self.bind(sd)

    def bind(self, sd: Dict[str, torch.Tensor]):
        dev, k = self.device, self._keep
        k.clear()
        def w(name):
            t = _f32(sd[name], dev); k.append(t); return _ptr(t)
        st = _stream()
        _check(self.lib.sdvar_model_bind_embed(self.h, w("class_emb.weight"), w("pos_start"), w("pos_1LC"), w("lvl_embed.weight"),
                                               w("word_embed.weight"), w("word_embed.bias"), st))
        shared = "shared_ada_lin.1.weight" in sd                 # shared_aln=True checkpoints (VAR-d36-s): var.py:16-19, 81
        if shared:
            _check(self.lib.sdvar_model_bind_shared_aln(self.h, w("shared_ada_lin.1.weight"), w("shared_ada_lin.1.bias")))
        for i in range(self.depth):
            p = f"blocks.{i}."
            _check(self.lib.sdvar_model_bind_block(
                self.h, i, None if shared else w(p + "ada_lin.1.weight"), w(p + "ada_gss") if shared else w(p + "ada_lin.1.bias"),
                w(p + "attn.mat_qkv.weight"), w(p + "attn.q_bias"),
                w(p + "attn.v_bias"), w(p + "attn.scale_mul_1H11") if (p + "attn.scale_mul_1H11") in sd else None,      # absent: attn_l2_norm=False (basic_var.py:66-72)
                w(p + "attn.proj.weight"), w(p + "attn.proj.bias"),
                w(p + "ffn.fc1.weight"), w(p + "ffn.fc1.bias"), w(p + "ffn.fc2.weight"), w(p + "ffn.fc2.bias"), st))
        _check(self.lib.sdvar_model_bind_head(self.h, w("head_nm.ada_lin.1.weight"), w("head_nm.ada_lin.1.bias"), w("head.weight"), w("head.bias"), st))

    # thin wrappers ----------------------------------------------------------------------------------------------
    def begin(self, labels: torch.Tensor):
        assert labels.dtype == torch.int64 and labels.is_cuda
        self.B = labels.shape[0]
        _check(self.lib.sdvar_model_begin(self.h, self.B, _ptr(labels), _stream()))

    def begin_cond(self, cond: torch.Tensor):
        """The prologue from the conditioning rows (2B, C) themselves (`sos` of var.py:319-345) instead of labels."""
        assert cond.dtype == torch.float32 and cond.is_cuda and cond.dim() == 2 and cond.shape[1] == self.Cw and cond.shape[0] % 2 == 0
        self.B = cond.shape[0] // 2
        _check(self.lib.sdvar_model_begin_cond(self.h, self.B, _ptr(cond.contiguous()), _stream()))

    def export_prologue(self):
        """(cond (2B,C), lvl_pos (1,L,C), first_token_map (2B,1,C)) of the current call, as SDVAR.init_param returns them."""
        f = dict(device=self.device, dtype=torch.float32)
        cond, lvl, first = torch.empty(2 * self.B, self.Cw, **f), torch.empty(1, self.lad.L, self.Cw, **f), torch.empty(2 * self.B, 1, self.Cw, **f)
        _check(self.lib.sdvar_model_export_prologue(self.h, _ptr(cond), _ptr(lvl), _ptr(first), _stream()))
        return cond, lvl, first

    def place_first(self, x: torch.Tensor, ltot: int):
        _check(self.lib.sdvar_model_place_first(self.h, _ptr(x), ltot, _stream()))

    def embed_next(self, nxt: torch.Tensor, s_next: int, x: torch.Tensor, ltot: int, tok_off: int):
        _check(self.lib.sdvar_embed_next(self.h, _ptr(nxt), s_next, _ptr(x), ltot, tok_off, _stream()))

    def embed_next_at(self, nxt: torch.Tensor, s_next: int, pos_begin: int, x: torch.Tensor, ltot: int, tok_off: int):
        """embed_next with explicit lvl_pos rows (the resumed sampler of var.py:319-443 counts positions from the stage it starts at)."""
        _check(self.lib.sdvar_embed_next_at(self.h, _ptr(nxt), s_next, pos_begin, _ptr(x), ltot, tok_off, _stream()))

    def forward(self, x: torch.Tensor, s0: int, n: int, logits: torch.Tensor, bias: Optional[torch.Tensor] = None):
        """All blocks + head over stages s0 .. s0+n-1.  bias: explicit additive mask (l, K) instead of the block-causal rows (mask ablations)."""
        if bias is None:
            _check(self.lib.sdvar_stage_forward(self.h, _ptr(x), s0, n, _ptr(logits), _stream()))
        else:
            _check(self.lib.sdvar_stage_forward_masked(self.h, _ptr(x), s0, n, _ptr(bias), _ptr(logits), _stream()))

    def kv_len(self) -> int:
        return self.lib.sdvar_kv_len(self.h)

    def kv_set_len(self, n: int):
        _check(self.lib.sdvar_kv_set_len(self.h, n))

    def kv_set_origin(self, stage: int):
        """Empty cache whose first key will be the first token of `stage` (hand-off sampler, var.py:817-824)."""
        _check(self.lib.sdvar_kv_set_origin(self.h, stage))

    def head_forward(self, x: torch.Tensor, l: int, logits: torch.Tensor):
        """VAR.get_logits (var.py:119-125) on x (2B, l, C) -> logits (2B, l, V)."""
        _check(self.lib.sdvar_head_forward(self.h, _ptr(x), l, _ptr(logits), _stream()))

    def close(self):
        if self.h:
            self.lib.sdvar_model_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class QuantCtx:
    """sdvar_quant_t from the VQVAE state_dict (quantize.embedding / quantize.quant_resi.*, any of the three Phi layouts)."""

    def __init__(self, vae_sd: Dict[str, torch.Tensor], patch_nums: Sequence[int], max_batch: int, device, prefix: str = "quantize."):
        self.lib = load_library()
        self.lad = as_ladder(patch_nums)
        self.device = torch.device(device)
        self.codebook = _f32(vae_sd[prefix + "embedding.weight"], self.device)
        self.V, self.Cv = self.codebook.shape
        from .weights import phi_names_in
        names = phi_names_in(vae_sd, prefix + "quant_resi.")          # PhiPartiallyShared (qresi_ls.<k>), PhiShared (qresi) or PhiNonShared (<k>): quant.py:27-32
        n_phi = len(names)
        if n_phi == 0:
            raise SdvarError(f"no Phi convolution under {prefix}quant_resi.* in the state_dict")
        self.pw = [_f32(vae_sd[n + ".weight"], self.device) for n in names]
        self.pb = [_f32(vae_sd[n + ".bias"], self.device) for n in names]
        pn = (_I * self.lad.S)(*self.lad.patch_nums)
        self.h = C.c_void_p()
        self.max_batch = int(max_batch)
        with torch.cuda.device(self.device):
            _check(self.lib.sdvar_quant_create(self.lad.S, pn, self.Cv, self.V, max_batch, n_phi, C.byref(self.h)))
        aw = (_P * n_phi)(*[t.data_ptr() for t in self.pw]); ab = (_P * n_phi)(*[t.data_ptr() for t in self.pb])
        _check(self.lib.sdvar_quant_bind(self.h, _ptr(self.codebook), aw, ab))

    def next(self, si: int, ids: torch.Tensor, ids_stride: int, f_hat: torch.Tensor, nxt: Optional[torch.Tensor], B: int, f_in: Optional[torch.Tensor] = None):
        """quant.py:187-196 for stage si.  f_in given: f_hat = f_in + Phi(...) (f_in is left untouched), else f_hat += Phi(...)."""
        nx = _ptr(nxt) if nxt is not None else None
        if f_in is None:
            _check(self.lib.sdvar_quant_next(self.h, si, C.c_void_p(ids.data_ptr()), ids_stride, _ptr(f_hat), nx, B, _stream()))
        else:
            _check(self.lib.sdvar_quant_next_from(self.h, si, C.c_void_p(ids.data_ptr()), ids_stride, _ptr(f_in), _ptr(f_hat), nx, B, _stream()))

    def next_h(self, si: int, h: torch.Tensor, f_hat: torch.Tensor, nxt: Optional[torch.Tensor], B: int):
        """The same from feature vectors h (B, pn^2, Cvae) instead of ids (more_smooth=True, var.py:206-210)."""
        _check(self.lib.sdvar_quant_next_h(self.h, si, _ptr(h), _ptr(f_hat), _ptr(nxt) if nxt is not None else None, B, _stream()))

    def gumbel_mix(self, masked: torch.Tensor, B: int, l: int, ratio: float, tau: float, e: Optional[torch.Tensor], seed: int, draw: int, image_offset: int,
                   h_out: torch.Tensor):
        """var.py:206-208 + helpers.py:22-36: soft codebook mix of the masked CFG logits under gumbel noise -> h (B, l, Cvae)."""
        _check(self.lib.sdvar_gumbel_mix(self.h, _ptr(masked), B, l, float(ratio), float(tau), _ptr(e) if e is not None else None, seed & (2**64 - 1), draw,
                                         image_offset, _ptr(h_out), _stream()))

    def close(self):
        if self.h:
            self.lib.sdvar_quant_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VaeCtx:
    """sdvar_vae_t from the VQVAE state_dict: `fhat_to_img` (vqvae.py:62-63) as hand-written HIP (csrc/conv.hip, csrc/vae.hip)."""

    def __init__(self, vae_sd: Dict[str, torch.Tensor], max_batch: int, device, latent_hw: int = 16, ch: Optional[int] = None,
                 ch_mult: Sequence[int] = (1, 1, 2, 2, 4), num_res_blocks: int = 2, conv_mode: Optional[str] = None):
        self.lib = load_library()
        self.device = torch.device(device)
        z = vae_sd["post_quant_conv.weight"].shape[0]
        ch = ch if ch is not None else vae_sd["decoder.norm_out.weight"].shape[0] // ch_mult[0]
        d = _VaeDesc()
        d.ch, d.z_channels, d.n_mult, d.num_res_blocks, d.max_batch, d.latent_hw = ch, z, len(ch_mult), num_res_blocks, max_batch, latent_hw
        self.conv_mode = conv_mode or os.environ.get("SDVAR_CONV_MODE", DEFAULT_GEMM_MODE if DEFAULT_GEMM_MODE != "f32" else "f16x2")   # operands of the convolutions
        if self.conv_mode not in ("bf16x3", "f16x2"):
            raise SdvarError(f"conv_mode {self.conv_mode!r}: expected 'bf16x3' or 'f16x2'")
        d.plane_format = 3 if self.conv_mode == "bf16x3" else 2
        for i, m in enumerate(ch_mult):
            d.ch_mult[i] = m
        self.desc, self.max_batch, self.latent_hw, self.z = d, max_batch, latent_hw, z
        self.out_hw = latent_hw << (len(ch_mult) - 1)
        names = self.tensor_names(vae_sd, ch_mult, num_res_blocks)
        self.tensors = []                                   # keeps the device copies alive: biases and GroupNorm affine stay borrowed
        for n in names:
            self.tensors += [_f32(vae_sd[n + ".weight"], self.device), _f32(vae_sd[n + ".bias"], self.device)]
        want = self.lib.sdvar_vae_tensor_count(C.byref(d))
        if want != len(self.tensors):
            raise SdvarError(f"VQVAE decoder layout mismatch: the library expects {want} tensors, the state_dict walk found {len(self.tensors)}")
        self.h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib.sdvar_vae_create(C.byref(d), C.byref(self.h)))
            arr = (_P * len(self.tensors))(*[t.data_ptr() for t in self.tensors])
            _check(self.lib.sdvar_vae_bind(self.h, arr, len(self.tensors), _stream()))

    @staticmethod
    def tensor_names(vae_sd, ch_mult: Sequence[int] = (1, 1, 2, 2, 4), num_res_blocks: int = 2):
        """Module prefixes (each contributes .weight then .bias) in the order sdvar_vae_bind consumes them (include/sdvar_hip.h)."""
        names = ["post_quant_conv", "decoder.conv_in"]
        res = lambda p, sc: [p + ".norm1", p + ".conv1", p + ".norm2", p + ".conv2"] + ([p + ".nin_shortcut"] if sc else [])
        att = lambda p: [p + ".norm", p + ".qkv", p + ".proj_out"]
        names += res("decoder.mid.block_1", False) + att("decoder.mid.attn_1") + res("decoder.mid.block_2", False)
        for lv in reversed(range(len(ch_mult))):
            for i in range(num_res_blocks + 1):
                p = f"decoder.up.{lv}.block.{i}"
                names += res(p, p + ".nin_shortcut.weight" in vae_sd)
                if lv == len(ch_mult) - 1:
                    names += att(f"decoder.up.{lv}.attn.{i}")
            if lv != 0:
                names.append(f"decoder.up.{lv}.upsample.conv")
        return names + ["decoder.norm_out", "decoder.conv_out"]

    def decode(self, f_hat: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """f_hat (B, Cvae, h, w) fp32 on the device -> image (B, 3, H, W) in [-1, 1]; runs on torch's current stream."""
        B = f_hat.shape[0]
        if f_hat.dtype != torch.float32 or not f_hat.is_contiguous() or f_hat.device != self.device:
            f_hat = f_hat.to(device=self.device, dtype=torch.float32).contiguous()
        if tuple(f_hat.shape[1:]) != (self.z, self.latent_hw, self.latent_hw) or B > self.max_batch:
            raise SdvarError(f"decode: f_hat {tuple(f_hat.shape)} does not fit (max_batch {self.max_batch}, {self.z} x {self.latent_hw}^2)")
        if out is None:
            out = torch.empty(B, 3, self.out_hw, self.out_hw, device=self.device, dtype=torch.float32)
        _check(self.lib.sdvar_vae_decode(self.h, _ptr(f_hat), B, _ptr(out), _stream()))
        return out

    def close(self):
        if self.h:
            self.lib.sdvar_vae_destroy(self.h); self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------------- noise
class Noise:
    """Where the Exp(1) draw noise q comes from (sdvar_amd/noise.py explains why it is explicit).
    kind = 'device': Philox generated inside the sampler kernel (fast path, no host traffic);
           'host'  : the same Philox stream computed on the host in float64 and uploaded (portable parity mode);
           'torch' : torch.empty(B*l, V).exponential_(generator=cpu_gen), the reference's own CPU stream on this host;
           callable: fn(draw, B, l, V) -> float32 array/tensor (B*l, V) or (B, l, V)."""

    def __init__(self, kind="device", seed: int = 0, image_offset: int = 0, fn: Optional[Callable] = None, generator: Optional[torch.Generator] = None):
        self.kind, self.seed, self.image_offset, self.fn, self.gen = kind, int(seed), int(image_offset), fn, generator
        if kind == "torch" and generator is None:
            self.gen = torch.Generator(device="cpu"); self.gen.manual_seed(self.seed)

    def tensor(self, draw: int, B: int, l: int, V: int, device) -> Optional[torch.Tensor]:
        if self.kind == "device":
            return None
        if self.kind == "host":
            q = torch.from_numpy(exponential_noise(self.seed, draw, B, l, V, self.image_offset))
        elif self.kind == "torch":
            q = torch.empty(B * l, V).exponential_(1, generator=self.gen)
        else:
            q = self.fn(draw, B, l, V)
            q = torch.from_numpy(np.ascontiguousarray(q)) if isinstance(q, np.ndarray) else q
        return q.reshape(B * l, V).to(device=device, dtype=torch.float32, non_blocking=False).contiguous()


def cfg_sample(logits: torch.Tensor, B: int, l: int, V: int, t: float, top_k: int, top_p: float, q: Optional[torch.Tensor], seed: int, draw: int,
               image_offset: int, ids_out: torch.Tensor, ids_off: int, ids_stride: int, dbg_masked: Optional[torch.Tensor] = None):
    lib = load_library()
    _check(lib.sdvar_cfg_sample(_ptr(logits), B, l, V, float(t), int(top_k), float(top_p), _ptr(q) if q is not None else None, seed & (2**64 - 1),
                                draw, image_offset, C.c_void_p(ids_out.data_ptr() + 8 * ids_off), ids_stride,
                                _ptr(dbg_masked) if dbg_masked is not None else None, _stream()))


def handoff_mask(lad: Ladder, entry_num: int, sd_mask: int) -> torch.Tensor:
    """The (pindex, pindex) additive mask of SDVAR.sdvar_autoregressive_infer_cfg_sd_test3 for sd_mask 1, 2, 4, 5 (var.py:557-578, 777-798), pindex =
    tokens of stages 0 .. entry_num.  attn_bias_for_sdmasking: token i sees itself and every token of EARLIER stages (not its own stage's other
    tokens); attn_bias_for_block: token i sees exactly its own stage.  Masks 2 and 5 open the entry stage's rows completely."""
    p, s0 = lad.cum[entry_num], lad.begin(entry_num)
    blk = torch.cat([torch.full((n,), i) for i, n in enumerate(lad.lens)])[:p]
    bi, bj = blk.view(p, 1), blk.view(1, p)
    if sd_mask in (1, 2):
        ok = (bj < bi) | torch.eye(p, dtype=torch.bool)
    else:
        ok = bi == bj
    m = torch.where(ok, 0.0, float("-inf")).to(torch.float32)
    if sd_mask in (2, 5):
        m[s0:p, :] = 0.0
    return m.contiguous()


GUMBEL_DRAW = 0x40000000      # Philox `draw` of a stage's gumbel noise = its sampler draw | GUMBEL_DRAW (the reference draws it right after the multinomial)


@dataclass
class MatchRule:
    """Token rule of the acceptance scan.  'top1' is basic_token_matching (var.py:1199-1203, what the reference's advanced_token_matching
    stub falls back to); 'topk' / 'kl' / token_level are the rules its docstring sketches (var.py:1229-1243), reference-unpinned:
      topk        draft id among the target's `top_k` best CFG logits
      kl          KL(softmax target || softmax draft) of the token's two CFG distributions <= kl_thr
      token_level the first stage that fails the batch threshold is not re-drafted: its matching tokens are kept, the others take the
                  target's argmax, and the stage is committed (the target contributes tokens; no gamma decay, no forced accepts)."""
    rule: str = "top1"
    top_k: int = 1
    kl_thr: float = 0.0
    token_level: bool = False

    @property
    def code(self) -> int:
        return {"top1": 0, "topk": 1, "kl": 2}[self.rule]

    @property
    def basic(self) -> bool:
        return self.rule == "top1" and not self.token_level


def verify_accept(logits: torch.Tensor, B: int, lens: Sequence[int], V: int, ts: Sequence[float], ids: torch.Tensor, ids_off: int, ids_stride: int,
                  thr: float, counts: torch.Tensor, argmax_out: Optional[torch.Tensor] = None, rule: Optional[MatchRule] = None,
                  draft_logits: Optional[torch.Tensor] = None, match_out: Optional[torch.Tensor] = None, corrected_out: Optional[torch.Tensor] = None):
    lib = load_library()
    n = len(lens)
    o = lambda t: _ptr(t) if t is not None else None
    r = rule or MatchRule()
    _check(lib.sdvar_verify_accept_ex(_ptr(logits), B, int(sum(lens)), V, n, (_I * n)(*lens), (_D * n)(*[float(t) for t in ts]),
                                      C.c_void_p(ids.data_ptr() + 8 * ids_off), ids_stride, float(thr), r.code, int(r.top_k), float(r.kl_thr),
                                      o(draft_logits), _ptr(counts), o(argmax_out), o(match_out), o(corrected_out), _stream()))


def cfg_combine(logits: torch.Tensor, B: int, lens: Sequence[int], V: int, ts: Sequence[float]) -> List[torch.Tensor]:
    """var.py:1062-1067 on a verified chunk's raw logits (2B, sum(lens), V): the per-stage CFG logits (B, l_j, V), one kernel (csrc/sampler.hip)."""
    n, lsum = len(lens), int(sum(lens))
    out = torch.empty(B, lsum, V, dtype=torch.float32, device=logits.device)
    _check(load_library().sdvar_cfg_combine(_ptr(logits), B, lsum, V, n, (_I * n)(*lens), (_D * n)(*[float(t) for t in ts]), _ptr(out), _stream()))
    return list(out.split([int(x) for x in lens], dim=1))


def last_gemm_cfg() -> Dict[str, int]:
    """Test aid: row tile and K split of the last f16x2 GEMM call of this thread, and how many launches since the last read took the hybrid tail split /
    finished q and k in the QKV epilogue (reading resets the two counters)."""
    v = (_I * 4)()
    _check(load_library().sdvar_debug_get_gemm_cfg(v))
    return dict(bm=v[0], split=v[1], tail_launches=v[2], fused_qkv_launches=v[3])


_GUARD_ON = False


def f16x2_guard(on: bool):
    """Debug switch (mode f16x2): count saturated / non-finite / tiny elements of every GEMM operand plane written inside stage_forward.
    While it is on, every sampler call ends with a device sync and reports the counters of that call in `SampleResult.stats["f16x2_guard"]`."""
    global _GUARD_ON
    _check(load_library().sdvar_debug_set_f16x2_guard(1 if on else 0))
    _GUARD_ON = bool(on)
    if on:
        f16x2_guard_collect(reset=True)


def f16x2_guard_collect(reset: bool = True) -> Dict[str, int]:
    v = (_U64 * 4)()
    _check(load_library().sdvar_debug_get_f16x2_guard(v, 1 if reset else 0))
    return dict(elements=int(v[0]), saturated=int(v[1]), non_finite=int(v[2]), tiny=int(v[3]))


def prof_enable(on: bool):
    _check(load_library().sdvar_prof_enable(1 if on else 0))


def prof_collect() -> Dict[str, Dict[str, float]]:
    k = len(PROF_CLASSES)
    ms, n, fl, by = (_D * k)(), (C.c_int64 * k)(), (_D * k)(), (_D * k)()
    _check(load_library().sdvar_prof_collect(ms, n, fl, by))
    return {PROF_CLASSES[i]: dict(ms=ms[i], launches=int(n[i]), flops=fl[i], bytes=by[i]) for i in range(k)}


# ------------------------------------------------------------------------------------------------------- sampling loops
@dataclass
class SampleResult:
    ids: torch.Tensor                      # (B, L) int64 accepted token ids, stage s at columns [begin(s), begin(s)+l_s)
    f_hat: torch.Tensor                    # (B, Cvae, HW, HW)
    stats: Dict[str, object] = field(default_factory=dict)
    trace: Dict[str, list] = field(default_factory=dict)


class Sampler:
    """Buffers + loops for one (draft, target) pair (or a single model for plain AR) on one GPU."""

    def __init__(self, target: ModelCtx, quant: QuantCtx, draft: Optional[ModelCtx] = None):
        self.t, self.d, self.q = target, draft, quant
        self.lad: Ladder = target.lad
        self.dev = target.device
        B, V, lad = target.max_batch, target.V, self.lad
        lens = lad.lens
        self.lmax_t = max(sum(lens[s:s + target.max_chunk]) for s in range(lad.S))
        f = dict(device=self.dev, dtype=torch.float32)
        self.x_t = torch.empty(2 * B * self.lmax_t * target.Cw, **f)
        self._xt = [self.x_t, None]             # second target input buffer: allocated when the run-ahead path first needs it
        self._verify_stream = None
        self.logits_t = torch.empty(2 * B * self.lmax_t * V, **f)
        if draft is not None:
            assert draft.lad.patch_nums == lad.patch_nums and draft.V == V
            self.x_d = torch.empty(2 * B * lens[-1] * draft.Cw, **f)
            self.logits_d = torch.empty(2 * B * lens[-1] * V, **f)
        g = target.max_chunk
        self.ids = torch.zeros(B, lad.L, device=self.dev, dtype=torch.int64)
        self.f_work = torch.zeros(B, quant.Cv, lad.HW, lad.HW, **f)
        self.f_acc = torch.zeros_like(self.f_work)
        self._fs = [[torch.zeros_like(self.f_work) for _ in range(g)], None]      # f_hat snapshots / next-stage inputs of a round; the second
        self._nx = [[torch.empty(B * lens[-1] * quant.Cv, **f) for _ in range(g)], None]    # slot exists once the optimistic path runs
        self._slot = 0
        self.nxt_cur = torch.empty(B * lens[-1] * quant.Cv, **f)
        self.counts = torch.zeros(40, device=self.dev, dtype=torch.int32)
        self.counts_ra = torch.zeros(4 * lad.S, 40, device=self.dev, dtype=torch.int32)     # one counter row per enqueued verification
        self._counts_pin = torch.zeros(4 * lad.S, 40, dtype=torch.int32).pin_memory()
        self._row = 0
        self.counts_host = torch.zeros(40, dtype=torch.int32).pin_memory()
        self._lazy: Dict[str, torch.Tensor] = {}      # buffers only some paths need (more_smooth, KL rule, token-level acceptance, hand-off)

    def _buf(self, name: str, numel: int, dtype=torch.float32) -> torch.Tensor:
        t = self._lazy.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = self._lazy[name] = torch.empty(numel, device=self.dev, dtype=dtype)
        return t

    def _sample_stage(self, logits: torch.Tensor, B: int, si: int, cfg: float, top_k: int, top_p: float, noise: Noise, draw: int,
                      more_smooth: bool, f_hat: torch.Tensor, nxt: Optional[torch.Tensor], f_in: Optional[torch.Tensor] = None, keep_masked: bool = False):
        """CFG + top-k/top-p + multinomial (var.py:199-202), then the token -> feature step (var.py:205-211): codebook rows of the sampled ids,
        or with more_smooth=True the gumbel-softmax mix of the masked logits (var.py:206-208)."""
        lad, V, L, qz = self.lad, self.t.V, self.lad.L, self.q
        l = lad.lens[si]
        q = noise.tensor(draw, B, l, V, self.dev)
        masked = self._buf("masked", B * lad.lens[-1] * V) if (more_smooth or keep_masked) else None          # the CFG logits after top-k / top-p (-inf = removed)
        cfg_sample(logits, B, l, V, lad.cfg_t(cfg, si), top_k, top_p, q, noise.seed, draw, noise.image_offset, self.ids, lad.begin(si), L, masked)
        last = si == lad.S - 1
        if not more_smooth:
            qz.next(si, self.ids[:, lad.begin(si):], L, f_hat, None if last else nxt, B, f_in=f_in)
            return masked
        assert f_in is None
        ratio = si / (lad.S - 1)
        tau = max(0.27 * (1 - ratio * 0.95), 0.005)                           # var.py:207
        e = noise.tensor(draw | GUMBEL_DRAW, B, l, V, self.dev)                # the generator's next (B, l, V) exponential draw (helpers.py:26)
        h = self._buf("h_soft", B * lad.lens[-1] * qz.Cv)
        qz.gumbel_mix(masked, B, l, ratio, tau, e, noise.seed, draw | GUMBEL_DRAW, noise.image_offset, h)
        qz.next_h(si, h, f_hat, None if last else nxt, B)
        return masked

    @property
    def f_snap(self):
        return self._fs[self._slot]

    @property
    def nxt(self):
        return self._nx[self._slot]

    # ---- VAR.autoregressive_infer_cfg (var.py:127-215) up to the decode
    def plain_ar(self, labels: torch.Tensor, cfg: float, top_k: int, top_p: float, noise: Noise, trace: bool = False,
                 more_smooth: bool = False) -> SampleResult:
        m, qz, lad = self.t, self.q, self.lad
        B, V, S, L = labels.shape[0], m.V, lad.S, lad.L
        res = SampleResult(ids=self.ids[:B], f_hat=self.f_work[:B])
        with torch.cuda.device(self.dev):
            m.begin(labels)
            f_hat = self.f_work[:B]; f_hat.zero_()
            m.place_first(self.x_t, lad.lens[0])
            for si in range(S):
                l = lad.lens[si]
                m.forward(self.x_t, si, 1, self.logits_t)
                if trace:
                    res.trace.setdefault("logits", []).append(self.logits_t[:2 * B * l * V].view(2 * B, l, V).clone())
                self._sample_stage(self.logits_t, B, si, cfg, top_k, top_p, noise, si, more_smooth, f_hat, self.nxt[0])
                if si != S - 1:
                    m.embed_next(self.nxt[0], si + 1, self.x_t, lad.lens[si + 1], 0)
            m.kv_set_len(0)
        res.stats = dict(target_calls=S, draft_stage_calls=0, forced_accepts=0, accepted_tokens=0)
        if _GUARD_ON:
            res.stats["f16x2_guard"] = f16x2_guard_collect()
        return res

    # ---- VAR.autoregressive_infer_cfg_sd_helper1 (var.py:319-443): a run of `step` stages of the plain sampler from a handed-in state
    def resume_ar(self, cond: torch.Tensor, current_step: int, step: int, next_map: Optional[torch.Tensor], f_hat: torch.Tensor, cfg: float, top_k: int,
                  top_p: float, noise: Noise, more_smooth: bool = False):
        """Stages current_step .. current_step + step - 1 with the conditioning rows `cond` (2B, C) = `sos`, the next-scale map `next_map`
        (B, Cvae, pn, pn) of stage current_step and the running f_hat, which is updated IN PLACE as the reference's quantizer does (quant.py:191).
        The KV cache starts EMPTY: var.py:368 toggles kv_caching(True), which drops it (basic_var.py:87), so the resumed stages attend to themselves
        and to each other, not to the stages before current_step; and `cur_L` starts at 0 there too (var.py:352: the skipped stages do not advance
        it), so stage si is embedded with the lvl_pos rows begin(si) - begin(current_step) .. - both are the reference's behaviour, pinned by
        tests/golden/ar_d4_256_helper1.npz.  Draw index of stage si = si.
        Returns the four histories of var.py:436-443: input maps (B, l, Cvae) of the stages run that are not stage 0, then the raw next map
        (B, Cvae, pn', pn') after the last one (f_hat itself after the final stage); f_hat (the same tensor step + 1 times - the reference appends
        the object it then updates in place); CFG-combined logits (B, l, V) as the sampler left them (-inf at the entries top-k / top-p removed:
        helpers.py:10,15 mask the appended tensor in place); token ids (B, l)."""
        m, qz, lad = self.t, self.q, self.lad
        S, V, L, lens, Cv = lad.S, m.V, lad.L, lad.lens, qz.Cv
        B = cond.shape[0] // 2
        if not (0 <= current_step < S and step >= 1):
            raise SdvarError(f"resume_ar: current_step {current_step}, step {step} (ladder has {S} stages)")
        if f_hat.shape != (B, Cv, lad.patch_nums[-1], lad.patch_nums[-1]) or not f_hat.is_cuda or f_hat.dtype != torch.float32 or not f_hat.is_contiguous():
            raise SdvarError("resume_ar: f_hat must be a contiguous fp32 GPU tensor (B, Cvae, pn_last, pn_last)")
        if current_step > 0 and (next_map is None or next_map.numel() != B * Cv * lens[current_step]):
            raise SdvarError(f"resume_ar: stage {current_step} needs its next-scale map (B, Cvae, {lad.patch_nums[current_step]}, {lad.patch_nums[current_step]})")
        inputs, f_hist, logit_hist, id_hist = [], [], [], []
        end = min(current_step + step, S)
        with torch.cuda.device(self.dev):
            m.begin_cond(cond)
            m.kv_set_origin(current_step)
            nxt = self.nxt[0]
            for si in range(current_step, end):
                l = lens[si]
                if si == 0:
                    m.place_first(self.x_t, l)
                else:
                    if si == current_step:
                        rows = next_map.to(torch.float32).reshape(B, Cv, l).transpose(1, 2).contiguous()          # var.py:382
                        nxt[:B * l * Cv].view(B, l, Cv).copy_(rows)
                    inputs.append(nxt[:B * l * Cv].view(B, l, Cv).clone())
                    m.embed_next_at(nxt, si, lad.begin(si) - lad.begin(current_step), self.x_t, l, 0)              # var.py:385 with cur_L counted from current_step
                f_hist.append(f_hat)
                m.forward(self.x_t, si, 1, self.logits_t)
                masked = self._sample_stage(self.logits_t, B, si, cfg, top_k, top_p, noise, si, more_smooth, f_hat, nxt, keep_masked=True)
                logit_hist.append(masked[:B * l * V].view(B, l, V).clone())           # var.py:405-408: the CFG logits, masked in place by helpers.py:10,15 before the caller sees them
                id_hist.append(self.ids[:B, lad.begin(si):lad.begin(si) + l].clone())
            f_hist.append(f_hat)
            if end == S:
                inputs.append(f_hat)                                                                               # quant.py:196: the last stage hands back f_hat twice
            else:
                pn = lad.patch_nums[end]
                inputs.append(nxt[:B * pn * pn * Cv].view(B, pn * pn, Cv).transpose(1, 2).reshape(B, Cv, pn, pn).clone())
            m.kv_set_len(0)
        return inputs, f_hist, logit_hist, id_hist

    # ---- SDVAR.sdvar_autoregressive_infer_cfg_sd_test3 (var.py:604-865): the draft samples stages < entry_num, the target the rest
    def handoff(self, labels: torch.Tensor, cfg: float, top_k: int, top_p: float, noise: Noise, entry_num: int, sd_mask: int = 0,
                more_smooth: bool = False) -> SampleResult:
        """sd_mask 0: the target starts at the entry stage with an empty KV cache - it does not condition on the draft's prefix
        (var.py:817-824).  sd_mask 3: the target first runs the whole prefix + entry stage through its blocks under the block-causal
        mask (var.py:789, 802-804: this fills its cache), then - literally as the reference does - takes the entry stage's logits from
        the INPUT token map, not from the block output (var.py:809-811).  One noise stream: draw index = stage (var.py:642, 689, 831)."""
        assert self.d is not None, "the hand-off sampler needs a draft model"
        d, t, qz, lad = self.d, self.t, self.q, self.lad
        B, V, S, L, lens = labels.shape[0], t.V, lad.S, lad.L, lad.lens
        if not (0 <= entry_num <= S):
            raise SdvarError(f"entry_num {entry_num} outside [0, {S}]")
        if sd_mask not in (0, 1, 2, 3, 4, 5):
            raise SdvarError(f"sd_mask {sd_mask}: the reference defines 0 .. 5 (var.py:777-798)")
        res = SampleResult(ids=self.ids[:B], f_hat=self.f_work[:B])
        prefill = sd_mask != 0 and entry_num < S
        pindex = lad.cum[entry_num] if entry_num < S else L
        if prefill and t.max_chunk < entry_num + 1:
            raise SdvarError(f"sd_mask={sd_mask} prefills stages 0..{entry_num} in one pass: the target context needs max_chunk >= {entry_num + 1}")
        with torch.cuda.device(self.dev):
            f_hat = self.f_work[:B]; f_hat.zero_()
            d.begin(labels); t.begin(labels)
            x_pre = self._buf("x_prefill", 2 * B * pindex * t.Cw) if prefill else None
            if prefill:
                t.place_first(x_pre, pindex)
            d.place_first(self.x_d, lens[0])
            for si in range(min(entry_num, S)):                                  # var.py:669-723
                d.forward(self.x_d, si, 1, self.logits_d)
                self._sample_stage(self.logits_d, B, si, cfg, top_k, top_p, noise, si, more_smooth, f_hat, self.nxt[0])
                if si != S - 1:
                    d.embed_next(self.nxt[0], si + 1, self.x_d, lens[si + 1], 0)
                    if prefill:                                                 # draft_token_hub -> target embedding of the prefix (var.py:713, 753)
                        t.embed_next(self.nxt[0], si + 1, x_pre, pindex, lad.begin(si + 1))
            d.kv_set_len(0)
            for si in range(entry_num, S):                                       # var.py:768-859
                l = lens[si]
                if si == entry_num:
                    if prefill:
                        ent = self._buf("x_entry", 2 * B * l * t.Cw)             # the entry stage's slice of the input map, kept: the forward clobbers x
                        src = x_pre[:2 * B * pindex * t.Cw].view(2 * B, pindex, t.Cw)[:, lad.begin(si):pindex]
                        ent[:2 * B * l * t.Cw].view(2 * B, l, t.Cw).copy_(src)
                        bias = None if sd_mask == 3 else handoff_mask(lad, entry_num, sd_mask).to(self.dev)         # 3 = the block-causal rows the kernels derive themselves
                        t.forward(x_pre, 0, entry_num + 1, self._buf("logits_prefill", 2 * B * pindex * V), bias)   # fills the cache; its logits are not used
                        t.head_forward(ent, l, self.logits_t)
                    else:
                        t.kv_set_origin(si)
                        if si == 0:
                            t.place_first(self.x_t, l)
                        else:
                            t.embed_next(self.nxt[0], si, self.x_t, l, 0)
                        t.forward(self.x_t, si, 1, self.logits_t)
                else:
                    t.forward(self.x_t, si, 1, self.logits_t)
                self._sample_stage(self.logits_t, B, si, cfg, top_k, top_p, noise, si, more_smooth, f_hat, self.nxt[0])
                if si != S - 1:
                    t.embed_next(self.nxt[0], si + 1, self.x_t, lens[si + 1], 0)
            t.kv_set_len(0)
        res.stats = dict(target_calls=S - min(entry_num, S), draft_stage_calls=min(entry_num, S), forced_accepts=0, accepted_tokens=0, entry_num=entry_num, sd_mask=sd_mask)
        return res

    # ---- speculative draft -> verify loop, as the reference's four steps
    def spec_begin(self, labels: torch.Tensor, cfg: float, gamma: int, top_k: int, top_p: float, noise: Noise, thr: float = 0.5,
                   match: Optional[MatchRule] = None) -> "SpecState":
        """SDVAR._initialize_inference_state (var.py:871-947): both prologues, empty caches, counters."""
        assert self.d is not None, "the speculative loop needs a draft model"
        assert 1 <= gamma <= self.t.max_chunk, f"gamma {gamma} exceeds the engine's max_chunk {self.t.max_chunk}"
        with torch.cuda.device(self.dev):
            self.d.begin(labels); self.t.begin(labels)
            self.f_acc[:labels.shape[0]].zero_()
        self._row, self._slot = 0, 0
        return SpecState(sampler=self, labels=labels, B=labels.shape[0], cfg=cfg, gamma=gamma, top_k=top_k, top_p=top_p, noise=noise, thr=thr,
                         total_stages=self.lad.S, patch_nums=self.lad.patch_nums, match=match or MatchRule())

    def spec_draft(self, st: "SpecState") -> int:
        """SDVAR.draft_generate_batch (var.py:949-1024): g = min(gamma, S - cur) draft stages (forward, CFG, sample, quant,
        next-stage inputs for BOTH models).  Token ids land in self.ids, f_hat snapshots in self.f_snap."""
        d, t, qz, lad = self.d, self.t, self.q, self.lad
        B, V, S, L, lens, cur = st.B, t.V, lad.S, lad.L, lad.lens, st.current_stage
        g = min(st.gamma, S - cur)
        if g <= 0:
            return 0
        st.g, st.glen = g, lens[cur:cur + g]
        lsum = sum(st.glen)
        offs = [sum(st.glen[:j]) for j in range(g)]
        keep_logits = st.match.rule == "kl"        # the KL rule compares the two models' distributions: keep the draft's logits of the round
        dl = self._buf("draft_logits", 2 * self.t.max_batch * self.lmax_t * V) if keep_logits else None
        with torch.cuda.device(self.dev):
            x_t = self._xt[st.xt_idx]
            if cur == 0:
                d.place_first(self.x_d, lens[0]); t.place_first(x_t, lsum)
            else:
                d.embed_next(self.nxt_cur, cur, self.x_d, lens[cur], 0); t.embed_next(self.nxt_cur, cur, x_t, lsum, 0)
            for j in range(g):
                s = cur + j
                lg = dl[2 * B * V * offs[j]:] if keep_logits else self.logits_d
                d.forward(self.x_d, s, 1, lg)
                st.stats["draft_stage_calls"] += 1
                q = st.noise.tensor(st.draw, B, lens[s], V, self.dev)
                cfg_sample(lg, B, lens[s], V, lad.cfg_t(st.cfg, s), st.top_k, st.top_p, q, st.noise.seed, st.draw, st.noise.image_offset,
                           self.ids, lad.begin(s), L)
                st.draw += 1
                last = s == S - 1
                # f_hat snapshot j = snapshot j-1 (the accepted prefix for j = 0) + this stage: written directly, nothing is copied
                qz.next(s, self.ids[:, lad.begin(s):], L, self.f_snap[j][:B], None if last else self.nxt[j], B, f_in=self.f_acc[:B] if j == 0 else self.f_snap[j - 1][:B])
                if j + 1 < g:
                    d.embed_next(self.nxt[j], s + 1, self.x_d, lens[s + 1], 0)
                    t.embed_next(self.nxt[j], s + 1, x_t, lsum, offs[j + 1])
        st.drafted = True
        return g

    def spec_verify_forward(self, st: "SpecState") -> torch.Tensor:
        """SDVAR.target_verify_batch (var.py:1026-1070): ONE target forward over the drafted stages under the block-causal
        rows; returns the raw logits view (2B, lsum, V)."""
        assert st.drafted, "draft_generate_batch must run before target_verify_batch"
        lsum = sum(st.glen)
        with torch.cuda.device(self.dev):
            self.t.forward(self._xt[st.xt_idx], st.current_stage, st.g, self.logits_t)
        st.stats["target_calls"] += 1
        st.target_calls += 1
        st.verified = True
        return self.logits_t[:2 * st.B * lsum * self.t.V].view(2 * st.B, lsum, self.t.V)

    def spec_accept(self, st: "SpecState"):
        """SDVAR.basic_token_matching (var.py:1160-1227) on the verified chunk: (n_accept, matched per stage)."""
        assert st.verified
        lad, cur, g = self.lad, st.current_stage, st.g
        mr = st.match
        with torch.cuda.device(self.dev):
            corr = self._buf("ids_corr", self.t.max_batch * self.lmax_t, torch.int64) if mr.token_level else None
            verify_accept(self.logits_t, st.B, st.glen, self.t.V, [lad.cfg_t(st.cfg, cur + j) for j in range(g)], self.ids, lad.begin(cur), lad.L, st.thr, self.counts,
                          rule=mr, draft_logits=self._lazy.get("draft_logits") if mr.rule == "kl" else None, corrected_out=corr)
            self.counts_host.copy_(self.counts, non_blocking=False)            # the one host sync of the round
        c = self.counts_host.tolist()
        if st.accept_scope == "global":            # batch-wide decision across ranks (reference-literal for one big batch)
            from . import dist as D
            n, _, _ = D.global_accept(c[:g], c[17:17 + g], st.thr, self.dev)
            # the DECISION is global; the counts handed back stay this rank's own: every statistic built from them (accepted / corrected tokens,
            # rounds[].matched) is per rank and summed once by dist.gather_counters - global counts here would be added world-size times
            return n, c[:g]
        return c[16], c[:g]

    def spec_correct(self, st: "SpecState", n_acc: int, matched: Sequence[int]) -> int:
        """Token-level partial acceptance (the 'partial accept / token-level rollback' item of var.py:1229-1243): stage cur + n_acc failed the
        batch threshold; keep its tokens that satisfy the rule, give the others the target's argmax (both already chosen by the verify kernel),
        rebuild that stage's f_hat snapshot and next-scale input from the corrected ids, and return the number of stages to commit."""
        lad, B, cur, j = self.lad, st.B, st.current_stage, n_acc
        s, l, lsum, off = cur + j, st.glen[j], sum(st.glen), sum(st.glen[:j])
        with torch.cuda.device(self.dev):
            corr = self._lazy["ids_corr"][:B * lsum].view(B, lsum)
            self.ids[:B, lad.begin(s):lad.begin(s) + l].copy_(corr[:, off:off + l])
            last = s == lad.S - 1
            self.q.next(s, self.ids[:, lad.begin(s):], lad.L, self.f_snap[j][:B], None if last else self.nxt[j], B, f_in=self.f_acc[:B] if j == 0 else self.f_snap[j - 1][:B])
        st.stats["corrected_tokens"] = st.stats.get("corrected_tokens", 0) + B * l - matched[j]
        st.stats["corrected_stages"] = st.stats.get("corrected_stages", 0) + 1
        return n_acc + 1

    def spec_commit(self, st: "SpecState", n_acc: int, forced: bool = False, accepted_tokens: Optional[int] = None):
        """SDVAR.update_state_with_accepted_tokens (var.py:1245-1282) + `current_stage += n` (var.py:1349-1350), and the
        rollback of both KV caches to the accepted prefix (absent in the reference)."""
        lad, B, cur = self.lad, st.B, st.current_stage
        with torch.cuda.device(self.dev):
            if n_acc > 0:
                self.f_acc[:B].copy_(self.f_snap[n_acc - 1][:B])
                if not forced:
                    st.stats["accepted_tokens"] += sum(st.glen[:n_acc]) if accepted_tokens is None else accepted_tokens
                if cur + n_acc < lad.S:
                    self.nxt_cur.copy_(self.nxt[n_acc - 1])
                st.current_stage = cur = cur + n_acc
                st.accept_count += n_acc
            keep = lad.begin(cur) if cur < lad.S else lad.L
            self.d.kv_set_len(keep); self.t.kv_set_len(keep)
        st.drafted = st.verified = False

    def spec_end(self, st: "SpecState"):
        with torch.cuda.device(self.dev):
            self.d.kv_set_len(0); self.t.kv_set_len(0)
        st.stats["gamma_final"] = st.gamma

    def spec_decode(self, labels: torch.Tensor, cfg: float, gamma: int, top_k: int, top_p: float, noise: Noise, thr: float = 0.5,
                    trace: bool = False, run_ahead: bool = True, accept_scope: str = "shard", match: Optional[MatchRule] = None) -> SampleResult:
        """The whole loop with the policy of var.py:1318-1372 (gamma only decreases; forced accept at gamma == 1; never break).
        run_ahead: once gamma has dropped to 1 the draft no longer waits for the verifier (see _spec_run_ahead); same results.
        accept_scope "global": the batch-wide decision is taken across ranks (one all-reduce per round), always in lock-step."""
        st = self.spec_begin(labels, cfg, gamma, top_k, top_p, noise, thr, match)
        st.accept_scope = accept_scope
        run_ahead = run_ahead and st.match.basic        # the richer rules run in lock-step
        res = SampleResult(ids=self.ids[:st.B], f_hat=self.f_acc[:st.B], stats=st.stats)
        optimistic = False                     # speculate on acceptance only after a round that was accepted in full
        while st.current_stage < st.total_stages:
            if run_ahead and st.accept_scope == "shard" and not trace:
                if st.gamma == 1:
                    self._spec_run_ahead(st)
                    break
                if optimistic:
                    self._spec_optimistic(st)
                    optimistic = False
                    continue
            cur = st.current_stage
            g = self.spec_draft(st)
            lg = self.spec_verify_forward(st)
            if trace:
                res.trace.setdefault("target_logits", []).append((cur, g, lg.clone()))
            n_acc, matched = self.spec_accept(st)
            forced, acc_tok = False, None
            if n_acc < g and st.match.token_level:
                acc_tok = sum(st.glen[:n_acc]) + matched[n_acc]
                n_acc = self.spec_correct(st, n_acc, matched)
            elif n_acc == 0:                                                    # var.py:1353-1364
                if st.gamma > 1:
                    st.gamma -= 1
                else:
                    n_acc, forced = 1, True
                    st.stats["forced_accepts"] += 1
            st.stats["rounds"].append(dict(stage=cur, g=g, matched=matched, total=[st.B * n for n in st.glen], n_accept=n_acc, forced=forced))
            optimistic = n_acc == g and not forced
            self.spec_commit(st, n_acc, forced, acc_tok)
        self.spec_end(st)
        if _GUARD_ON:
            res.stats["f16x2_guard"] = f16x2_guard_collect()
        return res

    def _ensure_second_stream(self):
        if self._verify_stream is None:
            self._verify_stream = torch.cuda.Stream(device=self.dev, priority=torch.cuda.current_stream(self.dev).priority)
            self._xt[1] = torch.empty_like(self.x_t)
        if self._fs[1] is None:
            self._fs[1] = [torch.zeros_like(t) for t in self._fs[0]]
            self._nx[1] = [torch.empty_like(t) for t in self._nx[0]]
            self._f_prev = [torch.zeros_like(self.f_acc) for _ in range(2)]
            self._nxt_prev = [torch.empty_like(self.nxt_cur) for _ in range(2)]
        return self._verify_stream

    def _spec_optimistic(self, st: "SpecState"):
        """gamma > 1 and the previous round was accepted in full: draft the next round before the verdict on the current one is
        known.  Every round is committed as if accepted (per-round slots keep its f_hat snapshots, next-stage inputs and the
        state before it); the verifier runs on the second stream, its counters come back through pinned memory, and the host
        reads the verdict of round r after it has enqueued round r+1.  A verdict short of full acceptance rolls the state back
        to what the lock-step loop would hold (accepted prefix, gamma policy, draw counter, both KV cursors), discards the
        speculative round and returns to lock-step.  Results and counters equal the lock-step loop's."""
        lad, B, V, S = self.lad, st.B, self.t.V, self.lad.S
        with torch.cuda.device(self.dev):
            D = torch.cuda.current_stream(self.dev)
            T = self._ensure_second_stream()
            T.wait_stream(D)
            t_done, pend, n = [], None, 0
            while st.current_stage < S and st.gamma > 1:
                slot = n & 1
                if n >= 2:
                    D.wait_event(t_done[n - 2])                 # last verification that read this slot's target input
                self._slot = st.xt_idx = slot
                self._f_prev[slot][:B].copy_(self.f_acc[:B]); self._nxt_prev[slot].copy_(self.nxt_cur)
                cur = st.current_stage
                g = self.spec_draft(st)
                ready = torch.cuda.Event(); ready.record(D)
                T.wait_event(ready)
                row = self._row; self._row += 1
                with torch.cuda.stream(T):
                    self.t.forward(self._xt[slot], cur, g, self.logits_t)
                    verify_accept(self.logits_t, B, st.glen, V, [lad.cfg_t(st.cfg, cur + j) for j in range(g)], self.ids, lad.begin(cur), lad.L, st.thr,
                                  self.counts_ra[row])
                    self._counts_pin[row].copy_(self.counts_ra[row], non_blocking=True)
                    ev = torch.cuda.Event(); ev.record(T); t_done.append(ev)
                st.stats["target_calls"] += 1
                st.target_calls += 1
                info = dict(cur=cur, g=g, glen=list(st.glen), slot=slot, draw_after=st.draw, row=row, ev=ev)
                self.spec_commit(st, g, forced=True)            # optimistic: all g stages; the counters are settled by _resolve
                n += 1
                if pend is not None and not self._resolve(st, pend):
                    st.stats["draft_stage_calls"] -= info["g"]  # the speculative round is discarded, as if never drafted
                    st.stats["target_calls"] -= 1
                    st.target_calls -= 1
                    st.stats["discarded_speculative_rounds"] = st.stats.get("discarded_speculative_rounds", 0) + 1
                    pend = None
                    break
                pend = info
            if pend is not None:
                self._resolve(st, pend)
            self._slot = st.xt_idx = 0
            D.wait_stream(T)

    def _resolve(self, st: "SpecState", p: dict) -> bool:
        """Verdict of an optimistically committed round; rolls back when it was not accepted in full.  True = accepted in full."""
        lad, B = self.lad, st.B
        p["ev"].synchronize()
        c = self._counts_pin[p["row"]].tolist()
        g, glen, n_acc, matched = p["g"], p["glen"], c[16], c[:p["g"]]
        st.stats["accepted_tokens"] += sum(glen[:n_acc])
        st.stats["rounds"].append(dict(stage=p["cur"], g=g, matched=matched, total=[B * m for m in glen], n_accept=n_acc, forced=False))
        if n_acc == g:
            return True
        slot = p["slot"]
        if n_acc > 0:
            self.f_acc[:B].copy_(self._fs[slot][n_acc - 1][:B])
            self.nxt_cur.copy_(self._nx[slot][n_acc - 1])
        else:
            self.f_acc[:B].copy_(self._f_prev[slot][:B]); self.nxt_cur.copy_(self._nxt_prev[slot])
            st.gamma -= 1                                      # var.py:1353-1356 (this path only runs with gamma > 1)
        st.accept_count -= (st.current_stage - (p["cur"] + n_acc))
        st.current_stage = p["cur"] + n_acc
        st.draw = p["draw_after"]
        keep = lad.begin(st.current_stage)
        self.d.kv_set_len(keep); self.t.kv_set_len(keep)
        st.drafted = st.verified = False
        return False

    def _spec_run_ahead(self, st: "SpecState"):
        """The tail of the loop at gamma == 1.  There a round always advances one stage with the DRAFT's tokens - accepted, or
        force-accepted (var.py:1357-1364) - so the verify result only feeds the counters, and since gamma never grows again
        this holds to the end.  The draft of stage s+1 therefore need not wait for the verification of stage s: the target
        forwards + acceptance scans run on a second HIP stream one round behind (target input double-buffered, one counter row
        per round, read back once at the end); the launch-bound early stages of one model fill the gaps of the other."""
        lad, B, V = self.lad, st.B, self.t.V
        with torch.cuda.device(self.dev):
            D = torch.cuda.current_stream(self.dev)
            T = self._ensure_second_stream()
            T.wait_stream(D)                                   # earlier rounds ran the target on the caller's stream
            row0 = self._row
            t_done, meta = [], []
            r = 0
            while st.current_stage < st.total_stages:
                cur = st.current_stage
                if r >= 2:
                    D.wait_event(t_done[r - 2])                # the target forward that last used this input buffer
                st.xt_idx = r & 1
                g = self.spec_draft(st)                        # gamma == 1: one stage
                ready = torch.cuda.Event(); ready.record(D)
                T.wait_event(ready)
                with torch.cuda.stream(T):
                    self.t.forward(self._xt[st.xt_idx], cur, 1, self.logits_t)
                    verify_accept(self.logits_t, B, st.glen, V, [lad.cfg_t(st.cfg, cur)], self.ids, lad.begin(cur), lad.L, st.thr, self.counts_ra[row0 + r])
                    ev = torch.cuda.Event(); ev.record(T); t_done.append(ev)
                st.stats["target_calls"] += 1
                st.target_calls += 1
                meta.append((cur, g, list(st.glen)))
                self.spec_commit(st, 1, forced=True)           # data movement of an accepted and of a forced stage is the same; counters below
                r += 1
            st.xt_idx = 0
            D.wait_stream(T)
            self._row = row0 + r
            c = self.counts_ra[row0:row0 + r].cpu().tolist()   # the one host sync of the tail
        for (cur, g, glen), row in zip(meta, c):
            n_acc, matched, forced = row[16], row[:g], False
            if n_acc == 0:
                n_acc, forced = 1, True
                st.stats["forced_accepts"] += 1
            else:
                st.stats["accepted_tokens"] += sum(glen[:n_acc])
            st.stats["rounds"].append(dict(stage=cur, g=g, matched=matched, total=[B * n for n in glen], n_accept=n_acc, forced=forced))


@dataclass
class SpecState:
    """Run state of the speculative loop; the public counters carry the reference's names (var.py:912-943)."""
    sampler: "Sampler"
    labels: torch.Tensor
    B: int
    cfg: float
    gamma: int
    top_k: int
    top_p: float
    noise: Noise
    thr: float
    total_stages: int
    patch_nums: tuple
    current_stage: int = 0
    accept_count: int = 0
    reject_count: int = 0
    target_calls: int = 0
    more_smooth: bool = False                     # stored and never read on the speculative path, as in the reference (var.py:1315 vs 949-1024)
    match: "MatchRule" = field(default_factory=lambda: MatchRule())
    accept_scope: str = "shard"                   # "shard": this process decides alone; "global": all-reduce of the match counts
    draw: int = 0
    xt_idx: int = 0                               # which target input buffer the current round uses (run-ahead double buffer)
    g: int = 0
    glen: list = field(default_factory=list)
    drafted: bool = False
    verified: bool = False
    stats: Dict[str, object] = field(default_factory=lambda: dict(target_calls=0, draft_stage_calls=0, forced_accepts=0, accepted_tokens=0, rounds=[], gamma_final=0))

    @property
    def draft_f_hat(self) -> torch.Tensor:        # f_hat of the accepted prefix (after the last commit)
        return self.sampler.f_acc[:self.B]

    @property
    def target_f_hat(self) -> torch.Tensor:
        return self.sampler.f_acc[:self.B]
