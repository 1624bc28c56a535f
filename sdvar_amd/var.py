"""Host-side mirror of the reference's `models.var` API for the sampling path: VAR and SDVAR.

Same constructor arguments, attribute names, sampler signatures and state_dict keys as
/root/reference/models/var.py:22-215 (VAR) and :535-1383 (SDVAR), so existing callers and upstream checkpoints work
unchanged - but the modules here only HOLD parameters; every sampler call runs on the gfx950 kernels through
sdvar_amd.engine (C ABI).  Training-side methods (VAR.forward with teacher forcing, progressive training) are outside
the scope table and raise NotImplementedError.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from . import engine as E
from .ladder import as_ladder
from .vqvae import VQVAE


class _SelfAttention(nn.Module):                   # parameter names of basic_var.py:58-87
    def __init__(self, C, H, attn_l2_norm=True):
        super().__init__()
        self.num_heads, self.head_dim, self.attn_l2_norm = H, C // H, attn_l2_norm
        if attn_l2_norm:                                # basic_var.py:66-72: otherwise plain scaled dot-product attention, scale 0.25 / sqrt(head_dim)
            self.scale_mul_1H11 = nn.Parameter(torch.full((1, H, 1, 1), 4.0).log())
        self.mat_qkv = nn.Linear(C, 3 * C, bias=False)
        self.q_bias, self.v_bias = nn.Parameter(torch.zeros(C)), nn.Parameter(torch.zeros(C))
        self.register_buffer("zero_k_bias", torch.zeros(C))
        self.proj = nn.Linear(C, C)
        self.caching = False
        self._owner = None

    def kv_caching(self, enable: bool):             # basic_var.py:87: toggling resets the cache
        self.caching = enable
        if self._owner is not None:
            self._owner()._kv_reset()


class _FFN(nn.Module):
    def __init__(self, C):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(C, 4 * C), nn.Linear(4 * C, C)


class _Block(nn.Module):
    def __init__(self, C, H, shared_aln=False, attn_l2_norm=True):
        super().__init__()
        self.attn, self.ffn = _SelfAttention(C, H, attn_l2_norm), _FFN(C)
        self.shared_aln = shared_aln
        if shared_aln:                                                   # basic_var.py:143-144
            self.ada_gss = nn.Parameter(torch.randn(1, 1, 6, C) / C ** 0.5)
        else:
            self.ada_lin = nn.Sequential(nn.SiLU(inplace=False), nn.Linear(C, 6 * C))


class _HeadNorm(nn.Module):
    def __init__(self, C):
        super().__init__()
        self.ada_lin = nn.Sequential(nn.SiLU(inplace=False), nn.Linear(C, 2 * C))


class VAR(nn.Module):
    def __init__(self, vae_local: VQVAE, num_classes=1000, depth=16, embed_dim=1024, num_heads=16, mlp_ratio=4., drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., norm_eps=1e-6, shared_aln=False, cond_drop_rate=0.1, attn_l2_norm=False,
                 patch_nums=(1, 2, 3, 4, 5, 6, 8, 10, 13, 16), flash_if_available=True, fused_if_available=True):
        super().__init__()
        assert embed_dim % num_heads == 0
        if embed_dim != 64 * num_heads or num_heads != depth or mlp_ratio != 4.:
            raise NotImplementedError("the HIP path assumes head_dim 64, heads == depth, mlp_ratio 4 (models/__init__.py:26-27)")
        self.Cvae, self.V = vae_local.Cvae, vae_local.vocab_size
        self.depth, self.C, self.D, self.num_heads = depth, embed_dim, embed_dim, num_heads
        self.patch_nums: Tuple[int] = tuple(patch_nums)
        lad = as_ladder(patch_nums)
        self.L, self.first_l = lad.L, lad.lens[0]
        self.begin_ends = [(lad.begin(s), lad.cum[s]) for s in range(lad.S)]
        self.num_stages_minus_1 = lad.S - 1
        self.num_classes = num_classes
        self.vae_proxy, self.vae_quant_proxy = (vae_local,), (vae_local.quantize,)
        C = embed_dim
        self.word_embed = nn.Linear(self.Cvae, C)
        self.class_emb = nn.Embedding(num_classes + 1, C)
        self.pos_start = nn.Parameter(torch.zeros(1, self.first_l, C))
        self.pos_1LC = nn.Parameter(torch.zeros(1, self.L, C))
        self.lvl_embed = nn.Embedding(lad.S, C)
        self.shared_aln = bool(shared_aln)
        self.shared_ada_lin = nn.Sequential(nn.SiLU(inplace=False), nn.Linear(C, 6 * C)) if shared_aln else nn.Identity()   # var.py:81
        self.attn_l2_norm = bool(attn_l2_norm)
        self.blocks = nn.ModuleList([_Block(C, num_heads, shared_aln, self.attn_l2_norm) for _ in range(depth)])
        lvl = torch.cat([torch.full((n,), i, dtype=torch.int64) for i, n in enumerate(lad.lens)]).view(1, self.L)
        self.register_buffer("lvl_1L", lvl)
        d = lvl.view(1, self.L, 1)
        self.register_buffer("attn_bias_for_masking", torch.where(d >= d.transpose(1, 2), 0., -torch.inf).reshape(1, 1, self.L, self.L).contiguous())
        self.head_nm = _HeadNorm(C)
        self.head = nn.Linear(C, self.V)
        self.rng = torch.Generator(device="cpu")
        import weakref
        for b in self.blocks:
            b.attn._owner = weakref.ref(self)
        self._ctx: Optional[E.ModelCtx] = None
        self._sampler: Optional[E.Sampler] = None
        self._quant: Optional[E.QuantCtx] = None
        self.noise_kind = "device"                  # 'device' | 'host' | 'torch' (sdvar_amd.engine.Noise)
        self.last_result: Optional[E.SampleResult] = None

    # ---- engine plumbing ---------------------------------------------------------------------------------------
    def _device(self):
        return self.lvl_1L.device

    def _kv_reset(self):
        if self._ctx is not None and self._ctx.kv_len() > 0:
            self._ctx.kv_set_len(0)

    def invalidate_engine(self):
        """Call after changing parameters in place (load_state_dict does it automatically)."""
        self._ctx = self._sampler = self._quant = None

    def load_state_dict(self, *a, **kw):
        out = super().load_state_dict(*a, **kw)
        self.invalidate_engine()
        return out

    def _apply(self, fn, *a, **kw):                 # .to()/.cuda() move the tensors the engine points at
        self.invalidate_engine()
        return super()._apply(fn, *a, **kw)

    def engine_ctx(self, max_batch: int, max_chunk: int = 1) -> E.ModelCtx:
        dev = self._device()
        if dev.type != "cuda":
            raise E.SdvarError("the sampler runs on an MI355X: move the model to a cuda (HIP) device; there is no CPU path")
        c = self._ctx
        if c is None or c.max_batch < max_batch or c.max_chunk < max_chunk or c.device != dev:
            if c is not None:
                c.close()
            self._ctx = E.ModelCtx(self.state_dict(), self.depth, self.patch_nums, max_batch, max_chunk, dev, self.num_classes)
            self._sampler = None
        return self._ctx

    def quant_ctx(self, max_batch: int) -> E.QuantCtx:
        q = self._quant
        if q is None or q.lad.patch_nums != self.patch_nums or q.device != self._device() or q.max_batch < max_batch:
            if q is not None:                                  # a larger batch than the cached quantizer was sized for: rebuild it
                q.close()
                self._sampler = None
            sd = {"quantize." + k: v for k, v in self.vae_quant_proxy[0].state_dict().items()}
            self._quant = E.QuantCtx(sd, self.patch_nums, max(max_batch, 1), self._device())
        return self._quant

    def _labels(self, B, label_B, rng):             # var.py:147-150
        dev = self._device()
        if label_B is None:
            p = torch.full((1, self.num_classes), 1.0 / self.num_classes)
            label_B = torch.multinomial(p, num_samples=B, replacement=True, generator=rng).reshape(B)
        elif isinstance(label_B, int):
            label_B = torch.full((B,), fill_value=self.num_classes if label_B < 0 else label_B)
        label_B = label_B.to(device=dev, dtype=torch.int64).contiguous()
        assert label_B.shape == (B,)
        return label_B

    def _noise(self, g_seed, image_offset=0) -> E.Noise:
        seed = int(torch.seed() & 0x7FFFFFFFFFFFFFFF) if g_seed is None else int(g_seed)
        if self.noise_kind == "torch":
            return E.Noise("torch", seed, generator=self.rng if g_seed is not None else None)
        return E.Noise(self.noise_kind, seed, image_offset)

    # ---- the sampler (var.py:127-215) ----------------------------------------------------------------------------
    @torch.no_grad()
    def autoregressive_infer_cfg(self, B: int, label_B: Optional[Union[int, torch.LongTensor]], g_seed: Optional[int] = None, cfg=1.5,
                                 top_k=0, top_p=0.0, more_smooth=False) -> torch.Tensor:
        if g_seed is None:
            rng = None
        else:
            self.rng.manual_seed(g_seed); rng = self.rng
        labels = self._labels(B, label_B, rng)
        ctx = self.engine_ctx(B, 1)
        qc = self.quant_ctx(ctx.max_batch)
        if self._sampler is None or self._sampler.t is not ctx or self._sampler.q is not qc:
            self._sampler = E.Sampler(ctx, qc)
        res = self._sampler.plain_ar(labels, cfg, top_k, top_p, self._noise(g_seed), more_smooth=bool(more_smooth))   # more_smooth: var.py:206-208
        self.last_result = res
        return self.vae_proxy[0].fhat_to_img(res.f_hat.clone()).add_(1).mul_(0.5)

    @torch.no_grad()
    def autoregressive_infer_cfg_sd_helper1(self, B: int, current_step: int, step: int, next_token_map, f_hat, rng, sos, lvl_pos, cfg=1.5, top_k=0, top_p=0.0,
                                            more_smooth=False):
        """var.py:319-443: `step` stages of the sampler from stage `current_step`, from a handed-in (next_token_map, f_hat) and the conditioning rows `sos`
        (2B, C) of SDVAR.init_param; returns (input_token_history, f_hat_history, logits_history, token_id_history) as the reference builds them.  `lvl_pos`
        is the model's own table (init_param hands it back unchanged, var.py:598): it is checked for shape and otherwise taken from the bound weights.
        `rng`: a CPU torch.Generator (the reference's stream on this host), an engine.Noise, or None (device Philox, fresh seed)."""
        S = len(self.patch_nums)
        assert 0 <= current_step < S and step >= 1
        assert sos.shape == (2 * B, self.C), tuple(sos.shape)
        assert lvl_pos is None or tuple(lvl_pos.shape[-2:]) == (self.L, self.C), tuple(lvl_pos.shape)
        dev = self._device()
        ctx = self.engine_ctx(B, 1)
        qc = self.quant_ctx(ctx.max_batch)
        if self._sampler is None or self._sampler.t is not ctx or self._sampler.q is not qc:
            self._sampler = E.Sampler(ctx, qc)
        if isinstance(rng, E.Noise):
            noise = rng
        elif rng is not None:
            noise = E.Noise("torch", 0, generator=rng)
        else:
            noise = self._noise(None)
        if not (f_hat.is_cuda and f_hat.dtype == torch.float32 and f_hat.is_contiguous()):
            raise E.SdvarError("autoregressive_infer_cfg_sd_helper1: f_hat must be a contiguous fp32 GPU tensor (it is updated in place, quant.py:191)")
        nm = None if (next_token_map is None or current_step == 0) else next_token_map.to(dev)
        return self._sampler.resume_ar(sos.to(device=dev, dtype=torch.float32).contiguous(), current_step, step, nm, f_hat, cfg, top_k, top_p, noise,
                                       more_smooth=bool(more_smooth))

    def forward(self, *a, **kw):
        raise NotImplementedError("teacher-forced training forward (var.py:217-259) is outside the sampling hot path")

    def init_weights(self, init_adaln=0.5, init_adaln_gamma=1e-5, init_head=0.02, init_std=0.02, conv_std_or_gain=0.02, seed: int = 1234):
        """Distributions of var.py:261-311 drawn from a seeded generator (sdvar_amd.weights 'perf' init): trunc-normal(init_std; < 0 =
        sqrt(1/C/3)) Linear / Embedding / positional tensors, head x init_head, head_nm and the adaLN scale/shift rows x init_adaln, the
        adaLN gamma rows x init_adaln_gamma, proj / fc2 / sqrt(2 depth), zero biases.  conv_std_or_gain only concerns convolutions, of
        which a VAR has none (var.py:275)."""
        from .weights import var_state_dict
        sd = var_state_dict(self.depth, self.patch_nums, "perf", seed, V=self.V, Cvae=self.Cvae, num_classes=self.num_classes, shared_aln=self.shared_aln, attn_l2_norm=self.attn_l2_norm,
                            init_adaln=init_adaln, init_adaln_gamma=init_adaln_gamma, init_head=init_head, init_std=init_std)
        self.load_state_dict({k: v.to(self._device()) for k, v in sd.items()})


try:                                                     # optional: only the from_pretrained / save_pretrained conveniences need it
    from huggingface_hub import PyTorchModelHubMixin as _HubMixin
except Exception:                                        # pragma: no cover - huggingface_hub not installed
    class _HubMixin:                                     # type: ignore[no-redef]
        pass


class VARHF(VAR, _HubMixin):
    """models/var.py:513-533: VAR that builds its own VQVAE from `vae_kwargs` and carries the Hugging Face hub mixin
    (`VARHF.from_pretrained(local_dir)` loads a `config.json` + weights directory; there is no network on the GPU boxes)."""

    def __init__(self, vae_kwargs, num_classes=1000, depth=16, embed_dim=1024, num_heads=16, mlp_ratio=4., drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_eps=1e-6, shared_aln=False, cond_drop_rate=0.1, attn_l2_norm=False,
                 patch_nums=(1, 2, 3, 4, 5, 6, 8, 10, 13, 16), flash_if_available=True, fused_if_available=True):
        super().__init__(vae_local=VQVAE(**vae_kwargs), num_classes=num_classes, depth=depth, embed_dim=embed_dim, num_heads=num_heads,
                         mlp_ratio=mlp_ratio, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate, norm_eps=norm_eps,
                         shared_aln=shared_aln, cond_drop_rate=cond_drop_rate, attn_l2_norm=attn_l2_norm, patch_nums=patch_nums,
                         flash_if_available=flash_if_available, fused_if_available=fused_if_available)


class SDVAR(nn.Module):
    """Speculative draft -> verify sampler over a (draft, target) VAR pair (models/var.py:535-1383)."""

    def __init__(self, draft_model: VAR, target_model: VAR, similarity_thresh: float = 0.8):
        super().__init__()
        self.draft_model, self.target_model = draft_model, target_model
        self.similarity_thresh = similarity_thresh     # stored and never read, as in the reference (var.py:546)
        self.match_threshold = 0.5                      # the constant of var.py:1215
        self.match_rule = E.MatchRule()                 # advanced_token_matching's rule (var.py:1229-1243); the default is the reference's fall-back
        self.noise_kind = "device"
        self.accept_scope = "shard"
        self._sampler: Optional[E.Sampler] = None
        self.last_result: Optional[E.SampleResult] = None

    # ---- the reference's helper methods (var.py:580-601, 871-1282), each a thin call into the engine --------------
    def init_param(self, model: VAR, B: int, label_B):
        """var.py:580-601 -> (sos, cond_BD, cond_BD_or_gss, lvl_pos, first_token_map, first_f_hat), computed by
        sdvar_model_begin on the device."""
        labels = model._labels(B, label_B, None)
        ctx = model.engine_ctx(B, 1)
        with torch.cuda.device(ctx.device):
            ctx.begin(labels)
            cond, lvl_pos, first = ctx.export_prologue()
        f_hat = cond.new_zeros(B, model.Cvae, model.patch_nums[-1], model.patch_nums[-1])
        return cond, cond, cond, lvl_pos, first, f_hat

    def _get_sampler(self, B: int, chunk: int) -> E.Sampler:
        d, t = self.draft_model, self.target_model
        assert d.patch_nums == t.patch_nums                                   # var.py:877
        dc, tc = d.engine_ctx(B, 1), t.engine_ctx(B, max(chunk, 1))
        qc = t.quant_ctx(max(tc.max_batch, dc.max_batch))
        s = self._sampler
        if s is None or s.t is not tc or s.d is not dc or s.q is not qc:
            self._sampler = E.Sampler(tc, qc, dc)
        return self._sampler

    def _prepare(self, B: int, label_B, g_seed: Optional[int], chunk: int, rng: Optional[torch.Generator] = None):
        """Labels, sampler objects and the noise source of one call (var.py:641-656, 880-902)."""
        t = self.target_model
        if g_seed is not None and rng is None:
            rng = torch.Generator(device="cpu"); rng.manual_seed(g_seed)      # var.py:882-887
        labels = t._labels(B, label_B, rng)
        smp = self._get_sampler(B, chunk)
        seed = int(torch.seed() & 0x7FFFFFFFFFFFFFFF) if g_seed is None else int(g_seed)
        noise = E.Noise("torch", seed, generator=rng) if self.noise_kind == "torch" else E.Noise(self.noise_kind, seed)
        return labels, smp, noise

    def _initialize_inference_state(self, B: int, label_B, g_seed: Optional[int], cfg: float, gamma: int) -> E.SpecState:
        """var.py:871-947.  The returned state carries the reference's field names (current_stage, gamma, total_stages,
        accept_count, target_calls, patch_nums, cfg, top_k, top_p, draft_f_hat, target_f_hat)."""
        labels, smp, noise = self._prepare(B, label_B, g_seed, gamma)
        st = smp.spec_begin(labels, cfg, gamma, 0, 0.0, noise, thr=self.match_threshold, match=self.match_rule)   # top_k / top_p set by the caller (var.py:937-938)
        st.accept_scope = self.accept_scope
        return st

    def draft_generate_batch(self, state: E.SpecState, B: int):
        """var.py:949-1024 -> list of g token tensors (B, pn^2) int64 for stages current_stage .. current_stage+g-1."""
        smp, lad = state.sampler, state.sampler.lad
        g = smp.spec_draft(state)
        cur = state.current_stage
        return [smp.ids[:B, lad.begin(cur + j):lad.begin(cur + j) + lad.lens[cur + j]].clone() for j in range(g)]

    def target_verify_batch(self, draft_tokens, state: E.SpecState, B: int):
        """var.py:1026-1070 -> (list of per-stage CFG logits (B, pn^2, V), gamma).  The verified stages are the ones the last
        draft_generate_batch produced (their next-scale inputs were built while drafting)."""
        if not draft_tokens:
            return [], 0
        assert len(draft_tokens) == state.g, "target_verify_batch verifies the stages of the last draft_generate_batch"
        smp, lad, cur = state.sampler, state.sampler.lad, state.current_stage
        lg = smp.spec_verify_forward(state)
        lens = [lad.lens[cur + j] for j in range(state.g)]
        with torch.cuda.device(lg.device):                                                           # var.py:1062-1067 (csrc/sampler.hip cfg_combine_kernel)
            out = E.cfg_combine(lg, B, lens, lg.shape[-1], [lad.cfg_t(state.cfg, cur + j) for j in range(state.g)])
        return out, state.g

    def _match(self, draft_tokens, target_logits, B: int, rule: E.MatchRule, draft_logits=None):
        if not draft_tokens or not target_logits or len(draft_tokens) != len(target_logits):
            return 0, None
        dev = target_logits[0].device
        lens = [int(t.shape[1]) for t in draft_tokens]
        ids = torch.cat([t.to(dev) for t in draft_tokens], 1).contiguous()
        lg = torch.cat(target_logits, 1)
        lg2 = torch.cat([lg, lg], 0).contiguous()          # t = 0 makes the kernel's CFG the identity on the first B rows
        dl = None
        if rule.rule == "kl":
            if draft_logits is None or len(draft_logits) != len(draft_tokens):
                raise ValueError("the KL rule compares distributions: pass the draft's CFG logits per stage (draft_logits=[(B, pn^2, V), ...])")
            dl = torch.cat([torch.cat([d, d], 0).reshape(-1) for d in draft_logits]).to(dev).contiguous()     # stage j as (2B, l_j, V)
        counts = torch.zeros(40, dtype=torch.int32, device=dev)
        match = torch.zeros(B, ids.shape[1], dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            E.verify_accept(lg2, B, lens, lg.shape[-1], [0.0] * len(lens), ids, 0, ids.shape[1], self.match_threshold, counts, rule=rule, draft_logits=dl,
                            match_out=match)
        return int(counts[16].item()), match

    def basic_token_matching(self, draft_tokens, target_logits, state, B: int) -> int:
        """var.py:1160-1227 on arbitrary (tokens, CFG logits) lists: leading stages whose batch match rate is >= 0.5."""
        return self._match(draft_tokens, target_logits, B, E.MatchRule())[0]

    def advanced_token_matching(self, draft_tokens, target_logits, state, B: int, draft_logits=None) -> int:
        """var.py:1229-1243.  The reference's body is a stub that returns basic_token_matching; with the default `self.match_rule` this does
        the same.  Setting `self.match_rule = MatchRule('topk', top_k=k)` / `MatchRule('kl', kl_thr=x)` switches on the rules its docstring
        lists (top-k membership, KL threshold); `token_level=True` (partial acceptance) acts in the sampling loop, see
        sdvar_amd.engine.Sampler.spec_correct.  The per-token verdicts of the last call are kept in `self.last_match`."""
        n, self.last_match = self._match(draft_tokens, target_logits, B, self.match_rule, draft_logits)
        return n

    def update_state_with_accepted_tokens(self, draft_tokens, accept_length: int, state: E.SpecState, B: int):
        """var.py:1245-1282 (+ the stage advance of var.py:1349-1350 and the KV rollback the reference lacks)."""
        state.sampler.spec_commit(state, max(int(accept_length), 0))

    @torch.no_grad()
    def sdvar_autoregressive_infer_cfg_parallel_v1(self, B: int, label_B: Optional[Union[int, torch.LongTensor]] = None, g_seed: Optional[int] = None,
                                                   cfg: float = 1.5, gamma: int = 2, top_k: int = 0, top_p: float = 0.0, more_smooth: bool = False) -> torch.Tensor:
        """var.py:1284-1383 with the resolved semantics of SURVEY.md App. C.1: ONE call of engine.Sampler.spec_decode, i.e. the loop the
        step-wise helpers above spell out, with the verifier running behind / ahead of the draft where the policy allows it (same tokens,
        same counters: tests/test_gpu_e2e.py).  more_smooth is accepted and, exactly as in the reference, has no effect on this path
        (var.py:1315 stores it, draft_generate_batch var.py:949-1024 never reads it)."""
        t = self.target_model
        labels, smp, noise = self._prepare(B, label_B, g_seed, gamma)
        res = smp.spec_decode(labels, cfg, gamma, top_k, top_p, noise, thr=self.match_threshold, run_ahead=True, accept_scope=self.accept_scope,
                              match=self.match_rule)
        self.last_result = res
        return t.vae_proxy[0].fhat_to_img(res.f_hat.clone()).add_(1).mul_(0.5)

    @torch.no_grad()
    def sdvar_autoregressive_infer_cfg_sd_test3(self, B: int, label_B: Optional[Union[int, torch.LongTensor]], g_seed: Optional[int] = None, cfg: float = 1.5,
                                                top_k: int = 0, top_p: float = 0.0, more_smooth: bool = False, entry_num: int = 10, sd_mask: int = 0) -> torch.Tensor:
        """var.py:604-865: the draft samples stages 0 .. entry_num-1, the target continues from the shared f_hat.  One generator, the target
        model's (var.py:641-642); labels as in var.py:647-656.  All six sd_mask values are built (engine.Sampler.handoff;
        1, 2, 4, 5 run the prefill under the explicit block-wise masks of var.py:557-578)."""
        t = self.target_model
        rng = None
        if g_seed is not None:
            t.rng.manual_seed(g_seed); rng = t.rng
        S = len(t.patch_nums)
        labels, smp, noise = self._prepare(B, label_B, g_seed, entry_num + 1 if (sd_mask != 0 and entry_num < S) else 1, rng=rng)
        res = smp.handoff(labels, cfg, top_k, top_p, noise, entry_num, sd_mask, more_smooth=bool(more_smooth))
        self.last_result = res
        return t.vae_proxy[0].fhat_to_img(res.f_hat.clone()).add_(1).mul_(0.5)


def _factory_checks(patch_nums: Sequence[int]):
    if len(patch_nums) > E.MAX_STAGES:
        raise NotImplementedError(f"ladders longer than {E.MAX_STAGES} stages are not supported")


def build_vae_var(device, patch_nums=(1, 2, 3, 4, 5, 6, 8, 10, 13, 16), V=4096, Cvae=32, ch=160, share_quant_resi=4, num_classes=1000, depth=16,
                  shared_aln=False, attn_l2_norm=True, flash_if_available=True, fused_if_available=True, init_adaln=0.5, init_adaln_gamma=1e-5,
                  init_head=0.02, init_std=-1):
    """models/__init__.py:16-46 (same arguments and return value)."""
    _factory_checks(patch_nums)
    vae = VQVAE(vocab_size=V, z_channels=Cvae, ch=ch, test_mode=True, share_quant_resi=share_quant_resi, v_patch_nums=patch_nums).to(device)
    var = VAR(vae_local=vae, num_classes=num_classes, depth=depth, embed_dim=depth * 64, num_heads=depth, drop_rate=0., attn_drop_rate=0.,
              drop_path_rate=0.1 * depth / 24, norm_eps=1e-6, shared_aln=shared_aln, cond_drop_rate=0.1, attn_l2_norm=attn_l2_norm,
              patch_nums=patch_nums, flash_if_available=flash_if_available, fused_if_available=fused_if_available).to(device)
    var.init_weights(init_adaln=init_adaln, init_adaln_gamma=init_adaln_gamma, init_head=init_head, init_std=init_std)
    _init_vae(vae)
    return vae, var


def build_vae_var_speculative_decoding(device, patch_nums=(1, 2, 3, 4, 5, 6, 8, 10, 13, 16), V=4096, Cvae=32, ch=160, share_quant_resi=4,
                                       num_classes=1000, depth_draft=16, depth_target=30, shared_aln=False, attn_l2_norm=True,
                                       flash_if_available=True, fused_if_available=True, init_adaln=0.5, init_adaln_gamma=1e-5, init_head=0.02,
                                       init_std=-1, similarity_thresh=0.8):
    """models/__init__.py:51-97 (same arguments; returns (vae, draft, target, sd_var); one VQVAE shared by both models)."""
    _factory_checks(patch_nums)
    vae = VQVAE(vocab_size=V, z_channels=Cvae, ch=ch, test_mode=True, share_quant_resi=share_quant_resi, v_patch_nums=patch_nums).to(device)
    models = []
    for depth in (depth_draft, depth_target):
        m = VAR(vae_local=vae, num_classes=num_classes, depth=depth, embed_dim=depth * 64, num_heads=depth, drop_path_rate=0.1 * depth / 24,
                shared_aln=shared_aln, attn_l2_norm=attn_l2_norm, patch_nums=patch_nums, flash_if_available=flash_if_available,
                fused_if_available=fused_if_available).to(device)
        m.init_weights(init_adaln=init_adaln, init_adaln_gamma=init_adaln_gamma, init_head=init_head, init_std=init_std)
        models.append(m)
    _init_vae(vae)
    return vae, models[0], models[1], SDVAR(models[0], models[1], similarity_thresh)


def _init_vae(vae: VQVAE, seed: int = 1234):
    """The reference leaves the VQVAE uninitialised when no checkpoint is loaded; give it the seeded
    'perf' init so random-weight runs are finite and reproducible."""
    from .weights import vae_state_dict
    dev = next(vae.parameters()).device
    sd = vae_state_dict(vae.quantize.v_patch_nums, "perf", seed, V=vae.vocab_size, Cvae=vae.Cvae, ch=vae.decoder.conv_out.in_channels,
                        with_encoder=hasattr(vae, "encoder"))
    vae.load_state_dict({k: v.to(dev) for k, v in sd.items()})
