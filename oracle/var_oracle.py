"""CPU ORACLE - TEST INFRASTRUCTURE ONLY.  Never imported by the product path (sdvar_amd/*).

A plain PyTorch-CPU fp32 restatement of the reference's sampling path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module, and only as the checker / the reported CPU baseline.

Pinning: tests/golden/make_golden.py imports the reference (CPU) in the build container, runs it beside this file on
the same weights and noise and stores the outputs as fixtures; tests/test_oracle_golden.py replays them.  For plain
AR and every op the oracle is pinned bit-exactly on token ids and <=1e-6 on logits.  The speculative loop has no
runnable reference (its entry point crashes) => for `spec_decode` "parity unpinned" by the reference;
it is pinned by reference *components* plus the invariants I1-I6 of SURVEY.md App. C.4.

Reference lines followed (all under /root/reference/models/):
  prologue            var.py:162-170, 580-601
  block / attention   basic_var.py:90-119, 152-159 ; FFN basic_var.py:44-52
  head + CFG          var.py:119-125, 199-200 ; basic_var.py:172-174
  sampler             helpers.py:6-19   (multinomial == argmax(p/q), q~Exp(1): SURVEY.md F6)
  quant next-input    quant.py:187-196, 199-206, 218-226 ; var.py:186-188, 205-211
  draft round         var.py:949-1024 ; acceptance var.py:1160-1227 ; loop policy var.py:1318-1372
  hand-off sampler    var.py:604-865 (sd_mask 0 and 3) ; more_smooth var.py:206-208 + helpers.py:22-36
  richer acceptance   var.py:1229-1243 is a stub in the reference (it returns the basic result): top-k membership, KL threshold
                      and token-level partial acceptance below are this build's statement of its docstring - "parity unpinned"
  decode              vqvae.py:62-63, basic_vae.py:163-226, var.py:215
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
NoiseFn = Callable[[int, int, int, int], Tensor]      # (draw, B, l, V) -> q (B*l, V) float32


# ------------------------------------------------------------------------------------------------ model container
class OracleVAR:
    """Weights (reference state_dict names) + dims of one VAR transformer, with an explicit KV cache."""

    def __init__(self, sd: Dict[str, Tensor], depth: int, patch_nums: Sequence[int], num_classes: int = 1000, kv_fp16: bool = False):
        self.kv_fp16 = kv_fp16        # BASELINE config P4: cache entries rounded to fp16 (all arithmetic stays fp32)
        self.sd = {k: v.detach().to(torch.float32) if v.is_floating_point() else v for k, v in sd.items()}
        self.depth, self.C, self.H = depth, 64 * depth, depth
        self.patch_nums = tuple(patch_nums)
        self.S = len(self.patch_nums)
        self.lens = [p * p for p in self.patch_nums]
        self.cum = list(np.cumsum(self.lens))
        self.L = int(self.cum[-1])
        self.num_classes = num_classes
        self.V = self.sd["head.weight"].shape[0]
        self.Cvae = self.sd["word_embed.weight"].shape[1]
        self.kv: List[Optional[Tuple[Tensor, Tensor]]] = [None] * depth
        self.kv_base = 0              # token position of the cache's first key (hand-off sampler with sd_mask 0: the prefix is absent)

    def begin(self, s: int) -> int:
        return 0 if s == 0 else int(self.cum[s - 1])

    # -- prologue (var.py:162-170 / 580-601)
    def prologue(self, label_B: Tensor):
        B = label_B.shape[0]
        sd = self.sd
        lab = torch.cat((label_B, torch.full_like(label_B, self.num_classes)), dim=0)
        cond = sd["class_emb.weight"][lab]                                    # (2B, C)
        lvl_1L = torch.cat([torch.full((n,), i, dtype=torch.int64) for i, n in enumerate(self.lens)])
        lvl_pos = sd["lvl_embed.weight"][lvl_1L].unsqueeze(0) + sd["pos_1LC"]  # (1, L, C)
        first = cond.unsqueeze(1).expand(2 * B, 1, -1) + sd["pos_start"].expand(2 * B, 1, -1) + lvl_pos[:, :1]
        return cond, lvl_pos, first

    def kv_reset(self):
        self.kv = [None] * self.depth
        self.kv_base = 0

    def kv_len(self) -> int:
        return 0 if self.kv[0] is None else self.kv[0][0].shape[2]

    def kv_truncate(self, n: int):
        self.kv = [None if (kv is None or n == 0) else (kv[0][:, :, :n].clone(), kv[1][:, :, :n].clone()) for kv in self.kv]

    # -- one transformer block (basic_var.py:152-159, 90-119, 44-52)
    def _block(self, i: int, x: Tensor, cond: Tensor, mask: Optional[Tensor]) -> Tensor:
        sd, C, H = self.sd, self.C, self.H
        p = f"blocks.{i}."
        if p + "ada_gss" in sd:      # shared_aln (var.py:81,192 + basic_var.py:153-154): per-block offset + one shared SiLU-Linear of cond
            ada = sd[p + "ada_gss"] + F.linear(F.silu(cond), sd["shared_ada_lin.1.weight"], sd["shared_ada_lin.1.bias"]).view(-1, 1, 6, C)
        else:
            ada = F.linear(F.silu(cond), sd[p + "ada_lin.1.weight"], sd[p + "ada_lin.1.bias"]).view(-1, 1, 6, C)
        g1, g2, s1, s2, sh1, sh2 = ada.unbind(2)
        R, l, _ = x.shape
        h = F.layer_norm(x, (C,), eps=1e-6).mul(s1.add(1)).add_(sh1)
        qkv = F.linear(h, sd[p + "attn.mat_qkv.weight"],
                       torch.cat((sd[p + "attn.q_bias"], torch.zeros(C), sd[p + "attn.v_bias"]))).view(R, l, 3, H, 64)
        q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)                       # (R, H, l, 64)
        l2 = (p + "attn.scale_mul_1H11") in sd           # attn_l2_norm (basic_var.py:66-72, 101-105); False: plain attention, scale 0.25 / sqrt(64)
        if l2:
            scale_mul = sd[p + "attn.scale_mul_1H11"].clamp_max(math.log(100.0)).exp()
            q = F.normalize(q, dim=-1).mul(scale_mul)
            k = F.normalize(k, dim=-1)
        if self.kv_fp16:
            k, v = k.half().float(), v.half().float()
        if self.kv[i] is not None:
            k = torch.cat((self.kv[i][0], k), dim=2); v = torch.cat((self.kv[i][1], v), dim=2)
        self.kv[i] = (k, v)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, scale=1.0 if l2 else 0.25 / math.sqrt(64)).transpose(1, 2).reshape(R, l, C)
        x = x + F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"]).mul_(g1)
        h = F.layer_norm(x, (C,), eps=1e-6).mul(s2.add(1)).add_(sh2)
        h = F.linear(F.gelu(F.linear(h, sd[p + "ffn.fc1.weight"], sd[p + "ffn.fc1.bias"]), approximate="tanh"),
                     sd[p + "ffn.fc2.weight"], sd[p + "ffn.fc2.bias"])
        return x + h.mul(g2)

    def chunk_mask(self, s0: int, n: int) -> Optional[Tensor]:
        """Rows of the block-causal mask (var.py:108-113) for a query chunk covering stages s0..s0+n-1 given that the
        KV cache holds exactly stages < s0: shape (1,1,l_chunk, cum[s0+n-1]); None for a single stage."""
        if n == 1:
            return None
        bg, ed = self.begin(s0), int(self.cum[s0 + n - 1])
        lvl = torch.cat([torch.full((m,), i) for i, m in enumerate(self.lens)])[:ed]
        d = lvl.view(ed, 1)
        full = torch.where(d >= d.t(), 0.0, -torch.inf)
        return full[bg:ed, :ed].reshape(1, 1, ed - bg, ed).contiguous()

    # -- all blocks + head for one call (var.py:195-197, 119-125)
    def forward(self, x: Tensor, cond: Tensor, s0: int, n_stages: int = 1, mask: Optional[Tensor] = None) -> Tensor:
        """mask: explicit (1, 1, l, K) additive mask instead of the block-causal rows (the ablation masks of the hand-off sampler)."""
        assert self.kv_len() == self.begin(s0) - self.kv_base, (self.kv_len(), self.begin(s0), self.kv_base)
        if mask is None:
            mask = self.chunk_mask(s0, n_stages)
        assert mask is None or self.kv_base == 0
        for i in range(self.depth):
            x = self._block(i, x, cond, mask)
        return self.head(x, cond)

    def head(self, x: Tensor, cond: Tensor) -> Tensor:
        """VAR.get_logits (var.py:119-125, basic_var.py:172-174)."""
        sd, C = self.sd, self.C
        sc, sh = F.linear(F.silu(cond), sd["head_nm.ada_lin.1.weight"], sd["head_nm.ada_lin.1.bias"]).view(-1, 1, 2, C).unbind(2)
        h = F.layer_norm(x, (C,), eps=1e-6).mul(sc.add(1)).add_(sh)
        return F.linear(h, sd["head.weight"], sd["head.bias"])

    def embed_next(self, nxt: Tensor, lvl_pos: Tensor, s_next: int, pos_begin: Optional[int] = None) -> Tensor:
        """var.py:186-188: next (B,Cvae,pn,pn) -> (2B, pn^2, C).  pos_begin: first lvl_pos row if not the stage's own (var.py:385 in the resumed sampler)."""
        B = nxt.shape[0]
        t = nxt.view(B, self.Cvae, -1).transpose(1, 2)
        bg = self.begin(s_next) if pos_begin is None else pos_begin
        t = F.linear(t, self.sd["word_embed.weight"], self.sd["word_embed.bias"]) + lvl_pos[:, bg:bg + self.lens[s_next]]
        return t.repeat(2, 1, 1)


class OracleQuant:
    """VectorQuantizer2 inference side: codebook + shared Phi convs (quant.py:187-229)."""

    def __init__(self, vae_sd: Dict[str, Tensor], patch_nums: Sequence[int], n_phi: int = None):
        self.patch_nums = tuple(patch_nums)
        self.S = len(patch_nums)
        self.codebook = vae_sd["quantize.embedding.weight"].to(torch.float32)
        # the three Phi layouts of quant.py:27-32: PhiShared keeps ONE conv ("qresi", quant.py:209-216: __getitem__ ignores its argument),
        # PhiPartiallyShared ("qresi_ls.<k>", quant.py:219-229) and PhiNonShared (an nn.ModuleList, "<k>", quant.py:232-243) pick by the nearest tick
        pre = "quantize.quant_resi."
        if pre + "qresi.weight" in vae_sd:
            names = [pre + "qresi"]
        else:
            stem = pre + "qresi_ls." if pre + "qresi_ls.0.weight" in vae_sd else pre
            names, k = [], 0
            while f"{stem}{k}.weight" in vae_sd:
                names.append(f"{stem}{k}"); k += 1
        assert names and (n_phi is None or n_phi == len(names)), (len(names), n_phi)
        self.phi = [(vae_sd[n + ".weight"].float(), vae_sd[n + ".bias"].float()) for n in names]
        K = len(names)
        self.ticks = np.linspace(1 / 3 / K, 1 - 1 / 3 / K, K) if K == 4 else np.linspace(1 / 2 / K, 1 - 1 / 2 / K, K)

    def phi_of(self, si: int) -> int:
        return int(np.argmin(np.abs(self.ticks - si / (self.S - 1))))

    def embed_ids(self, ids: Tensor, pn: int) -> Tensor:                     # var.py:205,210
        B = ids.shape[0]
        return self.codebook[ids].transpose(1, 2).reshape(B, self.codebook.shape[1], pn, pn)

    def next_input(self, si: int, f_hat: Tensor, h: Tensor) -> Tuple[Tensor, Tensor]:
        """quant.py:187-196 without the in-place aliasing: returns (new f_hat, next-scale map)."""
        HW = self.patch_nums[-1]
        w, b = self.phi[self.phi_of(si)]
        if si != self.S - 1:
            h = F.interpolate(h, size=(HW, HW), mode="bicubic")
        h = h.mul(0.5) + F.conv2d(h, w, b, padding=1).mul_(0.5)
        f_hat = f_hat + h
        if si != self.S - 1:
            pn = self.patch_nums[si + 1]
            return f_hat, F.interpolate(f_hat, size=(pn, pn), mode="area")
        return f_hat, f_hat


# ------------------------------------------------------------------------------------------------ sampler / accept
def cfg_combine(logits_2B: Tensor, B: int, t: float) -> Tensor:               # var.py:199-200
    return (1 + t) * logits_2B[:B] - t * logits_2B[B:]


def sample_topk_topp(logits: Tensor, top_k: int, top_p: float, q: Tensor, margin_out: Optional[list] = None) -> Tuple[Tensor, Tensor]:
    """helpers.py:6-19 with the multinomial replaced by its exact equivalent argmax(p/q).
    Returns (ids (B,l) int64, masked logits)."""
    B, l, V = logits.shape
    logits = logits.clone()
    if top_k > 0:
        kth = logits.topk(top_k, largest=True, sorted=False, dim=-1)[0].amin(dim=-1, keepdim=True)
        logits.masked_fill_(logits < kth, -torch.inf)
    if top_p > 0:
        s, idx = logits.sort(dim=-1, descending=False)
        rm = s.softmax(dim=-1).cumsum_(dim=-1) <= (1 - top_p)
        rm[..., -1:] = False
        logits.masked_fill_(rm.scatter(idx.ndim - 1, idx, rm), -torch.inf)
    p = logits.softmax(dim=-1).view(-1, V)
    ratio = p / q.view(-1, V)
    ids = torch.argmax(ratio, dim=-1).view(B, l)
    if margin_out is not None:      # smallest relative gap between the winner and the runner-up: how close a draw was to a tie
        top2 = ratio.topk(2, dim=-1)[0]
        margin_out.append(float(((top2[:, 0] - top2[:, 1]) / top2[:, 0]).min()))
    return ids, logits


def accept_scan(draft_ids: List[Tensor], cfg_logits: List[Tensor], thr: float = 0.5) -> Tuple[int, List[int], List[int]]:
    """var.py:1199-1222: per stage batch-mean of (draft == argmax_V target) >= thr, stop at the first failure.
    Returns (n_accept, matched[j], total[j]) with matched/total filled for every stage (not only the scanned ones)."""
    matched, total, n, alive = [], [], 0, True
    for ids, lg in zip(draft_ids, cfg_logits):
        m = (ids == torch.argmax(lg, dim=-1))
        matched.append(int(m.sum().item())); total.append(m.numel())
        rate = m.float().mean().item()
        if alive and rate >= thr:
            n += 1
        else:
            alive = False
    return n, matched, total


@dataclass
class MatchRule:
    """Mirror of sdvar_amd.engine.MatchRule (the token rule of the acceptance scan)."""
    rule: str = "top1"          # 'top1' (var.py:1199-1203) | 'topk' | 'kl'
    top_k: int = 1
    kl_thr: float = 0.0
    token_level: bool = False


def token_matches(ids: Tensor, cfg_logits: Tensor, rule: MatchRule, draft_cfg_logits: Optional[Tensor] = None) -> Tensor:
    """(B, l) bool: does the draft token satisfy the rule against the target's CFG logits (B, l, V)."""
    if rule.rule == "top1":
        return ids == torch.argmax(cfg_logits, dim=-1)
    if rule.rule == "topk":         # fewer than k vocabulary entries score strictly higher than the draft token
        xd = cfg_logits.gather(-1, ids.unsqueeze(-1))
        return (cfg_logits > xd).sum(-1) < rule.top_k
    if rule.rule == "kl":           # KL(softmax target || softmax draft), evaluated in float64 from the float32 logits
        lt = cfg_logits.double().log_softmax(-1); ld = draft_cfg_logits.double().log_softmax(-1)
        return (lt.exp() * (lt - ld)).sum(-1) <= float(np.float32(rule.kl_thr))
    raise ValueError(rule.rule)


def accept_scan_ex(draft_ids: List[Tensor], cfg_logits: List[Tensor], thr: float, rule: MatchRule, draft_cfg_logits: Optional[List[Tensor]] = None):
    """accept_scan with an arbitrary token rule; also returns the per-stage match masks and the corrected ids
    (draft id where the rule holds, target argmax elsewhere)."""
    matched, total, masks, corrected, n, alive = [], [], [], [], 0, True
    for j, (ids, lg) in enumerate(zip(draft_ids, cfg_logits)):
        m = token_matches(ids, lg, rule, None if draft_cfg_logits is None else draft_cfg_logits[j])
        masks.append(m); corrected.append(torch.where(m, ids, torch.argmax(lg, dim=-1)))
        matched.append(int(m.sum().item())); total.append(m.numel())
        rate = m.float().mean().item()
        if alive and rate >= thr:
            n += 1
        else:
            alive = False
    return n, matched, total, masks, corrected


def gumbel_mix(masked_cfg_logits: Tensor, ratio: float, e: Tensor, codebook: Tensor) -> Tensor:
    """var.py:206-208 + helpers.py:22-36 (rng given, hard=False): softmax((logits * (1 + ratio) + (-log E)) / tau) @ codebook -> (B, l, Cvae).
    `masked_cfg_logits` are the logits after the sampler masked them in place (helpers.py:10,15); e ~ Exp(1), shape (B, l, V)."""
    tau = max(0.27 * (1 - ratio * 0.95), 0.005)
    gumbels = (masked_cfg_logits.mul(1 + ratio) + (-e.log())) / tau
    return gumbels.softmax(-1) @ codebook.unsqueeze(0)


GUMBEL_DRAW = 0x40000000      # noise(draw | GUMBEL_DRAW, ...) is the stage's gumbel exponential, drawn right after its multinomial


# ------------------------------------------------------------------------------------------------ plain AR
@dataclass
class ARTrace:
    ids: List[Tensor] = field(default_factory=list)          # per stage (B, pn^2) int64
    logits: List[Tensor] = field(default_factory=list)       # per stage raw (2B, pn^2, V)
    cfg_logits: List[Tensor] = field(default_factory=list)   # per stage (B, pn^2, V) before masking
    x_in: List[Tensor] = field(default_factory=list)         # per stage (2B, pn^2, C)
    f_hat: Optional[Tensor] = None
    stats: Dict[str, object] = field(default_factory=dict)
    margins: List[float] = field(default_factory=list)      # per sampler call: min relative top-2 gap of p/q (tie detector)


def _stage_features(quant: OracleQuant, cl: Tensor, pn: int, top_k: int, top_p: float, noise: NoiseFn, draw: int, more_smooth: bool, ratio: float,
                    margins: Optional[list]):
    """Sample one stage (helpers.py:6-19) and return (ids, h (B, Cvae, pn, pn)): codebook rows, or the gumbel mix when more_smooth."""
    B, l, V = cl.shape
    ids, masked = sample_topk_topp(cl, top_k, top_p, noise(draw, B, l, V), margins)
    if not more_smooth:
        return ids, quant.embed_ids(ids, pn)
    h = gumbel_mix(masked, ratio, noise(draw | GUMBEL_DRAW, B, l, V).view(B, l, V), quant.codebook)
    return ids, h.transpose(1, 2).reshape(B, quant.codebook.shape[1], pn, pn)


def plain_ar(model: OracleVAR, quant: OracleQuant, label_B: Tensor, cfg: float, top_k: int, top_p: float,
             noise: NoiseFn, keep: bool = True, more_smooth: bool = False) -> ARTrace:
    """var.py:127-215 up to (not including) the image decode."""
    B = label_B.shape[0]
    S = model.S
    cond, lvl_pos, x = model.prologue(label_B)
    f_hat = torch.zeros(B, model.Cvae, model.patch_nums[-1], model.patch_nums[-1])
    model.kv_reset()
    tr = ARTrace()
    for si, pn in enumerate(model.patch_nums):
        logits = model.forward(x, cond, si, 1)
        t = cfg * (si / (S - 1))
        cl = cfg_combine(logits, B, t)
        ids, h = _stage_features(quant, cl, pn, top_k, top_p, noise, si, more_smooth, si / (S - 1), tr.margins)
        if keep:
            tr.logits.append(logits); tr.cfg_logits.append(cl); tr.x_in.append(x)
        tr.ids.append(ids)
        f_hat, nxt = quant.next_input(si, f_hat, h)
        if si != S - 1:
            x = model.embed_next(nxt, lvl_pos, si + 1)
    model.kv_reset()
    tr.f_hat = f_hat
    return tr


def resume_ar(model: OracleVAR, quant: OracleQuant, cond: Tensor, current_step: int, step: int, next_map: Optional[Tensor], f_hat: Tensor, cfg: float,
              top_k: int, top_p: float, noise: NoiseFn, more_smooth: bool = False, margins: Optional[list] = None):
    """VAR.autoregressive_infer_cfg_sd_helper1 (var.py:319-443): stages current_step .. current_step + step - 1 from a handed-in state.  The KV cache is
    EMPTY at entry (var.py:368 -> basic_var.py:87) and the position rows are counted from the stage the call starts at (var.py:352: `cur_L = 0`, and the
    skipped stages `continue` before var.py:389 advances it).  Histories as var.py:436-443; f_hat snapshots are CLONES here (the reference's list holds one aliased
    tensor, quant.py:191) - compare the last entry; the logits history holds the CFG logits AFTER the in-place top-k / top-p masking, as the reference's
    does.  Draw index of stage si = si."""
    B, S = cond.shape[0] // 2, model.S
    sd = model.sd
    lvl_1L = torch.cat([torch.full((n,), i, dtype=torch.int64) for i, n in enumerate(model.lens)])
    lvl_pos = sd["lvl_embed.weight"][lvl_1L].unsqueeze(0) + sd["pos_1LC"]
    model.kv_reset()
    model.kv_base = model.begin(current_step)
    inputs, f_hist, logit_hist, id_hist = [], [], [], []
    nxt = next_map
    end = min(current_step + step, S)
    for si in range(current_step, end):
        pn = model.patch_nums[si]
        if si == 0:
            x = cond.unsqueeze(1).expand(2 * B, 1, -1) + sd["pos_start"].expand(2 * B, 1, -1) + lvl_pos[:, :1]          # var.py:374-378
        else:
            inputs.append(nxt.reshape(B, model.Cvae, -1).transpose(1, 2).clone())                                      # var.py:382-384
            x = model.embed_next(nxt.reshape(B, model.Cvae, pn, pn), lvl_pos, si, model.begin(si) - model.begin(current_step))   # var.py:352, 385, 389: cur_L counts from current_step
        f_hist.append(f_hat.clone())
        logits = model.forward(x, cond, si, 1)
        cl = cfg_combine(logits, B, cfg * (si / (S - 1)))
        ids, masked = sample_topk_topp(cl, top_k, top_p, noise(si, B, pn * pn, model.V), margins)
        logit_hist.append(masked)                      # var.py:408 appends the tensor helpers.py:10,15 then mask IN PLACE: the history holds -inf at the removed entries
        id_hist.append(ids)
        if not more_smooth:
            h = quant.embed_ids(ids, pn)
        else:
            ratio = si / (S - 1)
            h = gumbel_mix(masked, ratio, noise(si | GUMBEL_DRAW, B, pn * pn, model.V).view(B, pn * pn, model.V), quant.codebook)
            h = h.transpose(1, 2).reshape(B, quant.codebook.shape[1], pn, pn)
        f_hat, nxt = quant.next_input(si, f_hat, h)
    f_hist.append(f_hat.clone())
    inputs.append(nxt)                                                                                                 # var.py:430 (raw (B, Cvae, pn', pn'); f_hat after the last stage)
    model.kv_reset()
    return inputs, f_hist, logit_hist, id_hist


# ------------------------------------------------------------------------------------------------ hand-off sampler
def handoff_mask(model: OracleVAR, entry_num: int, sd_mask: int) -> Tensor:
    """(1, 1, p, p) additive masks of var.py:557-578 cut to the prefix + entry stage (var.py:780-798).  sd_mask 1 / 2: attn_bias_for_sdmasking
    (j <= i and not (same stage, i != j)); 4 / 5: attn_bias_for_block (same stage only); 2 and 5 set the entry stage's rows to 0."""
    p, s0 = int(model.cum[entry_num]), model.begin(entry_num)
    blk = torch.cat([torch.full((m,), i) for i, m in enumerate(model.lens)])[:p]
    out = torch.full((p, p), -torch.inf)
    for i in range(p):
        for j in range(p):
            if sd_mask in (1, 2):
                if j > i or (blk[i] == blk[j] and i != j):
                    continue
            elif blk[i] != blk[j]:
                continue
            out[i, j] = 0.0
    if sd_mask in (2, 5):
        out[s0:p, :] = 0.0
    return out.reshape(1, 1, p, p)


def handoff(draft: OracleVAR, target: OracleVAR, quant: OracleQuant, label_B: Tensor, cfg: float, top_k: int, top_p: float, noise: NoiseFn,
            entry_num: int, sd_mask: int = 0, more_smooth: bool = False) -> ARTrace:
    """SDVAR.sdvar_autoregressive_infer_cfg_sd_test3 (var.py:604-865), up to the decode (sd_mask 1, 2, 4, 5: as 3 with the explicit masks above).  The draft samples stages
    < entry_num (var.py:669-723); the target takes over the shared f_hat.  sd_mask 0: entry stage on an empty target cache (var.py:817-824).
    sd_mask 3: prefix + entry stage through the target blocks under the block-causal mask (var.py:789, 802-804), entry-stage logits from
    the INPUT token map (var.py:809-811, literal).  One noise stream, draw = stage index."""
    assert sd_mask in (0, 1, 2, 3, 4, 5)
    B, S, pns = label_B.shape[0], draft.S, draft.patch_nums
    tr = ARTrace()
    d_cond, d_lvl, d_x = draft.prologue(label_B)
    f_hat = torch.zeros(B, draft.Cvae, pns[-1], pns[-1])
    draft.kv_reset(); target.kv_reset()
    hub = []                                                   # draft_token_hub: the next-scale maps (B, Cvae, pn', pn') of stages 1..entry_num
    nxt = None
    for si in range(min(entry_num, S)):
        lg = draft.forward(d_x, d_cond, si, 1)
        cl = cfg_combine(lg, B, cfg * (si / (S - 1)))
        ids, h = _stage_features(quant, cl, pns[si], top_k, top_p, noise, si, more_smooth, si / (S - 1), tr.margins)
        tr.ids.append(ids)
        f_hat, nxt = quant.next_input(si, f_hat, h)
        if si != S - 1:
            hub.append(nxt)
            d_x = draft.embed_next(nxt, d_lvl, si + 1)
    draft.kv_reset()
    if entry_num < S:
        t_cond, t_lvl, t_first = target.prologue(label_B)
        xs = [t_first] + [target.embed_next(n, t_lvl, i + 1) for i, n in enumerate(hub)]        # target inputs of stages 0..entry_num
        for si in range(entry_num, S):
            if si == entry_num:
                x = xs[entry_num]
                if sd_mask == 0:
                    target.kv_base = target.begin(si)           # empty cache: the entry stage only sees itself
                    lg = target.forward(x, t_cond, si, 1)
                else:
                    mask = None if sd_mask == 3 else handoff_mask(target, entry_num, sd_mask)
                    target.forward(torch.cat(xs, dim=1), t_cond, 0, entry_num + 1, mask)          # fills the cache; logits unused
                    lg = target.head(x, t_cond)
            else:
                lg = target.forward(x, t_cond, si, 1)
            cl = cfg_combine(lg, B, cfg * (si / (S - 1)))
            ids, h = _stage_features(quant, cl, pns[si], top_k, top_p, noise, si, more_smooth, si / (S - 1), tr.margins)
            tr.ids.append(ids)
            f_hat, nxt = quant.next_input(si, f_hat, h)
            if si != S - 1:
                x = target.embed_next(nxt, t_lvl, si + 1)
        target.kv_reset()
    tr.f_hat = f_hat
    return tr


# ------------------------------------------------------------------------------------------------ speculative loop
def spec_decode(draft: OracleVAR, target: OracleVAR, quant: OracleQuant, label_B: Tensor, cfg: float, gamma: int,
                top_k: int, top_p: float, noise: NoiseFn, thr: float = 0.5, keep: bool = False, match: Optional[MatchRule] = None) -> ARTrace:
    """Resolved semantics of SDVAR.sdvar_autoregressive_infer_cfg_parallel_v1: draft gamma
    stages (var.py:949-1024), ONE target forward over them under the block-causal rows (var.py:1026-1070 intent),
    batch-level acceptance (var.py:1160-1227), commit / rollback, gamma policy (var.py:1353-1367, never break)."""
    assert draft.patch_nums == target.patch_nums
    B, S = label_B.shape[0], draft.S
    pns = draft.patch_nums
    d_cond, d_lvl, d_x = draft.prologue(label_B)
    t_cond, t_lvl, t_x = target.prologue(label_B)
    f_acc = torch.zeros(B, draft.Cvae, pns[-1], pns[-1])
    draft.kv_reset(); target.kv_reset()
    cur, draw = 0, 0
    tr = ARTrace()
    ids_acc: List[Tensor] = []
    st = dict(target_calls=0, draft_stage_calls=0, forced_accepts=0, accepted_tokens=0, rounds=[], gamma_final=gamma)
    while cur < S:
        g = min(gamma, S - cur)
        # ---- draft g stages
        ids_r, fh_r, dx_r, tx_r, dcl_r, nx_r = [], [], [d_x], [t_x], [], []
        f = f_acc
        for j in range(g):
            s = cur + j
            lg = draft.forward(dx_r[j], d_cond, s, 1)
            st["draft_stage_calls"] += 1
            cl = cfg_combine(lg, B, cfg * (s / (S - 1)))
            ids, _ = sample_topk_topp(cl, top_k, top_p, noise(draw, B, pns[s] ** 2, draft.V), tr.margins); draw += 1
            ids_r.append(ids); dcl_r.append(cl)
            f, nxt = quant.next_input(s, f, quant.embed_ids(ids, pns[s]))
            fh_r.append(f); nx_r.append(nxt)
            if s + 1 < S:
                dx_r.append(draft.embed_next(nxt, d_lvl, s + 1)); tx_r.append(target.embed_next(nxt, t_lvl, s + 1))
        # ---- one target forward over the g stages
        tl = target.forward(torch.cat(tx_r[:g], dim=1), t_cond, cur, g)
        st["target_calls"] += 1
        cls, off = [], 0
        for j in range(g):
            n = pns[cur + j] ** 2
            cls.append(cfg_combine(tl[:, off:off + n], B, cfg * ((cur + j) / (S - 1)))); off += n
        rule = match or MatchRule()
        n_acc, matched, total, _, corrected = accept_scan_ex(ids_r, cls, thr, rule, dcl_r)
        forced, acc_tok = False, None
        if n_acc < g and rule.token_level:
            # token-level partial acceptance: stage cur + n_acc keeps its matching tokens, the others take the target's argmax
            j = n_acc
            acc_tok = sum(pns[cur + i] ** 2 for i in range(j)) + matched[j]
            st["corrected_tokens"] = st.get("corrected_tokens", 0) + total[j] - matched[j]
            st["corrected_stages"] = st.get("corrected_stages", 0) + 1
            ids_r[j] = corrected[j]
            fh_r[j], nxt = quant.next_input(cur + j, f_acc if j == 0 else fh_r[j - 1], quant.embed_ids(ids_r[j], pns[cur + j]))
            if cur + j + 1 < S:
                dx, tx = draft.embed_next(nxt, d_lvl, cur + j + 1), target.embed_next(nxt, t_lvl, cur + j + 1)
                if len(dx_r) > j + 1:
                    dx_r[j + 1], tx_r[j + 1] = dx, tx
                else:
                    dx_r.append(dx); tx_r.append(tx)
            n_acc = j + 1
        elif n_acc == 0:
            if gamma > 1:
                gamma -= 1
            else:
                n_acc, forced = 1, True
                st["forced_accepts"] += 1
        st["rounds"].append(dict(stage=cur, g=g, matched=matched, total=total, n_accept=n_acc, forced=forced))
        if keep:
            tr.cfg_logits.append(cls)
        # ---- commit n_acc stages, roll both caches back to the accepted prefix
        if n_acc > 0:
            f_acc = fh_r[n_acc - 1]
            ids_acc.extend(ids_r[:n_acc])
            if not forced:
                st["accepted_tokens"] += sum(pns[cur + j] ** 2 for j in range(n_acc)) if acc_tok is None else acc_tok
            cur += n_acc
            if cur < S:
                d_x, t_x = dx_r[n_acc], tx_r[n_acc]
        keep_len = draft.begin(cur) if cur < S else draft.L
        draft.kv_truncate(keep_len); target.kv_truncate(keep_len)
    st["gamma_final"] = gamma
    draft.kv_reset(); target.kv_reset()
    tr.ids, tr.f_hat, tr.stats = ids_acc, f_acc, st
    return tr


# ------------------------------------------------------------------------------------------------ VQVAE decode
def _gn(x, sd, name): return F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], eps=1e-6)
def _conv(x, sd, name, pad): return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=pad)


def _res(x, sd, name):
    h = _conv(F.silu(_gn(x, sd, name + ".norm1")), sd, name + ".conv1", 1)
    h = _conv(F.silu(_gn(h, sd, name + ".norm2")), sd, name + ".conv2", 1)
    if (name + ".nin_shortcut.weight") in sd:
        x = _conv(x, sd, name + ".nin_shortcut", 0)
    return x + h


def _attn(x, sd, name):
    B, C, H, W = x.shape
    q, k, v = _conv(_gn(x, sd, name + ".norm"), sd, name + ".qkv", 0).reshape(B, 3, C, H * W).unbind(1)
    w = torch.bmm(q.transpose(1, 2), k).mul_(C ** -0.5).softmax(dim=2)        # (B, HWq, HWk)
    h = torch.bmm(v, w.transpose(1, 2)).view(B, C, H, W)
    return x + _conv(h, sd, name + ".proj_out", 0)


def decode_image(vae_sd: Dict[str, Tensor], f_hat: Tensor) -> Tensor:
    """vqvae.py:62-63 + basic_vae.py:207-226 + var.py:215 -> (B,3,H,W) in [0,1]."""
    sd = vae_sd
    h = _conv(_conv(f_hat, sd, "post_quant_conv", 1), sd, "decoder.conv_in", 1)
    h = _res(h, sd, "decoder.mid.block_1"); h = _attn(h, sd, "decoder.mid.attn_1"); h = _res(h, sd, "decoder.mid.block_2")
    n_lv = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("decoder.up."))
    for lv in reversed(range(n_lv)):
        ib = 0
        while f"decoder.up.{lv}.block.{ib}.norm1.weight" in sd:
            h = _res(h, sd, f"decoder.up.{lv}.block.{ib}")
            if f"decoder.up.{lv}.attn.{ib}.norm.weight" in sd:
                h = _attn(h, sd, f"decoder.up.{lv}.attn.{ib}")
            ib += 1
        if lv != 0:
            h = _conv(F.interpolate(h, scale_factor=2, mode="nearest"), sd, f"decoder.up.{lv}.upsample.conv", 1)
    h = _conv(F.silu(_gn(h, sd, "decoder.norm_out")), sd, "decoder.conv_out", 1)
    return h.clamp_(-1, 1).add_(1).mul_(0.5)


# ------------------------------------------------------------------------------------------------ noise sources
def torch_noise(gen: torch.Generator) -> NoiseFn:
    """The reference's own stream on this host: q = empty(B*l, V).exponential_(1, gen)."""
    def fn(draw, B, l, V):
        return torch.empty(B * l, V).exponential_(1, generator=gen)
    return fn


def array_noise(fn_np) -> NoiseFn:
    """Wrap a numpy generator (draw, B, l, V) -> (B, l, V) float32, e.g. sdvar_amd.noise.exponential_noise."""
    def fn(draw, B, l, V):
        return torch.from_numpy(np.ascontiguousarray(fn_np(draw, B, l, V))).view(B * l, V)
    return fn
