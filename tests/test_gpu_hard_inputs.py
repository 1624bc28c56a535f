"""GPU: the benched arithmetic (GEMM mode f16x2: two fp16 planes per operand, activations limited to +-65504) on HARD inputs.

Every other end-to-end fixture uses random-init weights with O(1) activations.  Real checkpoints have heavy tails: a few residual-stream channels orders of
magnitude above the rest, adaLN scales far from 0, FFN pre-activations in the thousands.  `conftest.heavy_tailed` builds such a model (calibration in its
docstring: the fp32 oracle is within 2e-5 of an fp64 evaluation on it) and this file holds the HIP path to the north-star bar on it - token ids bit-exact or a
first flip on a draw the oracle itself had within 1e-3 of a tie, logits <= 1e-3 (basic_var.py:90-159, helpers.py:6-19) - with the f16x2 guard proving that no
operand saturated; then shows the guard firing, and the exact-split mode (bf16x3) still meeting the bar, when one activation is pushed out of fp16's range."""
import numpy as np
import pytest
import torch

from conftest import heavy_tailed, oracle_memo, state_dicts
from oracle import var_oracle as orc
from sdvar_amd import engine as E
from sdvar_amd.ladder import LADDER_256, LADDER_512, as_ladder
from sdvar_amd.noise import exponential_noise

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
LOGIT_TOL = 1e-3          # BASELINE.json north_star
FHAT_TOL = 1e-4


def _noise_o(seed):
    return orc.array_noise(lambda d, B, l, V: exponential_noise(seed, d, B, l, V))


def _run_case(dev, depth, pns, B, seed, gm, inject=0.0, guard=False, check=True):
    lad = as_ladder(pns)
    sd0, sd_v = state_dicts(depth, pns)
    sd = heavy_tailed(sd0, depth, inject=inject)
    labels = (torch.arange(B) * 331 + 17) % 1000
    tr = oracle_memo(("heavy", depth, tuple(pns), B, seed, inject),
                     lambda: orc.plain_ar(orc.OracleVAR(sd, depth, pns), orc.OracleQuant(sd_v, pns), labels, 1.5, 900, 0.96, _noise_o(seed), keep=True))
    ctx = E.ModelCtx(sd, depth, pns, B, 1, dev, gemm_mode=gm); qc = E.QuantCtx(sd_v, pns, B, dev)
    if guard:
        E.f16x2_guard(True)
    try:
        res = E.Sampler(ctx, qc).plain_ar(labels.to(dev), 1.5, 900, 0.96, E.Noise("host", seed), trace=True)
    finally:
        if guard:
            E.f16x2_guard(False)
    ids, want = res.ids.cpu().numpy(), torch.cat(tr.ids, 1).numpy()
    flip = None
    if not np.array_equal(ids, want):
        t = sorted(map(tuple, np.argwhere(ids != want)), key=lambda x: x[1])[0][1]
        flip = next(i for i in range(lad.S) if t < lad.cum[i])
        assert not check or tr.margins[flip] < 1e-3, f"ids differ from stage {flip} on although the oracle's draw there had a top-2 margin of {tr.margins[flip]:.2e}"
    errs = [float((res.trace["logits"][s].cpu() - tr.logits[s]).abs().max()) for s in range(lad.S if flip is None else flip + 1)]
    if flip is None:
        assert (res.f_hat.cpu() - tr.f_hat).abs().max().item() <= FHAT_TOL
    ctx.close(); qc.close()
    return errs, flip, res.stats.get("f16x2_guard"), min(tr.margins)


@pytest.mark.parametrize("depth,pns,B,seed", [(4, LADDER_256, 2, 14), (6, LADDER_512, 1, 13)])
def test_heavy_tailed_init_f16x2_vs_oracle(dev, depth, pns, B, seed):
    """Residual channels x1e3, adaLN scales x30, FFN pre-activations to +-1e4, both ladders, in the BENCHED mode: no operand saturates (guard), ids as the oracle's,
    logits <= 1e-3.  Prints what it measured so a drift shows up in the log before it reaches the bar."""
    errs, flip, guard, mm = _run_case(dev, depth, pns, B, seed, "f16x2", guard=True)
    print(f"\n[heavy d{depth} L={as_ladder(pns).L}] per-stage max|dlogit| {['%.1e' % e for e in errs]} first flip {flip} oracle min margin {mm:.1e} guard {guard}")
    assert guard is not None and guard["elements"] > 0 and guard["saturated"] == 0 and guard["non_finite"] == 0, guard
    assert max(errs) <= LOGIT_TOL, errs


def test_out_of_range_activation_trips_the_guard_and_bf16x3_stays_exact(dev):
    """One FFN hidden unit pushed past 65504 (fp32 handles it; fp16 planes cannot): the f16x2 guard reports saturated operand elements - the mode's documented limit,
    visible instead of silent - and the exact three-plane mode still meets the bar on the same weights."""
    errs_b, flip_b, _, _ = _run_case(dev, 4, LADDER_256, 2, 11, "bf16x3", inject=24.0)
    assert max(errs_b) <= LOGIT_TOL, errs_b
    _, flip_h, guard, _ = _run_case(dev, 4, LADDER_256, 2, 11, "f16x2", inject=24.0, guard=True, check=False)      # saturated operands: this run MAY leave the oracle's ids
    print(f"\n[injected] bf16x3 per-stage max|dlogit| {['%.1e' % e for e in errs_b]} (first flip {flip_b}); f16x2 guard {guard}, f16x2 first flip {flip_h}")
    assert guard["saturated"] > 0, guard
