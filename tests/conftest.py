import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu() -> bool:
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def dev():
    if not has_gpu():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rnd(seed, shape, scale=1.0):
    """Portable test inputs (numpy Philox), float32 torch tensor."""
    g = np.random.Generator(np.random.Philox(key=[seed, 4242]))
    return torch.from_numpy(g.standard_normal(size=tuple(shape), dtype=np.float32) * np.float32(scale))


_SD_CACHE = {}


def state_dicts(depth, patch_nums, mode="stress", seed=1234, vae=True, shared_aln=False, attn_l2_norm=True):
    from sdvar_amd.weights import var_state_dict, vae_state_dict
    key = (depth, tuple(patch_nums), mode, seed, shared_aln, attn_l2_norm)
    if key not in _SD_CACHE:
        _SD_CACHE[key] = var_state_dict(depth, patch_nums, mode, seed, shared_aln=shared_aln, attn_l2_norm=attn_l2_norm)
    vkey = ("vae", tuple(patch_nums), mode, seed)
    if vae and vkey not in _SD_CACHE:
        _SD_CACHE[vkey] = vae_state_dict(patch_nums, mode, seed, with_encoder=False)
    return _SD_CACHE[key], (_SD_CACHE[vkey] if vae else None)


_ORACLE_CACHE = {}


def oracle_memo(key, fn):
    """CPU-oracle results do not depend on the GEMM mode of the HIP path under test: compute each (weights, arguments) case once per session."""
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = fn()
    return _ORACLE_CACHE[key]


def heavy_tailed(sd, depth, hot_scale=1e3, ada_scale=30.0, ffn_scale=240.0, inject=0.0):
    """A heavy-tailed variant of a VAR state_dict (the activation ranges real checkpoints show and O(1) random inits do not): three residual-stream channels x1e3
    (position / level / word-embedding tables), the adaLN scale rows x30, three FFN hidden units per block x240 (pre-activations reach +-1e4, hidden values likewise:
    inside fp16's range, far outside O(1)).  inject > 0 multiplies one fc1 row of block 1 further so that its hidden value exceeds 65504 (the f16x2 mode's activation
    range).  Calibrated on the CPU: the oracle in fp32 and in fp64 agree to 2e-5 on the logits of this init (stages 0-5, d4 / d6), so 1e-3 is a meaningful bar."""
    sd = type(sd)((k, v.clone()) for k, v in sd.items())
    C = 64 * depth
    hot = [3, C // 2 + 1, C - 2]
    for k in ("pos_start", "pos_1LC", "lvl_embed.weight"):
        sd[k][..., hot] *= hot_scale
    sd["word_embed.bias"][hot] *= hot_scale
    for i in range(depth):
        p = f"blocks.{i}."
        sd[p + "ada_lin.1.weight"][2 * C:4 * C] *= ada_scale
        rows = [7 + 11 * i, 2 * C + 5, 4 * C - 3 - i]
        sd[p + "ffn.fc1.weight"][rows] *= ffn_scale
        if inject and i == 1:
            sd[p + "ffn.fc1.weight"][rows[1]] *= inject
    return sd
