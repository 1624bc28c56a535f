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
