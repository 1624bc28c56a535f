"""The C-ABI library loads without a GPU and exports every symbol include/sdvar_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "sdvar_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sdvar_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    from sdvar_amd import engine as E
    lib = E.load_library()
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/sdvar_hip.h but not exported by libsdvar_hip.so"
        assert s in E._SIGNATURES, f"{s} has no ctypes prototype in sdvar_amd/engine.py"
    assert set(E._SIGNATURES) == set(syms)
    assert lib.sdvar_abi_version() == E.ABI_VERSION == 4


def test_argument_errors_are_reported_without_gpu():
    from sdvar_amd import engine as E
    lib = E.load_library()
    h = ctypes.c_void_p()
    d = E._ModelDesc()
    d.depth, d.n_stages, d.vocab, d.cvae, d.num_classes, d.max_batch, d.max_chunk_stages = 0, 10, 4096, 32, 1000, 1, 1
    rc = lib.sdvar_model_create(ctypes.byref(d), ctypes.byref(h))
    assert rc == 1 and b"depth" in lib.sdvar_last_error()
    assert lib.sdvar_kv_len(None) == -1


def test_missing_library_fails_loudly(tmp_path):
    from sdvar_amd import engine as E
    import pytest
    saved = E._lib
    E._lib = None
    try:
        with pytest.raises(E.SdvarError):
            E.load_library(str(tmp_path / "nope.so"))
    finally:
        E._lib = saved


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sdvar_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports the oracle"


def test_product_has_no_torch_math():
    """sdvar_amd/ uses torch for device memory, streams and process groups only: no torch.nn.functional, matmul-like or softmax calls, and no nn.Module with a
    forward() (the PyTorch decoder the parity tests compare against is tests/torch_ref.py)."""
    pkg = os.path.join(ROOT, "sdvar_amd")
    pat = re.compile(r"torch\.nn\.functional|\bF\.[a-z_]+\(|torch\.(bmm|matmul|mm|einsum|softmax|conv2d|addmm)\b|\.softmax\(|\.bmm\(|\.matmul\(")
    import io
    import tokenize
    for f in sorted(os.listdir(pkg)):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            by_line = {}
            for tok in tokenize.generate_tokens(io.StringIO(src).readline):          # code only: strings (docstrings cite torch ops) and comments dropped
                if tok.type not in (tokenize.STRING, tokenize.COMMENT):
                    by_line.setdefault(tok.start[0], []).append(tok.string)
            for ln, toks in sorted(by_line.items()):
                code = "".join(toks)
                assert not pat.search(code), f"sdvar_amd/{f}:{ln}: torch math in the product package: {code.strip()}"
    import torch.nn as nn
    from sdvar_amd import var as V, vqvae as Q
    for mod in (V, Q):
        for name, cls in vars(mod).items():
            if isinstance(cls, type) and issubclass(cls, nn.Module) and cls.__module__ == mod.__name__ and "forward" in cls.__dict__:
                assert name == "VAR", f"{mod.__name__}.{name} defines forward(): parameter containers only"      # VAR.forward raises (training is out of scope)


def test_vae_bind_order_matches_the_state_dict_walk():
    """sdvar_vae_tensor_count (host logic of csrc/vae.hip) == the tensors engine.VaeCtx would hand to sdvar_vae_bind, and every name
    exists in the decoder's state_dict (the key set of vae_ch160v4096z32.pth)."""
    import ctypes as C
    from sdvar_amd import engine as E
    from sdvar_amd.weights import vae_state_dict
    lib = E.load_library()
    for ch in (160, 32):
        sd = vae_state_dict((1, 2, 3, 4, 5, 6, 8, 10, 13, 16), "perf", 1, ch=ch, with_encoder=False)
        names = E.VaeCtx.tensor_names(sd)
        assert all(n + ".weight" in sd and n + ".bias" in sd for n in names)
        used = {n + s for n in names for s in (".weight", ".bias")}
        assert used == {k for k in sd if k.startswith(("decoder.", "post_quant_conv."))}          # nothing of the decoder is left unbound
        d = E._VaeDesc()
        d.ch, d.z_channels, d.n_mult, d.num_res_blocks, d.max_batch, d.latent_hw = ch, 32, 5, 2, 1, 16
        for i, m in enumerate((1, 1, 2, 2, 4)):
            d.ch_mult[i] = m
        assert lib.sdvar_vae_tensor_count(C.byref(d)) == 2 * len(names)


def test_varhf_local_round_trip(tmp_path):
    """models/var.py:513-533: VARHF builds its VQVAE from vae_kwargs and saves / loads through the hub mixin from a LOCAL directory."""
    import torch
    from sdvar_amd import VARHF
    m = VARHF(vae_kwargs=dict(vocab_size=64, ch=32, with_encoder=False), depth=2, embed_dim=128, num_heads=2, attn_l2_norm=True)
    with torch.no_grad():
        m.pos_start.normal_()
    m.save_pretrained(tmp_path)
    m2 = VARHF.from_pretrained(tmp_path)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
