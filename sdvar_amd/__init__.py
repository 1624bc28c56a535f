"""sdvar_amd - MI355X-native speculative draft-verify sampler for VAR (drop-in for the reference's `models` package on
the sampling path).  See DESIGN.md for the path, the boundary and the kernels; include/sdvar_hip.h for the C ABI."""
from .var import SDVAR, VAR, VARHF, build_vae_var, build_vae_var_speculative_decoding
from .vqvae import VQVAE

__all__ = ["VQVAE", "VAR", "VARHF", "SDVAR", "build_vae_var", "build_vae_var_speculative_decoding"]
