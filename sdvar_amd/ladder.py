"""Scale ladders and the constant resampling tables of the VAR token pyramid.

Reference behaviour restated here (nothing is imported from the reference):
  * ladders: /root/reference/models/__init__.py:18 (256^2), utils/arg_util.py:244-249 (512^2, 1024^2)
  * begin/end offsets: models/var.py:41-47
  * Phi selection per stage: models/quant.py:218-226 (PhiPartiallyShared ticks)
  * bicubic-up / area-down as separable linear maps: SURVEY.md App. A.7 (ATen upsample_bicubic2d with
    A=-0.75, align_corners=False, clamped taps; adaptive_avg_pool2d windows)
These tables are what the HIP quant_next kernel consumes (sdvar_amd/csrc/quant_next.hip).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

LADDER_256 = (1, 2, 3, 4, 5, 6, 8, 10, 13, 16)
LADDER_512 = (1, 2, 3, 4, 6, 9, 13, 18, 24, 32)
LADDER_1024 = (1, 2, 3, 4, 5, 7, 9, 12, 16, 21, 27, 36, 48, 64)


def _cubic1(x: float, a: float) -> float:   # |x| <= 1
    return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0


def _cubic2(x: float, a: float) -> float:   # 1 < |x| < 2
    return ((a * x - 5.0 * a) * x + 8.0 * a) * x - 4.0 * a


def bicubic_up_matrix(n_in: int, n_out: int) -> np.ndarray:
    """(n_out, n_in) fp32 matrix W with up(h) = W h W^T  (one axis at a time).

    ATen computes the four tap weights in the tensor's dtype (fp32) from t = src - floor(src) with
    src = (dst + 0.5) * (n_in / n_out) - 0.5 evaluated in fp32 as well; we do the same so the table matches
    F.interpolate(mode='bicubic') to the last bit of each weight, then merge clamped taps by addition.
    """
    A = np.float32(-0.75)
    W = np.zeros((n_out, n_in), dtype=np.float64)
    scale = np.float32(n_in) / np.float32(n_out)
    for o in range(n_out):
        src = np.float32(scale * np.float32(o + 0.5) - np.float32(0.5))
        i0 = int(math.floor(float(src)))
        t = np.float32(src - np.float32(i0))
        x2 = np.float32(np.float32(1.0) - t)
        x = [np.float32(t + np.float32(1.0)), t, x2, np.float32(x2 + np.float32(1.0))]
        w = [
            np.float32(((A * x[0] - np.float32(5) * A) * x[0] + np.float32(8) * A) * x[0] - np.float32(4) * A),
            np.float32(((A + np.float32(2)) * x[1] - (A + np.float32(3))) * x[1] * x[1] + np.float32(1)),
            np.float32(((A + np.float32(2)) * x[2] - (A + np.float32(3))) * x[2] * x[2] + np.float32(1)),
            np.float32(((A * x[3] - np.float32(5) * A) * x[3] + np.float32(8) * A) * x[3] - np.float32(4) * A),
        ]
        for k in range(4):
            idx = min(max(i0 - 1 + k, 0), n_in - 1)
            W[o, idx] += float(w[k])
    return W.astype(np.float32)


def area_down_matrix(n_in: int, n_out: int) -> np.ndarray:
    """(n_out, n_in) fp32 matrix of adaptive average pooling (F.interpolate mode='area')."""
    W = np.zeros((n_out, n_in), dtype=np.float32)
    for o in range(n_out):
        s = (o * n_in) // n_out
        e = -((-(o + 1) * n_in) // n_out)
        W[o, s:e] = np.float32(1.0) / np.float32(e - s)
    return W


def phi_index(si: int, n_stages: int, n_phi: int = 4) -> int:
    """Which shared Phi conv stage `si` uses (models/quant.py:223-226)."""
    ticks = np.linspace(1 / 3 / n_phi, 1 - 1 / 3 / n_phi, n_phi) if n_phi == 4 else np.linspace(1 / 2 / n_phi, 1 - 1 / 2 / n_phi, n_phi)
    return int(np.argmin(np.abs(ticks - si / (n_stages - 1))))


@dataclass(frozen=True)
class Ladder:
    patch_nums: Tuple[int, ...]

    @property
    def S(self) -> int:
        return len(self.patch_nums)

    @property
    def lens(self) -> List[int]:
        return [p * p for p in self.patch_nums]

    @property
    def cum(self) -> List[int]:          # cum[s] = tokens in stages 0..s
        out, c = [], 0
        for n in self.lens:
            c += n
            out.append(c)
        return out

    @property
    def L(self) -> int:
        return sum(self.lens)

    @property
    def HW(self) -> int:
        return self.patch_nums[-1]

    def begin(self, s: int) -> int:
        return 0 if s == 0 else self.cum[s - 1]

    def cfg_t(self, cfg: float, s: int) -> float:      # models/var.py:190,199
        return cfg * (s / (self.S - 1))

    def tables(self, n_phi: int = 4):
        """Flattened constant tables for the device: up[s] (HW x pn_s), dn[s] (pn_{s+1} x HW), phi[s]."""
        up = [bicubic_up_matrix(p, self.HW) if s < self.S - 1 else np.eye(self.HW, dtype=np.float32)
              for s, p in enumerate(self.patch_nums)]
        dn = [area_down_matrix(self.HW, self.patch_nums[s + 1]) for s in range(self.S - 1)]
        phi = [phi_index(s, self.S, n_phi) for s in range(self.S)]
        return up, dn, phi


def as_ladder(patch_nums: Sequence[int]) -> Ladder:
    return Ladder(tuple(int(p) for p in patch_nums))
