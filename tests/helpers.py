"""Shared helpers for the parity tests: golden-fixture summaries and tolerance checks."""
from __future__ import annotations

import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fp32 parity tolerance stated by the north star: 1e-5 relative (norm-wise for large tensors,
# elementwise with an absolute floor for sampled values).
RTOL = 1e-5


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def summary(t: torch.Tensor, nsamp: int = 256):
    a = t.detach().to("cpu", torch.float64).reshape(-1).numpy()
    if a.size <= 4096:
        samp = a.astype(np.float32)
    else:
        idx = (np.arange(nsamp, dtype=np.int64) * (a.size // nsamp) + (a.size // nsamp) // 2)
        samp = a[idx].astype(np.float32)
    return np.array([a.sum(), (a * a).sum()], dtype=np.float64), samp


def check_summary(gold, name: str, t: torch.Tensor, rtol: float = RTOL, what: str = "", atol: float = 0.0):
    """Compare tensor `t` against the (sum, sumsq, strided samples) the fixture holds for `name`.
    Sampled values must agree to 4*rtol relative to the tensor's RMS (norm-wise) plus atol, the sum
    of squares and the plain sum to the bounds those per-element errors imply."""
    stat, samp = summary(t)
    gstat, gsamp = gold[name + "#stat"], gold[name + "#samp"]
    n = t.numel()
    rms = float(np.sqrt(max(gstat[1], 1e-300) / n))
    err = float(np.max(np.abs(samp.astype(np.float64) - gsamp.astype(np.float64)))) if samp.size else 0.0
    assert err <= rtol * max(rms, 1e-30) * 4 + atol + 1e-12, f"{what}{name}: sample err {err:.3e} vs rms {rms:.3e}"
    tol2 = 4 * rtol * gstat[1] + 2 * atol * np.sqrt(n * gstat[1]) + n * atol * atol + 1e-20
    assert abs(stat[1] - gstat[1]) <= tol2, f"{what}{name}: sumsq {stat[1]} vs {gstat[1]}"
    scale = np.sqrt(n * max(gstat[1], 1e-300))
    assert abs(stat[0] - gstat[0]) <= rtol * scale + n * atol + 1e-12, f"{what}{name}: sum {stat[0]} vs {gstat[0]}"


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """Norm-wise relative error |a-b| / |b| in fp64."""
    a = a.detach().to("cpu", torch.float64)
    b = b.detach().to("cpu", torch.float64)
    den = float(b.norm())
    num = float((a - b).norm())
    return num / den if den > 0 else num


def max_rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / rms(b): elementwise error measured against the tensor's own scale."""
    a = a.detach().to("cpu", torch.float64)
    b = b.detach().to("cpu", torch.float64)
    rms = float(b.pow(2).mean().sqrt())
    return float((a - b).abs().max()) / (rms if rms > 0 else 1.0)


def poison_gpu_cache(device=None, mbytes: int = 768) -> None:
    """Fill the blocks torch's caching allocator will hand out next with NaN: a large block and a run of small ones are
    written and freed, so later torch.empty tensors start as NaN instead of as whatever (often zeros) the driver or an
    earlier test left there.  A kernel that reads memory nothing wrote then shows up as NaN, not as a rare mismatch."""
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    big = torch.full((mbytes << 18,), float("nan"), device=dev)
    small = [torch.full((1 << 16,), float("nan"), device=dev) for _ in range(512)]        # 256 KB each: the small pools
    tiny = [torch.full((256,), float("nan"), device=dev) for _ in range(2048)]
    torch.cuda.synchronize(dev)
    del big, small, tiny
