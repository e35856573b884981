"""Which kernel classes bound the iteration: bench config 1 (tape route) with one class of launches left out at a time
(results are then meaningless -- timing experiment only).  usage (GPU box, repo root): python tools/skipclass.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pg-alignments-gan_amd")]
import numpy as np
import torch
import pgan_hip
from pgan_hip import HipOps
from pgan_trainer import FusedTrainer
from pgan_engine import disc_param_shapes, gen_param_shapes
from bench import synthetic_pool

CLASSES = {
    "none": (),
    "groupnorm (all four ops)": ("pgan_gn_fwd", "pgan_gn_fwd_presummed", "pgan_gn_bwd", "pgan_gn_jvp", "pgan_gn_bwd2"),
    "groupnorm, stages of <= 8x8 only": "gn_small",
    "batchnorm": ("pgan_bn_fwd", "pgan_bn_bwd"),
    "1-D towers": ("pgan_d1d_fwd_len", "pgan_d1d_bwd_len", "pgan_ct1d_up", "pgan_ct1d_down", "pgan_ct1d_wgrad"),
    "single-channel convolutions": ("pgan_conv_down", "pgan_conv_up", "pgan_conv_wgrad", "pgan_conv_down_gn_sums"),
    "adam + weight preparation": ("pgan_adam", "pgan_conv_h2_prep_multi", "pgan_conv_h2_prep"),
    "row reductions, head, penalty, loss": ("pgan_rows_reduce", "pgan_rows_reduce_multi", "pgan_head_fwd", "pgan_head_seed",
                                            "pgan_head_wgrad", "pgan_gp_penalty", "pgan_critic_loss", "pgan_interpolate"),
    "inputs (batch, draws, copies)": ("pgan_take_batch_slice", "pgan_philox_slice", "pgan_affine"),
    "DOWN (matrix pipe)": ("pgan_conv_down_h2",), "UP (matrix pipe)": ("pgan_conv_up_h2",),
    "WGRAD (matrix pipe)": ("pgan_conv_wgrad_h2",),
}
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)


class Skip:
    def __init__(self, fn, pred=None):
        self.fn, self.pred = fn, pred

    def __call__(self, *a):
        if self.pred is None or self.pred(a):
            return 0
        return self.fn(*a)


def run(skip):
    lib = pgan_hip._lib
    saved = {}
    names = skip if isinstance(skip, tuple) else ("pgan_gn_fwd", "pgan_gn_bwd", "pgan_gn_jvp", "pgan_gn_bwd2")
    for n in names:
        saved[n] = getattr(lib, n)
        # GroupNorm entry points end (..., N, C, HW, amax, stream): HW <= 64 are the 8x8 and 4x4 stages
        setattr(lib, n, Skip(saved[n], (lambda a: a[-3] <= 64) if skip == "gn_small" else None))
    try:
        tr = FusedTrainer(HipOps(dev), 64, dev, seed=1234)
        gen = torch.Generator().manual_seed(20260)
        for store, shapes in ((tr.eng.G, gen_param_shapes()), (tr.eng.D, disc_param_shapes())):
            for k, sh in shapes.items():
                if len(sh) == 1:
                    v = torch.ones(sh) if k.endswith("weight") and not k.startswith("out") else torch.zeros(sh)
                else:
                    v = (torch.rand(sh, generator=gen) * 2 - 1) / int(np.prod(sh[1:])) ** 0.5
                store.p[k].copy_(v)
        tr.weights_changed()
        tr.set_pool(*synthetic_pool(4096, 20261))
        tr.record_tape()
        for _ in range(10):
            tr.replay_tape()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            tr.replay_tape()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / 40, len(tr.tape.entries)
    finally:
        for n, f in saved.items():
            setattr(lib, n, f)


base = None
print("class left out | ms per iteration | saved | calls left")
for name, skip in CLASSES.items():
    ms, calls = run(skip)
    base = ms if base is None else base
    print(f"{name} | {ms:.3f} | {base - ms:+.3f} | {calls}")
ms, _ = run(())
print(f"none (again) | {ms:.3f} | {base - ms:+.3f} |")
