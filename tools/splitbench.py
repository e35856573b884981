"""Accuracy (against fp64 on the CPU) and speed of the split-operand convolution kernels per launch shape, for the
bf16 three-plane split ("x3") and the fp16 two-plane split ("h2").  usage: python tools/splitbench.py [N ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pg-alignments-gan_amd"))
import torch
import torch.nn.functional as F
from pgan_hip import HipOps

dev = torch.device("cuda:0")
ops = HipOps(dev)
ops.static_weights = True
LAYERS = [(128, 64, 16), (256, 128, 8), (512, 256, 4)]
scale = float(os.environ.get("SB_SCALE", "1"))


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for N in [int(v) for v in sys.argv[1:]] or [64, 128, 256]:
    for Cs, Cl, hs in LAYERS:
        fl = 2.0 * N * Cs * Cl * 16 * hs * hs
        g = torch.Generator().manual_seed(N + hs)
        y = (torch.randn(N, Cs, hs, hs, generator=g) * scale).to(dev)
        x = (torch.randn(N, Cl, 2 * hs, 2 * hs, generator=g) * scale).to(dev)
        w = (torch.randn(Cs, Cl, 4, 4, generator=g) * 0.05).to(dev)
        refd = F.conv2d(x.double().cpu(), w.double().cpu(), stride=2, padding=1) if N <= 64 else None
        refu = F.conv_transpose2d(y.double().cpu(), w.double().cpu(), stride=2, padding=1) if N <= 64 else None
        for prec in ("x3", "h2"):
            ops.precision = prec
            ops.invalidate_weight_planes()
            yo, xo, dw = torch.empty_like(y), torch.empty_like(x), torch.empty_like(w)
            ops.conv_down(x, w, yo)
            ops.conv_up(y, w, xo)
            ops.conv_wgrad(y, x, dw)
            err = ""
            if refd is not None:
                refw = torch.autograd.grad(F.conv2d(x.double().cpu(), (wd := w.double().cpu().requires_grad_()), stride=2, padding=1),
                                           wd, y.double().cpu())[0]
                e = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
                err = f" relerr down {e(yo, refd):.2e} up {e(xo, refu):.2e} wgrad {e(dw, refw):.2e}"
            td, tu, tw = timeit(lambda: ops.conv_down(x, w, yo)), timeit(lambda: ops.conv_up(y, w, xo)), timeit(lambda: ops.conv_wgrad(y, x, dw))
            print(f"N={N:3d} hs={hs:2d} {prec}: down {td:7.1f} us {fl / td / 1e6:6.1f} TF | up {tu:7.1f} us {fl / tu / 1e6:6.1f} TF | "
                  f"wgrad {tw:7.1f} us {fl / tw / 1e6:6.1f} TF{err}", flush=True)
