"""Tile / K-split sweep of the split-operand DOWN, UP and WGRAD kernels per launch shape (fp16 two-plane split).
usage: python tools/splitsweep.py [N ...]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pg-alignments-gan_amd"))
import torch
import pgan_hip
from pgan_hip import HipOps

dev = torch.device("cuda:0")
ops = HipOps(dev, precision=os.environ.get("SB_PREC", "h2"))
ops.static_weights = True
LAYERS = [(128, 64, 16), (256, 128, 8), (512, 256, 4)]


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


def sweep(label, fn, fl, combos):
    out = []
    for opts in combos:
        try:
            for k, v in opts.items():
                pgan_hip.set_x3_option(k, v)
            t = timeit(fn)
            out.append(f"{','.join(f'{k}={v}' for k, v in opts.items()) or 'auto'}: {t:6.1f} us {fl / t / 1e6:5.0f} TF")
        except Exception as ex:                      # a tile the shape does not divide into
            out.append(f"{opts}: {type(ex).__name__}")
        finally:
            for k in opts:
                pgan_hip.set_x3_option(k, 0)
    print(f"  {label}: " + " | ".join(out), flush=True)


for N in [int(v) for v in sys.argv[1:]] or [64, 128, 256]:
    for Cs, Cl, hs in LAYERS:
        fl = 2.0 * N * Cs * Cl * 16 * hs * hs
        g = torch.Generator().manual_seed(N + hs)
        y = torch.randn(N, Cs, hs, hs, generator=g).to(dev)
        x = torch.randn(N, Cl, 2 * hs, 2 * hs, generator=g).to(dev)
        w = (torch.randn(Cs, Cl, 4, 4, generator=g) * 0.05).to(dev)
        yo, xo, dw = torch.empty_like(y), torch.empty_like(x), torch.empty_like(w)
        print(f"N={N} Cs={Cs} Cl={Cl} hs={hs}")
        sweep("down", lambda: ops.conv_down(x, w, yo), fl,
              [{}] + [dict(down_order=o) for o in (1, 2)] + [dict(down_variant=v, down_ks=k) for v in (2, 4) for k in (1, 2, 4)])
        sweep("up", lambda: ops.conv_up(y, w, xo), fl,
              [{}] + [dict(up_variant=v, up_ks=k) for v in (1, 2, 3, 5) for k in (1, 2)] +
              [dict(up_variant=5, up_order=o) for o in (1, 2)])
        sweep("wgrad", lambda: ops.conv_wgrad(y, x, dw), fl, [{}] + [dict(wg_target=t) for t in (128, 256, 1024, 2048)])
