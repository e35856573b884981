"""Micro-benchmark of the conv kernel variants (tuning aid; run on the GPU box)."""
import sys, os, itertools
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pg-alignments-gan_amd"))
import torch
from pgan_hip import HipOps, set_option
dev = torch.device("cuda:0")
ops = HipOps(dev)
LAYERS = [(128, 64, 16), (256, 128, 8), (512, 256, 4)]
def timeit(fn, reps=30):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
which = sys.argv[1] if len(sys.argv) > 1 else "all"
for N in tuple(int(v) for v in os.environ.get('CB_N', '64,192').split(',')):
    for Cs, Cl, hs in LAYERS:
        fl = 2.0 * N * Cs * Cl * 16 * hs * hs
        y = torch.randn(N, Cs, hs, hs, device=dev); x = torch.randn(N, Cl, 2*hs, 2*hs, device=dev)
        w = torch.randn(Cs, Cl, 4, 4, device=dev) * 0.05; dw = torch.zeros_like(w)
        if which in ("all", "up"):
            for mode in (1, 2, 3):
                set_option("up_mode", mode)
                us = timeit(lambda: ops.conv_up(y, w, x))
                print(f"up   N={N:3d} hs={hs:2d} mode={mode}: {us:8.1f} us {fl/us/1e6:7.1f} TF")
            set_option("up_mode", 0)
        if which in ("all", "down"):
            for ks in ():
                for kg in (1, 2, 4):
                    set_option("down_kg", kg); set_option("down_ks", ks)
                    us = timeit(lambda: ops.conv_down(x, w, y))
                    print(f"down N={N:3d} hs={hs:2d} ks={ks} kg={kg}:   {us:8.1f} us {fl/us/1e6:7.1f} TF")
            set_option("down_kg", 0); set_option("down_ks", 0)
            us = timeit(lambda: ops.conv_down(x, w, y))
            print(f"down N={N:3d} hs={hs:2d} auto:        {us:8.1f} us {fl/us/1e6:7.1f} TF")
            for var in (1, 2, 3, 4, 5, 6, 7):
                set_option("down_variant", var)
                yy = torch.zeros_like(y); ops.conv_down(x, w, yy)
                set_option("down_variant", 0); ops.conv_down(x, w, y)
                err = float((yy - y).abs().max())
                set_option("down_variant", var)
                us = timeit(lambda: ops.conv_down(x, w, y))
                print(f"down N={N:3d} hs={hs:2d} variant={var}:   {us:8.1f} us {fl/us/1e6:7.1f} TF  maxdiff {err:.1e}")
            set_option("down_variant", 0)
        if which in ("x3",):
            from pgan_hip import set_x3_option
            refd = torch.nn.functional.conv2d(x.double(), w.double(), stride=2, padding=1)
            refu = torch.nn.functional.conv_transpose2d(y.double(), w.double(), stride=2, padding=1)
            ops.precision = "fp32"
            ops.conv_down(x, w, y2 := torch.empty_like(y)); e = float((y2.double() - refd).norm() / refd.norm())
            us = timeit(lambda: ops.conv_down(x, w, y2))
            print(f"down N={N:3d} hs={hs:2d} fp32 mfma:       {us:8.1f} us {fl/us/1e6:7.1f} TF  relerr {e:.2e}")
            ops.conv_up(y, w, x2 := torch.empty_like(x)); e = float((x2.double() - refu).norm() / refu.norm())
            us = timeit(lambda: ops.conv_up(y, w, x2))
            print(f"up   N={N:3d} hs={hs:2d} fp32 mfma:       {us:8.1f} us {fl/us/1e6:7.1f} TF  relerr {e:.2e}")
            ops.precision = "x3"; ops.static_weights = True
            us = timeit(lambda: (ops.invalidate_weight_planes(), ops._weight_planes(w)))
            print(f"prep Cs={Cs} Cl={Cl}: {us:8.1f} us")
            for var, ks in ((0, 0), (4, 1), (4, 2), (4, 4), (2, 1), (2, 2), (3, 1), (3, 2), (1, 1)):
                set_x3_option("down_variant", var); set_x3_option("down_ks", ks)
                yy = torch.full_like(y, float("nan"))
                try:
                    ops.conv_down(x, w, yy)
                except Exception as ex:
                    print(f"down N={N:3d} hs={hs:2d} x3 v={var} ks={ks}: {ex}"); continue
                err = float((yy.double() - refd).norm() / refd.norm())
                us = timeit(lambda: ops.conv_down(x, w, yy))
                print(f"down N={N:3d} hs={hs:2d} x3 v={var} ks={ks}:     {us:8.1f} us {fl/us/1e6:7.1f} TF  relerr {err:.2e}")
            set_x3_option("down_variant", 0); set_x3_option("down_ks", 0)
            for var, ks in ((0, 0), (1, 0), (5, 0), (2, 0), (3, 0)):
                set_x3_option("up_variant", var); set_x3_option("up_order", ks)
                xx = torch.full_like(x, float("nan"))
                ops.conv_up(y, w, xx)
                err = float((xx.double() - refu).norm() / refu.norm())
                us = timeit(lambda: ops.conv_up(y, w, xx))
                print(f"up   N={N:3d} hs={hs:2d} x3 v={var} ks={ks}:     {us:8.1f} us {fl/us/1e6:7.1f} TF  relerr {err:.2e}")
            set_x3_option("up_variant", 0); set_x3_option("up_order", 0)
        if which in ("wgt",):
            from pgan_hip import set_x3_option
            for tgt in (256, 384, 512, 768, 1024, 1536):
                set_x3_option("wg_target", tgt)
                try:
                    us = timeit(lambda: ops.conv_wgrad(y, x, dw))
                    print(f"wgrd N={N:3d} hs={hs:2d} target={tgt}: {us:8.1f} us {fl/us/1e6:7.1f} TF")
                except Exception as ex:
                    print(f"wgrd N={N:3d} hs={hs:2d} target={tgt}: {ex}")
            set_x3_option("wg_target", 0)
        if which in ("wg3",):
            ref = None
            for px in (False, True):
                ops.precision = "x3"; ops.wgrad_x3 = px
                ops.conv_wgrad(y, x, dw)
                if ref is None:
                    ref = torch.nn.grad.conv2d_weight(x.double(), w.shape, y.double(), stride=2, padding=1)
                err = float((dw.double() - ref).norm() / ref.norm())
                us = timeit(lambda: ops.conv_wgrad(y, x, dw))
                print(f"wgrd N={N:3d} hs={hs:2d} {'bf16 pipe' if px else 'fp32 pipe'}: {us:8.1f} us {fl/us/1e6:7.1f} TF  relerr {err:.2e}")
        if which in ("all", "wgrad"):
            us = timeit(lambda: ops.conv_wgrad(y, x, dw))
            print(f"wgrd N={N:3d} hs={hs:2d}:        {us:8.1f} us {fl/us/1e6:7.1f} TF")
