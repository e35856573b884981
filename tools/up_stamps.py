"""Where an UP launch spends its time, from inside: a debugging build of csrc/conv_x3.hip (-DPGAN_SPLIT_F16 -DPGAN_UP_TIMING,
linked as tools/_pk/libpgan_upt.so) has thread 0 of the first 64 blocks note the 100 MHz wall clock at six points -- kernel
entry, scale agreed, first K step staged, K loop done, group sums exchanged, stores issued.  Prints, per launch shape and
kernel variant, the median over blocks of each interval in microseconds, the spread of the blocks' entry times, and the
launch's duration by events (which adds the launch itself and the drain after the last store).
LIB names another build in tools/_pk (e.g. one with -DPGAN_ABL_UP=n: parts of the K loop left out); HS, NS, VARIANTS choose the cases."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
VP, I, LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_pk", os.environ.get("LIB", "libpgan_upt.so")))
lib.pgan_conv_h2_prep_bytes.restype = LL
lib.pgan_conv_h2_prep_bytes.argtypes = [I, I]
lib.pgan_conv_h2_prep.argtypes = [VP, VP, I, I, VP]
lib.pgan_conv_up_h2.argtypes = [VP, VP, VP, I, I, I, I, VP, I, I, VP]
lib.pgan_h2_set_option.argtypes = [ctypes.c_char_p, I]
lib.pgan_h2_debug_up_stamps.argtypes = [VP]
g = torch.Generator().manual_seed(2)
st = torch.cuda.Stream()
names = ["entry->scale", "scale->staged", "K loop", "exchange", "stores issued"]

for N in [int(v) for v in os.environ.get("NS", "8,64,320").split(",")]:
    for Cs, Cl, hs in [s for s in ((512, 256, 4), (256, 128, 8), (128, 64, 16)) if str(s[2]) in os.environ.get("HS", "4,8,16").split(",")]:
        y = torch.randn(N, Cs, hs, hs, generator=g).to(dev)
        w = (torch.randn(Cs, Cl, 4, 4, generator=g) * 0.05).to(dev)
        bnd = y.abs().reshape(N, 8, -1).amax(2).contiguous()
        planes = torch.zeros(lib.pgan_conv_h2_prep_bytes(Cs, Cl), dtype=torch.uint8, device=dev)
        x = torch.empty((N, Cl, 2 * hs, 2 * hs), device=dev)
        with torch.cuda.stream(st):
            assert lib.pgan_conv_h2_prep(w.data_ptr(), planes.data_ptr(), Cs, Cl, st.cuda_stream) == 0
        for var in [int(v) for v in os.environ.get("VARIANTS", "0,5").split(",")]:
            assert lib.pgan_h2_set_option(b"up_variant", var) == 0
            call = lambda: lib.pgan_conv_up_h2(y.data_ptr(), planes.data_ptr(), x.data_ptr(), N, Cs, Cl, hs, bnd.data_ptr(), 8,
                                               bnd.numel(), st.cuda_stream)
            with torch.cuda.stream(st):
                for _ in range(20):
                    assert call() == 0
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(100):
                    call()
                e.record()
                e.synchronize()
            us = s.elapsed_time(e) * 10
            out = np.zeros((64, 8), dtype=np.uint64)
            assert lib.pgan_h2_debug_up_stamps(out.ctypes.data) == 0
            t = out[:, :6].astype(np.int64)
            t = t[t[:, 0] > 0]
            d = np.diff(t, axis=1) / 100.0          # 100 MHz -> us
            line = f"UP N={N:3d} Cs={Cs} hs={hs:2d} variant {var}: {us:6.2f} us/launch by events; blocks noted {len(t)};"
            line += f" entries spread over {(t[:, 0].max() - t[:, 0].min()) / 100.0:5.2f} us; inside:"
            for nm, v in zip(names, np.median(d, axis=0)):
                line += f"  {nm} {v:5.2f}"
            line += f"  | entry->last stamp {np.median(t[:, 5] - t[:, 0]) / 100.0:5.2f}"
            print(line, flush=True)
lib.pgan_h2_set_option(b"up_variant", 0)
