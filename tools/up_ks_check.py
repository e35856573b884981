"""KIND=down: the same for DOWN over ("down_variant" 1-4 = 128x128, 128x64, 64x128, 64x64 tiles; "down_ks" 1, 2, 4 blocks x groups along K, 3 = two wave groups in one block), the
clear of a K-split launch's output included.
Isolated time of the fp16-split UP kernel on the launch shapes of an iteration per (tuning option "up_variant", "up_ks"):
variants 5 / 1 are 32- / 64-pixel tiles, 6 / 7 the same with two wave groups per block along K; up_ks > 1 splits K over
blocks that add into an output the launch clears first (measured 1.6-3x slower at every shape, no longer in the list).
Through the shipped library; prints microseconds per call and the largest difference from the first result."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
VP, I, LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
lib = ctypes.CDLL(os.environ.get("PGAN_HIP_LIBRARY", os.path.join(ROOT, "pg-alignments-gan_amd", "libpgan_hip.so")))
lib.pgan_conv_h2_prep_bytes.restype = LL
lib.pgan_conv_h2_prep_bytes.argtypes = [I, I]
lib.pgan_conv_h2_prep.argtypes = [VP, VP, I, I, VP]
lib.pgan_conv_up_h2.argtypes = [VP, VP, VP, I, I, I, I, VP, I, I, VP]
lib.pgan_conv_down_h2.argtypes = [VP, VP, VP, I, I, I, I, VP, I, I, VP]
KIND = os.environ.get("KIND", "up")
lib.pgan_h2_set_option.argtypes = [ctypes.c_char_p, I]
g = torch.Generator().manual_seed(2)
st = torch.cuda.Stream()


def timed(fn, reps=200):
    with torch.cuda.stream(st):
        for _ in range(10):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        e.synchronize()
    return s.elapsed_time(e) / reps * 1e3


combos = [(0, 0), (5, 0), (1, 0), (6, 0), (7, 0), (0, 0)]
if KIND == "down":
    combos = [(0, 0), (4, 1), (4, 3), (4, 4), (2, 1), (2, 3), (2, 4), (0, 0)]
for N in [int(v) for v in os.environ.get("NS", "64,128").split(",")]:
    for Cs, Cl, hs in ((512, 256, 4), (256, 128, 8), (128, 64, 16)):
        y = torch.randn(N, Cs, hs, hs, generator=g).to(dev)
        w = (torch.randn(Cs, Cl, 4, 4, generator=g) * 0.05).to(dev)
        bnd = y.abs().reshape(N, 8, -1).amax(2).contiguous()
        planes = torch.zeros(lib.pgan_conv_h2_prep_bytes(Cs, Cl), dtype=torch.uint8, device=dev)
        with torch.cuda.stream(st):
            assert lib.pgan_conv_h2_prep(w.data_ptr(), planes.data_ptr(), Cs, Cl, st.cuda_stream) == 0
        ref, line = None, f"{KIND.upper()} N={N:3d} Cs={Cs} hs={hs:2d}:"
        xin = torch.randn(N, Cl, 2 * hs, 2 * hs, generator=g).to(dev)
        bndx = xin.abs().reshape(N, 8, -1).amax(2).contiguous()
        for var, ks in combos:
            assert lib.pgan_h2_set_option(f"{KIND}_variant".encode(), var) == 0 and lib.pgan_h2_set_option(f"{KIND}_ks".encode(), ks) == 0
            if KIND == "down":
                x = torch.full((N, Cs, hs, hs), float("nan"), device=dev)
                call = lambda: lib.pgan_conv_down_h2(xin.data_ptr(), planes.data_ptr(), x.data_ptr(), N, Cs, Cl, hs, bndx.data_ptr(),
                                                     8, bndx.numel(), st.cuda_stream)
            else:
                x = torch.full((N, Cl, 2 * hs, 2 * hs), float("nan"), device=dev)
                call = lambda: lib.pgan_conv_up_h2(y.data_ptr(), planes.data_ptr(), x.data_ptr(), N, Cs, Cl, hs, bnd.data_ptr(), 8,
                                                   bnd.numel(), st.cuda_stream)
            timed(call, 50)
            us = timed(call)
            torch.cuda.synchronize()
            d = "" if ref is None else f" (d {float((x - ref).abs().max()):.1e})"
            ref = x if ref is None else ref
            line += f"  v{var}k{ks} {us:6.2f}{d}"
        print(line, flush=True)
lib.pgan_h2_set_option(f"{KIND}_variant".encode(), 0)
lib.pgan_h2_set_option(f"{KIND}_ks".encode(), 0)
