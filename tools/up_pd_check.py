"""Isolated time of the fp16-split UP (and DOWN) kernels from standalone builds of csrc/conv_x3.hip with other weight-fragment
prefetch depths (-DPGAN_UP_PD32=d -DPGAN_UP_PD64=d; libraries tools/_pk/libconv_h2_pd<d32><d64>.so), on the launch shapes
of an iteration; results compared with the default build's bit for bit (the depth only moves loads)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
VP, I, LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
g = torch.Generator().manual_seed(2)
tags = sys.argv[1:] or ["21", "31", "41", "42"]
libs = {}
for t in tags:
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_pk", f"libconv_h2_{os.environ.get('PREFIX', 'pd')}{t}.so"))
    lib.pgan_conv_h2_prep_bytes.restype = LL
    lib.pgan_conv_h2_prep_bytes.argtypes = [I, I]
    lib.pgan_conv_h2_prep.argtypes = [VP, VP, I, I, VP]
    lib.pgan_conv_up_h2.argtypes = [VP, VP, VP, I, I, I, I, VP, I, I, VP]
    lib.pgan_conv_down_h2.argtypes = [VP, VP, VP, I, I, I, I, VP, I, I, VP]
    libs[t] = lib
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


for N in [int(v) for v in os.environ.get("NS", "64,128,320").split(",")]:
    for Cs, Cl, hs in ((512, 256, 4), (256, 128, 8), (128, 64, 16)):
        y = torch.randn(N, Cs, hs, hs, generator=g).to(dev)
        w = (torch.randn(Cs, Cl, 4, 4, generator=g) * 0.05).to(dev)
        bnd = y.abs().reshape(N, 8, -1).amax(2).contiguous()
        KIND = os.environ.get("KIND", "up")
        xin = torch.randn(N, Cl, 2 * hs, 2 * hs, generator=g).to(dev)
        bndx = xin.abs().reshape(N, 8, -1).amax(2).contiguous()
        ref, line = None, f"{KIND.upper()} N={N:3d} Cs={Cs} hs={hs:2d}:"
        for t, lib in libs.items():
            planes = torch.zeros(lib.pgan_conv_h2_prep_bytes(Cs, Cl), dtype=torch.uint8, device=dev)
            x = torch.full((N, Cl, 2 * hs, 2 * hs), float("nan"), device=dev)
            with torch.cuda.stream(st):
                assert lib.pgan_conv_h2_prep(w.data_ptr(), planes.data_ptr(), Cs, Cl, st.cuda_stream) == 0
            if KIND == "up":
                call = lambda: lib.pgan_conv_up_h2(y.data_ptr(), planes.data_ptr(), x.data_ptr(), N, Cs, Cl, hs, bnd.data_ptr(), 8,
                                                   bnd.numel(), st.cuda_stream)
            else:
                x = torch.full((N, Cs, hs, hs), float("nan"), device=dev)
                call = lambda: lib.pgan_conv_down_h2(xin.data_ptr(), planes.data_ptr(), x.data_ptr(), N, Cs, Cl, hs, bndx.data_ptr(),
                                                     8, bndx.numel(), st.cuda_stream)
            timed(call, 50)
            us = timed(call)
            torch.cuda.synchronize()
            same = "" if ref is None else (" =" if torch.equal(x, ref) else " DIFFERS")
            ref = x if ref is None else ref
            line += f"  pd{t} {us:6.2f} us{same}"
        print(line, flush=True)
