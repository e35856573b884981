"""Fourth neighbour matrix for the packed-fp32 fault: which part of the fp16-split DOWN kernel disturbs the packed kernel?
Neighbours are standalone builds of csrc/conv_x3.hip with parts of the DOWN kernel's main loop left out
(-DPGAN_ABL_DOWN=mask: 1 no MFMAs, 2 no weight-fragment loads, 4 no staging stores (and with them no activation loads),
16 no LDS reads), all WITHOUT packed-fp32 instructions; the kernel under test is the packed build's single-channel DOWN.
  for a in 0 1 2 4 16 6; do hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -Xclang -target-feature -Xclang
      -packed-fp32-ops -DPGAN_NO_PACKED_FP32=1 -DPGAN_SPLIT_F16 -DPGAN_ABL_DOWN=$a -o tools/_pk/libconv_h2_abl$a.so
      pg-alignments-gan_amd/csrc/conv_x3.hip; done"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
VP, I, LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
pk = ctypes.CDLL(os.path.join(ROOT, "tools", "_pk", "libpgan_hip_pk.so"))
pk.pgan_conv_down.argtypes = [VP, VP, VP, I, I, I, I, VP]
g = torch.Generator(device="cpu").manual_seed(5)
N, REPS = 64, int(os.environ.get("REPS", "60"))


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev)


x, w1 = rnd(N, 1, 64, 64), rnd(64, 1, 4, 4, scale=0.25)
want = torch.empty(N, 64, 32, 32, device=dev)
outs = [torch.empty_like(want) for _ in range(8)]
main = torch.cuda.current_stream().cuda_stream


def victim(out):
    rc = pk.pgan_conv_down(x.data_ptr(), w1.data_ptr(), out.data_ptr(), N, 64, 1, 32, main)
    assert rc == 0, rc


victim(want)
torch.cuda.synchronize()
side = torch.cuda.Stream()
a2, wt2, y2 = rnd(128, 64, 32, 32), rnd(128, 64, 4, 4, scale=0.03), torch.empty(128, 128, 16, 16, device=dev)
names = {0: "whole kernel", 1: "no MFMAs", 2: "no weight-fragment loads", 4: "no staging stores, no activation loads",
         16: "no LDS reads", 6: "no weight loads, no staging, no activation loads"}
masks = [int(m) for m in os.environ.get("MASKS", "0,1,2,4,16,6").split(",")]
for mask in masks:
    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_pk", f"libconv_h2_abl{mask}.so"))
    lib.pgan_conv_h2_prep_bytes.restype = LL
    lib.pgan_conv_h2_prep_bytes.argtypes = [I, I]
    lib.pgan_conv_h2_prep.argtypes = [VP, VP, I, I, VP]
    lib.pgan_conv_down_h2.argtypes = [VP, VP, VP, I, I, I, I, VP, I, I, VP]
    planes = torch.zeros(lib.pgan_conv_h2_prep_bytes(128, 64), dtype=torch.uint8, device=dev)
    assert lib.pgan_conv_h2_prep(wt2.data_ptr(), planes.data_ptr(), 128, 64, side.cuda_stream) == 0
    torch.cuda.synchronize()
    bad_launches = bad_values = 0
    t_nb = 0.0
    for rep in range(REPS):
        for _ in range(4):
            rc = lib.pgan_conv_down_h2(a2.data_ptr(), planes.data_ptr(), y2.data_ptr(), 128, 128, 64, 16, None, 0, 0, side.cuda_stream)
            assert rc == 0, rc
        for out in outs:
            victim(out)
        torch.cuda.synchronize()
        for out in outs:
            n = int((out != want).sum())
            bad_launches += n > 0
            bad_values += n
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record()
        for _ in range(20):
            lib.pgan_conv_down_h2(a2.data_ptr(), planes.data_ptr(), y2.data_ptr(), 128, 128, 64, 16, None, 0, 0, side.cuda_stream)
        e1.record()
    torch.cuda.synchronize()
    print(f"beside DOWN 64->128 hs 16, {names[mask]:48s} ({e0.elapsed_time(e1) / 20 * 1e3:6.1f} us per launch): "
          f"{bad_launches:4d} of {REPS * len(outs)} launches differ, {bad_values} values", flush=True)
