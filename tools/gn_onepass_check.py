"""One-launch GroupNorm ops against the sums / apply pair: bit equality of every output and isolated time per call, on the
four stage shapes at the batch sizes of an iteration's chains (64 and 128 samples), back-to-back launches from a hipGraph."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("pg-alignments-gan_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch

from pgan_hip import HipOps, StepGraph

dev = torch.device("cuda:0")
o = HipOps(dev)
g = torch.Generator(device="cpu").manual_seed(3)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
REPS = 50


def timed(fn):
    cap = torch.cuda.Stream()
    o.reserve_stream(cap.cuda_stream)
    with torch.cuda.stream(cap):
        for _ in range(3):
            fn()
        cap.synchronize()
        gr = StepGraph()
        o.capturing = True
        try:
            with gr:
                for _ in range(REPS):
                    fn()
        finally:
            o.capturing = False
        gr.launch()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        gr.launch()
        e.record()
        e.synchronize()
    return s.elapsed_time(e) / REPS * 1e3


ops = sys.argv[1:] or ["bwd"]
for N in (64, 128):
    for C, H in ((64, 32), (128, 16), (256, 8), (512, 4)):
        a, u = rnd(N, C, H, H), rnd(N, C, H, H)
        gamma, beta = 1 + 0.1 * rnd(C), 0.1 * rnd(C)
        h, stats = torch.empty_like(a), torch.empty(N, 2, device=dev)
        o.gn_fwd(a, gamma, beta, h, stats)
        res = {}
        for mode in (False, True):
            HipOps.gn_onepass = 2 if mode else 0
            out = dict(d=torch.full_like(a, float("nan")), sb=torch.full((N, 2), float("nan"), device=dev),
                       pg=torch.full((N, C), float("nan"), device=dev), pb=torch.full((N, C), float("nan"), device=dev),
                       am=torch.full((N, 8), float("nan"), device=dev))
            call = lambda: o.gn_bwd(u, None, a, stats, gamma, out["d"], out["sb"], out["pg"], out["pb"], amax=out["am"], beta=beta)
            call()
            torch.cuda.synchronize()
            res[mode] = ({k: v.clone() for k, v in out.items()}, timed(call))
        same = all(torch.equal(res[False][0][k], res[True][0][k]) for k in res[False][0])
        if not same:
            for k in res[False][0]:
                x, y = res[False][0][k], res[True][0][k]
                if not torch.equal(x, y):
                    bad = (x != y) & ~(torch.isnan(x) & torch.isnan(y))
                    idx = bad.nonzero()
                    print(f"    {k}: {int(bad.sum())} of {x.numel()} differ, max |diff| {float((x - y).abs().nan_to_num(nan=9e9).max()):.3e}, "
                          f"first at {idx[0].tolist()}, last at {idx[-1].tolist()}, NaN in one-launch {int(torch.isnan(y).sum())}")
        print(f"gn_bwd [{N}, {C}, {H}, {H}]: two launches {res[False][1]:7.2f} us, one launch {res[True][1]:7.2f} us, "
              f"outputs {'bit-identical' if same else 'DIFFER'}", flush=True)
print("a one-launch call gave up waiting:", o.gn_onepass_error())
