"""Does every kernel of the critic step see its producer's finished output?  Before each run the slots the step
itself fills (interpolates, tangent inputs, their activations) are poisoned with NaN; afterwards stage 1 of the penalty
chain is recomputed from the final inputs with the same kernels and compared bit for bit."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("pg-alignments-gan_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch

import dcwgan_oracle as orc
import pgan_tape
from pgan_engine import Engine
from pgan_hip import HipOps

dev = torch.device("cuda:0")
B = int(os.environ.get("B", "8"))
MODE = os.environ.get("REPLAY", "tape")
eng = Engine(HipOps(dev, precision=os.environ.get("PREC", "h2")), B, dev, loss=os.environ.get("LOSS", "gp"))
o = eng.ops
eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
rs, rp = orc.synthetic_alignments(2 * B, 60)
g = np.random.Generator(np.random.PCG64(62))
eng.X0[:2 * B].copy_(rs.to(dev))
eng.P0[:2 * B].copy_(rp.to(dev))
eng.tbuf.copy_(torch.from_numpy(g.random(B)).float().to(dev))
NS, NT = eng.NS, eng.NT
eng.critic_grads_resident(run_generator=False)
torch.cuda.synchronize()
pgan_tape.tape = entries = []
eng.critic_grads_resident(run_generator=False)
pgan_tape.tape = None
torch.cuda.synchronize()
tmp = torch.empty_like(eng.dA[1][:B])
want_x = torch.empty_like(eng.X0[:B])
bad = 0
for rep in range(int(os.environ.get("REPS", "200"))):
    if os.environ.get("POISON", "1") == "1":
        eng.X0[2 * B:].fill_(float("nan"))
        eng.G0.fill_(float("nan"))
        for k in range(1, eng.nS + 1):
            eng.dA[k][2 * B:].fill_(float("nan"))
            eng.dH[k][2 * B:].fill_(float("nan"))
    torch.cuda.synchronize()
    if MODE == "tape":
        pgan_tape.replay(entries)
    else:
        eng.critic_grads_resident(run_generator=False)
    torch.cuda.synchronize()
    msgs = []
    o.interpolate(eng.tbuf, eng.X0[:B], eng.X0[B:2 * B], want_x)
    torch.cuda.synchronize()
    if not torch.equal(want_x, eng.X0[2 * B:NS]):
        msgs.append(f"interpolates differ from a recomputation ({int((want_x != eng.X0[2 * B:NS]).sum())} elements)")
    for name, lo, hi in (("penalised", 2 * B, NS), ("tangent", NS, NT)):
        o.conv_down(eng.X0[lo:hi], eng.dconv_w(1), tmp)
        torch.cuda.synchronize()
        got = eng.dA[1][lo:hi]
        if not torch.equal(torch.nan_to_num(tmp, nan=-7.0), torch.nan_to_num(got, nan=-7.0)):
            d = torch.nan_to_num(tmp, nan=-7.0) != torch.nan_to_num(got, nan=-7.0)
            slots = torch.nonzero(d.reshape(d.shape[0], -1).any(1)).flatten().tolist()
            nz = torch.nonzero(d)
            truth = torch.nn.functional.conv2d(eng.X0[lo:hi].double(), eng.dconv_w(1).double(), stride=2, padding=1)
            e_got = float((got.double() - truth).abs().max())
            e_tmp = float((tmp.double() - truth).abs().max())
            ex = []
            for n_, c_, h_, w_ in nz[:: max(1, nz.shape[0] // 12)].tolist():
                ex.append(f"({n_},{c_},{h_},{w_}) got {float(got[n_, c_, h_, w_]):.6f} again {float(tmp[n_, c_, h_, w_]):.6f} "
                          f"truth {float(truth[n_, c_, h_, w_]):.6f}")
            hist_h = torch.bincount(nz[:, 2], minlength=32).tolist()
            hist_w = torch.bincount(nz[:, 3], minlength=32).tolist()
            hist_c = torch.bincount(nz[:, 1], minlength=64).tolist()
            msgs.append(f"max error vs fp64: in-step {e_got:.2e}, recomputed {e_tmp:.2e}; by h {hist_h}; by w {hist_w}; by c {hist_c}")
            msgs.extend(ex)
            # what the wrong values are made of: the fp64 contribution of every kernel row kh to the element, for its own
            # channel and for the channels next to it (a stale weight row would show as one row taken from a neighbour)
            wd = eng.dconv_w(1).double()
            xin = torch.nn.functional.pad(eng.X0[lo:hi].double(), (1, 1, 1, 1))
            for n_, c_, h_, w_ in nz[:: max(1, nz.shape[0] // 6)].tolist():
                terms = [[float(xin[n_, 0, 2 * h_ + kh, 2 * w_ + kw] * wd[c_, 0, kh, kw]) for kw in range(4)] for kh in range(4)]
                delta = float(got[n_, c_, h_, w_]) - float(truth[n_, c_, h_, w_])
                msgs.append(f"({n_},{c_},{h_},{w_}) got - truth {delta:+.6f}; the 16 products x[kh][kw] * w[kh][kw]: "
                            + " | ".join(" ".join("%+.5f" % v for v in row) for row in terms))
    nan_any = {n: int(torch.isnan(t).sum()) for n, t in (("X0", eng.X0), ("G0", eng.G0), ("Dg", eng.D.gflat), ("closs", eng.closs))}
    if any(nan_any.values()):
        msgs.append(f"NaN left: {nan_any}")
    if msgs:
        bad += 1
        print(f"rep {rep}:", flush=True)
        for m in msgs:
            print("   " + m, flush=True)
        if bad >= 2:
            break
print("bad repetitions:", bad)
