"""Eager / graph / tape routes of the trainer after poisoning the allocator cache; OPTS="down_ks=2"."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("pg-alignments-gan_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch

import dcwgan_oracle as orc
import pgan_hip
from helpers import poison_gpu_cache
from pgan_hip import HipOps
from pgan_trainer import FusedTrainer

dev = torch.device("cuda:0")
for item in filter(None, os.environ.get("OPTS", "").split(",")):
    k, v = item.split("=")
    pgan_hip.set_x3_option(k, int(v))
loss = os.environ.get("LOSS", "gp")
hip = HipOps(dev, precision=os.environ.get("PREC", "h2"))
rs, rp = orc.synthetic_alignments(64, 9)
for trial in range(int(os.environ.get("TRIALS", "3"))):
    poison_gpu_cache()
    out = {}
    for route in os.environ.get("ROUTES", "eager,graph,tape").split(","):
        tr = FusedTrainer(hip, 8, dev, loss=loss, critic_iter=2, seed=77)
        tr.eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
        tr.eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
        tr.weights_changed()
        tr.set_pool(rs, rp)
        tr.use_graph, tr.use_tape = route == "graph", route == "tape"
        ls = []
        for _ in range(3):
            tr.step()
            ls.append(tr.losses()[0])
        torch.cuda.synchronize()
        out[route] = (tr.eng.D.flat.clone(), tr.eng.G.flat.clone(), ls)
        del tr
    a = next(iter(out.values()))
    for r, b in out.items():
        print(trial, r, "D equal", torch.equal(a[0], b[0]), "G equal", torch.equal(a[1], b[1]), b[2], flush=True)
