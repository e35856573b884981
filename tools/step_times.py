"""Where an iteration's time goes, unprofiled: timing events recorded on the caller's stream inside the replayed launch tape --
at the start of the iteration, after every critic step's Adam update, before and after the generator step -- averaged over
ITERS replays (batch 64, gp, 5 critic steps).  The events are Python steps between native runs of the tape.
OPTS="name=value ..." sets tuning options of the split-operand convolution kernels first."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("pg-alignments-gan_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch

import dcwgan_oracle as orc
import pgan_tape
from pgan_hip import HipOps, set_x3_option
from pgan_trainer import FusedTrainer

dev = torch.device("cuda:0")
for kv in os.environ.get("OPTS", "").split():
    set_x3_option(kv.split("=")[0], int(kv.split("=")[1]))
B, CI, ITERS = int(os.environ.get("B", "64")), 5, int(os.environ.get("ITERS", "40"))
tr = FusedTrainer(HipOps(dev), B, dev, loss=os.environ.get("LOSS", "gp"), critic_iter=CI, seed=5)
tr.eng.D.load(orc.make_params(orc.disc_param_shapes(), 7))
tr.eng.G.load(orc.make_params(orc.gen_param_shapes(), 8))
tr.weights_changed()
rs, rp = orc.synthetic_alignments(512, 9)
tr.set_pool(rs, rp)
marks = []            # (label, [events per replay])


def mark(label):
    evs = []
    marks.append((label, evs))

    def rec():
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        evs.append(ev)
    pgan_tape.call(rec)


e = tr.eng
adam_d, gen, adam_g = e.adam_d, e.generator_grads_resident, e.adam_g
count = [0]


def adam_d_marked():
    adam_d()
    count[0] += 1
    mark(f"after critic step {count[0]}")


def gen_marked(**kw):
    mark("generator step begins")
    gen(**kw)


def adam_g_marked():
    adam_g()
    mark("after the generator step")


for _ in range(2):
    tr.iteration()
torch.cuda.synchronize()
e.adam_d, e.generator_grads_resident, e.adam_g = adam_d_marked, gen_marked, adam_g_marked
body = tr._iteration_body_batched


def body_marked():
    mark("iteration begins")
    body()


tr._iteration_body_batched = body_marked
tr.record_tape()
for _, evs in marks:
    evs.clear()
torch.cuda.synchronize()
for _ in range(ITERS):
    tr.replay_tape()
torch.cuda.synchronize()
print(f"tape: {tr.tape.native_calls} library calls, {tr.tape.python_steps} Python steps; {ITERS} replays")
prev_label, prev = marks[0]
total = 0.0
for label, evs in marks[1:]:
    dt = sum(a.elapsed_time(b) for a, b in zip(prev, evs)) / len(evs)
    print(f"{prev_label:28s} -> {label:28s} {dt * 1e3:8.1f} us")
    total += dt
    prev_label, prev = label, evs
wrap = sum(a.elapsed_time(b) for a, b in zip(marks[-1][1][:-1], marks[0][1][1:])) / (ITERS - 1)
print(f"{'after the generator step':28s} -> {'next iteration begins':28s} {wrap * 1e3:8.1f} us")
print(f"iteration: {(total + wrap) * 1e3:.1f} us")
