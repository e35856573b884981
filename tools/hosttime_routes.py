"""Host time per iteration of every way an iteration can be enqueued (GPU box, repo root): eager Python, the launch tape
replayed call by call from Python, the native tape walked in C by the library, and the hipGraph launch -- next to the
iteration's wall time on each route.  Same trainer set-up as bench.py config 1."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pg-alignments-gan_amd")]
import numpy as np
import torch
import pgan_tape
from pgan_hip import HipOps
from pgan_trainer import FusedTrainer
from pgan_engine import disc_param_shapes, gen_param_shapes
from bench import synthetic_pool

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
tr = FusedTrainer(HipOps(dev), 64, dev, seed=1234)
gen = torch.Generator().manual_seed(20260)
for store, shapes in ((tr.eng.G, gen_param_shapes()), (tr.eng.D, disc_param_shapes())):
    for k, sh in shapes.items():
        if len(sh) == 1:
            v = torch.ones(sh) if k.endswith("weight") and not k.startswith("out") else torch.zeros(sh)
        else:
            v = (torch.rand(sh, generator=gen) * 2 - 1) / int(np.prod(sh[1:])) ** 0.5
        store.p[k].copy_(v)
tr.weights_changed()
tr.set_pool(*synthetic_pool(4096, 20261))


def measure(fn, n=30):
    """(host ms to enqueue one iteration on an idle device, wall ms per iteration back to back)"""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    host = 0.0
    for _ in range(n):          # one iteration at a time: the queues are empty, so the call never waits for the device
        a = time.perf_counter()
        fn()
        host += time.perf_counter() - a
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * host / n, 1e3 * (time.perf_counter() - t0) / n


rows = [("eager (Python drives every launch)", measure(tr.iteration))]
os.environ["PGAN_NATIVE_TAPE"] = "0"
tr.record_tape()
rows.append((f"launch tape replayed from Python ({len(tr.tape)} calls)", measure(tr.replay_tape)))
os.environ["PGAN_NATIVE_TAPE"] = "1"
tr.record_tape()
rows.append((f"native tape walked in C ({tr.tape.native_calls} calls, {tr.tape.python_steps} Python steps)", measure(tr.replay_tape)))
tr.tape = None
tr.capture()
rows.append(("hipGraph launch", measure(tr.replay)))
print("route | host ms to enqueue one iteration (idle device) | wall ms per iteration (back to back)")
for name, (h, w) in rows:
    print(f"{name} | {h:.3f} | {w:.3f}")
