#!/usr/bin/env python3
"""Benchmark of the fused DCWGAN-GP training step on MI355X.

One "step" = one full iteration of the reference trainer's inner loop
(/root/reference/src/train_wgan_v2.py:516-634): critic_iter (5) critic updates + 1 generator
update.  --config picks the BASELINE.json configuration (default 1 = configs[1]: sigmoid generator, gp loss,
batch 64 per GPU; 2 = configs[2]: tanh generator + --normalize, global batch 512 over the GPUs; 3 = configs[3]: div
loss + replay buffer of 20 batches + row-permuted input, global batch 256).  Inputs are synthetic ms-style 64x64
alignments resident in HBM before the timed region; z and t are drawn on the device.
Metric: train alignments/sec = critic_iter * global_batch * steps / wall time.

  python bench.py --gpus 1 --steps 50 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel family (implicit-GEMM convolutions; all three
kinds run on the 16-bit matrix pipe: by default as three exact fp16 partial products per fp32 product, with
PGAN_CONV_PRECISION=x3 as six bf16 ones): `achieved` counts the MFMA flops the family executes on its pipe against
that pipe's dense peak; `algorithmic_tflops` is the fp32-equivalent rate.  After the timed region one iteration's
launches are noted and every launch shape is timed with HIP events on the launch stream around back-to-back launches
replayed from a hipGraph (eager per-launch events would time host launch gaps).  `hbm_kernels` does the same for the
bandwidth-bound kernels (norms, single-channel convolutions, Adam) against their algorithmic bytes.
`cpu_baseline` times the CPU oracle (stock PyTorch restatement of the reference step, with the reference's wasted
backward passes included) on this box's host cores over a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pg-alignments-gan_amd"))

import numpy as np
import torch

def forward_flops(data=64, latent=128):
    """(discriminator, generator) forward FLOP per sample, 2 per multiply-add, convolutions and linear layers only
    (data = 64: 204,593,664 and 205,703,168, the reference networks' counts)."""
    from pgan_engine import net_dims
    dm = net_dims(data, latent)
    d = sum(dm["d2c"][k] * dm["d2c"][k - 1] * 16 * dm["d2h"][k] ** 2 for k in range(1, len(dm["d2c"])))
    d += sum(dm["d1c"][k] * dm["d1c"][k - 1] * 4 * dm["d1l"][k] for k in range(1, 4)) + dm["feat2"] + dm["feat1"]
    g = dm["g2c"][0] * dm["g2c"][1] * 16 + sum(dm["g2c"][k] * dm["g2c"][k + 1] * 16 * dm["g2h"][k] ** 2
                                               for k in range(1, len(dm["g2c"]) - 1))
    g += dm["g1c"][0] * dm["g1c"][1] * 4 + sum(dm["g1c"][k] * dm["g1c"][k + 1] * 4 * dm["g1l"][k]
                                               for k in range(1, len(dm["g1c"]) - 1))
    return 2 * d, 2 * g


def flop_per_alignment(data=64, critic_iter=5):
    """Algorithmic FLOP of one iteration per real alignment: critic step F_G + 12 F_D per alignment (gp and div alike),
    generator step 3 F_G + 2 F_D per latent sample, one generator step per critic_iter critic steps."""
    f_d, f_g = forward_flops(data)
    return (critic_iter * (f_g + 12 * f_d) + 3 * f_g + 2 * f_d) / critic_iter


PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense
PEAK_HBM_GBS = 8000.0
# sustained by a register-only MFMA loop on random operands on this part (tools/mfma_peak.hip,
# tools/mfma_bf16_peak.hip): the clocks drop under data-dependent power
SUSTAINED_RANDOM_TFLOPS = {"fp32": 92.0, "bf16": 1770.0, "fp16": 1770.0}      # fp16 runs at the bf16 rate


def synthetic_pool(n, seed, size=64):
    """ms-style synthetic alignments: every column polymorphic, derived-allele count ~ 1/i."""
    rng = np.random.Generator(np.random.PCG64(seed))
    w = 1.0 / np.arange(1, size)
    w /= w.sum()
    counts = rng.choice(np.arange(1, size), size=(n, size), p=w)
    rank = np.argsort(np.argsort(rng.random((n, size, size)), axis=1), axis=1)
    sites = (rank < counts[:, None, :]).astype(np.float32)
    pos = np.sort(rng.random((n, size)), axis=1).astype(np.float32)
    return torch.from_numpy(sites[:, None]), torch.from_numpy(pos[:, None])


def _numel(t):
    n = 1
    for d in t.shape:
        n *= d
    return n


class RecordingOps:
    """Proxy around HipOps that, while `recording`, notes every launch of the kernels listed in DESCRIBE (kernel,
    shape, the actual argument tensors) of an iteration so that each launch shape can be timed afterwards."""

    _OWN = ("_ops", "recording", "calls")

    # name -> (key, algorithmic flops, algorithmic bytes) of one call
    @staticmethod
    def _conv(kind):
        def d(a, k):
            if kind == "conv_down":
                x, w = a[0], a[1]
                n, cs, cl, hs = x.shape[0], w.shape[0], w.shape[1], x.shape[2] // 2
            elif kind == "conv_up":
                y, w = a[0], a[1]
                n, cs, cl, hs = y.shape[0], w.shape[0], w.shape[1], y.shape[2]
            else:
                ya, x = a[0], a[1]
                n, cs, cl, hs = ya.shape[0], ya.shape[1], x.shape[1], ya.shape[2]
            small, large, wts = n * cs * hs * hs, n * cl * 4 * hs * hs, cs * cl * 16
            return (kind, n, cs, cl, hs), 2.0 * n * cs * cl * 16 * hs * hs, 4.0 * (small + large + wts)
        return d

    @staticmethod
    def _stream_op(kind, tensors):
        """tensors: how many activation-sized tensors the op reads + writes"""
        def d(a, k):
            t = a[0] if a[0] is not None else a[1]
            n = tensors - (1 if (kind == "gn_bwd2" and a[0] is None) else 0)
            return (kind,) + tuple(t.shape), 0.0, 4.0 * n * _numel(t)
        return d

    DESCRIBE = {
        "conv_down": _conv.__func__("conv_down"), "conv_up": _conv.__func__("conv_up"),
        "conv_down_gn_sums": _conv.__func__("conv_down"),      # first stage: counted (and timed) as the convolution it is
        "conv_wgrad": _conv.__func__("conv_wgrad"),
        # the GroupNorm backward ops recompute the LeakyReLU mask from the pre-norm input: the forward output is not read
        "gn_fwd": _stream_op.__func__("gn_fwd", 2), "gn_bwd": _stream_op.__func__("gn_bwd", 3),
        "gn_jvp": _stream_op.__func__("gn_jvp", 3), "gn_bwd2": _stream_op.__func__("gn_bwd2", 5),
        "bn_fwd": _stream_op.__func__("bn_fwd", 2), "bn_bwd": _stream_op.__func__("bn_bwd", 4),
        "adam": lambda a, k: (("adam", _numel(a[0])), 0.0, 28.0 * _numel(a[0])),
    }

    def __init__(self, ops):
        self._ops = ops
        self.recording = False
        self.calls = {}

    def __setattr__(self, name, value):        # settings made through the proxy reach the real backend
        if name in self._OWN:
            object.__setattr__(self, name, value)
        else:
            setattr(self._ops, name, value)

    def __getattr__(self, name):
        fn = getattr(self._ops, name)
        if name not in self.DESCRIBE or not callable(fn):
            return fn

        def wrapped(*a, **k):
            if self.recording:
                key, flops, nbytes = self.DESCRIBE[name](a, k)
                rec = self.calls.setdefault(key, dict(count=0, args=(a, k), flops=flops, bytes=nbytes))
                rec["count"] += 1
            return fn(*a, **k)
        return wrapped

    def time_shapes(self, reps=20, stream=None):
        """Average duration of every recorded launch shape: HIP events on the launch stream around `reps` back-to-back
        launches replayed from a hipGraph (no host work between the launches: the shortest kernels here take 3 us, a
        launch through Python 5-10); shapes whose launches cannot be captured are timed with the queue fed from Python.
        stream: an existing (idle) stream to capture on -- a new one would shift which hardware queue later streams get."""
        import pgan_hip
        from pgan_hip import StepGraph
        out = {}
        cap = stream if stream is not None else torch.cuda.Stream()
        # markers for a kernel trace of this run: one launch of a kernel nothing else here uses (the loader's bit packer on
        # one alignment) right before and right after every timed region, so that tools/trace_shapes.py can cut the trace
        # into the same regions, in the order of the `per_shape` table
        mark_in = torch.zeros(64 * 64, device="cuda")
        mark_out = torch.zeros(512, dtype=torch.uint8, device=mark_in.device)

        def marker():
            pgan_hip._lib.pgan_pack_sites(mark_in.data_ptr(), mark_out.data_ptr(), 1, cap.cuda_stream)

        if hasattr(self._ops, "reserve_stream"):
            self._ops.reserve_stream(cap.cuda_stream)
        for key, rec in self.calls.items():
            a, k = rec["args"]
            fn = getattr(self._ops, key[0])
            with torch.cuda.stream(cap):
                for _ in range(3):
                    fn(*a, **k)
                cap.synchronize()
                graph = StepGraph()
                self._ops.capturing = True
                try:
                    with graph:
                        for _ in range(reps):
                            fn(*a, **k)
                except Exception:
                    graph = None
                finally:
                    self._ops.capturing = False
                if graph is not None:
                    graph.launch()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                marker()
                s.record()
                if graph is not None:
                    graph.launch()
                else:
                    for _ in range(reps):
                        fn(*a, **k)
                e.record()
                marker()
                e.synchronize()
            ms = s.elapsed_time(e) / reps
            out[key] = dict(calls=rec["count"], ms=ms, total_ms=ms * rec["count"], flops=rec["flops"], bytes=rec["bytes"],
                            timed="graph" if graph is not None else "queue")
        return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(loss, act, batch, critic_iter, iters=12, data=64, one_thread_batch=None):
    """The reference's step on the host cores: the CPU oracle with the reference's wasted work included (gp: fakes
    left attached, so every critic backward also runs through the generator; the generator step's backward also
    computes the discriminator's weight gradients), on all cores of this process and on one thread (the reference's
    SLURM script asks for one core, /root/reference/slurm/train_gan_conda.sh:8), plus the same without the wasted
    work."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import dcwgan_oracle as orc
    torch.manual_seed(0)
    rs, rp = orc.synthetic_alignments(batch * critic_iter, 20263, size=data)
    if act == "tanh":
        rs, rp = (rs - 0.5) / 0.5, (rp - 0.5) / 0.5
    full_batch = batch

    def run(n, reference_cost, batch=full_batch):
        batches = [(rs[i * batch:(i + 1) * batch], rp[i * batch:(i + 1) * batch]) for i in range(critic_iter)]
        gp_ = orc.make_params(orc.gen_param_shapes(data=data), 20261)
        dp = orc.make_params(orc.disc_param_shapes(data), 20262)
        gbuf = orc.make_gen_buffers(data=data)
        d_opt, g_opt = orc.AdamState(dp), orc.AdamState(gp_)

        def one():
            zs = [torch.randn(batch, 128, 1, 1) for _ in range(critic_iter + 1)]
            ts = [torch.rand(batch, 1, 1, 1) for _ in range(critic_iter)]
            orc.train_iteration(dp, gp_, gbuf, d_opt, g_opt, batches, zs, ts if loss == "gp" else None, loss, 10.0, act,
                                reference_cost=reference_cost)
        if n > 1:
            one()
        t0 = time.perf_counter()
        for _ in range(n):
            one()
        return (time.perf_counter() - t0) / n

    cores = torch.get_num_threads()
    per_it = run(iters, True)                      # the headline sample: about 10 s of work on the host cores at batch 64
    lean_n = max(1, iters // 4)
    lean = run(lean_n, False)
    b1 = one_thread_batch or batch
    torch.set_num_threads(1)
    try:
        one_thread = run(1, True, b1)
    finally:
        torch.set_num_threads(cores)
    rate = lambda sec, b=batch: round(critic_iter * b / sec, 2)
    return dict(value=rate(per_it), unit="alignments/sec", cores=cores, kind="port", cpu_model=_cpu_model(),
                one_thread=dict(value=rate(one_thread, b1), cores=1,
                                sample=f"1 full iteration at batch {b1}, {one_thread:.2f} s"),
                without_wasted_backward=dict(value=rate(lean), cores=cores, sample=f"{lean_n} full iterations, {lean:.2f} s each"),
                sample=f"{iters} full iterations (5 critic + 1 generator step, batch {batch}, {loss} loss, {act} generator) "
                       f"after 1 warm-up, stock-PyTorch CPU oracle doing the reference's work (fakes attached in gp critic "
                       f"steps, discriminator weight gradients in the generator step), {per_it:.2f} s/iteration")


CONFIGS = {
    # BASELINE.json configs[i] -> defaults (explicit flags override them)
    1: dict(loss="gp", act="sigmoid", batch=64, global_batch=None, buffer=0, permute=False,
            name="configs[1]: DCWGAN-GP sigGen 64x64, batch 64, fp32, single MI355X"),
    2: dict(loss="gp", act="tanh", batch=64, global_batch=512, buffer=0, permute=False,
            name="configs[2]: DCWGAN-GP tanNorm + --normalize, 64x64, global batch 512 data-parallel"),
    3: dict(loss="div", act="sigmoid", batch=32, global_batch=256, buffer=20, permute=True,
            name="configs[3]: DCWGAN-GP div loss + --use_buffer + --permute, 64x64, global batch 256"),
    4: dict(loss="gp", act="sigmoid", batch=128, global_batch=1024, buffer=0, permute=False, data=128,
            name="configs[4]: scaled shape 128 inds x 128 sites (build-defined networks, one more stride-2 stage), "
                 "global batch 1024"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS),
                    help="BASELINE.json configuration: 1 sigGen gp batch 64 (default), 2 tanNorm + normalize, "
                         "3 div + replay buffer + permute, 4 the scaled 128 x 128 networks at batch 128 per GPU")
    ap.add_argument("--batch", type=int, default=None, help="batch per GPU (default: the configuration's)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="fixed total batch split over the GPUs (strong scaling); with --config 2 / 3 on several GPUs "
                         "the configuration's global batch is the default")
    ap.add_argument("--loss", default=None, choices=["gp", "div"])
    ap.add_argument("--act", default=None, choices=["sigmoid", "tanh"])
    ap.add_argument("--use-buffer", type=int, default=None, help="replay buffer of this many batches (0: none)")
    ap.add_argument("--permute", action="store_true", default=None, help="row-permuted real alignments")
    ap.add_argument("--critic-iter", type=int, default=5)
    ap.add_argument("--route", default="tape", choices=["auto", "eager", "tape", "graph"],
                    help="how an iteration is enqueued: the launch tape replayed by the library (default; the fastest route in "
                         "every same-box comparison, and the one that also carries collectives), eager Python calls, a "
                         "hipGraph replay (one GPU only), or auto = time all available ones first and take the fastest")
    ap.add_argument("--no-graph", action="store_true", help="same as --route eager")
    ap.add_argument("--separate-generator-forwards", action="store_true",
                    help="one generator forward per critic step instead of the iteration's forwards batched")
    ap.add_argument("--local-bn", action="store_true",
                    help="several GPUs: BatchNorm statistics per GPU instead of over the global batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of a multi-rank run: nccl (= RCCL, the measured configuration) or gloo -- a "
                         "rehearsal of the multi-rank control flow where the ranks have to share GPUs (collectives staged "
                         "through host memory; ranks beyond the visible devices wrap around)")
    ap.add_argument("--profile-iters", type=int, default=20, help="back-to-back launches per shape for per-kernel timing")
    ap.add_argument("--no-fp32-pipe", action="store_true", help="skip the side measurement with every convolution on the fp32 matrix pipe")
    ap.add_argument("--kernels-only", action="store_true",
                    help="one untimed iteration to collect the launch shapes, then only the isolated per-shape timing "
                         "(the command to put under rocprofv3 --kernel-trace --stats when comparing its per-kernel "
                         "average durations with roofline.per_shape)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="kernel-variant override of the split-operand convolutions for comparison runs; may repeat")
    args = ap.parse_args()
    if args.kernels_only:
        args.steps, args.warmup, args.no_graph, args.no_cpu_baseline, args.no_fp32_pipe = 1, 0, True, True, True
    if args.no_graph:
        args.route = "eager"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    cfg = CONFIGS[args.config]
    args.loss = args.loss or cfg["loss"]
    args.act = args.act or cfg["act"]
    buffer_n = cfg["buffer"] if args.use_buffer is None else args.use_buffer
    permute = cfg["permute"] if args.permute is None else args.permute
    scaling = "weak"
    gb = args.global_batch
    if gb is None and args.batch is None and world > 1 and cfg["global_batch"]:
        gb = cfg["global_batch"]           # the configuration is quoted on a fixed global batch
    if gb is not None:
        if gb % world:
            raise SystemExit(f"--global-batch {gb} is not a multiple of the {world} GPUs")
        args.batch, scaling = gb // world, "strong"
    elif args.batch is None:
        args.batch = cfg["batch"]
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1) if args.dist_backend != "nccl" else local)
    torch.cuda.set_device(dev)
    pg = None
    use_dist = world > 1 or bool(os.environ.get("PGAN_FORCE_DIST"))   # PGAN_FORCE_DIST: exercise the RCCL path on one GPU

    from pgan_hip import HipOps
    from pgan_trainer import FusedTrainer

    from pgan_engine import disc_param_shapes, gen_param_shapes
    data = cfg.get("data", 64)
    # one pool for all ranks: the ranks of a data-parallel run split one run (same pool order, same generator streams;
    # rank r takes its rows of every global batch and draw), so N GPUs train on the samples one GPU at N x batch would
    ps, pp = synthetic_pool(4096 if data == 64 else 2048, 20260 + 1, size=data)
    flop_pa = flop_per_alignment(data, args.critic_iter)

    def make_trainer(backend):
        t = FusedTrainer(backend, args.batch, dev, loss=args.loss, act=args.act, critic_iter=args.critic_iter,
                         seed=1234, rank=rank, world=world, batched_generator=not args.separate_generator_forwards,
                         data=data)
        gen = torch.Generator().manual_seed(20260)   # identical initial weights on every rank
        for store, shapes in ((t.eng.G, gen_param_shapes(data=data)), (t.eng.D, disc_param_shapes(data))):
            for k, sh in shapes.items():
                if len(sh) == 1:
                    v = torch.ones(sh) if k.endswith("weight") and not k.startswith("out") else torch.zeros(sh)
                else:
                    fan = int(np.prod(sh[1:]))
                    v = (torch.rand(sh, generator=gen) * 2 - 1) / fan ** 0.5
                store.p[k].copy_(v)
        t.set_pool(ps, pp, normalize=(args.act == "tanh"))
        if permute:          # --permute: every alignment's individuals in a new random order at the start of an epoch
            t.shuffle_pool(rows=True)
        if buffer_n:
            t.enable_fake_buffer(buffer_n * args.batch)
        return t

    ops = HipOps(dev, max_samples=max(1024, 4 * args.batch))
    for item in args.opt:
        import pgan_hip
        name, value = item.split("=")
        pgan_hip.set_x3_option(name, int(value))
    timed = RecordingOps(ops)
    tr = make_trainer(timed)
    if use_dist:
        # after the trainer: its four streams hold the process's four hardware queues, RCCL's streams share them
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29531"
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
        pg = dist.group.WORLD
        tr.attach_process_group(pg, sync_bn=not args.local_bn)

    # Three routes enqueue the same kernels on the same streams: eager launches from Python (about 8 ms of host time per
    # iteration), a launch tape (the same calls replayed from a flat list, about a third of that) and a hipGraph replay
    # (no host work, but the runtime's own placement of the graph's branches; one GPU only -- no capture around RCCL).
    # The tape is the default; --route auto times 10 untimed iterations of each and takes the fastest.
    def rate(fn, n=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    def max_over_ranks(values):
        """element-wise maximum of a few host numbers over the ranks"""
        import torch.distributed as dist
        t = torch.tensor(values, dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    routes = {}
    if args.route in ("auto", "eager"):
        routes["eager"] = tr.iteration
    if args.route in ("auto", "tape"):
        try:
            tr.record_tape()
            routes["tape"] = tr.replay_tape
        except RuntimeError as ex:            # op backend without tape support: the eager route remains
            if args.route == "tape":
                raise
            print(f"launch tape unavailable: {ex}", file=sys.stderr)
    if args.route in ("auto", "graph") and world == 1 and pg is None:
        tr.capture()
        routes["graph"] = tr.replay
    if not routes:
        raise SystemExit(f"--route {args.route} is not available here")
    times = {k: rate(fn) for k, fn in routes.items()}
    if world > 1:      # every rank takes the same route: the slowest rank's timing decides
        times = dict(zip(sorted(times), max_over_ranks([times[k] for k in sorted(times)])))
    best = min(times, key=times.get)
    step, use_graph = routes[best], best == "graph"
    route = best + " (calibration, ms per iteration: " + ", ".join(f"{k} {v * 1e3:.2f}" for k, v in sorted(times.items())) + ")"
    ct = getattr(tr, "tape", None)
    if best == "tape" and hasattr(ct, "native_calls"):
        route += f"; {ct.native_calls} library calls replayed from C, {ct.python_steps} Python steps between them"
    for _ in range(args.warmup):
        step()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = max_over_ranks([dt])[0]
    losses = tr.losses()

    # per-kernel timing: note one iteration's launches, then time each launch shape on its own
    timed.recording = True
    tr.iteration()
    timed.recording = False
    torch.cuda.synchronize()
    # the per-shape launches are captured into a graph on an idle stream of the engine -- except under a process group:
    # RCCL's watchdog thread polls the events of finished collectives, which were recorded on the engine's streams, and HIP
    # refuses a query of an event whose stream is capturing (the capture is then invalidated): a fresh stream there
    summ = timed.time_shapes(args.profile_iters, stream=None if pg is not None else getattr(tr.eng.side, "stream", None))

    if rank == 0:
        conv = {k: v for k, v in summ.items() if k[0].startswith("conv_")}
        fam = {}
        for (name, n, cs, cl, hs), r in conv.items():
            f = fam.setdefault(name, dict(total_ms=0.0, flops=0.0, calls=0))
            f["total_ms"] += r["total_ms"]
            f["flops"] += r["flops"] * r["calls"]
            f["calls"] += r["calls"]
        dom = max(fam, key=lambda k: fam[k]["total_ms"]) if fam else None
        prec = getattr(ops, "precision", "fp32")
        x3 = prec in ("x3", "h2")               # split operands on the 16-bit matrix pipe
        wg3 = x3 and getattr(ops, "wgrad_x3", False)
        pipe16 = "fp16" if prec == "h2" else "bf16"
        pipe = {k: (pipe16 if (x3 and (k != "conv_wgrad" or wg3)) else "fp32") for k in fam}
        cname = lambda k: f"{k[0]}[N={k[1]},Cs={k[2]},Cl={k[3]},hs={k[4]}]"
        # bandwidth-bound kernels: algorithmic bytes (tensors read + written once; Adam 28 B per parameter) over the
        # isolated launch duration, against the 8 TB/s HBM peak
        hbm = {}
        for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"]):
            single = k[0].startswith("conv_") and k[3] == 1
            if k[0].startswith("conv_") and not single:
                continue
            label = cname(k) if single else f"{k[0]}{list(k[1:])}"
            gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            hbm[label] = dict(calls=v["calls"], avg_us=round(v["ms"] * 1e3, 2), algorithmic_MB=round(v["bytes"] / 1e6, 3),
                              GBps=round(gbs, 1), frac_of_hbm_peak=round(gbs / PEAK_HBM_GBS, 4), timed=v["timed"])
        roofline = None
        if dom:
            f = fam[dom]
            alg = f["flops"] / (f["total_ms"] * 1e-3) / 1e12
            mult = {"bf16": 6.0, "fp16": 3.0, "fp32": 1.0}[pipe[dom]]      # MFMA products executed per fp32 product
            peak = PEAK_FP32_MFMA_TFLOPS if pipe[dom] == "fp32" else PEAK_BF16_MFMA_TFLOPS
            ach = alg * mult
            per_shape = {cname(k): dict(calls=v["calls"], avg_us=round(v["ms"] * 1e3, 2),
                                        tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), timed=v["timed"])
                         for k, v in sorted(conv.items())}
            # HBM traffic of the family's largest launch shape from the committed PMC passes (bytes per launch):
            # rocprofv3 --pmc runs of tools/onekernel.py, separate passes for FETCH_SIZE and WRITE_SIZE; the file names
            # the commit its kernels were built from
            # traffic: FETCH_SIZE is doubled (gfx950 tallies the 128-byte requests of 16-byte-per-lane loads, which carry
            # these kernels' reads, at 64 bytes), WRITE_SIZE is exact; both include re-reads served by the on-die cache
            traffic, traffic_of, traffic_at, traffic_raw, traffic_ratio = None, None, None, None, None
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                traffic_at = pm.get("measured_at")
                have = [k for k in conv if k[0] == dom and cname(k) in pm]
                if have:       # the measured shape with the largest time share
                    big = max(have, key=lambda k: conv[k]["total_ms"])
                    row = pm[cname(big)]
                    traffic_raw = (row["FETCH_SIZE_KB"] + row["WRITE_SIZE_KB"]) * 1024
                    traffic = (2 * row["FETCH_SIZE_KB"] + row["WRITE_SIZE_KB"]) * 1024
                    traffic_ratio = round(traffic / ((row["algorithmic_in_KB"] + row["algorithmic_out_KB"]) * 1024), 3)
                    traffic_of = cname(big)
            except (OSError, ValueError, KeyError):
                pass
            kname = dom + "_x3_kernel" if pipe[dom] != "fp32" else dom + "_kernel"
            roofline = dict(bound="mfma", kernel=f"{kname} ({pipe[dom]} matrix pipe, implicit GEMM)",
                            achieved=round(ach, 3), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                            traffic=traffic, traffic_shape=traffic_of, traffic_measured_at=traffic_at,
                            traffic_over_algorithmic=traffic_ratio, traffic_raw_counters=traffic_raw,
                            traffic_note="bytes per launch of traffic_shape: 2 x FETCH_SIZE + WRITE_SIZE (rocprofv3 --pmc, "
                                         "separate passes; FETCH_SIZE doubled for 16-byte-per-lane loads on gfx950); "
                                         "algorithmic = operands read once (fp16 weight planes, fp32 activations) + output",
                            algorithmic_tflops=round(alg, 3),
                            algorithmic_frac_of_fp32_mfma_peak=round(alg / PEAK_FP32_MFMA_TFLOPS, 4),
                            note=(f"achieved / peak / frac price the {pipe[dom]} MFMA work actually executed ({mult:g} partial "
                                  "products per fp32 product) against the dense 16-bit matrix peak; algorithmic_* price the "
                                  "fp32 convolution flops (2 N Cs Cl 16 hs^2 per launch) against the fp32 matrix pipe's "
                                  "157.3 TFLOP/s" if pipe[dom] != "fp32" else "fp32 convolution flops on the fp32 matrix pipe"),
                            executed_per_algorithmic_flop=mult,
                            sustained_on_random_operands=SUSTAINED_RANDOM_TFLOPS[pipe[dom]],
                            frac_of_sustained=round(ach / SUSTAINED_RANDOM_TFLOPS[pipe[dom]], 4),
                            algorithmic_ceiling_tflops=round(SUSTAINED_RANDOM_TFLOPS[pipe[dom]] / mult, 1),
                            avg_launch_us=round(f["total_ms"] * 1e3 / f["calls"], 2), launches=f["calls"],
                            flops_per_launch_avg=f["flops"] / f["calls"],
                            measured=f"HIP events on the launch stream around {args.profile_iters} back-to-back launches (replayed "
                                     f"from a hipGraph) of every launch shape of one iteration, after the timed region; per-launch "
                                     f"average weighted by the shape's launches per iteration", families={k: dict(
                                         share_ms_per_iter=round(v["total_ms"], 3), pipe=pipe[k],
                                         algorithmic_tflops=round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12, 2))
                                         for k, v in fam.items()},
                            per_shape=per_shape)
        nalign = args.critic_iter * args.batch * world * args.steps
        value = nalign / dt
        extras = (f", replay buffer of {buffer_n} batches" if buffer_n else "") + (", row-permuted input" if permute else "")
        out = dict(metric=f"train alignments/sec (WGAN-GP step, {data}x{data})", value=round(value, 2), unit="alignments/sec",
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 4),
                   higher_is_better=True, scaling=scaling, vs_baseline=None, dtype="f32", data="synthetic",
                   arithmetic={"h2": "fp32 storage and accumulation; 4x4 convolution products as three exact fp16 partial products "
                                     "(two-plane split with per-block power-of-two scaling, operand error below 2^-22) on the "
                                     "16-bit matrix pipe",
                               "x3": "fp32 storage and accumulation; 4x4 convolution products as six exact bf16 partial products "
                                     "(3-way split, error below an fp32 multiply's rounding) on the bf16 matrix pipe",
                               "fp32": "fp32 throughout (fp32 matrix pipe)"}[prec],
                   config=dict(workload=f"{cfg['name']} -- run as {args.act} generator"
                                        f"{' + normalize' if args.act == 'tanh' else ''}, {args.loss} loss, batch {args.batch}/GPU"
                                        f"{extras}, {args.critic_iter} critic + 1 generator step per step",
                               baseline_config=args.config, global_batch=args.batch * world, critic_iter=args.critic_iter,
                               loss=args.loss, hip_graph=bool(use_graph), launch=route, parallelism=f"dp{world}",
                               batchnorm=("global batch statistics (per-channel sums all-reduced)" if tr.sync_bn
                                          else "per-GPU batch statistics")),
                   step_tflops=round(value * flop_pa / 1e12, 3), flop_per_alignment=flop_pa,
                   step_frac_of_fp32_mfma_peak=round(value * flop_pa / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world), 4),
                   losses=dict(critic=losses[0], d_real=losses[1], d_fake=losses[2], penalty=losses[3], generator=losses[4]),
                   roofline=roofline, hbm_kernels=hbm,
                   # the launch shapes in the order they were timed (= the order of the marker pairs in a kernel trace)
                   timed_shapes=[(cname(k) if k[0].startswith("conv_") else f"{k[0]}{list(k[1:])}") + f" {v['ms'] * 1e3:.2f}us"
                                 for k, v in summ.items()])
        if world == 1 and x3 and not args.no_fp32_pipe and data == 64:
            # the same step with the other arithmetic choices, for reference: three bf16 planes / six products, and
            # every convolution on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32)
            for other, label in (("x3", "bf16_split"), ("fp32", "fp32_pipe_only")):
                if other == prec:
                    continue
                t32 = make_trainer(HipOps(dev, precision=other, max_samples=max(1024, 4 * args.batch)))
                n32 = max(5, args.steps // 2)
                if best == "graph":
                    t32.capture()
                elif best == "tape":
                    t32.record_tape()
                step32 = {"graph": t32.replay, "tape": t32.replay_tape, "eager": t32.iteration}[best]
                for _ in range(3):
                    step32()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(n32):
                    step32()
                torch.cuda.synchronize()
                d32 = time.perf_counter() - t1
                out[label] = dict(value=round(args.critic_iter * args.batch * n32 / d32, 2), unit="alignments/sec",
                                  ms_per_step=round(d32 / n32 * 1e3, 4), steps=n32,
                                  note=f"PGAN_CONV_PRECISION={other}: " + ("three bf16 planes, six products per fp32 product"
                                                                          if other == "x3" else
                                                                          "all convolutions on the fp32 matrix pipe"))
                del t32
        if world == 1 and not args.no_cpu_baseline:
            if data == 64:
                out["cpu_baseline"] = cpu_baseline(args.loss, args.act, args.batch, args.critic_iter)
            else:      # bounded sample: a smaller batch of the same networks (the rate per alignment barely depends on it)
                out["cpu_baseline"] = cpu_baseline(args.loss, args.act, min(args.batch, 16), args.critic_iter, iters=8,
                                                   data=data, one_thread_batch=2)
        print(json.dumps(out))
    if pg is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
