"""FusedTrainer: the whole critic_iter loop + generator step of PG-Alignments-GAN's trainer
(/root/reference/src/train_wgan_v2.py:516-634) as one sequence of HIP kernel launches on static
buffers, optionally captured into a hipGraph and replayed.

Inside one iteration nothing touches the host: real batches come from a pool resident in HBM
(pgan_take_batch), z ~ N(0,1) and t ~ U[0,1) from device-side Philox state, Adam's step counts live
in device memory, and the losses stay in device buffers until the caller reads them.

Data parallel (one process per GPU, torch.distributed with the RCCL backend): every rank holds full
replicas of both networks and optimiser states, and the flat gradient buffers are summed over the ranks
before each Adam step, which applies the 1 / ranks itself.  The generator's BatchNorm layers take their batch
statistics over all ranks' samples (per-channel partial sums all-reduced inside forward and backward, the
"global batch" of a single-device run) unless sync_bn=False keeps them per rank.
With global_stream (the default on several ranks) the ranks split ONE run: every rank holds the whole pool and keeps
the same generator states and pool cursor, which advance by the global batch, and rank r of P takes rows
[r B/P, (r+1) B/P) of every real batch, z draw, t draw and augmentation draw -- P ranks at batch B/P then train on
exactly the samples one device at batch B would (the weights agree up to the summation order of the all-reduce).
Without it every rank draws from generator streams and a pool of its own.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

import pgan_tape
from pgan_engine import Engine


class FusedTrainer:
    def __init__(self, ops, batch: int, device, loss: str = "gp", act: str = "sigmoid", lam: float = 10.0,
                 critic_iter: int = 5, seed: int = 1234, lr_d: float = 5e-5, lr_g: float = 5e-5,
                 latent: int = 128, process_group=None, rank: int = 0, world: int = 1, sync_bn: Optional[bool] = None,
                 batched_generator: bool = True, data: int = 64, global_stream: Optional[bool] = None):
        """data: side of the square alignments, 64 (the reference's networks) or 128 (the scaled networks).
        global_stream: the ranks take their rows of one global batch / draw (default on several ranks; see above)."""
        self.ops, self.dev, self.critic_iter = ops, torch.device(device), int(critic_iter)
        self.global_stream = (world > 1) if global_stream is None else bool(global_stream)
        self.eng = Engine(ops, batch, device, loss=loss, act=act, lam=lam, latent=latent, lr_d=lr_d, lr_g=lr_g,
                          data=data)
        self.eng.use_device_step_counters()
        # the generator forwards of an iteration after the first (critic steps 2.., the generator step's own) share
        # their weights and run as ONE pass over (critic_iter) x batch samples beside the first critic step
        self.batched_generator = bool(batched_generator)
        if self.batched_generator:
            self.eng.enable_gen_chunks(self.critic_iter + 1)
        self.rank, self.world, self.pg = rank, world, process_group
        # data parallel: gradients are summed over the ranks before every Adam step, in two collectives per step.  The
        # large 2-D convolution weights sit at the end of the flat buffers; the critic's last stage (8.4 of 11.1 MB,
        # complete first in the backward) and the generator's three transposed-convolution weights (11 of 15.4 MB,
        # complete before its first layer and 1-D tower) are reduced as one early bucket each, started when the last of
        # the bucket's weight-gradient kernels has been enqueued, beside the rest of the backward; the remainder (one
        # contiguous range) follows after the backward.  More, smaller buckets cost more than they hide: every
        # collective runs on the collective library's own stream, which shares a hardware queue with one of the
        # engine's four (measured on one rank over RCCL: +0.55 ms per iteration with one bucket per critic layer,
        # +0.34 ms with one bucket per generator layer and the gaps between them reduced one by one).
        self._early = []                      # (store, (lo, hi) range of the flat gradient, pending reduction)
        dn, gn = self.eng.early_d_names(self.eng.nS), self.eng.early_g_names(self.eng.nG)
        self.early_buckets = {}               # trigger name -> (store, lo, hi)
        for store, names, trigger in ((self.eng.D, dn[-1:], dn[-1] if dn else None),
                                      (self.eng.G, gn, gn[-1] if gn else None)):
            if trigger is None:
                continue
            lo = min(store.offsets[n] for n in names)
            hi = max(store.offsets[n] + store.p[n].numel() for n in names)
            # the bucket is usable only if nothing but its own tensors lies inside the range
            inside = [n for n in store.offsets if lo <= store.offsets[n] < hi]
            if sorted(inside) == sorted(names):
                self.early_buckets[trigger] = (store, lo, hi)
        self.dist_active, self.sync_bn, self._drain_known = False, False, None
        if world > 1 or process_group is not None:
            self.attach_process_group(process_group, sync_bn)
        # Philox state: (seed lo, seed hi, offset lo, offset hi); ranks with streams of their own differ in the seed's
        # high word, ranks that share one global stream hold identical states
        self.seed = int(seed)
        hi = 0 if self.global_stream else rank
        self.stream_id = hi
        self.rng_z = torch.tensor([seed & 0x7FFFFFFF, hi, 0, 0], dtype=torch.int32, device=device)
        self.rng_t = torch.tensor([(seed + 1) & 0x7FFFFFFF, hi, 0, 0], dtype=torch.int32, device=device)
        self.rng_aug = torch.tensor([(seed + 3) & 0x7FFFFFFF, hi, 0, 0], dtype=torch.int32, device=device)
        self.rng_mono = torch.tensor([(seed + 4) & 0x7FFFFFFF, hi, 0, 0], dtype=torch.int32, device=device)
        # this rank's rows of a global batch / draw: (first row, rows of the global batch)
        self.row0, self.rows = (rank * self.eng.B, world * self.eng.B) if self.global_stream else (0, self.eng.B)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=device)
        self.pool_normalized = False
        self.label_noise: Optional[float] = None     # enable_augment
        self.mono_bounds = None
        self.aug_given = None                        # tests: (ey, (n, cols, u)) used instead of the device draws
        self._pool_gen = None
        self.pool_s: Optional[torch.Tensor] = None
        self.pool_p: Optional[torch.Tensor] = None
        self.buf_s: Optional[torch.Tensor] = None    # replay buffer of generated alignments (enable_fake_buffer)
        self.graph = None
        self.tape = None
        self.cap_stream = None       # capture(): the stream the step graph is recorded on
        # generator-forwards-done mark of the batched iteration body (created here: nothing is created while a
        # tape or a graph is being recorded)
        self.gen_done = self._new_event()
        self.use_graph = False       # step(): capture on the first call, replay afterwards
        self.use_tape = False        # step(): record a launch tape on the first call, replay it afterwards
        self.iterations = 0
        # weights change only through the engine's Adam steps here, which mark prepared weight copies stale;
        # anything else that writes parameters in place must call weights_changed()
        if hasattr(ops, "static_weights"):
            ops.static_weights = True

    def _new_event(self):
        """An event of the library on the GPU (created up front, reused: see pgan_hip.Event), None on the host."""
        if self.dev.type != "cuda":
            return None
        import pgan_hip
        return pgan_hip.Event()

    def _cur(self) -> int:
        import pgan_hip
        return pgan_hip._stream()

    def attach_process_group(self, process_group=None, sync_bn: Optional[bool] = None) -> None:
        """Switch on the data-parallel reductions over `process_group` (None: the default group once it exists).  A
        trainer may be built before torch.distributed is initialised and attached afterwards: its streams then get
        their hardware queues before the collective library creates streams of its own.
        sync_bn: generator BatchNorm over the global batch (default; equals a single-device run at the global batch
        size) or over each rank's own samples (False: no collectives inside the generator)."""
        self.pg, self.dist_active = process_group, True
        self.tape = None
        self.eng.grad_scale = 1.0 / self.world      # the gradient buffers hold sums over the ranks; Adam reads them scaled
        # Only RCCL reduces device memory in place, enqueued on the calling stream.  With any other backend (gloo: a
        # rehearsal of the data-parallel path on a box with one GPU) the trainer stages each collective through host
        # memory itself -- drain the device, copy out, reduce on the host, copy back -- instead of handing device
        # tensors to the backend: gloo's own device staging (copies on streams of its own, completed from a helper
        # thread) was seen to stall two ranks that share one GPU, both between the engine's four busy streams and at
        # the end of a long-lived test process.
        # Decided at the first collective: the trainer may be attached before torch.distributed is initialised.
        self._drain_known = None
        # collectives run on "the current stream": a launch tape replays them under the stream they were issued on
        self.eng.grad_ready = lambda name: pgan_tape.call(self._early_reduce, name, on_current_stream=True)
        self.eng.gen_grad_ready = lambda name: pgan_tape.call(self._early_reduce_g, name, on_current_stream=True)
        self.sync_bn = True if sync_bn is None else bool(sync_bn)
        if self.sync_bn:
            self.eng.bn_sync = lambda part: pgan_tape.call(self._bn_allreduce, part, on_current_stream=True)
            self.eng.bn_world = self.world
        else:
            self.eng.bn_sync, self.eng.bn_world = None, 1

    def weights_changed(self) -> None:
        """Tell the op backend that parameters were written outside the engine's optimizer steps."""
        if hasattr(self.ops, "invalidate_weight_planes"):
            self.ops.invalidate_weight_planes(reset=True)
        self.graph = None
        self.tape = None

    def set_pool(self, sites: torch.Tensor, pos: torch.Tensor, normalize: bool = False) -> None:
        """Real alignments resident on the device: sites [P,1,64,64], positions [P,1,64], fp32.
        normalize applies (x - 0.5) / 0.5 once here (src/train_wgan_v2.py:453,519-521)."""
        assert sites.shape[0] == pos.shape[0] and sites.shape[0] >= self.eng.B
        self.pool_s = sites.to(self.dev, torch.float32).contiguous().clone()
        self.pool_p = pos.to(self.dev, torch.float32).contiguous().clone()
        self.graph = self.tape = None          # both hold the old pool's addresses
        self.pool_normalized = bool(normalize)
        if normalize:
            self.ops.affine(self.pool_s, 2.0, -1.0, self.pool_s)
            self.ops.affine(self.pool_p, 2.0, -1.0, self.pool_p)
        self.cursor.zero_()

    def set_pool_packed(self, bits, pos, normalize: bool = False) -> None:
        """set_pool from packed alignments (bits [P, 512] uint8, positions [P, 64] float32, see alignment_slab):
        768 bytes per alignment cross to the GPU through pinned double buffers and are expanded there."""
        import alignment_slab
        assert bits.shape[0] == pos.shape[0] and bits.shape[0] >= self.eng.B
        self.pool_s, self.pool_p = alignment_slab.to_device(bits, pos, self.dev, normalize)
        self.graph = self.tape = None
        self.pool_normalized = bool(normalize)
        self.cursor.zero_()

    def shuffle_pool(self, rows: bool = False) -> None:
        """Start of an epoch: new random order of the pool (the loader's on_epoch_end shuffle,
        src/train_wgan_v2.py:239-242) and, with rows=True, an independent permutation of the 64
        individuals of every alignment (np.random.shuffle(s), :257-258).  Device-side, outside
        the captured step."""
        n = self.pool_s.shape[0]
        # a generator of the trainer's own, seeded like its device streams: ranks that split one run hold the whole pool
        # and shuffle it the same way, and a run's sample order is a function of its seed alone
        if self._pool_gen is None:
            self._pool_gen = torch.Generator(device=self.dev)
            self._pool_gen.manual_seed(self.seed + 5 + 1000003 * self.stream_id)
        gen = self._pool_gen
        perm = torch.randperm(n, device=self.dev, generator=gen)
        self.pool_s.copy_(self.pool_s[perm])
        self.pool_p.copy_(self.pool_p[perm])
        if rows:
            S = self.pool_s.shape[-1]
            order = torch.rand(n, S, device=self.dev, generator=gen).argsort(dim=1)
            idx = order.view(n, 1, S, 1).expand(n, 1, S, S)
            self.pool_s.copy_(torch.gather(self.pool_s, 2, idx))
        self.cursor.zero_()

    def enable_fake_buffer(self, size: int) -> None:
        """Replay buffer of generated alignments (FakeBuffer, src/train_wgan_v2.py:98-117): every critic step adds its
        fresh fakes to a bounded FIFO resident in HBM and trains on a batch drawn from it without replacement.  Append,
        draw (device Philox) and gather are kernels of the iteration like the others, so the buffered step runs on the
        launch tape / step graph too.  buf_picks (int32[B] on the device, optional) replaces the random draws with given
        logical indices (0 = oldest): a test drives it beside the reference's host deque."""
        B = self.eng.B
        assert size >= B
        S = self.eng.data
        self.buf_s = torch.zeros(size, 1, S, S, device=self.dev)
        self.buf_p = torch.zeros(size, 1, S, device=self.dev)
        self.buf_state = torch.zeros(4, dtype=torch.int32, device=self.dev)     # head, count, error flag, spare
        self.buf_slots = torch.zeros(B, dtype=torch.int32, device=self.dev)
        seed = int(self.rng_z[0]) + 2
        self.rng_buf = torch.tensor([seed & 0x7FFFFFFF, self.rank, 0, 0], dtype=torch.int32, device=self.dev)
        self.buf_picks = None
        self.graph = self.tape = None

    def enable_augment(self, label_noise: Optional[float] = None, mono_switch=None) -> None:
        """Per-batch augmentations of the real alignments, drawn afresh for every critic batch on the device inside the
        recorded step.  label_noise: --label_smooth with this --label_noise (sites (1 - ey) + noise ey, ey ~ U(0, noise);
        DataGeneratorDisk.get_batch, src/train_wgan_v2.py:265-268).  mono_switch: (lower, upper) bounds of
        switch_monomorphic (:129-160; the reference's main passes 0, 10 at :523-524), applied after the smoothing and the
        pool's normalisation, as the reference orders them."""
        self.label_noise = None if label_noise is None else float(label_noise)
        self.mono_bounds = None if mono_switch is None else (int(mono_switch[0]), int(mono_switch[1]))
        self.graph = self.tape = None

    def _take_real(self) -> None:
        """The next real batch into the discriminator's input slots (this rank's rows of it), augmented; and the critic
        step's interpolation draws."""
        e, o, B = self.eng, self.ops, self.eng.B
        S = e.data
        slc = {} if not self.global_stream else {"offset": self.row0, "advance": self.rows}
        o.take_batch(self.pool_s, self.pool_p, e.X0[:B], e.P0[:B], self.cursor, **slc)
        given = self.aug_given or (None, None)
        if self.label_noise is not None:
            o.label_smooth(e.X0[:B], self.label_noise, self.label_noise, self.pool_normalized, self.rng_aug, ey=given[0],
                           skip=self.row0 * S * S, total=self.rows * S * S)
        if self.mono_bounds is not None:
            o.mono_switch(e.X0[:B], e.P0[:B], self.mono_bounds[0], self.mono_bounds[1], self.rng_mono, given=given[1],
                          first=self.row0, total=self.rows)
        if e.loss == "gp":
            self._draw(e.tbuf, self.rng_t, False, 1)

    def _draw(self, out: torch.Tensor, state: torch.Tensor, normal: bool, per_row: int) -> None:
        """out = this rank's rows of a draw of per_row values for every row of the (global) batch."""
        if self.global_stream:
            self.ops.philox(out, state, normal, skip=self.row0 * per_row, total=self.rows * per_row)
        else:
            self.ops.philox(out, state, normal)

    def _draw_z(self, Z: torch.Tensor) -> None:
        """Latent draws of one or several consecutive generator passes (Z: [passes B, latent]) in the reference's order:
        one draw per pass over the global batch."""
        B = self.eng.B
        if not self.global_stream:
            self.ops.philox(Z, self.rng_z, True)      # one long draw equals the passes' separate draws
            return
        for i in range(Z.shape[0] // B):
            self._draw(Z[i * B:(i + 1) * B], self.rng_z, True, self.eng.latent)

    def _hand_fakes(self, src_s: torch.Tensor, src_p: torch.Tensor) -> None:
        """This critic step's fake batch into the discriminator's input slots: the fresh fakes themselves, or (replay
        buffer) the fresh fakes appended to the buffer and a batch drawn from it."""
        e, o, B = self.eng, self.ops, self.eng.B
        if self.buf_s is not None:
            o.fakebuf_step(self.buf_s, self.buf_p, self.buf_state, self.buf_slots, src_s, src_p, e.X0[B:2 * B],
                           e.P0[B:2 * B], self.rng_buf, self.buf_picks)
        else:
            o.affine(src_s, 1.0, 0.0, e.X0[B:2 * B])      # copies
            o.affine(src_p, 1.0, 0.0, e.P0[B:2 * B])

    def step(self) -> None:
        """One iteration by the fastest available route."""
        if self.tape is not None:
            self.replay_tape()
        elif self.use_tape:
            self.record_tape()
        elif self.graph is not None:
            self.replay()
        elif self.use_graph and not self.dist_active:
            self.capture()           # runs this iteration eagerly, then records the graph for the next ones
        else:
            self.iteration()

    def _reduce(self, flat: torch.Tensor) -> None:
        if self.dist_active:
            pgan_tape.call(self._allreduce, flat, on_current_stream=True)

    @property
    def _drain(self) -> bool:
        """Collectives staged through host memory: every backend but RCCL on device tensors."""
        if self._drain_known is None:
            import torch.distributed as dist
            if not dist.is_initialized():
                raise RuntimeError("a data-parallel trainer needs torch.distributed initialised before its first step")
            self._drain_known = self.dev.type == "cuda" and dist.get_backend(self.pg) != "nccl"
        return self._drain_known

    def _sum_over_ranks(self, t: torch.Tensor, async_op: bool = False):
        """all-reduce(sum) of a device tensor in place: on the device over RCCL, through host memory otherwise."""
        import torch.distributed as dist
        if self._drain:
            torch.cuda.synchronize()
            host = t.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.pg)
            t.copy_(host)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)

    def _bn_allreduce(self, part: torch.Tensor) -> None:
        self._sum_over_ranks(part)

    def _early_reduce(self, name: str) -> None:
        """Called when gradient `name` has been enqueued: starts the reduction of the bucket it completes."""
        b = self.early_buckets.get(name)
        if b is not None:
            store, lo, hi = b
            side = getattr(self.eng.side, "stream", None)      # None on the host (CPU op backend)
            if self._drain or side is None:
                work = self._sum_over_ranks(store.gflat[lo:hi])
            else:
                # A blocking collective is enqueued on the caller's current stream; a non-blocking one goes through the
                # process group's own stream, whose hand-over costs ~30 us of idle queue per call.  The 1-D towers'
                # stream is idle by now and is joined at the end of the pass, so the bucket is reduced there, beside
                # the rest of the backward pass.
                self.eng.side.ev_fork.edge(self._cur(), side.cuda_stream)
                with torch.cuda.stream(side):
                    work = self._sum_over_ranks(store.gflat[lo:hi])
            self._early.append((store, (lo, hi), work))

    _early_reduce_g = _early_reduce

    def _allreduce(self, flat: torch.Tensor) -> None:
        """Sum `flat` (a store's whole gradient buffer) over the ranks; ranges whose reduction was already started by
        _early_reduce are only waited for.  The division by the number of ranks happens inside the Adam step that
        follows (Engine.grad_scale): no separate pass over the buffer."""
        if not self.dist_active:
            return
        store = self.eng.D if flat.data_ptr() == self.eng.D.gflat.data_ptr() else self.eng.G
        mine = [(rng, w) for st, rng, w in self._early if st is store]
        if flat.data_ptr() == store.gflat.data_ptr() and mine:
            done = sorted(rng for rng, _ in mine)
            lo = 0
            for a, b in done + [(flat.numel(), flat.numel())]:
                if a > lo:
                    self._sum_over_ranks(flat[lo:a])
                lo = b
            for _, work in mine:
                if work is not None:
                    work.wait()
            self._early = [e for e in self._early if e[0] is not store]
        else:
            self._sum_over_ranks(flat)

    def _iteration_body(self) -> None:
        if self.batched_generator:
            return self._iteration_body_batched()
        if self.sync_bn:
            return self._iteration_body_ahead()
        return self._iteration_body_one_ahead()

    def _iteration_body_batched(self) -> None:
        """The generator's weights do not change during the critic steps.  The first critic step's fakes come from a
        forward of their own; all the others (critic steps 2.. and the generator step's own forward, whose activations
        stay in the last slice of the workspace for its backward) are ONE forward over critic_iter x batch samples
        -- one launch per layer, every pass with its own BatchNorm statistics, running statistics stepped in pass
        order -- on the generator stream beside the first critic step.  z is drawn in the reference's order (the
        device generator's stream of one long draw equals that of the separate draws)."""
        e, o = self.eng, self.ops
        B, gs, n = e.B, e.gside, self.critic_iter
        m = n + 1
        gs.fork()
        with gs:
            self._draw_z(e.Zw[:B])
            e.gen_forward_chunks(0, 1)

        def step_inputs(i):
            """Critic step i's real batch, interpolation draws and fakes into the discriminator's input slots."""
            self._take_real()
            if i == 0:
                gs.join()
            elif i == 1 and self.gen_done:
                self.gen_done.wait(self._cur())
            self._hand_fakes(e.fake_s[i * B:(i + 1) * B], e.fake_p[i * B:(i + 1) * B])

        # A critic step ends in a serial run of small launches (gradient reductions, Adam, weight preparation) and used to
        # begin with another (batch gather, cursor, draws, fake copies): six launches that depend on nothing the optimizer
        # writes.  They are enqueued on the 1-D towers' stream right after the previous step's kernels have been joined,
        # beside that step's Adam update, and joined before the next step starts.
        # (PGAN_INPUTS_AHEAD=1.  Same-box A/B with the current kernels: 6.99 ms per iteration with it, 6.96 without, and 6.89
        # without it and with one weight-gradient pass per stage -- the default now; it had won 0.06 ms before.)
        ahead = os.environ.get("PGAN_INPUTS_AHEAD", "0") != "0"
        step_inputs(0)
        for i in range(n):
            if i == 0:
                gs.fork()
                with gs:
                    self._draw_z(e.Zw[B:m * B])
                    e.gen_forward_chunks(1, m)
                    if self.gen_done:
                        self.gen_done.record(gs.raw)
            e.critic_grads_resident(run_generator=False)
            if ahead and i + 1 < n:
                e.side.fork()
                with e.side:
                    step_inputs(i + 1)
            self._reduce(e.D.gflat)
            e.adam_d()
            if i + 1 < n:
                if ahead:
                    e.side.join()
                else:
                    step_inputs(i + 1)
        gs.join()
        e.generator_grads_resident(forward_done=True)
        self._reduce(e.G.gflat)
        e.adam_g()

    def _iteration_body_ahead(self) -> None:
        """Variant for BatchNorm statistics over several ranks: every generator forward then holds 8 small
        collectives, a latency chain of about a critic step's length.  All of the iteration's generator forwards
        (one per critic step, then the generator step's own) are therefore enqueued up front on their stream and run
        as far ahead as they can; each critic step waits for its own fakes only.  z is drawn in the reference's order
        and the running statistics see the forwards in that order."""
        e, o = self.eng, self.ops
        B, gs, n = e.B, e.gside, self.critic_iter
        if getattr(self, "fa_s", None) is None or len(self.fa_s) != n:
            S = e.data
            self.fa_s = [torch.zeros(B, 1, S, S, device=self.dev) for _ in range(n)]
            self.fa_p = [torch.zeros(B, 1, S, device=self.dev) for _ in range(n)]
            self.fa_done = [self._new_event() for _ in range(n)]
        gs.fork()
        with gs:
            for i in range(n):
                self._draw_z(e.Z)
                e.gen_forward_resident(self.fa_s[i], self.fa_p[i])
                if self.fa_done[i] is not None:
                    self.fa_done[i].record(gs.raw)
            self._draw_z(e.Z)
            e.gen_forward_resident(e.gXs, e.gXp)       # its activations stay in the workspace for the generator step
        for i in range(n):
            self._take_real()
            if self.fa_done[i] is not None:       # this step's fakes are complete
                self.fa_done[i].wait(self._cur())
            self._hand_fakes(self.fa_s[i], self.fa_p[i])
            e.critic_grads_resident(run_generator=False)
            self._reduce(e.D.gflat)
            e.adam_d()
        gs.join()
        e.generator_grads_resident(forward_done=True)
        self._reduce(e.G.gflat)
        e.adam_g()

    def _iteration_body_one_ahead(self) -> None:
        """The generator's weights do not change during the critic steps, so every generator forward
        (a chain of small launches) is made one step ahead on its own stream, beside the previous
        critic step's large kernels: fakes alternate between two buffers, and the generator step's own
        forward runs beside the last critic step.  z is still drawn in the reference's order."""
        e, o = self.eng, self.ops
        B, gs = e.B, e.gside
        if getattr(self, "fk_s", None) is None:
            S = e.data
            self.fk_s = [torch.zeros(B, 1, S, S, device=self.dev) for _ in range(2)]
            self.fk_p = [torch.zeros(B, 1, S, device=self.dev) for _ in range(2)]
        gs.fork()
        with gs:
            self._draw_z(e.Z)
            e.gen_forward_resident(self.fk_s[0], self.fk_p[0])
        for i in range(self.critic_iter):
            self._take_real()
            gs.join()                                   # this step's fakes are complete
            self._hand_fakes(self.fk_s[i & 1], self.fk_p[i & 1])
            gs.fork()
            with gs:                                    # next forward, beside this critic step
                self._draw_z(e.Z)
                if i + 1 < self.critic_iter:
                    e.gen_forward_resident(self.fk_s[(i + 1) & 1], self.fk_p[(i + 1) & 1])
                else:
                    e.gen_forward_resident(e.gXs, e.gXp)
            e.critic_grads_resident(run_generator=False)
            self._reduce(e.D.gflat)
            e.adam_d()
        gs.join()
        e.generator_grads_resident(forward_done=True)
        self._reduce(e.G.gflat)
        e.adam_g()

    def iteration(self) -> None:
        """One full iteration, launched eagerly on the current stream."""
        self._iteration_body()
        self.iterations += 1

    def capture(self) -> None:
        """Run ONE iteration eagerly (it counts as a training iteration), then record the same launches into a hipGraph
        for replay().  The eager pass runs on the stream that is captured afterwards, with the same side streams, so every
        lazily created object of an iteration (per-stream scratch, prepared weight planes, events, variant buffers)
        exists before the capture opens; inside it nothing may allocate (checked: the op backend refuses, and the
        allocator's request count must not move).  The capture goes through the library's own entry points
        (pgan_step_graph_begin / _end), not through torch's graph machinery, and every cross-stream edge inside it is a
        record / wait on an event of the library that existed before (checked: pgan_events_created must not move)."""
        if self.dist_active:
            raise RuntimeError("graph capture is used on the single-GPU path only")
        from pgan_hip import StepGraph
        if self.cap_stream is None:
            self.cap_stream = torch.cuda.Stream(device=self.dev)
            self.ops.reserve_stream(self.cap_stream.cuda_stream)
        cap, cur = self.cap_stream, torch.cuda.current_stream()
        torch.cuda.synchronize()
        self.ops.invalidate_weight_planes()
        cap.wait_stream(cur)
        with torch.cuda.stream(cap):
            self._iteration_body()
        cur.wait_stream(cap)
        torch.cuda.synchronize()
        self.iterations += 1
        tracked, dstep, gstep = self.eng.g_batches_tracked, self.eng.D.step, self.eng.G.step
        self.ops.invalidate_weight_planes()      # so that the recorded iteration starts by preparing them
        with torch.cuda.stream(cap):
            self.ops.settle_weight_planes()      # ... and holds no reset of the running weight maxima
        import pgan_hip
        requests = torch.cuda.memory_stats(self.dev).get("allocation.all.allocated", 0)
        events = pgan_hip.events_created()
        graph = StepGraph()
        self.ops.capturing = True
        try:
            with torch.cuda.stream(cap):
                with graph:
                    self._iteration_body()
        finally:
            self.ops.capturing = False
            # capture records launches without running them: undo the host-side counters it bumped
            self.eng.g_batches_tracked, self.eng.D.step, self.eng.G.step = tracked, dstep, gstep
        if torch.cuda.memory_stats(self.dev).get("allocation.all.allocated", 0) != requests:
            raise RuntimeError("device memory was requested while the step graph was being captured")
        if pgan_hip.events_created() != events:
            raise RuntimeError("an event was created while the step graph was being captured")
        self.graph = graph

    def record_tape(self) -> None:
        """Run ONE iteration eagerly (it counts) while noting every enqueue call with its resolved arguments;
        replay_tape() repeats them from the flat list: the same launches on the same streams as iteration(), with a
        third of its host time.  The op backend must be the HIP library (its entry points feed the tape)."""
        if not getattr(self.ops, "feeds_tape", False):
            raise RuntimeError("launch tapes need the HIP op backend")
        torch.cuda.synchronize()
        # the recorded iteration must start by preparing the weight planes: planes left fresh by an earlier forward
        # (an evaluation pass, say) would put no preparation call on the tape, and every replay would convolve with
        # the planes of old weights while Adam keeps stepping the fp32 ones
        if hasattr(self.ops, "invalidate_weight_planes"):
            self.ops.invalidate_weight_planes()
            self.ops.settle_weight_planes()      # pending resets of the weight maxima happen now, not in every replay
        pgan_tape.tape = entries = []
        try:
            self._iteration_body()
        finally:
            pgan_tape.tape = None
        self.iterations += 1
        # replay form: runs of library calls as native tapes walked in C (PGAN_NATIVE_TAPE=0: call by call from Python)
        self.tape = pgan_tape.compile(entries) if os.environ.get("PGAN_NATIVE_TAPE", "1") != "0" else entries

    def replay_tape(self) -> None:
        if isinstance(self.tape, list):
            pgan_tape.replay(self.tape)
        else:
            self.tape.replay()
        self.iterations += 1
        self.eng.g_batches_tracked += self.critic_iter + 1
        self.eng.D.step += self.critic_iter
        self.eng.G.step += 1

    def replay(self) -> None:
        self.graph.launch()
        self.iterations += 1
        self.eng.g_batches_tracked += self.critic_iter + 1
        self.eng.D.step += self.critic_iter
        self.eng.G.step += 1

    def losses(self):
        """(critic loss W, mean real, mean fake, penalty, generator loss) of the last step: one D2H read."""
        c = self.eng.closs.tolist()
        return c[0], c[1], c[2], c[3], float(self.eng.gloss)
