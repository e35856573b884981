"""Fused DCWGAN-GP training-step engine: buffer layout + op order on top of the HIP C-ABI.

One Engine owns, on one GPU, the generator and discriminator parameters (flat fp32 buffers
with named views = the reference modules' state_dict entries), both Adam states, and a static
workspace; its methods enqueue hand-written HIP kernels (pgan_hip.HipOps) on the current
stream and never allocate after construction, so a whole iteration can be captured in a
hipGraph.  What it computes, per reference line:

  critic step     /root/reference/src/train_wgan_v2.py:534-612  (gp: layers.py:34-55, div: :60-77)
  generator step  /root/reference/src/train_wgan_v2.py:619-634
  Adam            /root/reference/src/train_wgan_v2.py:489-490

The penalty terms are differentiated analytically instead of through an autograd graph.  With
v = d(penalty)/d(g) held constant, grad_theta(penalty) = grad_theta sum_b JVP_x(D; v)_b, so the
discriminator's 2-D tower is evaluated on four "streams" of samples that live side by side in
every per-layer buffer:

  gp :  [ real | fake | interpolate | tangent of interpolate ]   NS = 3B primal, NP = B penalised
  div:  [ real | fake | tangent of real | tangent of fake ]       NS = 2B primal, NP = 2B penalised

forward (primal), first-order backward (input gradient g of the penalised samples and, in gp
mode, the Wasserstein gradients of real/fake), tangent forward of v, second-order backward of
the penalised samples, then ONE weight-gradient pass per layer over all 4B sample slots.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch

LATENT = 128
DATA = 64
D2_CH = (1, 64, 128, 256, 512)      # discriminator 2-D tower channels (layers.py:441-463)
D2_HW = (64, 32, 16, 8, 4)
D1_CH = (1, 16, 32, 64)             # discriminator 1-D tower (layers.py:470-478)
D1_LEN = (64, 61, 58, 55)
FEAT2 = 512 * 16
FEAT1 = 64 * 55


def net_dims(data: int = DATA, latent: int = LATENT) -> Dict[str, Tuple[int, ...]]:
    """Layer tables of the two networks for `data` x `data` alignments.  64: the reference's networks
    (layers.py:91-140,439-482).  128: the scaled networks of this package (the reference has none: its
    discriminator head is hard-wired to 64 x 64 inputs, layers.py:482) -- one more stride-2 stage at the
    large end of each 2-D tower (discriminator 1->32->64->128->256->512 over 128->...->4, generator
    ...->64->32->1 over ...->32->64->128) and of the generator's 1-D tower; the discriminator's 1-D tower keeps
    its three k=4 convolutions on the longer position vector."""
    if data == 64:
        d2c, g2c = (1, 64, 128, 256, 512), (latent, 512, 256, 128, 64, 1)
        g1c = (latent, latent // 2, latent // 4, latent // 8, latent // 16, 1)
    elif data == 128:
        d2c, g2c = (1, 32, 64, 128, 256, 512), (latent, 512, 256, 128, 64, 32, 1)
        g1c = (latent, latent // 2, latent // 4, latent // 8, latent // 16, latent // 32, 1)
    else:
        raise ValueError(f"alignments of {data} x {data} are not supported (64 or 128)")
    ns = len(d2c) - 1
    d2h = tuple(data >> k for k in range(ns + 1))
    g2h = (1,) + tuple(4 << k for k in range(len(g2c) - 1))
    return dict(d2c=d2c, d2h=d2h, g2c=g2c, g2h=g2h, g1c=g1c, g1l=g2h, d1c=D1_CH,
                d1l=(data, data - 3, data - 6, data - 9), feat2=d2c[-1] * d2h[-1] ** 2, feat1=64 * (data - 9))


def gen_param_shapes(latent: int = LATENT, data: int = DATA) -> Dict[str, Tuple[int, ...]]:
    """state_dict entries of DC_Generator_Leaky_{Sig,Tan} (layers.py:91-140), parameters only."""
    dm = net_dims(data, latent)
    s: Dict[str, Tuple[int, ...]] = {}
    ch2, ch1 = dm["g2c"], dm["g1c"]
    n = len(ch2) - 1
    for i in range(n):
        s[f"twod_conv_layers.{3 * i}.weight"] = (ch2[i], ch2[i + 1], 4, 4)
        if i < n - 1:
            s[f"twod_conv_layers.{3 * i + 1}.weight"] = (ch2[i + 1],)
            s[f"twod_conv_layers.{3 * i + 1}.bias"] = (ch2[i + 1],)
    for i in range(n):
        s[f"oned_conv_layers.{3 * i}.weight"] = (ch1[i], ch1[i + 1], 4)
        if i < n - 1:
            s[f"oned_conv_layers.{3 * i + 1}.weight"] = (ch1[i + 1],)
            s[f"oned_conv_layers.{3 * i + 1}.bias"] = (ch1[i + 1],)
    return s


def gen_buffer_shapes(latent: int = LATENT, data: int = DATA) -> Dict[str, Tuple[int, ...]]:
    dm = net_dims(data, latent)
    s: Dict[str, Tuple[int, ...]] = {}
    for tower, chs in (("twod_conv_layers", dm["g2c"][1:-1]), ("oned_conv_layers", dm["g1c"][1:-1])):
        for i, c in enumerate(chs):
            s[f"{tower}.{3 * i + 1}.running_mean"] = (c,)
            s[f"{tower}.{3 * i + 1}.running_var"] = (c,)
    return s


def disc_param_shapes(data: int = DATA) -> Dict[str, Tuple[int, ...]]:
    """state_dict entries of DC_Discriminator (layers.py:439-482)."""
    dm = net_dims(data)
    s: Dict[str, Tuple[int, ...]] = {}
    ch2 = dm["d2c"]
    for i in range(len(ch2) - 1):
        s[f"twod_conv_layers.{3 * i}.weight"] = (ch2[i + 1], ch2[i], 4, 4)
        s[f"twod_conv_layers.{3 * i + 1}.weight"] = (ch2[i + 1],)
        s[f"twod_conv_layers.{3 * i + 1}.bias"] = (ch2[i + 1],)
    for i in range(3):
        s[f"oned_conv_layers.{3 * i}.weight"] = (D1_CH[i + 1], D1_CH[i], 4)
        s[f"oned_conv_layers.{3 * i}.bias"] = (D1_CH[i + 1],)
        s[f"oned_conv_layers.{3 * i + 1}.weight"] = (D1_CH[i + 1],)
        s[f"oned_conv_layers.{3 * i + 1}.bias"] = (D1_CH[i + 1],)
    s["out.weight"] = (1, dm["feat2"] + dm["feat1"])
    s["out.bias"] = (1,)
    return s


class FlatStore:
    """Named fp32 tensors carved out of one flat buffer (256-byte aligned slices) so that Adam,
    gradient zeroing and the data-parallel all-reduce each touch a single contiguous range."""

    def __init__(self, shapes: Dict[str, Tuple[int, ...]], device, with_opt: bool = True):
        self.shapes = dict(shapes)
        self.offsets: Dict[str, int] = {}
        off = 0
        for k, sh in shapes.items():
            n = 1
            for d in sh:
                n *= d
            self.offsets[k] = off
            off += (n + 63) // 64 * 64
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.p = self._views(self.flat)
        if with_opt:
            self.gflat = torch.zeros_like(self.flat)
            self.mflat = torch.zeros_like(self.flat)
            self.vflat = torch.zeros_like(self.flat)
            self.g = self._views(self.gflat)
            self.m = self._views(self.mflat)
            self.v = self._views(self.vflat)
        self.step = 0
        self.on_load = []       # callables run after load(): holders of derived copies of the weights (prepared planes)

    def _views(self, flat):
        out = {}
        for k, sh in self.shapes.items():
            n = 1
            for d in sh:
                n *= d
            out[k] = flat[self.offsets[k]:self.offsets[k] + n].view(sh)
        return out

    def load(self, params: Dict[str, torch.Tensor]) -> None:
        for k, v in self.p.items():
            v.copy_(params[k].to(v.device, torch.float32))
        for fn in self.on_load:
            fn()


class SideStream:
    """Second HIP stream for the small position-vector towers: they are latency-bound chains of tiny
    kernels that depend only on parameters and their own inputs, so they run beside the 2-D towers'
    large convolution kernels (also under graph capture, where fork/join become graph edges).
    Fork, join and mark are edges over three events of the library that exist before anything is enqueued
    (pgan_hip.Event): no edge creates or destroys an object, also not while a capture is open, and a launch tape
    notes them like any other library call.  On the CPU backend every method is a no-op."""

    def __init__(self, device):
        # Default priority on purpose: with streams of different priority classes the step deadlocked on the GPU (a
        # higher-class hardware queue blocked on an event of a lower-class queue kept that queue from being served).
        self.enabled = torch.device(device).type == "cuda"
        if self.enabled:
            import pgan_hip
            # PGAN_SERIAL_STREAMS=1 (debugging aid): every side stream is the caller's stream, so nothing overlaps
            serial = os.environ.get("PGAN_SERIAL_STREAMS", "0") == "1"
            self.stream = torch.cuda.current_stream(device) if serial else torch.cuda.Stream(device=device)
            self.raw = self.stream.cuda_stream
            self._cur = pgan_hip._stream
            self.ev_fork, self.ev_join, self.mark = pgan_hip.Event(), pgan_hip.Event(), pgan_hip.Event()

    def fork(self):
        """The side stream waits for what the current stream holds now."""
        if self.enabled:
            self.ev_fork.edge(self._cur(), self.raw)

    def join(self):
        """The current stream waits for what the side stream holds now."""
        if self.enabled:
            self.ev_join.edge(self.raw, self._cur())

    def set_mark(self):
        """Remember this point of the side stream."""
        if self.enabled:
            self.mark.record(self.raw)

    def wait_mark(self):
        """The current stream waits for the side stream's mark."""
        if self.enabled:
            self.mark.wait(self._cur())

    def __enter__(self):
        if self.enabled:
            self._ctx = torch.cuda.stream(self.stream)
            self._ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.enabled:
            self._ctx.__exit__(*a)
        return False


class Engine:
    def __init__(self, ops, batch: int, device, loss: str = "gp", act: str = "sigmoid", lam: float = 10.0,
                 latent: int = LATENT, lr_d: float = 5e-5, lr_g: float = 5e-5,
                 betas: Tuple[float, float] = (0.5, 0.999), eps: float = 1e-8,
                 g_store: Optional["FlatStore"] = None, g_buffers: Optional["FlatStore"] = None,
                 d_store: Optional["FlatStore"] = None, generator_only: bool = False, data: int = DATA):
        """generator_only: no discriminator workspace (an engine that only runs generator forwards / backwards, e.g.
        the 1000-sample plotting forward of the trainer)."""
        if loss not in ("gp", "div"):
            raise ValueError(f"loss must be 'gp' or 'div', got {loss!r}")
        if act not in ("sigmoid", "tanh"):
            raise ValueError(f"act must be 'sigmoid' or 'tanh', got {act!r}")
        self.ops, self.B, self.dev = ops, int(batch), device
        self.data = int(data)
        dm = self.dims = net_dims(self.data, int(latent))
        self.d2c, self.d2h, self.nS = dm["d2c"], dm["d2h"], len(dm["d2c"]) - 1
        self.nG = len(dm["g2c"]) - 2                      # BatchNorm stages of each generator tower
        self.feat2, self.feat1 = dm["feat2"], dm["feat1"]
        self.loss, self.act, self.lam, self.latent = loss, act, float(lam), int(latent)
        self.act_id = 1 if act == "sigmoid" else 2
        self.lr_d, self.lr_g, self.betas, self.eps = lr_d, lr_g, betas, eps
        B = self.B
        self.NS = 3 * B if loss == "gp" else 2 * B      # primal sample slots
        self.NP = B if loss == "gp" else 2 * B          # penalised samples (tangent slots)
        self.NT = self.NS + self.NP                     # = 4B
        f = dict(dtype=torch.float32, device=device)

        # parameter stores may be handed in (the nn.Module front-ends own theirs) or created here
        # Layout of the flat buffers: the large 2-D convolution weights sit at the end.  Their gradients are complete
        # early in a backward pass (last stage first) and a data-parallel run reduces each on its own as soon as it
        # is; everything else is then one contiguous range reduced after the backward.
        def tail_last(shapes, names):
            for n in names:
                shapes[n] = shapes.pop(n)
            return shapes
        if d_store is None:
            d_store = FlatStore(tail_last(disc_param_shapes(self.data), self.early_d_names(self.nS)), device)
        self.D = d_store
        self.G = g_store if g_store is not None else FlatStore(
            tail_last(gen_param_shapes(latent, self.data), self.early_g_names(self.nG)), device)
        if g_buffers is not None:
            self.Gbuf = g_buffers
        else:
            self.Gbuf = FlatStore(gen_buffer_shapes(latent, self.data), device, with_opt=False)
            for k, v in self.Gbuf.p.items():
                if k.endswith("running_var"):
                    v.fill_(1.0)
        # Prepared copies of convolution weights are cached by the op backend under the weights' address: a new engine's
        # buffers can sit where a freed engine's did, and load() rewrites them in place -- both make cached copies stale.
        stale = getattr(ops, "invalidate_weight_planes", None)
        if stale is not None:
            for st in (self.D, self.G):
                stale(st.flat, reset=True)
                st.on_load.append(lambda st=st: stale(st.flat, reset=True))
        self.g_batches_tracked = 0   # every BatchNorm of the generator sees the same count
        # data parallel with batch statistics over all ranks' samples: bn_sync(partial sums, float64) adds them over
        # the ranks in place on the current stream, bn_world = number of ranks; None: statistics of this rank's batch
        self.bn_sync = None
        self.bn_world = 1
        # factor Adam applies to the gradients as it reads them: 1 / ranks when the flat gradient buffers hold sums over
        # the ranks of a data-parallel run
        self.grad_scale = 1.0
        self.two_chains = os.environ.get("PGAN_TWO_CHAINS", "1") != "0"
        # PGAN_SPLIT_WGRAD=1: weight gradients of the Wasserstein samples right after their backward, the penalised samples'
        # share added later (two half-size passes per stage instead of one).  It won 0.16 ms per iteration when the
        # weight-gradient kernel was slower; with the current kernels one pass over all slots is 0.03-0.07 ms faster
        # (same-box A/B, 6.99 -> 6.91 / 6.92 ms), so one pass is the default.
        self.split_wgrad = os.environ.get("PGAN_SPLIT_WGRAD", "0") != "0"
        # loss / norm-gradient / head-gradient launches of a critic step beside its last weight-gradient passes
        # (PGAN_TAIL_BESIDE_WGRAD=0: after them, for comparison)
        self.tail_beside_wgrad = os.environ.get("PGAN_TAIL_BESIDE_WGRAD", "1") != "0"
        self._bn_parts = {}
        # The caller's stream and these three are the first streams a fresh process creates, one hardware queue each
        # (HIP serves a process's streams from 4 queues); streams created later (a collective library's) share them.
        self._zeroed = set()      # (address, elements) of convolution outputs cleared ahead of their launch in this step
        self.side = SideStream(device)
        self.wside = SideStream(device)     # weight-gradient passes of the critic step, beside the backward chain
        self.gside = SideStream(device)     # generator forwards made one critic step ahead (FusedTrainer)
        if hasattr(ops, "reserve_stream"):  # per-stream norm scratch exists before anything runs there
            for s in (self.side, self.wside, self.gside):
                if s.enabled:
                    ops.reserve_stream(s.stream.cuda_stream)
        # called with a parameter name right after that discriminator weight gradient has been enqueued (on
        # the stream it was enqueued on): lets a data-parallel driver start reducing it during the backward
        self.grad_ready = None
        # the same for the generator's 2-D convolution weight gradients (generator step's backward)
        self.gen_grad_ready = None

        NT = self.NT
        self.generator_only = bool(generator_only)
        NPl = self.NP
        if generator_only:
            NT = B = NPl = 0        # zero-sized discriminator buffers
        # ---- discriminator workspace: index k = 1..nS is the k-th conv stage, 0 the input ----
        S, nS, d2c, d2h = self.data, self.nS, self.d2c, self.d2h
        stage = lambda n: [None] + [torch.zeros(n, d2c[k], d2h[k], d2h[k], **f) for k in range(1, nS + 1)]
        self.X0 = torch.zeros(NT, 1, S, S, **f)          # sites: primal slots then v (tangent input)
        self.P0 = torch.zeros(2 * B, 1, S, **f)          # positions of real | fake
        self.dA = stage(NT)
        self.dH = stage(NT)
        self.dH[0] = self.X0
        self.dU = stage(NT)
        self.dJ = stage(NT)
        # largest magnitudes (8 per sample, written by the GroupNorm ops) of the activations dH[k] and the adjoints
        # dJ[k]: the scaling bounds of the convolutions that read them
        self.dHmax = [None] + [torch.zeros(NT, 8, **f) for _ in range(nS)]
        self.dJmax = [None] + [torch.zeros(NT, 8, **f) for _ in range(nS)]
        self.dStat = [None] + [torch.zeros(NT, 2, **f) for _ in range(nS)]    # mu, rstd
        self.dStatB = [None] + [torch.zeros(NT, 2, **f) for _ in range(nS)]   # mean(w), mean(w xhat)
        self.dStatT = [None] + [torch.zeros(NT, 2, **f) for _ in range(nS)]   # mean(t), mean(xhat t)
        # per-sample gamma | beta partial sums side by side, so one row reduction yields both gradients
        # (columns laid out like the [gamma | beta] stretch of the flat gradient buffer, whose slices start on multiples
        # of 64 floats: for a stage of fewer channels there is a gap between the two, which stays zero)
        gap = [None] + [self.D.offsets[f"twod_conv_layers.{3 * (k - 1) + 1}.bias"]
                        - self.D.offsets[f"twod_conv_layers.{3 * (k - 1) + 1}.weight"] for k in range(1, nS + 1)]
        self.dPGB = [None] + [torch.zeros(NT, gap[k] + d2c[k], **f) for k in range(1, nS + 1)]
        self.dPG = [None] + [self.dPGB[k][:, :d2c[k]] for k in range(1, nS + 1)]
        self.dPB = [None] + [self.dPGB[k][:, gap[k]:] for k in range(1, nS + 1)]
        self.G0 = torch.zeros(max(NPl, B), 1, S, S, **f)   # input gradient of penalised samples / gen step
        self.d1A = [torch.zeros(2 * B, D1_CH[i + 1], dm["d1l"][i + 1], **f) for i in range(3)]
        self.F1 = torch.zeros(2 * B, self.feat1, **f)
        self.dP0 = torch.zeros(B, 1, S, **f)             # d loss / d positions (generator step)
        self.score = torch.zeros(2 * B, **f)
        self.coef = torch.zeros(NT, **f)                 # per-slot seed / head-gradient weights
        self.coef_head = torch.zeros(NT, **f)
        self.coef_top = torch.zeros(NPl, **f)        # div: Wasserstein seed of the penalised samples
        self.HB4 = torch.zeros(NPl, d2c[nS], d2h[nS], d2h[nS], **f)  # div: primal adjoint entering the top stage
        self.pen = torch.zeros(1, **f)
        self.closs = torch.zeros(4, **f)                 # W, mean real, mean fake, penalty
        self.gloss = torch.zeros(1, **f)
        self.tbuf = torch.zeros(B, **f)
        self.coef_gen = torch.full((B,), -1.0 / max(B, 1), **f)
        self.ones2B = torch.ones(2 * B, 1, **f)
        B = self.B
        if not generator_only:
            self._init_coefs()

        # ---- generator workspace: index i = 1..nG is the i-th BatchNorm stage ----
        g2c, g2h, g1c, g1l, nG = dm["g2c"], dm["g2h"], dm["g1c"], dm["g1l"], self.nG
        self.g2c, self.g2h, self.g1c, self.g1l = g2c, g2h, g1c, g1l
        g2 = lambda: [None] + [torch.zeros(B, g2c[i], g2h[i], g2h[i], **f) for i in range(1, nG + 1)]
        g1 = lambda: [None] + [torch.zeros(B, g1c[i], g1l[i], **f) for i in range(1, nG + 1)]
        self.Z = torch.zeros(B, latent, **f)
        self.gA, self.gH = g2(), g2()                    # conv outputs; after BN+LReLU
        self.gStat = [None] + [torch.zeros(2, g2c[i], **f) for i in range(1, nG + 1)]
        self.gdH, self.gdA = g2(), g2()
        # per-channel largest magnitudes of gH[i] / gdA[i] (written by the BatchNorm ops), bounds for the convolutions
        self.gHmax = [None] + [torch.zeros(g2c[i], **f) for i in range(1, nG + 1)]
        self.gdAmax = [None] + [torch.zeros(g2c[i], **f) for i in range(1, nG + 1)]
        self.gXs = torch.zeros(B, 1, S, S, **f)          # generator-step output (sites)
        self.gXp = torch.zeros(B, 1, S, **f)
        self.gdX = torch.zeros(B, 1, S, S, **f)
        self.hA, self.hH = g1(), g1()
        self.hStat = [None] + [torch.zeros(2, g1c[i], **f) for i in range(1, nG + 1)]
        self.hdH, self.hdA = g1(), g1()
        self.hdX = torch.zeros(B, 1, S, **f)

    @staticmethod
    def early_d_names(nS: int):
        """Discriminator gradients worth reducing on their own: the 2-D stages' weights above the single-channel one."""
        return [f"twod_conv_layers.{3 * (k - 1)}.weight" for k in range(2, nS + 1)]

    @staticmethod
    def early_g_names(nG: int):
        """The generator's transposed-convolution weights below the single-channel output layer."""
        return [f"twod_conv_layers.{3 * i}.weight" for i in range(nG - 1, 0, -1)]

    # ------------------------------------------------------------------------------------------
    def _init_coefs(self, wasserstein: bool = True):
        """Per-slot seeds.  wasserstein=False drops the mean(D(fake)) - mean(D(real)) term so that
        critic_grads* produce the gradient of the penalty alone."""
        B, NS = self.B, self.NS
        ws = 1.0 / B if wasserstein else 0.0
        c = torch.zeros(self.NT)
        ch = torch.zeros(self.NT)
        if self.loss == "gp":
            c[:B], c[B:2 * B], c[2 * B:3 * B] = -ws, ws, 1.0
            ch[:B], ch[B:2 * B], ch[3 * B:] = -ws, ws, 1.0
        else:
            c[:2 * B] = 1.0
            ch[:B], ch[B:2 * B], ch[2 * B:] = -ws, ws, 1.0
            top = torch.cat([torch.full((B,), -ws), torch.full((B,), ws)])
            self.coef_top.copy_(top)
        self.coef.copy_(c)
        self.coef_head.copy_(ch)
        w1 = torch.cat([torch.full((B,), -ws), torch.full((B,), ws)])
        if not hasattr(self, "coef_w"):
            self.coef_w = torch.zeros(2 * B, dtype=torch.float32, device=self.dev)
        self.coef_w.copy_(w1)

    def dconv_w(self, k):
        return self.D.p[f"twod_conv_layers.{3 * (k - 1)}.weight"]

    def dgn(self, k):
        return (self.D.p[f"twod_conv_layers.{3 * (k - 1) + 1}.weight"], self.D.p[f"twod_conv_layers.{3 * (k - 1) + 1}.bias"])

    def _gn_grad_pairs(self):
        """[gamma_k | beta_k] gradient of each stage as one contiguous view of the flat gradient buffer
        (the two tensors are adjacent there: sizes are multiples of the 64-float slice alignment)."""
        if getattr(self, "_gnpairs", None) is None:
            pairs = []
            for k in range(1, self.nS + 1):
                n = f"twod_conv_layers.{3 * (k - 1) + 1}"
                off, c = self.D.offsets[n + ".weight"], self.d2c[k]
                pairs.append(self.D.gflat[off:self.D.offsets[n + ".bias"] + c])
            self._gnpairs = pairs
        return self._gnpairs

    def _reduce_norm_grads(self, r0, r1, r2, r3) -> None:
        """[gamma | beta] gradients of every stage from the per-sample partial sums of rows [r0,r1) and [r2,r3)
        (at most four stages per launch)."""
        pairs = self._gn_grad_pairs()
        for lo in range(0, self.nS, 4):
            self.ops.rows_reduce_multi(self.dPGB[1 + lo:5 + lo], pairs[lo:lo + 4], r0, r1, r2, r3)

    def _d1d_params(self, store):
        cw = [store[f"oned_conv_layers.{3 * i}.weight"] for i in range(3)]
        cb = [store[f"oned_conv_layers.{3 * i}.bias"] for i in range(3)]
        gw = [store[f"oned_conv_layers.{3 * i + 1}.weight"] for i in range(3)]
        gb = [store[f"oned_conv_layers.{3 * i + 1}.bias"] for i in range(3)]
        return cw, cb, gw, gb

    # ------------------------------------------------------------------------------------------
    # generator
    # ------------------------------------------------------------------------------------------
    def _fresh(self) -> None:
        """Entry points call this first: a critic step that raised between preparing and releasing the weight planes
        must not leave later calls reading planes of outdated weights."""
        rel = getattr(self.ops, "release_weights", None)
        if rel is not None:
            rel()
        self._zeroed.clear()      # outputs cleared ahead of their convolutions: only within the step that cleared them

    def gen_forward(self, z: torch.Tensor, out_s: torch.Tensor, out_p: torch.Tensor) -> None:
        """(sites, positions) = G(z), BatchNorm in training mode (layers.py:143-147).  Activations
        stay in the generator workspace for a following gen_backward."""
        self._fresh()
        if z.data_ptr() != self.Z.data_ptr():
            self.Z.copy_(z.reshape(self.B, self.latent))
        self.g_batches_tracked += 1
        self.side.fork()
        with self.side:
            self._gen_forward_1d(out_p)
        self._gen_forward_2d(out_s)
        self.side.join()

    def gen_forward_resident(self, out_s: torch.Tensor, out_p: torch.Tensor) -> None:
        """G(z) for the z already in self.Z, both towers on the current stream."""
        self.g_batches_tracked += 1
        self._gen_forward_1d(out_p)
        self._gen_forward_2d(out_s)

    def _bn_part(self, key, a):
        part = self._bn_parts.get(key)
        if part is None:
            hw = a.numel() // (a.shape[0] * a.shape[1])
            part = torch.zeros(self.ops.bn_partials(a.shape[0], a.shape[1], hw), dtype=torch.float64, device=a.device)
            self._bn_parts[key] = part
        return part

    def _bn_fwd(self, key, a, gamma, beta, rmean, rvar, h, stats, groups: int = 1, amax=None) -> None:
        """BatchNorm + LeakyReLU forward; with bn_sync set, over the global batch of all ranks (the partial sums of
        every layer go through bn_sync, which adds them over the ranks on the current stream).  groups > 1: a holds
        that many forward passes, each with its own statistics."""
        o = self.ops
        if self.bn_sync is None:
            if groups == 1:
                o.bn_fwd(a, gamma, beta, rmean, rvar, h, stats, amax=amax)
            else:
                o.bn_fwd(a, gamma, beta, rmean, rvar, h, stats, groups=groups, amax=amax)
            return
        part = self._bn_parts.get(("f", key, groups))
        if part is None:
            hw = a.numel() // (a.shape[0] * a.shape[1])
            n = o.bn_fwd_partials(a.shape[0] // groups, a.shape[1], hw, groups)
            part = self._bn_parts[("f", key, groups)] = torch.zeros(n, dtype=torch.float64, device=a.device)
        if groups == 1:
            o.bn_fwd_sums(a, part, amax=amax)
            self.bn_sync(part)
            o.bn_fwd_apply(a, gamma, beta, rmean, rvar, h, stats, part, self.bn_world, amax=amax)
        else:
            o.bn_fwd_sums(a, part, groups=groups, amax=amax)
            self.bn_sync(part)
            o.bn_fwd_apply(a, gamma, beta, rmean, rvar, h, stats, part, self.bn_world, groups=groups, amax=amax)

    def _bn_bwd(self, key, dh, h, a, stats, gamma, da, dgamma, dbeta, amax=None) -> None:
        o = self.ops
        if self.bn_sync is None:
            o.bn_bwd(dh, h, a, stats, gamma, da, dgamma, dbeta, amax=amax)
            return
        part = self._bn_part(("b", key), a)
        o.bn_bwd_sums(dh, h, a, stats, part, amax=amax)
        self.bn_sync(part)
        o.bn_bwd_apply(dh, h, a, stats, gamma, da, dgamma, dbeta, part, self.bn_world, amax=amax)

    def _gen_forward_2d(self, out_s: torch.Tensor, bufs=None, groups: int = 1) -> None:
        """2-D tower on (Z, conv outputs, activations, statistics) = bufs (default: the single-pass workspace);
        groups > 1: that many passes stacked along the batch axis."""
        o, G, Gb = self.ops, self.G.p, self.Gbuf.p
        Z, gA, gH, gStat = bufs if bufs is not None else (self.Z, self.gA, self.gH, self.gStat)
        if hasattr(o, "prepare_weights"):       # prepared copies of all stages' weights in one go
            o.prepare_weights([G[f"twod_conv_layers.{3 * i}.weight"] for i in range(1, self.nG + 1)], hold=False)
        o.gemm_nn(Z, G["twod_conv_layers.0.weight"].view(self.latent, -1), gA[1].view(Z.shape[0], -1))
        for i in range(1, self.nG + 1):
            n = f"twod_conv_layers.{3 * (i - 1) + 1}"
            # gHmax[i] always describes the forward made last, whose final pass is what the backward reads
            self._bn_fwd(n, gA[i], G[n + ".weight"], G[n + ".bias"], Gb[n + ".running_mean"],
                         Gb[n + ".running_var"], gH[i], gStat[i], groups, amax=self.gHmax[i])
            w = G[f"twod_conv_layers.{3 * i}.weight"]
            if i < self.nG:
                o.conv_up(gH[i], w, gA[i + 1], bound=self.gHmax[i])
            else:
                o.conv_up(gH[i], w, out_s, self.act_id)

    def _gen_forward_1d(self, out_p: torch.Tensor, bufs=None, groups: int = 1) -> None:
        o, G, Gb = self.ops, self.G.p, self.Gbuf.p
        Z, hA, hH, hStat = bufs if bufs is not None else (self.Z, self.hA, self.hH, self.hStat)
        o.gemm_nn(Z, G["oned_conv_layers.0.weight"].view(self.latent, -1), hA[1].view(Z.shape[0], -1))
        for i in range(1, self.nG + 1):
            n = f"oned_conv_layers.{3 * (i - 1) + 1}"
            self._bn_fwd(n, hA[i], G[n + ".weight"], G[n + ".bias"], Gb[n + ".running_mean"],
                         Gb[n + ".running_var"], hH[i], hStat[i], groups)
            w = G[f"oned_conv_layers.{3 * i}.weight"]
            if i < self.nG:
                o.convt1d_up(hH[i], w, hA[i + 1])
            else:
                o.convt1d_up(hH[i], w, out_p, self.act_id)

    def enable_gen_chunks(self, m: int) -> None:
        """Generator workspace for m forward passes side by side along the batch axis (FusedTrainer: the passes of one
        iteration share their weights, so several of them run as one launch per layer).  The single-pass buffers
        (Z, activations, statistics, gXs / gXp) become the LAST pass's slices, so gen_forward_resident, the backward and
        the generator step keep working on pass m - 1."""
        B, dev = self.B, self.dev
        f = dict(dtype=torch.float32, device=dev)
        self.gen_chunks = m
        self.Zw = torch.zeros(m * B, self.latent, **f)
        R, S = range(1, self.nG + 1), self.data
        self.gAw = [None] + [torch.zeros(m * B, *self.gA[i].shape[1:], **f) for i in R]
        self.gHw = [None] + [torch.zeros(m * B, *self.gH[i].shape[1:], **f) for i in R]
        self.gStatw = [None] + [torch.zeros(m, 2, self.gStat[i].shape[1], **f) for i in R]
        self.hAw = [None] + [torch.zeros(m * B, *self.hA[i].shape[1:], **f) for i in R]
        self.hHw = [None] + [torch.zeros(m * B, *self.hH[i].shape[1:], **f) for i in R]
        self.hStatw = [None] + [torch.zeros(m, 2, self.hStat[i].shape[1], **f) for i in R]
        self.fake_s = torch.zeros(m * B, 1, S, S, **f)
        self.fake_p = torch.zeros(m * B, 1, S, **f)
        lo = (m - 1) * B
        self.Z = self.Zw[lo:]
        for i in R:
            self.gA[i], self.gH[i], self.gStat[i] = self.gAw[i][lo:], self.gHw[i][lo:], self.gStatw[i][m - 1]
            self.hA[i], self.hH[i], self.hStat[i] = self.hAw[i][lo:], self.hHw[i][lo:], self.hStatw[i][m - 1]
        self.gXs, self.gXp = self.fake_s[lo:], self.fake_p[lo:]

    def gen_forward_chunks(self, c0: int, c1: int) -> None:
        """G(z) for passes c0 .. c1-1 at once (z in Zw[c0 B : c1 B]) into fake_s / fake_p[c0 B : c1 B]: one launch per
        layer over (c1 - c0) B samples, every pass with its own BatchNorm statistics; counts as c1 - c0 batches."""
        B, g = self.B, c1 - c0
        lo, hi = c0 * B, c1 * B
        self.g_batches_tracked += g
        sl = lambda bufs: [None] + [b[lo:hi] for b in bufs[1:]]
        st = lambda bufs: [None] + [(b[c0] if g == 1 else b[c0:c1]) for b in bufs[1:]]
        self._gen_forward_1d(self.fake_p[lo:hi], (self.Zw[lo:hi], sl(self.hAw), sl(self.hHw), st(self.hStatw)), g)
        self._gen_forward_2d(self.fake_s[lo:hi], (self.Zw[lo:hi], sl(self.gAw), sl(self.gHw), st(self.gStatw)), g)

    def gen_backward(self, xs, xp, dxs, dxp) -> None:
        """Gradients of every generator parameter given d loss/d sites and d loss/d positions of the
        batch produced by the preceding gen_forward (into self.G.g, overwritten)."""
        self.side.fork()
        with self.side:
            self._gen_backward_1d(xp, dxp)
        self._gen_backward_2d(xs, dxs)
        self.side.join()

    def _gen_backward_2d(self, xs, dxs) -> None:
        o, G, dG = self.ops, self.G.p, self.G.g
        o.act_bwd(dxs, xs, self.gdX, self.act_id)
        up, up_max = self.gdX, None           # the single-channel top stage takes no bound
        for i in range(self.nG, 0, -1):
            w = G[f"twod_conv_layers.{3 * i}.weight"]
            self.wside.fork()
            with self.wside:      # the weight gradient beside the input-gradient chain
                o.conv_wgrad(self.gH[i], up, dG[f"twod_conv_layers.{3 * i}.weight"], bound_s=self.gHmax[i],
                             bound_l=up_max)
                if self.gen_grad_ready is not None:
                    self.gen_grad_ready(f"twod_conv_layers.{3 * i}.weight")
            self._conv_down(up, w, self.gdH[i], bound=up_max)
            n = f"twod_conv_layers.{3 * (i - 1) + 1}"
            self._bn_bwd(n, self.gdH[i], self.gH[i], self.gA[i], self.gStat[i], G[n + ".weight"], self.gdA[i],
                         dG[n + ".weight"], dG[n + ".bias"], amax=self.gdAmax[i])
            up, up_max = self.gdA[i], self.gdAmax[i]
        o.gemm_tn(self.Z, self.gdA[1].view(self.B, -1), dG["twod_conv_layers.0.weight"].view(self.latent, -1))
        self.wside.join()

    def _gen_backward_1d(self, xp, dxp) -> None:
        o, G, dG = self.ops, self.G.p, self.G.g
        o.act_bwd(dxp, xp, self.hdX, self.act_id)
        up = self.hdX
        for i in range(self.nG, 0, -1):
            w = G[f"oned_conv_layers.{3 * i}.weight"]
            o.convt1d_wgrad(self.hH[i], up, dG[f"oned_conv_layers.{3 * i}.weight"])
            o.convt1d_down(up, w, self.hdH[i])
            n = f"oned_conv_layers.{3 * (i - 1) + 1}"
            self._bn_bwd(n, self.hdH[i], self.hH[i], self.hA[i], self.hStat[i], G[n + ".weight"], self.hdA[i],
                         dG[n + ".weight"], dG[n + ".bias"])
            up = self.hdA[i]
        o.gemm_tn(self.Z, self.hdA[1].view(self.B, -1), dG["oned_conv_layers.0.weight"].view(self.latent, -1))

    # ------------------------------------------------------------------------------------------
    # discriminator pieces
    # ------------------------------------------------------------------------------------------
    def _hmax(self, k: int, lo: int, hi: int):
        """Bounds of dH[k][lo:hi] for the convolutions (stage 0 is the input image: single-channel kernels, no bound)."""
        return self.dHmax[k][lo:hi] if k >= 1 else None

    # Convolution launches with small grids split their reduction over two blocks per tile that ADD into the output, which
    # they otherwise clear first with a launch of their own in the middle of a chain.  The outputs of a step's launches of
    # that kind are cleared together in one launch at the step's start instead (PGAN_ZERO_AHEAD=0: not).
    zero_ahead = os.environ.get("PGAN_ZERO_AHEAD", "1") != "0"

    def _zero_ahead(self, jobs) -> None:
        """jobs: (output tensor, weight) of conv_down launches that follow in this step on streams ordered after this
        point; nothing else may write those outputs before their launch."""
        o = self.ops
        if not (self.zero_ahead and hasattr(o, "zero_multi")):
            return
        todo = []
        for y, w in jobs:
            if y.numel() and o.conv_down_adds(y.shape[0], w.shape[0], w.shape[1], y.shape[2]):
                todo.append(y)
                self._zeroed.add((y.data_ptr(), y.numel()))
        if todo:
            o.zero_multi(todo)

    def _conv_down(self, x, w, y, bound=None) -> None:
        key = (y.data_ptr(), y.numel())
        if key in self._zeroed:
            self._zeroed.discard(key)
            self.ops.conv_down(x, w, y, bound=bound, zeroed=True)
        else:
            self.ops.conv_down(x, w, y, bound=bound)

    def _d_forward2d(self, lo: int, hi: int) -> None:
        o = self.ops
        fused = getattr(o, "conv_down_gn_sums", None)      # first stage: the norm's statistics pass inside the convolution
        for k in range(1, self.nS + 1):
            gw, gb = self.dgn(k)
            if k == 1 and fused is not None:
                pre = fused(self.dH[0][lo:hi], self.dconv_w(1), self.dA[1][lo:hi])
                o.gn_fwd(self.dA[1][lo:hi], gw, gb, self.dH[1][lo:hi], self.dStat[1][lo:hi], amax=self.dHmax[1][lo:hi],
                         **({"presummed": True} if pre else {}))
                continue
            self._conv_down(self.dH[k - 1][lo:hi], self.dconv_w(k), self.dA[k][lo:hi], bound=self._hmax(k - 1, lo, hi))
            o.gn_fwd(self.dA[k][lo:hi], gw, gb, self.dH[k][lo:hi], self.dStat[k][lo:hi], amax=self.dHmax[k][lo:hi])

    def _d_backward1(self, lo: int, hi: int, g_lo: int, g_hi: int, g_out: Optional[torch.Tensor]) -> None:
        """First-order backward of sum_n coef[n] * score[n] through the 2-D tower for slots [lo, hi).
        The input gradient is produced only for slots [g_lo, g_hi) (into g_out)."""
        o = self.ops
        o.head_seed(self.D.p["out.weight"], self.coef_cur[lo:hi], self.dU[self.nS][lo:hi])
        for k in range(self.nS, 0, -1):
            gw, gb = self.dgn(k)      # beta given: the LeakyReLU mask is recomputed from dA, dH is not read
            o.gn_bwd(self.dU[k][lo:hi], self.dH[k][lo:hi], self.dA[k][lo:hi], self.dStat[k][lo:hi], gw,
                     self.dJ[k][lo:hi], self.dStatB[k][lo:hi], self.dPG[k][lo:hi], self.dPB[k][lo:hi],
                     amax=self.dJmax[k][lo:hi], beta=gb)
            if k > 1:
                o.conv_up(self.dJ[k][lo:hi], self.dconv_w(k), self.dU[k - 1][lo:hi], bound=self.dJmax[k][lo:hi])
            elif g_out is not None:
                o.conv_up(self.dJ[1][g_lo:g_hi], self.dconv_w(1), g_out)

    def disc_forward(self, xs: torch.Tensor, xp: torch.Tensor, score: torch.Tensor) -> None:
        """score = D(xs, xp) for up to 2B samples (layers.py:487-493)."""
        n = xs.shape[0]
        o = self.ops
        self._fresh()
        self.X0[:n].copy_(xs)
        self.P0[:n].copy_(xp)
        self._d_forward2d(0, n)
        cw, cb, gw, gb = self._d1d_params(self.D.p)
        o.d1d_fwd(self.P0[:n], cw, cb, gw, gb, [a[:n] for a in self.d1A], self.F1[:n])
        o.head_fwd(self.dH[self.nS][:n], self.F1[:n], self.D.p["out.weight"], self.D.p["out.bias"], score)

    def disc_backward(self, n: int, coef: torch.Tensor, dxs: Optional[torch.Tensor] = None,
                      dxp: Optional[torch.Tensor] = None, want_params: bool = True) -> None:
        """Plain backward of sum_i coef[i] * score[i] for the n samples of the preceding disc_forward:
        parameter gradients into self.D.g (overwritten) and, when given, the input gradients."""
        o, Dp, Dg = self.ops, self.D.p, self.D.g
        cw, cb, gw, gb = self._d1d_params(Dp)
        self.coef_cur = coef
        self._d_backward1(0, n, 0, n, dxs)
        a1 = [a[:n] for a in self.d1A]
        FEAT2 = self.feat2
        wt = Dp["out.weight"].view(-1)[FEAT2:]
        if want_params:
            for k in range(1, self.nS + 1):
                o.conv_wgrad(self.dJ[k][:n], self.dH[k - 1][:n], Dg[f"twod_conv_layers.{3 * (k - 1)}.weight"],
                             bound_s=self.dJmax[k][:n], bound_l=self._hmax(k - 1, 0, n))
            self._reduce_norm_grads(0, n, 0, 0)
            o.head_wgrad(self.dH[self.nS][:n], coef, Dg["out.weight"].view(-1)[:FEAT2])
            o.head_wgrad(self.F1[:n], coef, Dg["out.weight"].view(-1)[FEAT2:])
            o.head_wgrad(self.ones2B[:n], coef, Dg["out.bias"])
            dcw, dcb, dgw, dgb = self._d1d_params(Dg)
            o.d1d_bwd(self.P0[:n], cw, cb, gw, gb, a1, coef, wt, dcw, dcb, dgw, dgb, dxp)
        elif dxp is not None:
            o.d1d_bwd(self.P0[:n], cw, cb, gw, gb, a1, coef, wt, None, None, None, None, dxp)

    # ------------------------------------------------------------------------------------------
    # critic step
    # ------------------------------------------------------------------------------------------
    def critic_grads(self, real_s: torch.Tensor, real_p: torch.Tensor, z: torch.Tensor,
                     t: Optional[torch.Tensor] = None) -> None:
        """Loss W = mean D(fake) - mean D(real) + penalty and its gradient w.r.t. every
        discriminator parameter (into self.D.g, overwritten); self.closs = [W, mean real, mean
        fake, penalty].  `t` (B values in [0,1)) is the gp interpolation draw."""
        B = self.B
        self.X0[:B].copy_(real_s)
        self.P0[:B].copy_(real_p)
        self.Z.copy_(z.reshape(B, self.latent))
        if self.loss == "gp":
            self.tbuf.copy_(t.reshape(B))
        self.critic_grads_resident()

    def critic_grads_given_fake(self, real_s, real_p, fake_s, fake_p, t=None) -> None:
        """critic_grads for a fake batch that was generated elsewhere (the generator is not run)."""
        B = self.B
        self.X0[:B].copy_(real_s)
        self.P0[:B].copy_(real_p)
        self.X0[B:2 * B].copy_(fake_s)
        self.P0[B:2 * B].copy_(fake_p)
        if self.loss == "gp":
            self.tbuf.copy_(t.reshape(B))
        self.critic_grads_resident(run_generator=False)

    def critic_grads_resident(self, run_generator: bool = True) -> None:
        """critic_grads on inputs already placed in the workspace: real batch in X0[:B] / P0[:B],
        z in Z, t in tbuf (and, with run_generator=False, the fake batch in X0[B:2B] / P0[B:2B])."""
        o, B, NS, NP, NT = self.ops, self.B, self.NS, self.NP, self.NT
        Dp, Dg = self.D.p, self.D.g
        nS, FEAT2 = self.nS, self.feat2
        self._fresh()
        cw, cb, gw, gb = self._d1d_params(Dp)
        dcw, dcb, dgw, dgb = self._d1d_params(Dg)
        if run_generator:
            self.g_batches_tracked += 1
        # side stream: the whole position-vector chain (generator 1-D tower, discriminator 1-D tower
        # forward and backward); it depends on the main stream only through what is ready here
        self.side.fork()
        with self.side:
            if run_generator:
                self._gen_forward_1d(self.P0[B:2 * B])
            o.d1d_fwd(self.P0, cw, cb, gw, gb, self.d1A, self.F1)
            self.side.set_mark()
            o.d1d_bwd(self.P0, cw, cb, gw, gb, self.d1A, self.coef_w, Dp["out.weight"].view(-1)[FEAT2:],
                      dcw, dcb, dgw, dgb, None)
            o.head_wgrad(self.F1, self.coef_w, Dg["out.weight"].view(-1)[FEAT2:])
            o.head_wgrad(self.ones2B, self.coef_w, Dg["out.bias"])
        if run_generator:
            self._gen_forward_2d(self.X0[B:2 * B])
        self.coef_cur = self.coef
        two_chains = self.loss == "gp" and self.two_chains
        if two_chains:
            # gp: the Wasserstein term (real | fake: forward + first-order backward) and the penalty term
            # (interpolates: forward, backward, tangent forward, second backward) are independent chains of equal
            # length; they run side by side, each one's convolutions beside the other's small kernels.  The
            # Wasserstein chain goes first on the weight-gradient stream.
            if hasattr(o, "prepare_weights"):
                o.prepare_weights([self.dconv_w(k) for k in range(1, nS + 1)])
            self._zero_ahead([(self.dA[k][lo:hi], self.dconv_w(k)) for k in range(2, nS + 1)
                              for lo, hi in ((0, 2 * B), (2 * B, NS), (NS, NT))])
            self.wside.fork()
            with self.wside:
                self._d_forward2d(0, 2 * B)
                self.side.wait_mark()
                o.head_fwd(self.dH[nS][:2 * B], self.F1, Dp["out.weight"], Dp["out.bias"], self.score)
                self._d_backward1(0, 2 * B, 0, 0, None)
                self.wside.set_mark()       # scores, adjoints and per-sample norm gradients of the Wasserstein samples
                for k in (range(nS, 0, -1) if self.split_wgrad else ()):
                    o.conv_wgrad(self.dJ[k][:2 * B], self.dH[k - 1][:2 * B], Dg[f"twod_conv_layers.{3 * (k - 1)}.weight"],
                                 bound_s=self.dJmax[k][:2 * B], bound_l=self._hmax(k - 1, 0, 2 * B))
            o.interpolate(self.tbuf, self.X0[:B], self.X0[B:2 * B], self.X0[2 * B:3 * B])
            self._d_forward2d(2 * B, NS)
            self._d_backward1(2 * B, NS, NS - NP, NS, self.G0[:NP])
        else:
            if self.loss == "gp":
                o.interpolate(self.tbuf, self.X0[:B], self.X0[B:2 * B], self.X0[2 * B:3 * B])
            self._zero_ahead([(self.dA[k][lo:hi], self.dconv_w(k)) for k in range(2, nS + 1) for lo, hi in ((0, NS), (NS, NT))])
            # primal forward
            self._d_forward2d(0, NS)
            self.side.wait_mark()
            o.head_fwd(self.dH[nS][:2 * B], self.F1, Dp["out.weight"], Dp["out.bias"], self.score)
            # first-order backward (+ input gradient of the penalised samples)
            self._d_backward1(0, NS, NS - NP, NS, self.G0[:NP])
        # penalty value and v = d penalty / d g, written as the tangent input
        if self.loss == "gp":
            o.gp_penalty(self.G0[:NP], self.lam, self.X0[NS:NT], self.pen)
        else:
            o.div_penalty(self.G0[:NP], B, self.X0[NS:NT], self.pen)
        # tangent forward of v through the 2-D tower (primal state = the penalised samples' slots)
        pl, ph = NS - NP, NS
        for k in range(1, nS + 1):
            self._conv_down(self.dH[k - 1][NS:NT], self.dconv_w(k), self.dA[k][NS:NT], bound=self._hmax(k - 1, NS, NT))
            gw_k, gb_k = self.dgn(k)
            o.gn_jvp(self.dA[k][NS:NT], self.dH[k][pl:ph], self.dA[k][pl:ph], self.dStat[k][pl:ph], gw_k,
                     self.dH[k][NS:NT], self.dStatT[k][NS:NT], amax=self.dHmax[k][NS:NT], beta=gb_k)
        # second-order backward of the penalised samples
        hbar = None
        if self.loss == "div":
            o.head_seed(Dp["out.weight"], self.coef_top, self.HB4)
            hbar = self.HB4
        for k in range(nS, 0, -1):
            gw_k, gb_k = self.dgn(k)
            o.gn_bwd2(hbar, self.dU[k][pl:ph], self.dA[k][NS:NT], self.dH[k][pl:ph], self.dA[k][pl:ph],
                      self.dStat[k][pl:ph], self.dStatB[k][pl:ph], self.dStatT[k][NS:NT], gw_k,
                      self.dJ[k][NS:NT], self.dPG[k][NS:NT], self.dPB[k][NS:NT], amax=self.dJmax[k][NS:NT], beta=gb_k)
            # stage k's adjoints are complete over all 4B slots: its weight-gradient pass (one per stage,
            # over all slots) runs on its own stream beside the rest of the chain
            self.wside.fork()
            with self.wside:
                dw = Dg[f"twod_conv_layers.{3 * (k - 1)}.weight"]
                if two_chains and self.split_wgrad:
                    o.conv_wgrad(self.dJ[k][2 * B:], self.dH[k - 1][2 * B:], dw, 0, NP, accumulate=True,
                                 bound_s=self.dJmax[k][2 * B:], bound_l=self._hmax(k - 1, 2 * B, NT))
                else:
                    o.conv_wgrad(self.dJ[k], self.dH[k - 1], dw, NS - NP, NP, bound_s=self.dJmax[k],
                                 bound_l=self._hmax(k - 1, 0, NT))
                if self.grad_ready is not None:
                    self.grad_ready(f"twod_conv_layers.{3 * (k - 1)}.weight")
            if k > 1:
                o.conv_up(self.dJ[k][NS:NT], self.dconv_w(k), self.dU[k - 1][NS:NT], bound=self.dJmax[k][NS:NT])
                hbar = self.dU[k - 1][NS:NT]
        # The loss value, the norm layers' parameter gradients and the head's weight gradient need the chains' adjoints
        # but none of the convolution weight gradients: they run here beside the last weight-gradient passes (the
        # single-channel stage's comes last) instead of after them.
        if not self.tail_beside_wgrad:
            self.wside.join()
        elif two_chains:
            self.wside.wait_mark()
        o.critic_loss(self.score, B, self.pen, self.closs)
        r = (0, 2 * B, NS, NT) if self.loss == "gp" else (NS, NT, 0, 0)
        self._reduce_norm_grads(*r)
        o.head_wgrad(self.dH[nS], self.coef_head, Dg["out.weight"].view(-1)[:FEAT2])
        self.wside.join()
        if two_chains and hasattr(o, "release_weights"):
            o.release_weights()
        self.side.join()

    def critic_step(self, real_s, real_p, z, t=None) -> None:
        self.critic_grads(real_s, real_p, z, t)
        self.adam_d()

    # ------------------------------------------------------------------------------------------
    # generator step
    # ------------------------------------------------------------------------------------------
    def generator_grads(self, z: torch.Tensor) -> None:
        """loss = -mean D(G(z)) and its gradient w.r.t. the generator parameters (self.G.g);
        self.gloss holds the loss (src/train_wgan_v2.py:619-633)."""
        self.Z.copy_(z.reshape(self.B, self.latent))
        self.generator_grads_resident()

    def generator_grads_resident(self, forward_done: bool = False) -> None:
        """generator_grads with z already in self.Z (forward_done: G(z) is already in gXs / gXp with its
        activations in the generator workspace, from gen_forward_resident(self.gXs, self.gXp))."""
        o, B = self.ops, self.B
        Dp = self.D.p
        nS, FEAT2 = self.nS, self.feat2
        self._fresh()
        cw, cb, gw, gb = self._d1d_params(Dp)
        a1 = [a[:B] for a in self.d1A]
        if not forward_done:
            self.g_batches_tracked += 1
        self.side.fork()
        with self.side:   # positions: G 1-D forward, D 1-D forward, D 1-D input gradient, G 1-D backward
            if not forward_done:
                self._gen_forward_1d(self.gXp)
            o.affine(self.gXp, 1.0, 0.0, self.P0[:B])                      # copy
            o.d1d_fwd(self.P0[:B], cw, cb, gw, gb, a1, self.F1[:B])
            self.side.set_mark()
            o.d1d_bwd(self.P0[:B], cw, cb, gw, gb, a1, self.coef_gen, Dp["out.weight"].view(-1)[FEAT2:],
                      None, None, None, None, self.dP0)
            self._gen_backward_1d(self.gXp, self.dP0)
        if not forward_done:
            self._gen_forward_2d(self.gXs)
        o.affine(self.gXs, 1.0, 0.0, self.X0[:B])                          # copy
        if hasattr(o, "prepare_weights"):
            o.prepare_weights([self.dconv_w(k) for k in range(1, nS + 1)], hold=False)
        self._zero_ahead([(self.dA[k][:B], self.dconv_w(k)) for k in range(2, nS + 1)]
                         + [(self.gdH[i], self.G.p[f"twod_conv_layers.{3 * i}.weight"]) for i in range(self.nG, 0, -1)])
        self._d_forward2d(0, B)
        self.side.wait_mark()
        o.head_fwd(self.dH[nS][:B], self.F1[:B], Dp["out.weight"], Dp["out.bias"], self.score[:B])
        o.neg_mean(self.score[:B], self.gloss)
        self.coef_cur = self.coef_gen
        self._d_backward1(0, B, 0, B, self.G0[:B])
        self._gen_backward_2d(self.gXs, self.G0[:B])
        self.side.join()

    def generator_step(self, z) -> None:
        self.generator_grads(z)
        self.adam_g()

    # ------------------------------------------------------------------------------------------
    def use_device_step_counters(self) -> None:
        """Keep Adam's step counts in device memory (needed before capturing steps in a graph)."""
        self.D.step_dev = torch.full((1,), self.D.step, dtype=torch.int32, device=self.dev)
        self.G.step_dev = torch.full((1,), self.G.step, dtype=torch.int32, device=self.dev)

    def _adam(self, st: FlatStore, lr: float) -> None:
        st.step += 1
        scale = {} if self.grad_scale == 1.0 else {"grad_scale": self.grad_scale}
        if getattr(st, "step_dev", None) is not None:
            self.ops.adam(st.flat, st.gflat, st.mflat, st.vflat, 0, lr, self.betas[0], self.betas[1], self.eps,
                          st.step_dev, **scale)
        else:
            self.ops.adam(st.flat, st.gflat, st.mflat, st.vflat, st.step, lr, self.betas[0], self.betas[1], self.eps,
                          **scale)
        stale = getattr(self.ops, "invalidate_weight_planes", None)
        if stale is not None:     # prepared copies of this store's convolution weights are rebuilt at their next use
            stale(st.flat)

    def adam_d(self) -> None:
        self._adam(self.D, self.lr_d)

    def adam_g(self) -> None:
        self._adam(self.G, self.lr_g)

    def train_iteration(self, real_batches: List[Tuple[torch.Tensor, torch.Tensor]], zs: List[torch.Tensor],
                        ts: Optional[List[torch.Tensor]] = None) -> None:
        """critic_iter critic steps then one generator step (src/train_wgan_v2.py:516-634)."""
        for h, (rs, rp) in enumerate(real_batches):
            self.critic_step(rs, rp, zs[h], None if ts is None else ts[h])
        self.generator_step(zs[len(real_batches)])
