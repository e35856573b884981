"""CPU oracle for the DCWGAN-GP training step (TEST INFRASTRUCTURE ONLY).

This file is a checker, not a product path.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  It restates, with stock
torch.nn.functional ops + autograd on the CPU, the arithmetic of the reference's
hot path:

  generator      /root/reference/src/layers.py:82-147  (sigmoid), :223-288 (tanh)
  discriminator  /root/reference/src/layers.py:429-493
  gp penalty     /root/reference/src/layers.py:34-55   (= src/train_wgan_v2.py:41-62)
  div penalty    /root/reference/src/layers.py:60-77   (= src/train_wgan_v2.py:353-370)
  critic step    /root/reference/src/train_wgan_v2.py:534-612
  generator step /root/reference/src/train_wgan_v2.py:619-634
  Adam           torch.optim.Adam as configured at src/train_wgan_v2.py:489-490

Parity status: PINNED.  oracle/make_golden.py imports the reference's own
layers.py in the build container, drives it with the same portable-PRNG weights
and inputs, and writes tests/golden/*.npz; tests/test_oracle_golden.py checks
this restatement against those vectors.

Parameters live in plain dicts keyed by the reference's state_dict names so the
same dict can be loaded into the reference modules, this oracle and the HIP path.
Everything is dtype-generic: run it in float64 to arbitrate between two fp32
implementations.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
ParamDict = Dict[str, Tensor]

LATENT = 128
DATA = 64
SLOPE = 0.1
BN_EPS = 1e-5
GN_EPS = 1e-5
BN_MOMENTUM = 0.1

# ----------------------------------------------------------------------------------------
# parameter tables (names/shapes == the reference modules' state_dict, layers.py:91-140,439-482)
# ----------------------------------------------------------------------------------------

def net_dims(data: int = DATA, latent: int = LATENT):
    """Channel tables for `data` x `data` alignments.  64: the reference's modules.  128: the scaled networks the
    build defines (PARITY UNPINNED for that size: the reference has no 128 x 128 model -- its discriminator head is
    Linear(11712, 1), layers.py:482, and its generator always emits 64 x 64) -- one more stride-2 stage at the
    large end of each 2-D tower and of the generator's 1-D tower, same layer types and hyper-parameters."""
    if data == 64:
        return dict(d2c=[1, 64, 128, 256, 512], g2c=[latent, 512, 256, 128, 64, 1],
                    g1c=[latent, latent // 2, latent // 4, latent // 8, latent // 16, 1])
    if data == 128:
        return dict(d2c=[1, 32, 64, 128, 256, 512], g2c=[latent, 512, 256, 128, 64, 32, 1],
                    g1c=[latent, latent // 2, latent // 4, latent // 8, latent // 16, latent // 32, 1])
    raise ValueError(data)


def gen_param_shapes(latent: int = LATENT, data: int = DATA) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    dm = net_dims(data, latent)
    for tower, ch, k in (("twod_conv_layers", dm["g2c"], (4, 4)), ("oned_conv_layers", dm["g1c"], (4,))):
        n = len(ch) - 1
        for i in range(n):
            s[f"{tower}.{3 * i}.weight"] = (ch[i], ch[i + 1]) + k
            if i < n - 1:
                s[f"{tower}.{3 * i + 1}.weight"] = (ch[i + 1],)
                s[f"{tower}.{3 * i + 1}.bias"] = (ch[i + 1],)
    # the reference module registers the 2-D tower's layers first, then the 1-D tower's, layer by layer
    order = [k for k in s if k.startswith("twod")] + [k for k in s if k.startswith("oned")]
    return {k: s[k] for k in order}


def gen_buffer_shapes(latent: int = LATENT, data: int = DATA) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    dm = net_dims(data, latent)
    for tower, chs in (("twod_conv_layers", dm["g2c"][1:-1]), ("oned_conv_layers", dm["g1c"][1:-1])):
        for i, c in enumerate(chs):
            s[f"{tower}.{3 * i + 1}.running_mean"] = (c,)
            s[f"{tower}.{3 * i + 1}.running_var"] = (c,)
            s[f"{tower}.{3 * i + 1}.num_batches_tracked"] = ()
    return s


def disc_param_shapes(data: int = DATA) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    ch2 = net_dims(data)["d2c"]
    for i in range(len(ch2) - 1):
        s[f"twod_conv_layers.{3 * i}.weight"] = (ch2[i + 1], ch2[i], 4, 4)
        s[f"twod_conv_layers.{3 * i + 1}.weight"] = (ch2[i + 1],)
        s[f"twod_conv_layers.{3 * i + 1}.bias"] = (ch2[i + 1],)
    ch1 = [1, 16, 32, 64]
    for i in range(3):
        s[f"oned_conv_layers.{3 * i}.weight"] = (ch1[i + 1], ch1[i], 4)
        s[f"oned_conv_layers.{3 * i}.bias"] = (ch1[i + 1],)
        s[f"oned_conv_layers.{3 * i + 1}.weight"] = (ch1[i + 1],)
        s[f"oned_conv_layers.{3 * i + 1}.bias"] = (ch1[i + 1],)
    s["out.weight"] = (1, ch2[-1] * 16 + 64 * (data - 9))
    s["out.bias"] = (1,)
    return s


def _n_layers(p: ParamDict, tower: str) -> int:
    """Convolution layers of a tower, from the parameter names present."""
    n = 0
    while f"{tower}.{3 * n}.weight" in p:
        n += 1
    return n


def _is_norm_scale(name: str, shape: Tuple[int, ...]) -> bool:
    return len(shape) == 1 and name.endswith(".weight") and not name.startswith("out.")


def make_params(shapes: Dict[str, Tuple[int, ...]], seed: int, dtype=torch.float32) -> ParamDict:
    """Portable-PRNG parameters (numpy PCG64 -> fp32): conv/linear weights uniform in
    +-1/sqrt(fan_in)-ish bounds, norm scales near 1, norm/conv biases small and non-zero so
    every term of every gradient formula is exercised."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out: ParamDict = {}
    for name, shape in shapes.items():
        if _is_norm_scale(name, shape):
            v = 1.0 + 0.2 * (rng.random(shape, dtype=np.float32) - 0.5)
        elif len(shape) == 1:
            v = 0.2 * (rng.random(shape, dtype=np.float32) - 0.5)
        else:
            fan = int(np.prod(shape[1:])) if len(shape) > 2 else shape[1]
            bound = 1.0 / math.sqrt(fan)
            v = (2.0 * rng.random(shape, dtype=np.float32) - 1.0) * bound
        out[name] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dtype)
    return out


def make_gen_buffers(dtype=torch.float32, latent: int = LATENT, data: int = DATA) -> ParamDict:
    out: ParamDict = {}
    for name, shape in gen_buffer_shapes(latent, data).items():
        if name.endswith("running_mean"):
            out[name] = torch.zeros(shape, dtype=dtype)
        elif name.endswith("running_var"):
            out[name] = torch.ones(shape, dtype=dtype)
        else:
            out[name] = torch.zeros((), dtype=torch.int64)
    return out


def synthetic_alignments(n: int, seed: int, size: int = DATA) -> Tuple[Tensor, Tensor]:
    """ms-style synthetic batch: every column polymorphic with a derived-allele count drawn
    from the neutral SFS (prob ~ 1/i, i = 1..size-1) placed on random individuals; positions
    sorted U(0,1).  Shapes (n,1,size,size) / (n,1,size), fp32 (layout of
    src/train_wgan_v2.py:270)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    w = 1.0 / np.arange(1, size)
    w /= w.sum()
    counts = rng.choice(np.arange(1, size), size=(n, size), p=w)
    order = np.argsort(rng.random((n, size, size)), axis=1)  # per (n, column) a permutation of individuals
    rank = np.argsort(order, axis=1)
    sites = (rank < counts[:, None, :]).astype(np.float32)
    pos = np.sort(rng.random((n, size)), axis=1).astype(np.float32)
    return torch.from_numpy(sites[:, None]), torch.from_numpy(pos[:, None])


# ----------------------------------------------------------------------------------------
# forward passes
# ----------------------------------------------------------------------------------------

def _lrelu(x: Tensor, key: str, masks: Optional[Dict[str, Tensor]]) -> Tensor:
    """LeakyReLU(0.1).  When `masks` holds `key`, the slope pattern (1 or 0.1 per element) is taken
    from there instead of from sign(x): two fp32 implementations can disagree about the sign of a
    pre-activation that is within rounding of 0, which changes no forward value (LeakyReLU is
    continuous) but moves every gradient by O(1e-4); parity tests therefore evaluate the oracle on
    the linear region the implementation under test actually took, and check separately that the
    regions differ only at such ambiguous elements."""
    if masks is not None and key in masks:
        return x * masks[key].to(x.dtype)
    return F.leaky_relu(x, SLOPE)


def gen_forward(p: ParamDict, buf: Optional[ParamDict], z: Tensor, act: str = "sigmoid",
                acts: Optional[Dict[str, Tensor]] = None, masks: Optional[Dict[str, Tensor]] = None
                ) -> Tuple[Tensor, Tensor]:
    """layers.py:143-147.  BatchNorm always in training mode (the trainer never calls .eval(),
    src/train_wgan_v2.py:479,687): batch statistics normalise, running stats are updated with
    momentum 0.1 and the unbiased variance, num_batches_tracked += 1.  `buf` is updated in
    place when given."""
    final = torch.sigmoid if act == "sigmoid" else torch.tanh

    def bn(x, tower, idx):
        name = f"{tower}.{idx}"
        rm = rv = None
        if buf is not None:
            rm, rv = buf[name + ".running_mean"], buf[name + ".running_var"]
            buf[name + ".num_batches_tracked"] += 1
        return F.batch_norm(x, rm, rv, p[name + ".weight"], p[name + ".bias"], True, BN_MOMENTUM, BN_EPS)

    x = z
    n2 = _n_layers(p, "twod_conv_layers")
    for i in range(n2):
        w = p[f"twod_conv_layers.{3 * i}.weight"]
        x = F.conv_transpose2d(x, w, None, stride=1 if i == 0 else 2, padding=0 if i == 0 else 1)
        if acts is not None:
            acts[f"g2d.conv{i}"] = x
        if i < n2 - 1:
            x = _lrelu(bn(x, "twod_conv_layers", 3 * i + 1), f"g2d.{i}", masks)
            if acts is not None:
                acts[f"g2d.act{i}"] = x
    xs = final(x)
    y = z[:, :, :, 0]
    n1 = _n_layers(p, "oned_conv_layers")
    for i in range(n1):
        w = p[f"oned_conv_layers.{3 * i}.weight"]
        y = F.conv_transpose1d(y, w, None, stride=1 if i == 0 else 2, padding=0 if i == 0 else 1)
        if i < n1 - 1:
            y = _lrelu(bn(y, "oned_conv_layers", 3 * i + 1), f"g1d.{i}", masks)
            if acts is not None:
                acts[f"g1d.act{i}"] = y
    xp = final(y)
    return xs, xp


def disc_forward(p: ParamDict, xs: Tensor, xp: Tensor, acts: Optional[Dict[str, Tensor]] = None,
                 masks: Optional[Dict[str, Tensor]] = None, tag: str = "d") -> Tensor:
    """layers.py:487-493: 4x[Conv2d k4 s2 p1 no bias, GroupNorm(1,C), LeakyReLU 0.1], flatten;
    3x[Conv1d k4 bias, GroupNorm(1,C), LeakyReLU], flatten; cat; Linear(11712,1)."""
    x = xs
    for i in range(_n_layers(p, "twod_conv_layers")):
        x = F.conv2d(x, p[f"twod_conv_layers.{3 * i}.weight"], None, stride=2, padding=1)
        if acts is not None:
            acts[f"d2d.conv{i}"] = x
        x = F.group_norm(x, 1, p[f"twod_conv_layers.{3 * i + 1}.weight"], p[f"twod_conv_layers.{3 * i + 1}.bias"], GN_EPS)
        x = _lrelu(x, f"{tag}.{i}", masks)
        if acts is not None:
            acts[f"{tag}2d.act{i}"] = x
    y = xp
    for i in range(3):
        y = F.conv1d(y, p[f"oned_conv_layers.{3 * i}.weight"], p[f"oned_conv_layers.{3 * i}.bias"])
        y = F.group_norm(y, 1, p[f"oned_conv_layers.{3 * i + 1}.weight"], p[f"oned_conv_layers.{3 * i + 1}.bias"], GN_EPS)
        y = F.leaky_relu(y, SLOPE)
        if acts is not None:
            acts[f"{tag}1d.act{i}"] = y
    feat = torch.cat([x.flatten(1), y.flatten(1)], dim=-1)
    return F.linear(feat, p["out.weight"], p["out.bias"])


# ----------------------------------------------------------------------------------------
# penalties (quirks preserved: sites-gradient only; gp is ELEMENTWISE |g|)
# ----------------------------------------------------------------------------------------

def gp_penalty(p: ParamDict, real_s: Tensor, real_p: Tensor, fake_s: Tensor, fake_p: Tensor,
               lam: float, t: Tensor, aux: Optional[Dict[str, Tensor]] = None,
               masks: Optional[Dict[str, Tensor]] = None, acts: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """layers.py:34-55.  `t` is the (B,1,1,1) interpolation draw the reference takes from
    np.random.random (line 37).  The interpolates are detached leaves (lines 44-45); only the
    sites gradient is kept (`[0]`, line 53); `norm(2, dim=1)` runs over the size-1 channel
    axis, i.e. the penalty is mean over B*64*64 of (|g|-1)^2, times lambda (line 54)."""
    xi = (t * real_s + (1 - t) * fake_s).detach().requires_grad_(True)
    t1 = t[:, :, :, 0]
    pi = (t1 * real_p + (1 - t1) * fake_p).detach().requires_grad_(True)
    out = disc_forward(p, xi, pi, acts, masks, "d_gp")
    g = torch.autograd.grad(out, [xi, pi], torch.ones_like(out), create_graph=True, retain_graph=True)[0]
    if aux is not None:
        aux["interp_sites"] = xi.detach()
        aux["grad_sites"] = g.detach()
    return ((g.norm(2, dim=1) - 1) ** 2).mean() * lam


def div_penalty(d_real: Tensor, d_fake: Tensor, real_s: Tensor, real_p: Tensor,
                fake_s: Tensor, fake_p: Tensor) -> Tensor:
    """layers.py:60-77: k=2, p=6; per-sample (sum g^2)^3 of the sites gradient of the real and
    of the fake scores, mean over the batch, times k/2."""
    ones = torch.ones(real_s.size(0), 1, dtype=d_real.dtype)
    gr = torch.autograd.grad(d_real, [real_s, real_p], ones, create_graph=True, retain_graph=True)[0]
    gf = torch.autograd.grad(d_fake, [fake_s, fake_p], ones, create_graph=True, retain_graph=True)[0]
    nr = gr.reshape(gr.size(0), -1).pow(2).sum(1) ** 3
    nf = gf.reshape(gf.size(0), -1).pow(2).sum(1) ** 3
    return torch.mean(nr + nf) * 2 / 2


# ----------------------------------------------------------------------------------------
# losses and one optimiser step
# ----------------------------------------------------------------------------------------

def _leafify(p: ParamDict) -> ParamDict:
    return {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}


def critic_grads(dp: ParamDict, gp_: ParamDict, gbuf: Optional[ParamDict], real_s: Tensor, real_p: Tensor,
                 z: Tensor, loss: str = "gp", lam: float = 10.0, t: Optional[Tensor] = None,
                 act: str = "sigmoid", aux: Optional[Dict[str, Tensor]] = None,
                 masks: Optional[Dict[str, Tensor]] = None, acts: Optional[Dict[str, Tensor]] = None,
                 reference_cost: bool = False) -> Tuple[Tensor, ParamDict]:
    """One critic loss + its gradient w.r.t. every discriminator parameter
    (src/train_wgan_v2.py:534-609).  The generator forward runs in training mode and updates
    `gbuf`; its output is treated as a constant (in gp mode the reference leaves it attached,
    which only fills generator .grad fields that are zeroed before use, lines 538-539,619).
    reference_cost: do that wasted work too (gp: the fakes stay attached and the backward also runs
    through the generator) -- same results, the reference's running time; for CPU timing only."""
    B = real_s.size(0)
    GP = None
    if reference_cost and loss == "gp":
        GP = _leafify(gp_)
        fake_s, fake_p = gen_forward(GP, gbuf, z, act, acts, masks)
    else:
        with torch.no_grad():
            fake_s, fake_p = gen_forward(gp_, gbuf, z, act, acts, masks)
    P = _leafify(dp)
    if loss == "div":
        real_s = real_s.detach().clone().requires_grad_(True)
        real_p = real_p.detach().clone().requires_grad_(True)
        fake_s = fake_s.detach().clone().requires_grad_(True)
        fake_p = fake_p.detach().clone().requires_grad_(True)
    d = disc_forward(P, torch.cat([real_s, fake_s]), torch.cat([real_p, fake_p]), acts, masks, "d_main")
    d_real, d_fake = d[:B], d[B:]
    if loss == "gp":
        pen = gp_penalty(P, real_s, real_p, fake_s, fake_p, lam, t, aux, masks, acts)
    elif loss == "div":
        pen = div_penalty(d_real, d_fake, real_s, real_p, fake_s, fake_p)
    else:
        raise ValueError(loss)
    w = d_fake.mean() - d_real.mean() + pen
    names = list(P.keys())
    extra = [] if GP is None else list(GP.values())
    grads = torch.autograd.grad(w, [P[n] for n in names] + extra, allow_unused=True)[:len(names)]
    g = {n: (torch.zeros_like(P[n]) if gi is None else gi.detach()) for n, gi in zip(names, grads)}
    if aux is not None:
        aux.update(fake_sites=fake_s.detach(), fake_pos=fake_p.detach(), d_real=d_real.detach(),
                   d_fake=d_fake.detach(), penalty=pen.detach())
    return w.detach(), g


def generator_grads(dp: ParamDict, gp_: ParamDict, gbuf: Optional[ParamDict], z: Tensor,
                    act: str = "sigmoid", aux: Optional[Dict[str, Tensor]] = None,
                    masks: Optional[Dict[str, Tensor]] = None, acts: Optional[Dict[str, Tensor]] = None,
                    reference_cost: bool = False) -> Tuple[Tensor, ParamDict]:
    """src/train_wgan_v2.py:619-633: loss = -mean(D(G(z))), gradient w.r.t. generator params.
    reference_cost: the backward also computes the discriminator's parameter gradients, as the reference's
    gen_loss.backward() does before the next disc zero_grad discards them (line 567); for CPU timing only."""
    P = _leafify(gp_)
    xs, xp = gen_forward(P, gbuf, z, act, acts, masks)
    DP = _leafify(dp) if reference_cost else dp
    loss = -disc_forward(DP, xs, xp, acts, masks, "d_main").mean()
    names = list(P.keys())
    extra = list(DP.values()) if reference_cost else []
    grads = torch.autograd.grad(loss, [P[n] for n in names] + extra)[:len(names)]
    if aux is not None:
        aux.update(fake_sites=xs.detach(), fake_pos=xp.detach())
    return loss.detach(), {n: gi.detach() for n, gi in zip(names, grads)}


class AdamState:
    """torch.optim.Adam(lr, betas=(0.5, 0.999), eps=1e-8, weight_decay=0, amsgrad=False)
    restated: m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
    p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""

    def __init__(self, params: ParamDict, lr: float = 5e-5, betas=(0.5, 0.999), eps: float = 1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, betas[0], betas[1], eps
        self.step = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def apply(self, params: ParamDict, grads: ParamDict) -> None:
        self.step += 1
        bc1 = 1.0 - self.b1 ** self.step
        bc2 = 1.0 - self.b2 ** self.step
        for k, p in params.items():
            g = grads[k]
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def kink_margin(acts: Dict[str, Tensor]) -> float:
    """Smallest |LeakyReLU output| relative to that tensor's rms over every recorded activation:
    how far the nearest pre-activation is from the kink at 0, in units of the activation scale."""
    m = float("inf")
    for k, v in acts.items():
        if ".act" not in k:
            continue
        rms = float(v.detach().double().pow(2).mean().sqrt())
        m = min(m, float(v.detach().abs().min()) / max(rms, 1e-300))
    return m


def train_iteration(dp: ParamDict, gp_: ParamDict, gbuf: ParamDict, d_opt: AdamState, g_opt: AdamState,
                    real_batches: List[Tuple[Tensor, Tensor]], zs: List[Tensor], ts: Optional[List[Tensor]],
                    loss: str = "gp", lam: float = 10.0, act: str = "sigmoid", normalize: bool = False,
                    reference_cost: bool = False) -> Tuple[float, float]:
    """One iteration of src/train_wgan_v2.py:516-634: len(real_batches) critic steps, then one
    generator step with zs[-1].  `normalize` applies (x-0.5)/0.5 to sites and positions
    (lines 453,519-521).  Updates dp/gp_/gbuf/optimiser state in place."""
    w = torch.zeros(())
    for h, (rs, rp) in enumerate(real_batches):
        if normalize:
            rs, rp = (rs - 0.5) / 0.5, (rp - 0.5) / 0.5
        w, g = critic_grads(dp, gp_, gbuf, rs, rp, zs[h], loss, lam, None if ts is None else ts[h], act,
                            reference_cost=reference_cost)
        d_opt.apply(dp, g)
    gl, g = generator_grads(dp, gp_, gbuf, zs[len(real_batches)], act, reference_cost=reference_cost)
    g_opt.apply(gp_, g)
    return float(w), float(gl)
