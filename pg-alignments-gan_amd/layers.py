"""Drop-in model surface of PG-Alignments-GAN's DCWGAN-GP on MI355X.

Same class names, constructor arguments, forward signatures and state_dict keys/shapes as the
reference's /root/reference/src/layers.py (DC_Generator_Leaky_Sig :82-147, DC_Generator_Leaky_Tan
:223-288, DC_Discriminator :429-493, gp_gradient_penalty :34-55, div_gradient_penalty :60-77), so
the reference's training loop runs unchanged on these objects: forward returns autograd-attached
tensors, `.backward()` / `torch.optim.Adam` / `torch.save(state_dict)` work as usual.  Underneath,
every forward and backward is the hand-written HIP path (pgan_engine.Engine on pgan_hip.HipOps);
there is no PyTorch-op fallback, and a CPU tensor is rejected.

The penalties are evaluated analytically: their forward computes the penalty AND its parameter
gradients (no autograd graph through the critic), their backward just scales those.

Limits, stated: only the most recent forward of a module can be back-propagated (activations live
in one static workspace per batch size); data_size must be 64 and k_size 4 (the reference's
Linear(11712, 1) fixes the 64x64 input anyway, layers.py:482).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn as nn

from pgan_engine import (Engine, FlatStore, disc_param_shapes, gen_buffer_shapes, gen_param_shapes)

_OPS = {}


def _ops(device):
    from pgan_hip import HipOps
    key = (device.type, device.index)
    if key not in _OPS:
        _OPS[key] = HipOps(device)
    return _OPS[key]


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: this implementation runs on the GPU only (got a {t.device} tensor); "
                           f"move the module and its inputs to a HIP device first")


class _Slot(nn.Module):
    """Holder for the parameters/buffers of one submodule index of the reference's nn.Sequential."""


def _fan_in(shape: Tuple[int, ...]) -> int:
    n = shape[1]
    for d in shape[2:]:
        n *= d
    return n


class _FlatBacked(nn.Module):
    """nn.Module whose parameters are views into one flat fp32 buffer (so the HIP kernels, Adam and
    all-reduce see contiguous storage) registered under the reference's state_dict names."""

    def _build(self, shapes: Dict[str, Tuple[int, ...]], buffers: Dict[str, Tuple[int, ...]]):
        self._shapes, self._bshapes = shapes, buffers
        self._store = FlatStore(shapes, torch.device("cpu"))
        self._bstore = FlatStore(buffers, torch.device("cpu"), with_opt=False) if buffers else None
        self._engines = {}
        self._version = 0
        for name, view in self._store.p.items():
            self._slot(name).register_parameter(name.rsplit(".", 1)[1], nn.Parameter(view))
        if self._bstore is not None:
            for name, view in self._bstore.p.items():
                slot = self._slot(name)
                slot.register_buffer(name.rsplit(".", 1)[1], view)
                if name.endswith("running_var"):
                    view.fill_(1.0)
                    slot.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long))
        self._default_init()

    def _slot(self, name: str) -> nn.Module:
        mod = self
        for part in name.split(".")[:-1]:
            if not hasattr(mod, part):
                mod.add_module(part, _Slot())
            mod = getattr(mod, part)
        return mod

    def _default_init(self):
        """PyTorch's defaults for the reference's layer types: kaiming-uniform(a=sqrt 5) weights and
        U(+-1/sqrt(fan_in)) biases for conv/linear, ones/zeros for the norm layers."""
        with torch.no_grad():
            for name, p in self._store.p.items():
                if p.dim() > 1:
                    nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                elif name.endswith(".weight"):
                    p.fill_(1.0)
                else:
                    wname = name.rsplit(".", 1)[0] + ".weight"
                    w = self._store.p.get(wname)
                    if w is not None and w.dim() > 1:      # conv / linear bias
                        b = 1.0 / math.sqrt(_fan_in(tuple(w.shape)))
                        p.uniform_(-b, b)
                    else:
                        p.zero_()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._reflatten()
        return out

    def _reflatten(self):
        """After .to()/.cuda() each parameter is an independent tensor again: gather them back
        into one flat buffer on the new device and re-point the nn.Parameters at its views."""
        params = dict(self.named_parameters())
        dev = next(iter(params.values())).device
        if self._store.flat.device == dev and all(
                params[n].data_ptr() == v.data_ptr() for n, v in self._store.p.items()):
            return
        store = FlatStore(self._shapes, dev)
        for name, view in store.p.items():
            view.copy_(params[name].data)
            params[name].data = view
        self._store = store
        if self._bstore is not None:
            bufs = dict(self.named_buffers())
            bstore = FlatStore(self._bshapes, dev, with_opt=False)
            for name, view in bstore.p.items():
                view.copy_(bufs[name].data)
                self._slot(name)._buffers[name.rsplit(".", 1)[1]] = view
            self._bstore = bstore
        self._engines = {}

    def _param_list(self):
        return [dict(self.named_parameters())[n] for n in self._shapes]

    def _grad_tuple(self, store_g, scale=None):
        outs = []
        for n in self._shapes:
            g = store_g[n].clone()
            outs.append(g if scale is None else g * scale)
        return tuple(outs)


# --------------------------------------------------------------------------------------------
# generator
# --------------------------------------------------------------------------------------------
class _GenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, z, *params):
        _require_cuda(z, "generator forward")
        B = z.shape[0]
        eng = module._engine(B)
        xs = torch.empty(B, 1, 64, 64, device=z.device)
        xp = torch.empty(B, 1, 64, device=z.device)
        eng.gen_forward(z.detach().contiguous().float(), xs, xp)
        module._count_batch()
        module._version += 1
        ctx.module, ctx.eng, ctx.version = module, eng, module._version
        ctx.save_for_backward(xs, xp)
        return xs, xp

    @staticmethod
    def backward(ctx, dxs, dxp):
        m = ctx.module
        if ctx.version != m._version:
            raise RuntimeError("only the most recent forward of this generator can be back-propagated")
        xs, xp = ctx.saved_tensors
        dxs = torch.zeros_like(xs) if dxs is None else dxs.contiguous()
        dxp = torch.zeros_like(xp) if dxp is None else dxp.contiguous()
        ctx.eng.gen_backward(xs, xp, dxs, dxp)
        return (None, None) + m._grad_tuple(ctx.eng.G.g)


class _GeneratorBase(_FlatBacked):
    _act = "sigmoid"

    def __init__(self, data_size, latent_size, negative_slope, num_channels, k_size):
        super().__init__()
        if data_size != 64 or k_size != 4 or num_channels != 1 or abs(negative_slope - 0.1) > 1e-12:
            raise ValueError("the HIP path implements the trained configuration: data_size=64, k_size=4, "
                             "num_channels=1, negative_slope=0.1 (src/train_wgan_v2.py:479)")
        if latent_size % 16:
            raise ValueError("latent_size must be a multiple of 16")
        self.negative_slope, self.data_size, self.latent_size = negative_slope, data_size, latent_size
        self._build(gen_param_shapes(latent_size), gen_buffer_shapes(latent_size))

    def _engine(self, B):
        if B not in self._engines:
            dev = self._store.flat.device
            self._engines[B] = Engine(_ops(dev), B, dev, loss="gp", act=self._act, latent=self.latent_size,
                                      g_store=self._store, g_buffers=self._bstore)
        return self._engines[B]

    def _count_batch(self):
        for mod in self.modules():
            if "num_batches_tracked" in getattr(mod, "_buffers", {}):
                mod._buffers["num_batches_tracked"] += 1

    def forward(self, x):
        return _GenFn.apply(self, x, *self._param_list())


class DC_Generator_Leaky_Sig(_GeneratorBase):
    """layers.py:82-147: z (B, latent, 1, 1) -> sites (B,1,64,64) in (0,1), positions (B,1,64)."""
    _act = "sigmoid"


class DC_Generator_Leaky_Tan(_GeneratorBase):
    """layers.py:223-288: as above with tanh outputs."""
    _act = "tanh"


# --------------------------------------------------------------------------------------------
# discriminator
# --------------------------------------------------------------------------------------------
class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, xs, xp, *params):
        _require_cuda(xs, "discriminator forward")
        n = xs.shape[0]
        eng = module._engine(("fwd", n), n, "gp")
        score = torch.empty(n, device=xs.device)
        eng.disc_forward(xs.detach().contiguous().float(), xp.detach().contiguous().float(), score)
        module._version += 1
        ctx.module, ctx.eng, ctx.version, ctx.n = module, eng, module._version, n
        return score.view(n, 1)

    @staticmethod
    def backward(ctx, dscore):
        m, eng, n = ctx.module, ctx.eng, ctx.n
        if ctx.version != m._version:
            raise RuntimeError("only the most recent forward of this discriminator can be back-propagated")
        need_xs, need_xp = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        need_p = any(ctx.needs_input_grad[3:])
        coef = dscore.reshape(n).contiguous().float()
        dxs = torch.empty(n, 1, 64, 64, device=coef.device) if need_xs else None
        dxp = torch.empty(n, 1, 64, device=coef.device) if need_xp else None
        eng.disc_backward(n, coef, dxs, dxp, want_params=need_p)
        pg = m._grad_tuple(eng.D.g) if need_p else tuple(None for _ in m._shapes)
        return (None, dxs, dxp) + pg


class DC_Discriminator(_FlatBacked):
    """layers.py:429-493: (sites (N,1,64,64), positions (N,1,64)) -> score (N,1)."""

    def __init__(self, data_size, negative_slope, num_channels, k_size):
        super().__init__()
        if data_size != 64 or k_size != 4 or num_channels != 1 or abs(negative_slope - 0.1) > 1e-12:
            raise ValueError("the HIP path implements the trained configuration: data_size=64, k_size=4, "
                             "num_channels=1, negative_slope=0.1 (src/train_wgan_v2.py:480)")
        self.negative_slope, self.data_size = negative_slope, data_size
        self._build(disc_param_shapes(), {})

    def _engine(self, key, batch, loss):
        if key not in self._engines:
            dev = self._store.flat.device
            self._engines[key] = Engine(_ops(dev), batch, dev, loss=loss, d_store=self._store)
        return self._engines[key]

    def forward(self, xs, xp):
        return _DiscFn.apply(self, xs, xp, *self._param_list())


# --------------------------------------------------------------------------------------------
# penalties
# --------------------------------------------------------------------------------------------
class _PenaltyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, loss, lam, real_s, real_p, fake_s, fake_p, t, *params):
        B = real_s.shape[0]
        eng = module._engine((loss, B), B, loss)
        eng.lam = float(lam)
        eng._init_coefs(wasserstein=False)
        eng.critic_grads_given_fake(real_s.detach().float(), real_p.detach().float(), fake_s.detach().float(),
                                    fake_p.detach().float(), t)
        ctx.grads = module._grad_tuple(eng.D.g)
        return eng.pen.clone().reshape(())

    @staticmethod
    def backward(ctx, gout):
        return (None,) * 8 + tuple(g * gout for g in ctx.grads)


def gp_gradient_penalty(discriminator, real_sites, real_pos, fake_sites, fake_pos, _lambda, device):
    """layers.py:34-55.  t ~ U[0,1) per sample comes from np.random.random on the host, exactly where
    the reference draws it (line 37); interpolates are constants w.r.t. the generator (lines 44-45);
    the value is lambda * mean over B*64*64 of (|d D / d sites| - 1)^2 (line 54)."""
    _require_cuda(real_sites, "gp_gradient_penalty")
    t = torch.from_numpy(np.random.random((real_pos.size(0), 1, 1, 1))).float().to(real_sites.device)
    return _PenaltyFn.apply(discriminator, "gp", _lambda, real_sites, real_pos, fake_sites, fake_pos, t,
                            *discriminator._param_list())


def _find_disc(t: torch.Tensor):
    seen, stack = set(), [t.grad_fn]
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        if hasattr(fn, "module") and isinstance(fn.module, DC_Discriminator):
            return fn.module
        stack.extend(f for f, _ in fn.next_functions)
    raise RuntimeError("div_gradient_penalty: d_loss_real is not the output of a DC_Discriminator")


def div_gradient_penalty(d_loss_real, d_loss_fake, real_sites, real_pos, fake_sites, fake_pos, device):
    """layers.py:60-77 (k = 2, p = 6): mean over the batch of |d D/d real sites|^6 + |d D/d fake
    sites|^6.  The discriminator is taken from the graph of d_loss_real; its forward on the two
    batches is re-evaluated inside the fused second-order pass."""
    disc = _find_disc(d_loss_real)
    return _PenaltyFn.apply(disc, "div", 0.0, real_sites, real_pos, fake_sites, fake_pos, None, *disc._param_list())


def count_parameters(model):
    """layers.py:20-30 without the prettytable dependency: prints and returns the trainable count."""
    total = 0
    for name, p in model.named_parameters():
        if p.requires_grad:
            print(f"{name}\t{p.numel()}")
            total += p.numel()
    print(f"Total Trainable Params: {total}")
    return total
