"""Plain-torch statement of every device op the engine calls (TEST INFRASTRUCTURE).

Two uses: (1) the per-kernel reference the GPU tests compare each HIP op against, on the same
inputs; (2) a CPU backend that lets the `not gpu` tests run the engine's host logic (buffer
layout, op order, seeds, reductions) against the autograd oracle.  The product never imports
this file: its default and only backend is the HIP library.

Every method has the signature of the same-named method of pgan_hip.HipOps and writes its
results into caller-provided output tensors.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

SLOPE = 0.1
EPS = 1e-5


def _mask(h):
    return torch.where(h > 0, torch.ones_like(h), torch.full_like(h, SLOPE))


def _mean(x):
    return x.mean(dim=tuple(range(1, x.dim())), keepdim=True)


def _put_slice_amax(amax, out):
    """GroupNorm ops: amax[n, i] = largest magnitude of the i-th of 8 equal slices of sample n of the op's output."""
    if amax is not None:
        amax.copy_(out.reshape(out.shape[0], amax.shape[1], -1).abs().amax(2).to(amax.dtype))


def _put_channel_amax(amax, out):
    """BatchNorm ops: amax[c] = largest magnitude of channel c of the op's output."""
    if amax is not None:
        amax.copy_(out.abs().amax([d for d in range(out.dim()) if d != 1]).to(amax.dtype))


def philox_words(state, first_call: int, ncalls: int) -> np.ndarray:
    """Philox4x32-10 as the device generator runs it: call number (state offset + first_call + i) under the key
    (state[0], state[1]) gives four 32-bit words; returns uint32 [ncalls, 4].  state: 4 integers."""
    st = [int(x) & 0xFFFFFFFF for x in state]
    off = ((st[3] << 32) | st[2]) + int(first_call)
    ctr = (np.uint64(off) + np.arange(ncalls, dtype=np.uint64))
    c = [(ctr & np.uint64(0xFFFFFFFF)), (ctr >> np.uint64(32)), np.zeros(ncalls, np.uint64), np.zeros(ncalls, np.uint64)]
    k0, k1 = st[0], st[1]
    M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return np.stack(c, axis=1).astype(np.uint32)


def philox_advance(state, calls: int) -> None:
    """state's 64-bit offset += calls (state: an int32 tensor of 4 words, as the trainer keeps it)."""
    off = (((int(state[3]) & 0xFFFFFFFF) << 32) | (int(state[2]) & 0xFFFFFFFF)) + int(calls)
    lo, hi = off & 0xFFFFFFFF, (off >> 32) & 0xFFFFFFFF
    state[2] = lo - (1 << 32) if lo >= (1 << 31) else lo
    state[3] = hi - (1 << 32) if hi >= (1 << 31) else hi


def philox_values(state, skip: int, n: int, normal: bool) -> np.ndarray:
    """Values [skip, skip + n) of the draw that starts at the state's offset, as pgan_philox_slice produces them:
    uniforms (word >> 8) 2^-24; normals by Box-Muller on the word pairs of a call, in float32."""
    g0, g1 = skip // 4, (skip + n + 3) // 4
    w = philox_words(state, g0, g1 - g0)
    if normal:
        u1 = ((w[:, 0::2] >> 8).astype(np.float32) + np.float32(1)) * np.float32(2.0 ** -24)
        u2 = (w[:, 1::2] >> 8).astype(np.float32) * np.float32(2.0 ** -24)
        rad = np.sqrt(np.float32(-2) * np.log(u1, dtype=np.float32), dtype=np.float32)
        ang = np.float32(6.28318530717958647692) * u2
        v = np.stack([rad * np.cos(ang, dtype=np.float32), rad * np.sin(ang, dtype=np.float32)], axis=2).reshape(-1, 4)
    else:
        v = (w >> 8).astype(np.float32) * np.float32(2.0 ** -24)
    return v.reshape(-1)[skip - 4 * g0: skip - 4 * g0 + n].astype(np.float32)


MONO_CALLS, MONO_MAX = 8, 15      # Philox calls reserved per alignment / most columns switched (pgan_mono_switch)


def mono_switch_draws(state, first: int, B: int, lower: int, upper: int, S: int):
    """The draws pgan_mono_switch makes for alignments first .. first + B - 1: (n [B], cols [B, upper], u [B, upper])."""
    n_out = np.zeros(B, np.int32)
    cols = np.zeros((B, upper), np.int32)
    us = np.zeros((B, upper), np.float32)
    for a in range(B):
        w = philox_words(state, (first + a) * MONO_CALLS, MONO_CALLS).reshape(-1).astype(np.uint64)
        n = lower + int((w[0] * np.uint64(upper - lower + 1)) >> np.uint64(32))
        perm = list(range(S))
        for k in range(n):
            j = k + int((w[1 + k] * np.uint64(S - k)) >> np.uint64(32))
            perm[j], perm[k] = perm[k], perm[j]
            cols[a, k] = perm[k]
            us[a, k] = np.float32(int(w[1 + MONO_MAX + k]) >> 8) * np.float32(2.0 ** -24)
        n_out[a] = n
    return n_out, cols, us


def mono_switch_apply(sites, pos, n, cols, u):
    """switch_monomorphic (/root/reference/src/train_wgan_v2.py:129-160) with its draws given: per alignment the first
    n // 2 chosen columns all 0, the rest all 1; each chosen column's position redrawn, in the order drawn, as
    lo + (hi - lo) u between its current neighbours (0 / 1 at the ends), in double precision like np.random.uniform."""
    S = sites.shape[3]
    for a in range(sites.shape[0]):
        k_n = int(n[a])
        c = [int(x) for x in cols[a][:k_n]]
        for k, i in enumerate(c):
            sites[a, :, :, i] = 0.0 if k < k_n // 2 else 1.0
        for k, i in enumerate(c):
            lo = 0.0 if i == 0 else float(pos[a, 0, i - 1])
            hi = 1.0 if i == S - 1 else float(pos[a, 0, i + 1])
            pos[a, :, i] = float(lo + (hi - lo) * float(u[a][k]))
    return sites, pos


class TorchOps:
    name = "torch"

    # ---------------- conv family: weight w[Cs, Cl, 4(,4)], small side S, large side L ----------
    def conv_down(self, x, w, y, bound=None):        # bound (all conv_*): scaling hint of the fp16-split kernels; unused here
        y.copy_(F.conv2d(x, w, None, stride=2, padding=1))

    def conv_up(self, y, w, x, act=0, bound=None):
        o = F.conv_transpose2d(y, w, None, stride=2, padding=1)
        if act == 1:
            o = torch.sigmoid(o)
        elif act == 2:
            o = torch.tanh(o)
        x.copy_(o)

    def conv_wgrad(self, ya, x, dw, n_keep=None, n_swap=0, accumulate=False, bound_s=None, bound_l=None):
        """dw[cs,cl,kh,kw] = sum_n sum_{oh,ow} ya[n,cs,oh,ow] * x[r(n),cl,2oh-1+kh,2ow-1+kw] where
        r(n) = n for n < n_keep, n + n_swap for n_keep <= n < n_keep + n_swap, n - n_swap after."""
        n = ya.shape[0]
        if n_keep is None:
            n_keep = n
        idx = torch.arange(n)
        if n_swap:
            idx = torch.where(idx < n_keep, idx, torch.where(idx < n_keep + n_swap, idx + n_swap, idx - n_swap))
        xr = x[idx]
        g = torch.nn.grad.conv2d_weight(xr, dw.shape, ya, stride=2, padding=1)
        dw.copy_(dw + g if accumulate else g)

    # ---------------- GroupNorm(1, C) + LeakyReLU family (per-sample statistics) ----------------
    def gn_fwd(self, a, gamma, beta, h, stats, amax=None):
        mu = _mean(a)
        var = _mean((a - mu) ** 2)
        r = torch.rsqrt(var + EPS)
        g = gamma.view(1, -1, *([1] * (a.dim() - 2)))
        b = beta.view(1, -1, *([1] * (a.dim() - 2)))
        h.copy_(F.leaky_relu(g * ((a - mu) * r) + b, SLOPE))
        stats[:, 0] = mu.flatten()
        stats[:, 1] = r.flatten()
        _put_slice_amax(amax, h)

    def _xhat(self, a, stats):
        sh = (-1,) + (1,) * (a.dim() - 1)
        return (a - stats[:, 0].view(sh)) * stats[:, 1].view(sh), stats[:, 1].view(sh)

    def gn_bwd(self, u, h, a, stats, gamma, d, stats_b, pg, pb, amax=None, beta=None):      # beta: the HIP ops' switch to a mask recomputed from a; the mask is the same
        xh, r = self._xhat(a, stats)
        g = gamma.view(1, -1, *([1] * (a.dim() - 2)))
        e = _mask(h) * u
        w = g * e
        mw, mwx = _mean(w), _mean(w * xh)
        d.copy_(r * (w - mw - xh * mwx))
        _put_slice_amax(amax, d)
        stats_b[:, 0] = mw.flatten()
        stats_b[:, 1] = mwx.flatten()
        red = tuple(range(2, a.dim()))
        pg.copy_((e * xh).sum(red))
        pb.copy_(e.sum(red))

    def gn_jvp(self, t, h, a, stats, gamma, hd, stats_t, amax=None, beta=None):
        xh, r = self._xhat(a, stats)
        g = gamma.view(1, -1, *([1] * (a.dim() - 2)))
        tb, c = _mean(t), _mean(xh * t)
        hd.copy_(_mask(h) * g * (r * (t - tb - xh * c)))
        _put_slice_amax(amax, hd)
        stats_t[:, 0] = tb.flatten()
        stats_t[:, 1] = c.flatten()

    def gn_bwd2(self, hbar, u, t, h, a, stats, stats_b, stats_t, gamma, p, pg, pb, amax=None, beta=None):
        xh, r = self._xhat(a, stats)
        sh = (-1,) + (1,) * (a.dim() - 1)
        g = gamma.view(1, -1, *([1] * (a.dim() - 2)))
        m = _mask(h)
        mw, mwx = stats_b[:, 0].view(sh), stats_b[:, 1].view(sh)
        tb, c = stats_t[:, 0].view(sh), stats_t[:, 1].view(sh)
        w = g * m * u
        hb = torch.zeros_like(u) if hbar is None else hbar.view_as(u)
        q = g * m * hb
        mq, mqx, mwt = _mean(q), _mean(q * xh), _mean(w * t)
        mQ = mq - r * (c * mw + mwx * tb)
        mQx = mqx - 2 * r * c * mwx
        mwxd = r * (mwt - mw * tb - mwx * c)
        Q = q - r * (c * w + mwx * t)
        p.copy_(r * (Q - mQ - xh * (mQx + mwxd)))
        _put_slice_amax(amax, p)
        xd = r * (t - tb - xh * c)
        red = tuple(range(2, a.dim()))
        pg.copy_((m * u * xd + m * hb * xh).sum(red))
        pb.copy_((m * hb).sum(red))

    def rows_reduce(self, part, out, r0, r1, r2, r3, accumulate=False):
        """out[c] (+)= sum of part[n, c] over rows [r0, r1) and [r2, r3)."""
        v = part[r0:r1].sum(0) + part[r2:r3].sum(0)
        if accumulate:
            out.add_(v)
        else:
            out.copy_(v)

    def rows_reduce_multi(self, parts, outs, r0, r1, r2, r3):
        for part, out in zip(parts, outs):
            out.copy_(part[r0:r1].sum(0) + part[r2:r3].sum(0))

    # ---------------- linear head ----------------
    def head_fwd(self, h4, f1, w, b, score):
        """score[n] = <w[:8192], h4[n]> + <w[8192:], f1[n]> + b."""
        k = h4[0].numel()
        score.copy_(h4.flatten(1) @ w.flatten()[:k] + f1.flatten(1) @ w.flatten()[k:] + b)

    def head_seed(self, w, coef, u4):
        """u4[n] = coef[n] * w[:8192]."""
        k = u4[0].numel()
        u4.copy_((coef.view(-1, 1) * w.flatten()[:k].view(1, -1)).view_as(u4))

    def head_wgrad(self, x, coef, dw):
        """dw[j] = sum_n coef[n] * x[n, j]."""
        dw.copy_((coef.view(-1, 1) * x.flatten(1)).sum(0))

    # ---------------- penalties ----------------
    def interpolate(self, t, real, fake, out):
        sh = (-1,) + (1,) * (real.dim() - 1)
        tt = t.view(sh)
        out.copy_(tt * real + (1 - tt) * fake)

    def gp_penalty(self, g, lam, v, pen):
        """pen = lam * mean((|g|-1)^2); v = d pen / d g."""
        n = g.numel()
        a = g.abs()
        pen.copy_(((a - 1) ** 2).sum() * (lam / n))
        v.copy_((2.0 * lam / n) * (a - 1) * torch.sign(g))

    def div_penalty(self, g, nb, v, pen):
        """g holds 2*nb samples (real then fake).  pen = mean_b(|g_real,b|^6 + |g_fake,b|^6);
        v = d pen / d g = 6 |g_n|^4 g_n / nb."""
        s = g.flatten(1).pow(2).sum(1)
        pen.copy_((s ** 3).sum() / nb)
        v.copy_((6.0 / nb) * (s ** 2).view(-1, *([1] * (g.dim() - 1))) * g)

    # ---------------- discriminator 1-D tower ----------------
    def d1d_fwd(self, xp, cw, cb, gw, gb, a_save, feat):
        """Conv1d(k4)+bias -> GroupNorm(1,C) -> LeakyReLU, three times.  cw/cb/gw/gb are lists of 3.
        a_save[i] receives the conv outputs (pre-norm); feat the final activations (N, 64*55)."""
        y = xp
        for i in range(3):
            a = F.conv1d(y, cw[i], cb[i])
            a_save[i].copy_(a)
            y = F.leaky_relu(F.group_norm(a, 1, gw[i], gb[i], EPS), SLOPE)
        feat.copy_(y.flatten(1))

    def d1d_bwd(self, xp, cw, cb, gw, gb, a_save, coef, w_tail, d_cw, d_cb, d_gw, d_gb, dxp):
        """Backward of sum_n coef[n] * <w_tail, feat[n]> through the 1-D tower: parameter gradients
        (overwritten) and, when dxp is not None, the gradient w.r.t. the positions input."""
        xs = xp.detach().clone().requires_grad_(dxp is not None)
        ps = [t.detach().clone().requires_grad_(True) for t in list(cw) + list(cb) + list(gw) + list(gb)]
        cw_, cb_, gw_, gb_ = ps[0:3], ps[3:6], ps[6:9], ps[9:12]
        y = xs
        for i in range(3):
            y = F.leaky_relu(F.group_norm(F.conv1d(y, cw_[i], cb_[i]), 1, gw_[i], gb_[i], EPS), SLOPE)
        loss = (coef.view(-1, 1) * y.flatten(1) * w_tail.view(1, -1)).sum()
        ins = ps + ([xs] if dxp is not None else [])
        gr = torch.autograd.grad(loss, ins)
        if d_cw is not None:
            for dst, src in zip(list(d_cw) + list(d_cb) + list(d_gw) + list(d_gb), gr[:12]):
                dst.copy_(src)
        if dxp is not None:
            dxp.copy_(gr[12])

    # ---------------- generator pieces ----------------
    def gemm_nn(self, a, b, c):
        """c[M,N] = a[M,K] @ b[K,N]."""
        c.copy_(a @ b)

    def gemm_tn(self, a, b, c):
        """c[M,N] = a[K,M]^T @ b[K,N]."""
        c.copy_(a.t() @ b)

    def bn_fwd(self, a, gamma, beta, rmean, rvar, h, stats, momentum=0.1, groups=1, amax=None):
        """Training-mode BatchNorm + LeakyReLU.  stats[0,c]=batch mean, stats[1,c]=rstd.  Running
        stats are updated in place (momentum, unbiased variance).  groups > 1: that many consecutive forward passes
        stacked along the batch axis, each with its own statistics (stats [groups, 2, C])."""
        if groups > 1:
            B = a.shape[0] // groups
            for g in range(groups):
                self.bn_fwd(a[g * B:(g + 1) * B], gamma, beta, rmean, rvar, h[g * B:(g + 1) * B],
                            stats.view(groups, 2, -1)[g], momentum)
            _put_channel_amax(amax, h)
            return
        red = (0,) + tuple(range(2, a.dim()))
        sh = (1, -1) + (1,) * (a.dim() - 2)
        n = a.numel() // a.shape[1]
        mu = a.mean(red)
        var = ((a - mu.view(sh)) ** 2).mean(red)
        r = torch.rsqrt(var + EPS)
        h.copy_(F.leaky_relu(gamma.view(sh) * ((a - mu.view(sh)) * r.view(sh)) + beta.view(sh), SLOPE))
        stats[0] = mu
        stats[1] = r
        rmean.mul_(1 - momentum).add_(momentum * mu)
        rvar.mul_(1 - momentum).add_(momentum * var * (n / max(n - 1, 1)))
        _put_channel_amax(amax, h)

    # split form for batch statistics over several ranks: part[c] = (sum, sum of squares) resp. (sum dy, sum dy xhat)
    def bn_partials(self, B, C, HW) -> int:
        return 2 * C

    def bn_fwd_partials(self, B, C, HW, groups=1) -> int:
        return 2 * C * groups

    def bn_fwd_sums(self, a, part, groups=1, amax=None):
        if groups > 1:
            B = a.shape[0] // groups
            for g in range(groups):
                self.bn_fwd_sums(a[g * B:(g + 1) * B], part.view(groups, -1)[g])
            return
        red = (0,) + tuple(range(2, a.dim()))
        ad = a.double()
        part.view(-1, 2)[:, 0] = ad.sum(red)
        part.view(-1, 2)[:, 1] = (ad * ad).sum(red)

    def bn_fwd_apply(self, a, gamma, beta, rmean, rvar, h, stats, part, world, momentum=0.1, groups=1, amax=None):
        if groups > 1:
            B = a.shape[0] // groups
            for g in range(groups):
                self.bn_fwd_apply(a[g * B:(g + 1) * B], gamma, beta, rmean, rvar, h[g * B:(g + 1) * B],
                                  stats.view(groups, 2, -1)[g], part.view(groups, -1)[g], world, momentum)
            _put_channel_amax(amax, h)
            return
        sh = (1, -1) + (1,) * (a.dim() - 2)
        n = world * a.numel() // a.shape[1]
        mu = part.view(-1, 2)[:, 0] / n
        var = (part.view(-1, 2)[:, 1] / n - mu * mu).clamp_min(0)
        r = torch.rsqrt(var + EPS).float()
        mu32 = mu.float()
        h.copy_(F.leaky_relu(gamma.view(sh) * ((a - mu32.view(sh)) * r.view(sh)) + beta.view(sh), SLOPE))
        stats[0] = mu32
        stats[1] = r
        rmean.mul_(1 - momentum).add_(momentum * mu32)
        rvar.mul_(1 - momentum).add_(momentum * (var * (n / max(n - 1, 1))).float())
        _put_channel_amax(amax, h)

    def bn_bwd_sums(self, dh, h, a, stats, part, amax=None):
        red = (0,) + tuple(range(2, a.dim()))
        sh = (1, -1) + (1,) * (a.dim() - 2)
        xh = (a - stats[0].view(sh)) * stats[1].view(sh)
        dy = _mask(h) * dh
        part.view(-1, 2)[:, 0] = dy.double().sum(red)
        part.view(-1, 2)[:, 1] = (dy * xh).double().sum(red)

    def bn_bwd_apply(self, dh, h, a, stats, gamma, da, dgamma, dbeta, part, world, amax=None):
        sh = (1, -1) + (1,) * (a.dim() - 2)
        n = world * a.numel() // a.shape[1]
        xh = (a - stats[0].view(sh)) * stats[1].view(sh)
        dy = _mask(h) * dh
        sb, sg = part.view(-1, 2)[:, 0], part.view(-1, 2)[:, 1]
        dgamma.copy_((sg / world).float())
        dbeta.copy_((sb / world).float())
        da.copy_(gamma.view(sh) * stats[1].view(sh) * (dy - (sb / n).float().view(sh) - xh * (sg / n).float().view(sh)))
        _put_channel_amax(amax, da)

    def bn_bwd(self, dh, h, a, stats, gamma, da, dgamma, dbeta, amax=None):
        red = (0,) + tuple(range(2, a.dim()))
        sh = (1, -1) + (1,) * (a.dim() - 2)
        n = a.numel() // a.shape[1]
        xh = (a - stats[0].view(sh)) * stats[1].view(sh)
        dy = _mask(h) * dh
        sg = (dy * xh).sum(red)
        sb = dy.sum(red)
        dgamma.copy_(sg)
        dbeta.copy_(sb)
        da.copy_(gamma.view(sh) * stats[1].view(sh) * (dy - sb.view(sh) / n - xh * sg.view(sh) / n))
        _put_channel_amax(amax, da)

    def act_bwd(self, dy, y, dx, act):
        """dx = dy * act'(pre) expressed through the activation's output y (1 sigmoid, 2 tanh)."""
        if act == 1:
            dx.copy_(dy * y * (1 - y))
        else:
            dx.copy_(dy * (1 - y * y))

    def convt1d_up(self, y, w, x, act=0):
        o = F.conv_transpose1d(y, w, None, stride=2, padding=1)
        if act == 1:
            o = torch.sigmoid(o)
        elif act == 2:
            o = torch.tanh(o)
        x.copy_(o)

    def convt1d_down(self, x, w, y):
        """Input gradient of convt1d_up: y[n,cs,o] = sum_{cl,k} x[n,cl,2o-1+k] w[cs,cl,k]."""
        y.copy_(F.conv1d(x, w, None, stride=2, padding=1))

    def convt1d_wgrad(self, ya, x, dw):
        dw.copy_(torch.nn.grad.conv1d_weight(x, dw.shape, ya, stride=2, padding=1))

    # ---------------- optimiser / misc ----------------
    def adam(self, p, g, m, v, step, lr, b1, b2, eps, step_dev=None, grad_scale=1.0):
        if step_dev is not None:      # step count kept in a tensor: incremented, then used
            step_dev += 1
            step = int(step_dev)
        if grad_scale != 1.0:
            g = g * grad_scale
        bc1 = 1.0 - b1 ** step
        bc2 = 1.0 - b2 ** step
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        p.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(eps), value=-lr / bc1)

    def philox(self, out, state, normal, skip=0, total=None):
        """The device generator restated (Philox4x32-10, same words, same conversions): values [skip, skip + n) of a draw
        of `total`, the state's offset advanced by the whole draw."""
        n = out.numel()
        total = n if total is None else total
        out.copy_(torch.from_numpy(philox_values(state.tolist(), skip, n, bool(normal))).reshape(out.shape))
        philox_advance(state, (total + 3) // 4)

    def take_batch(self, pool_s, pool_p, out_s, out_p, cursor, offset=0, advance=None):
        n, B = pool_s.shape[0], out_s.shape[0]
        idx = (int(cursor) + offset + torch.arange(B)) % n
        out_s.copy_(pool_s[idx])
        out_p.copy_(pool_p[idx])
        cursor.copy_((cursor + (B if advance is None else advance)) % n)

    def label_smooth(self, sites, noise, centre, normalized, rng, ey=None, skip=0, total=None):
        """sites (1 - ey) + centre ey on the raw values (DataGeneratorDisk.get_batch,
        /root/reference/src/train_wgan_v2.py:265-268), ey = noise * uniform from the device generator's words."""
        n = sites.numel()
        total = n if total is None else total
        if ey is None:
            e = torch.from_numpy(philox_values(rng.tolist(), skip, n, False)).reshape(sites.shape) * noise
            philox_advance(rng, (total + 3) // 4)
        else:
            e = ey.reshape(sites.shape)
        raw = sites * 0.5 + 0.5 if normalized else sites
        sm = raw * (1 - e) + centre * e
        sites.copy_(sm * 2 - 1 if normalized else sm)

    def mono_switch(self, sites, pos, lower, upper, rng, given=None, first=0, total=None):
        B, S = sites.shape[0], sites.shape[3]
        total = B if total is None else total
        if given is None:
            n, cols, u = mono_switch_draws(rng.tolist(), first, B, lower, upper, S)
            philox_advance(rng, total * MONO_CALLS)
        else:
            n, cols, u = (t.cpu().numpy().reshape(B, -1) if i else t.cpu().numpy() for i, t in enumerate(given))
        mono_switch_apply(sites, pos, n, cols, u)

    def affine(self, x, scale, shift, out):
        out.copy_(x * scale + shift)

    def critic_loss(self, score, nb, pen, out):
        """out = [W, mean_real, mean_fake, pen] with W = mean(score[nb:2nb]) - mean(score[:nb]) + pen."""
        mr, mf = score[:nb].mean(), score[nb:2 * nb].mean()
        out[0] = mf - mr + pen.reshape(())
        out[1] = mr
        out[2] = mf
        out[3] = pen.reshape(())

    def neg_mean(self, score, out):
        out.copy_(-score.mean())
