"""Generate tests/golden/*.npz from the REFERENCE implementation (build container only).

TEST INFRASTRUCTURE.  Imports /root/reference/src/layers.py (with a stub for the absent
`prettytable`, which layers.py:11 imports only for count_parameters), loads portable-PRNG
parameters into the reference's own nn.Modules, drives them exactly as
src/train_wgan_v2.py:516-634 does (torch.optim.Adam at :489-490, gp/div penalties from
layers.py:34-77) and stores inputs' seeds + expected outputs.  The reference source never
ships: fixtures hold data only.  Re-run:  python oracle/make_golden.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import dcwgan_oracle as orc  # noqa: E402  (parameter tables + synthetic inputs only)

REF_SRC = "/root/reference/src"
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def import_reference():
    sys.dont_write_bytecode = True
    stub = types.ModuleType("prettytable")

    class PrettyTable:  # layers.py:20-30 only
        def __init__(self, *a, **k):
            pass

        def add_row(self, *a, **k):
            pass

    stub.PrettyTable = PrettyTable
    sys.modules["prettytable"] = stub
    sys.path.insert(0, REF_SRC)
    import layers as ref  # type: ignore
    return ref


def summary(t: torch.Tensor, nsamp: int = 256):
    """(sum, sumsq) in fp64 plus a strided sample; tensors up to 4096 elements are kept whole."""
    a = t.detach().to(torch.float64).reshape(-1).numpy()
    if a.size <= 4096:
        samp = a.astype(np.float32)
    else:
        idx = (np.arange(nsamp, dtype=np.int64) * (a.size // nsamp) + (a.size // nsamp) // 2)
        samp = a[idx].astype(np.float32)
    return np.array([a.sum(), (a * a).sum()], dtype=np.float64), samp


def put(store, name, t):
    s, samp = summary(t)
    store[name + "#stat"] = s
    store[name + "#samp"] = samp


def load_into(module, *dicts):
    sd = {}
    for d in dicts:
        sd.update({k: v.clone() for k, v in d.items()})
    module.load_state_dict(sd, strict=True)


def z_draw(seed, b, latent=orc.LATENT):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(rng.standard_normal((b, latent, 1, 1)).astype(np.float32))


def t_draw(seed, b):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random((b, 1, 1, 1))  # float64, as np.random.random gives the reference


class _FixedT:
    """Stands in for np.random.random inside the reference's gp_gradient_penalty (layers.py:37)."""

    def __init__(self, t):
        self.t = t

    def __call__(self, shape):
        assert tuple(shape) == self.t.shape
        return self.t


def forward_fixture(ref, B, seed):
    st = {}
    gp_ = orc.make_params(orc.gen_param_shapes(), seed)
    dp = orc.make_params(orc.disc_param_shapes(), seed + 1)
    z = z_draw(seed + 2, B)
    rs, rp = orc.synthetic_alignments(B, seed + 3)
    for act, cls in (("sigmoid", ref.DC_Generator_Leaky_Sig), ("tanh", ref.DC_Generator_Leaky_Tan)):
        G = cls(64, 128, 0.1, 1, 4)
        load_into(G, gp_, orc.make_gen_buffers())
        G.train()
        hooks, acts = [], {}
        for tower, tag in ((G.twod_conv_layers, "g2d"), (G.oned_conv_layers, "g1d")):
            for i, m in enumerate(tower):
                if isinstance(m, torch.nn.LeakyReLU):
                    hooks.append(m.register_forward_hook(
                        lambda mod, inp, out, key=f"{tag}.act{i // 3}": acts.__setitem__(key, out.detach().clone())))
        with torch.no_grad():
            xs, xp = G(z)
        for h in hooks:
            h.remove()
        put(st, f"{act}.xs", xs)
        put(st, f"{act}.xp", xp)
        for k, v in acts.items():
            put(st, f"{act}.{k}", v)
        for k, v in G.state_dict().items():
            if "running" in k:
                put(st, f"{act}.buf.{k}", v)
        if act == "sigmoid":
            fake_s, fake_p = xs, xp
    D = ref.DC_Discriminator(64, 0.1, 1, 4)
    load_into(D, dp)
    hooks, acts = [], {}
    for tower, tag in ((D.twod_conv_layers, "d2d"), (D.oned_conv_layers, "d1d")):
        for i, m in enumerate(tower):
            if isinstance(m, torch.nn.LeakyReLU):
                hooks.append(m.register_forward_hook(
                    lambda mod, inp, out, key=f"{tag}.act{i // 3}": acts.__setitem__(key, out.detach().clone())))
    with torch.no_grad():
        score = D(torch.cat([rs, fake_s]), torch.cat([rp, fake_p]))
    put(st, "d.score", score)
    for k, v in acts.items():
        put(st, f"d.{k}", v)
    st["meta"] = np.array([B, seed], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, f"forward_B{B}.npz"), **st)


def critic_fixture(ref, B, seed, loss, lam=10):
    st = {}
    gp_ = orc.make_params(orc.gen_param_shapes(), seed)
    dp = orc.make_params(orc.disc_param_shapes(), seed + 1)
    z = z_draw(seed + 2, B)
    rs, rp = orc.synthetic_alignments(B, seed + 3)
    t = t_draw(seed + 4, B)
    G = ref.DC_Generator_Leaky_Sig(64, 128, 0.1, 1, 4)
    D = ref.DC_Discriminator(64, 0.1, 1, 4)
    load_into(G, gp_, orc.make_gen_buffers())
    load_into(D, dp)
    G.train()
    device = torch.device("cpu")
    fs, fp = G(z)
    if loss == "div":  # src/train_wgan_v2.py:560-561,573-574
        rs = torch.autograd.Variable(rs, requires_grad=True)
        rp = torch.autograd.Variable(rp, requires_grad=True)
        fs = torch.autograd.Variable(fs, requires_grad=True)
        fp = torch.autograd.Variable(fp, requires_grad=True)
    D.zero_grad()
    d = D(torch.cat([rs, fs]), torch.cat([rp, fp]))
    if loss == "gp":  # :584-587
        real_rand = np.random.random
        np.random.random = _FixedT(t)
        try:
            pen = ref.gp_gradient_penalty(D, rs, rp, fs, fp, lam, device)
        finally:
            np.random.random = real_rand
        w = d[B:].mean() - d[:B].mean() + pen
    else:  # :589-597
        pen = ref.div_gradient_penalty(d[:B], d[B:], rs, rp, fs, fp, device)
        w = d[B:].mean() - d[:B].mean() + pen
    w.backward()
    st["w"] = np.array([float(w), float(pen), float(d[:B].mean()), float(d[B:].mean())], dtype=np.float64)
    st["t"] = t.astype(np.float32)
    for k, v in D.named_parameters():
        g = v.grad if v.grad is not None else torch.zeros_like(v)
        put(st, f"grad.{k}", g)
    # GP-only gradient: which parameters the penalty reaches at all
    if loss == "gp":
        D.zero_grad()
        np.random.random = _FixedT(t)
        try:
            pen2 = ref.gp_gradient_penalty(D, rs.detach(), rp.detach(), fs.detach(), fp.detach(), lam, device)
        finally:
            np.random.random = real_rand
        pen2.backward()
        untouched = [k for k, v in D.named_parameters() if v.grad is None]
        st["gp_untouched"] = np.array(sorted(untouched))
        for k, v in D.named_parameters():
            if v.grad is not None:
                put(st, f"gpgrad.{k}", v.grad)
    st["meta"] = np.array([B, seed, lam], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, f"critic_{loss}_B{B}.npz"), **st)


def generator_fixture(ref, B, seed, act):
    st = {}
    gp_ = orc.make_params(orc.gen_param_shapes(), seed)
    dp = orc.make_params(orc.disc_param_shapes(), seed + 1)
    z = z_draw(seed + 2, B)
    cls = ref.DC_Generator_Leaky_Sig if act == "sigmoid" else ref.DC_Generator_Leaky_Tan
    G = cls(64, 128, 0.1, 1, 4)
    D = ref.DC_Discriminator(64, 0.1, 1, 4)
    load_into(G, gp_, orc.make_gen_buffers())
    load_into(D, dp)
    G.train()
    G.zero_grad()
    xs, xp = G(z)
    gl = -1 * D(xs, xp).mean()  # src/train_wgan_v2.py:625-629
    gl.backward()
    st["loss"] = np.array([float(gl)], dtype=np.float64)
    for k, v in G.named_parameters():
        put(st, f"grad.{k}", v.grad)
    st["meta"] = np.array([B, seed], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, f"generator_{act}_B{B}.npz"), **st)


def train_fixture(ref, B, seed, loss, act, normalize, iters, critic_iter=5, lam=10, tag=None):
    """K full iterations with torch.optim.Adam(lr 5e-5, betas (0.5, 0.999))."""
    st = {}
    gp_ = orc.make_params(orc.gen_param_shapes(), seed)
    dp = orc.make_params(orc.disc_param_shapes(), seed + 1)
    cls = ref.DC_Generator_Leaky_Sig if act == "sigmoid" else ref.DC_Generator_Leaky_Tan
    G = cls(64, 128, 0.1, 1, 4)
    D = ref.DC_Discriminator(64, 0.1, 1, 4)
    load_into(G, gp_, orc.make_gen_buffers())
    load_into(D, dp)
    G.train()
    d_opt = torch.optim.Adam(D.parameters(), lr=5e-5, betas=(0.5, 0.999))
    g_opt = torch.optim.Adam(G.parameters(), lr=5e-5, betas=(0.5, 0.999))
    device = torch.device("cpu")
    real_s, real_p = orc.synthetic_alignments(B * critic_iter * iters, seed + 3)
    traj = []
    real_rand = np.random.random
    ctr = 0
    for it in range(iters):
        for h in range(critic_iter):
            rs = real_s[ctr * B:(ctr + 1) * B]
            rp = real_p[ctr * B:(ctr + 1) * B]
            if normalize:  # torchvision Normalize(0.5, 0.5), src/train_wgan_v2.py:453,519-521
                rs, rp = (rs - 0.5) / 0.5, (rp - 0.5) / 0.5
            z = z_draw(seed + 1000 + it * 16 + h, B)
            fs, fp = G(z)
            if loss == "div":
                rs = torch.autograd.Variable(rs, requires_grad=True)
                rp = torch.autograd.Variable(rp, requires_grad=True)
                fs = torch.autograd.Variable(fs, requires_grad=True)
                fp = torch.autograd.Variable(fp, requires_grad=True)
            d_opt.zero_grad()
            d = D(torch.cat([rs, fs]), torch.cat([rp, fp]))
            if loss == "gp":
                np.random.random = _FixedT(t_draw(seed + 5000 + it * 16 + h, B))
                try:
                    pen = ref.gp_gradient_penalty(D, rs, rp, fs, fp, lam, device)
                finally:
                    np.random.random = real_rand
            else:
                pen = ref.div_gradient_penalty(d[:B], d[B:], rs, rp, fs, fp, device)
            w = d[B:].mean() - d[:B].mean() + pen
            w.backward()
            d_opt.step()
            ctr += 1
        g_opt.zero_grad()
        z = z_draw(seed + 1000 + it * 16 + critic_iter, B)
        xs, xp = G(z)
        gl = -1 * D(xs, xp).mean()
        gl.backward()
        g_opt.step()
        traj.append([float(w), float(gl), float(pen)])
    st["traj"] = np.array(traj, dtype=np.float64)
    for k, v in D.state_dict().items():
        put(st, f"D.{k}", v)
    for k, v in G.state_dict().items():
        put(st, f"G.{k}", v.to(torch.float64) if v.dtype == torch.int64 else v)
    for name, opt, mod in (("D", d_opt, D), ("G", g_opt, G)):
        for k, v in mod.named_parameters():
            s = opt.state[v]
            put(st, f"{name}.exp_avg.{k}", s["exp_avg"])
            put(st, f"{name}.exp_avg_sq.{k}", s["exp_avg_sq"])
    st["meta"] = np.array([B, seed, lam, iters, critic_iter, int(normalize)], dtype=np.int64)
    tag = tag or f"{loss}_{act}{'_norm' if normalize else ''}"
    np.savez_compressed(os.path.join(OUT, f"train_{tag}_B{B}.npz"), **st)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    if "--one-step-only" in sys.argv:      # adds the one-step fixture without rewriting the others
        one_step_fixture(ref)
        return
    forward_fixture(ref, 4, 100)
    critic_fixture(ref, 4, 200, "gp")
    critic_fixture(ref, 4, 300, "div")
    critic_fixture(ref, 64, 400, "gp")
    generator_fixture(ref, 4, 500, "sigmoid")
    generator_fixture(ref, 4, 600, "tanh")
    train_fixture(ref, 8, 700, "gp", "sigmoid", False, iters=3)
    train_fixture(ref, 8, 800, "gp", "tanh", True, iters=2)
    train_fixture(ref, 8, 900, "div", "sigmoid", False, iters=2)
    one_step_fixture(ref)
    print("golden fixtures written to", OUT)


def one_step_fixture(ref):
    """One critic step + one generator step (critic_iter 1, one iteration): both optimizers have made exactly one step, so
    exp_avg = (1 - beta1) g and exp_avg_sq = (1 - beta2) g^2 of torch.optim.Adam pin the first gradients and the fused
    Adam's moment arithmetic; written as train_gp_sigmoid_onestep_B8.npz."""
    train_fixture(ref, 8, 1000, "gp", "sigmoid", False, iters=1, critic_iter=1, tag="gp_sigmoid_onestep")


if __name__ == "__main__":
    main()
