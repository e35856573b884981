"""DCWGAN-GP trainer for population-genetic alignments on MI355X: same command line, loaders and
checkpoint names as the reference's /root/reference/src/train_wgan_v2.py, with the training
loop (:509-640) replaced by the fused HIP step (pgan_trainer.FusedTrainer).

    python3 train_wgan_v2.py --idir sims/ --odir out/ --use_cuda --loss gp --batch_size 64

Differences from the reference, all deliberate:
  * the selected alignments are read from disk ONCE, packed into a pool resident in HBM and the
    critic_iter loop + generator step run as one hipGraph replay per iteration (no per-batch
    np.loadtxt, no per-iteration host synchronisation: losses are read back only when logged);
  * flags the reference parses but never reads are live: --gen (sigGen | tanGen | tanNorm),
    --use_buffer/--buffer_n (replay buffer of generated alignments, :98-117), --permute /
    --shuffle_inds (row shuffle of each real alignment), --label_smooth/--label_noise (fresh noise for every
    critic batch, as DataGeneratorDisk.get_batch draws it, :265-268); --mono_switch (live in the reference, :523-524)
    is drawn per critic batch on the device as well;
  * several GPUs (python -m torch.distributed.run --nproc-per-node N train_wgan_v2.py ...): --batch_size stays the
    global batch and the ranks split one run -- same pool, same generator streams, rank r takes rows
    [r B/N, (r+1) B/N) of every batch and draw -- so N GPUs train on the samples one GPU at that batch would;
  * an "epoch" is floor(num_in / batch / critic_iter) full iterations (the reference's extra
    generator step on the exhausted loader, :531-532,:639-640, is not reproduced);
  * of the --plot outputs (matplotlib / scikit-allel figures) only the 2DSWD number (w_dist_pca.txt) is produced.
"""
from __future__ import annotations

import argparse
import glob
import logging
import os
import random
import shutil
import subprocess
from collections import deque

import numpy as np
import torch

MS_PATH = os.path.join(os.getcwd(), "msdir/ms")


# ------------------------------------------------------------------------------------------------
# loaders (host side; same constructor / get_batch contracts as the reference)
# ------------------------------------------------------------------------------------------------
class DataGeneratorDisk(object):
    """src/train_wgan_v2.py:227-270.  `<n>_sites.csv` (64 lines of 64 comma-separated 0/1) and
    `<n>_pos.csv` (64 floats, one per line) pairs; `num_in` of them are drawn with random.sample
    (ValueError if the directory holds fewer), shuffled every epoch and served in batches."""

    def __init__(self, idir, num_in=10000, batch_size=8, data_size=64):
        self.data_size = int(data_size)      # individuals = sites per alignment (64; 128 for the scaled networks)
        pos = {os.path.basename(p)[:-len("_pos.csv")]: p for p in glob.glob(os.path.join(idir, "*_pos.csv"))}
        sites = {os.path.basename(p)[:-len("_sites.csv")]: p for p in glob.glob(os.path.join(idir, "*_sites.csv"))}
        keys = sorted(set(pos) & set(sites))
        self.positions = [pos[k] for k in keys]
        self.sites = [sites[k] for k in keys]
        self.indices = random.sample(list(range(len(self.positions))), k=num_in)
        self.length = len(self.indices) // batch_size
        self.batch_size = batch_size
        self.on_epoch_end()

    def on_epoch_end(self):
        random.shuffle(self.indices)
        self.ix = 0

    @staticmethod
    def read_pair(site_file, pos_file):
        s = np.loadtxt(site_file, delimiter=",", dtype=np.float32, ndmin=2)
        p = np.loadtxt(pos_file, dtype=np.float32, ndmin=1)
        return s, p

    def get_batch(self, label_smooth=False, shuffle=False):
        sel = self.indices[self.ix:self.ix + self.batch_size]
        self.ix += self.batch_size
        sites, pos = [], []
        for u in sel:
            s, p = self.read_pair(self.sites[u], self.positions[u])
            if shuffle:
                np.random.shuffle(s)
            sites.append(s)
            pos.append(p)
        S = self.data_size
        sites = np.array(sites, dtype=np.float32).reshape(len(sel), S, S) if sel else np.zeros((0, S, S), np.float32)
        if label_smooth:
            ey = np.random.uniform(0.0, 0.05, sites.shape)
            sites = sites * (1 - ey) + 0.05 * ey
        pos = np.array(pos, dtype=np.float32).reshape(len(sel), S) if sel else np.zeros((0, S), np.float32)
        return torch.FloatTensor(sites[:, None]), torch.FloatTensor(pos[:, None])

    def load_packed(self, threads=8):
        """Every selected alignment, in index order, through the native parser: (bits [num_in, 512] uint8,
        positions [num_in, 64] float32); alignment_slab.NotPackable if a sites file is not 0 / 1."""
        import alignment_slab
        return alignment_slab.parse_pairs([self.sites[u] for u in self.indices],
                                          [self.positions[u] for u in self.indices], threads)

    def load_all(self):
        """Every selected alignment, in index order: (num_in,1,64,64), (num_in,1,64) fp32.  Native parser when the
        sites are 0 / 1 (the values equal np.loadtxt's bit for bit), np.loadtxt otherwise."""
        import alignment_slab
        S = self.data_size
        if S == 64:          # the packed format holds 64 x 64 alignments
            try:
                bits, pos = self.load_packed()
                return (torch.from_numpy(alignment_slab.unpack_numpy(bits)[:, None]),
                        torch.from_numpy(pos[:, None].copy()))
            except alignment_slab.NotPackable:
                pass
        s = np.empty((len(self.indices), 1, S, S), np.float32)
        p = np.empty((len(self.indices), 1, S), np.float32)
        for i, u in enumerate(self.indices):
            s[i, 0], p[i, 0] = self.read_pair(self.sites[u], self.positions[u])
        return torch.from_numpy(s), torch.from_numpy(p)


class DataGenerator(object):
    """src/train_wgan_v2.py:272-302: batches from in-memory tensors, sites smoothed towards 0.5."""

    def __init__(self, sites_data, pos_data, batch_size=64):
        self.sites_data, self.pos_data = sites_data, pos_data
        self.length = self.sites_data.shape[0] // batch_size
        self.batch_size = batch_size
        self.on_epoch_end()

    def __len__(self):
        return self.length

    def on_epoch_end(self):
        self.indices = list(range(self.sites_data.shape[0]))
        random.shuffle(self.indices)

    def get_batch(self, sample=False):
        take = self.indices[:self.batch_size]
        xs = self.sites_data[take]
        ey = torch.FloatTensor(np.random.uniform(0.0, 0.05, xs.shape))
        xs = xs * (1 - ey) + 0.5 * ey
        xp = self.pos_data[take]
        if sample:
            random.shuffle(self.indices)
        else:
            del self.indices[:self.batch_size]
        return xs, xp


def read_ms_data(ms_lines):
    """Parse ms text output, statement for statement as src/train_wgan_v2.py:311-330: a replicate starts at every
    line containing '//'; its third line is 'positions: ...', split at single spaces with the first and the last
    token dropped (ms ends the line with a space); the lines after it, except the last one before the next '//', are
    one 0/1 string per individual; the last replicate is dropped (callers ask ms for batch_size + 1).
    Returns (R,1,64,S), (R,S,1)."""
    starts = [i for i, ln in enumerate(ms_lines) if "//" in ln] + [len(ms_lines)]
    chunks = [ms_lines[starts[k]:starts[k + 1]] for k in range(len(starts) - 1)]
    X, P = [], []
    for chunk in chunks[:-1]:
        P.append(np.array([u for u in chunk[2].split(" ")[1:-1] if u != ""], dtype=np.float32))
        X.append(np.array([[int(c) for c in r.replace("\n", "")] for r in chunk[3:-1]], dtype=np.float32))
    return np.expand_dims(np.array(X), 1), np.expand_dims(np.array(P), 2)


def read_ms_text(text, rows=64, cols=64):
    """read_ms_data on the whole text of an ms run through the native parser (pgan_parse_ms): same arrays
    (R,1,rows,cols), (R,cols,1); ValueError for text that is not `rows` individuals x `cols` sites per replicate."""
    import ctypes
    from pgan_hip import _lib
    raw = text.encode() if isinstance(text, str) else bytes(text)
    reps = max(0, raw.count(b"//") - 1)
    sites = np.empty((reps, 1, rows, cols), np.float32)
    pos = np.empty((reps, cols, 1), np.float32)
    got = _lib.pgan_parse_ms(raw, len(raw), rows, cols, reps, sites.ctypes.data_as(ctypes.c_void_p),
                             pos.ctypes.data_as(ctypes.c_void_p))
    if got < 0:
        raise ValueError(f"ms output is not {rows} individuals x {cols} sites per replicate (parser code {got})")
    return sites[:got], pos[:got]


class SimulatorGenerator(object):
    """src/train_wgan_v2.py:332-351: simulate batch_size + 1 replicates with `ms` on the fly
    (sim_class 0: constant size, 1: two demes of 32 with migration 0.01)."""

    def __init__(self, batch_size=64, sim_class=1):
        args = ["64", str(batch_size + 1), "-s", "64"]
        if sim_class == 1:
            args += ["-I", "2", "32", "32", "0.01"]
        elif sim_class != 0:
            raise ValueError("sim_class must be 0 or 1")
        self.cmd = [MS_PATH] + args

    def get_batch(self, label_smooth=True):
        sites, pos = read_ms_text(subprocess.run(self.cmd, capture_output=True, check=True).stdout)
        if label_smooth:   # formula as written in the reference (line 349)
            ey = np.random.uniform(0.0, 0.1, sites.shape)
            sites = sites * (1 - ey) + 0.5 * sites
        return torch.FloatTensor(sites), torch.FloatTensor(pos).transpose(1, 2)


class FakeBuffer(object):
    """src/train_wgan_v2.py:98-117: bounded FIFO of generated alignments; sample() draws a batch
    without replacement."""

    def __init__(self, max_size=2560):
        self.sites = deque(maxlen=max_size)
        self.pos = deque(maxlen=max_size)
        self.length = 0

    def add(self, sites, pos):
        self.sites.extend(list(sites))
        self.pos.extend(list(pos))
        self.length = len(self.sites)

    def sample(self, batch_size=64):
        ix = np.random.choice(range(self.length), batch_size, replace=False)
        return (torch.stack([torch.as_tensor(self.sites[u]) for u in ix]).float(),
                torch.stack([torch.as_tensor(self.pos[u]) for u in ix]).float())


def switch_monomorphic(sites, positions, lower_bound, upper_bound):
    """src/train_wgan_v2.py:129-160: per alignment, turn a random 0..n columns monomorphic (first
    half all-ancestral, second half all-derived) and redraw their positions between the neighbours."""
    for aln in range(sites.shape[0]):
        n_mono = random.randint(lower_bound, upper_bound)
        cols = np.random.choice(range(sites.shape[3]), n_mono, replace=False)
        half = len(cols) // 2
        sites[aln, :, :, cols[:half]] = 0.0
        sites[aln, :, :, cols[half:]] = 1.0
        last = positions.shape[2] - 1
        for i in cols:
            lo = 0.0 if i == 0 else float(positions[aln, 0, i - 1])
            hi = 1.0 if i == last else float(positions[aln, 0, i + 1])
            positions[aln, :, i] = float(np.random.uniform(lo, hi))
    return sites, positions


_GENOTYPE_RNG = {}        # device -> (op backend, Philox state): one per process and device
GENOTYPE_SEED = 20260     # seed of the device sampler's generator; set_genotype_sampler replaces it


def set_genotype_sampler(ops, seed: int) -> None:
    """Use `ops` (the trainer's op backend) and a generator state seeded with `seed` for the device path of
    sample_genotypes_bernoulli on that backend's GPU.  Without this call the first use builds a small backend of its
    own, seeded with GENOTYPE_SEED; Python's `random` stream is never touched."""
    dev = torch.device(ops.device)
    _GENOTYPE_RNG[dev] = (ops, torch.tensor([seed & 0x7FFFFFFF, (seed >> 31) & 0x7FFFFFFF, 0, 0], dtype=torch.int32,
                                            device=dev))


def sample_genotypes_bernoulli(sites_tensor, require_variation=True):
    """src/train_wgan_v2.py:162-192: Bernoulli genotypes from the generator's probabilities; a column
    that came out monomorphic is redrawn and one individual, picked in proportion to its probability
    of carrying the missing allele, is set to it.  Tensors on the GPU with 64 or 128 individuals are sampled there by one
    kernel (pgan_sample_genotypes, device Philox: see set_genotype_sampler); everything else by the loop below."""
    if sites_tensor.is_cuda and sites_tensor.dim() == 4 and sites_tensor.shape[1] == 1 and sites_tensor.shape[2] in (64, 128):
        from pgan_hip import HipOps
        dev = sites_tensor.device
        ent = _GENOTYPE_RNG.get(dev)
        if ent is None:
            with torch.cuda.device(dev):
                set_genotype_sampler(HipOps(dev, ws_floats=4, max_samples=1), GENOTYPE_SEED)
            ent = _GENOTYPE_RNG[dev]
        p = sites_tensor.detach().to(torch.float32).contiguous()
        out = torch.empty_like(p)
        with torch.cuda.device(dev):
            ent[0].sample_genotypes(p, out, ent[1], require_variation)
        return out
    out = torch.bernoulli(sites_tensor)
    if not require_variation:
        return out
    mono = (out.amin(dim=2) == out.amax(dim=2))[:, 0]           # (B, S)
    for aln, col in mono.nonzero().tolist():
        have = float(out[aln, 0, 0, col])
        probs = sites_tensor[aln, 0, :, col].detach().cpu().numpy().astype(np.float64)
        if have == 1.0:
            probs = 1 - probs
        probs = probs + 1e-7
        pick = np.random.choice(range(out.shape[2]), 1, p=probs / probs.sum())[0]
        out[aln, :, :, col] = torch.bernoulli(sites_tensor[aln, :, :, col])
        out[aln, :, pick, col] = 1 - have
    return out


# ------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--latent_size", default="128", help="size of latent/noise vector")
    p.add_argument("--idir", default="ms_sims/64ind_6sites_5.16e-04theta_10krep_pos/", help="input directory")
    p.add_argument("--odir", default="output/", help="output directory")
    p.add_argument("--plot", action="store_true", help="plot summaries in output")
    p.add_argument("--gen", default="sigGen", help="generator type: sigGen tanGen tanNorm")
    p.add_argument("--loss", default="gp", help="gp or div")
    p.add_argument("--gen_lr", default="0.00005", help="generator learning rate")
    p.add_argument("--disc_lr", default="0.00005", help="discriminator learning rate")
    p.add_argument("--num_in", default="10000", help="number of input alignments")
    p.add_argument("--use_cuda", action="store_true", help="run on the GPU (required)")
    p.add_argument("--save_freq", default="0", help="save model every save_freq epochs (0: never)")
    p.add_argument("--batch_size", default="64", help="batch size")
    p.add_argument("--epochs", default="10000", help="total number of epochs")
    p.add_argument("--critic_iter", default="5", help="critic updates per generator update")
    p.add_argument("--gp_lambda", default="10", help="lambda for gradient penalty")
    p.add_argument("--use_buffer", action="store_true", help="sample the critic's fake batch from a replay buffer")
    p.add_argument("--buffer_n", default="20", help="replay buffer size in batches")
    p.add_argument("--permute", action="store_true", help="permute real data along the individual axis")
    p.add_argument("--label_smooth", action="store_true", help="label smooth the real data")
    p.add_argument("--label_noise", default="0.05", help="upper bound of the label-smoothing noise")
    p.add_argument("--mono_switch", action="store_true", help="switch some input sites to monomorphic")
    p.add_argument("--normalize", action="store_true", help="normalize inputs to [-1, 1] for tanh")
    p.add_argument("--shuffle_inds", action="store_true", help="shuffle individuals in each input alignment")
    p.add_argument("--verbose", action="store_true", help="verbose output to log")
    p.add_argument("--slab", default="None",
                   help="packed copy of the selected input (bits + positions): read if the file exists, else written "
                        "after parsing --idir; later runs start without touching the CSV files")
    p.add_argument("--data_size", default="64",
                   help="individuals = sites per alignment: 64 (the reference's networks) or 128 (scaled networks with one "
                        "more stride-2 stage; no counterpart in the reference)")
    p.add_argument("--local_bn", action="store_true",
                   help="several GPUs: BatchNorm statistics of each GPU's own samples instead of the global batch")
    args = p.parse_args(argv)
    logging.basicConfig(level=logging.DEBUG if args.verbose else logging.INFO)
    if args.odir != "None" and int(os.environ.get("RANK", "0")) == 0:
        if not os.path.exists(args.odir):
            os.mkdir(args.odir)
        else:   # the reference empties an existing output directory (line 428)
            for entry in os.listdir(args.odir):
                path = os.path.join(args.odir, entry)
                shutil.rmtree(path) if os.path.isdir(path) else os.remove(path)
    return args


def default_init(store):
    """PyTorch's default initialisation of the layer types in `store` (conv / linear weights and biases uniform in
    +-1/sqrt(fan_in), norm scales 1 and shifts 0), for networks that have no nn.Module front-end here."""
    params = {}
    fans = {}
    for k, v in store.p.items():
        if v.dim() > 1:
            fan = v[0].numel()
            fans[k[:-len(".weight")]] = fan
            params[k] = (torch.rand(v.shape) * 2 - 1) / fan ** 0.5
    for k, v in store.p.items():
        if v.dim() == 1:
            base = k.rsplit(".", 1)[0]
            if k.endswith(".bias") and base in fans:          # bias of a convolution / the linear head
                params[k] = (torch.rand(v.shape) * 2 - 1) / fans[base] ** 0.5
            else:
                params[k] = torch.ones(v.shape) if k.endswith(".weight") else torch.zeros(v.shape)
    store.load(params)


def state_dicts(trainer):
    """(generator, discriminator) state_dicts with the reference modules' keys, shapes and dtypes."""
    e = trainer.eng
    g = {}
    for k, v in e.G.p.items():
        g[k] = v.detach().clone()
        if k.endswith(".bias"):   # BatchNorm: running stats follow the affine pair
            base = k[:-len(".bias")]
            g[base + ".running_mean"] = e.Gbuf.p[base + ".running_mean"].detach().clone()
            g[base + ".running_var"] = e.Gbuf.p[base + ".running_var"].detach().clone()
            g[base + ".num_batches_tracked"] = torch.tensor(e.g_batches_tracked, dtype=torch.long)
    d = {k: v.detach().clone() for k, v in e.D.p.items()}
    return g, d


def save_models(trainer, gen_path, disc_path):
    """src/data_processing_v2.py:26-28: torch.save of the two state_dicts."""
    g, d = state_dicts(trainer)
    torch.save(g, gen_path)
    torch.save(d, disc_path)


def plotting_forward(trainer, normalize, n=1000):
    """The generator pass the reference makes before every checkpoint, whether or not --plot is set
    (src/train_wgan_v2.py:686-693): n = 1000 latent draws through the generator in training mode -- its BatchNorm
    running statistics and batch counter step once more, and that is what the checkpoint then holds -- followed by
    undoing --normalize and genotyping.  Returns the genotyped sites (n,1,64,64) on the device."""
    from pgan_engine import Engine
    eng = trainer.eng
    ev = getattr(trainer, "_eval_engine", None)
    if ev is None or ev.B != n:
        ev = Engine(trainer.ops, n, trainer.dev, loss=eng.loss, act=eng.act, latent=eng.latent,
                    g_store=eng.G, g_buffers=eng.Gbuf, d_store=eng.D, generator_only=True, data=eng.data)
        trainer._eval_engine = ev
    z = torch.randn(n, eng.latent, device=trainer.dev)
    fs = torch.empty(n, 1, eng.data, eng.data, device=trainer.dev)
    fp = torch.empty(n, 1, eng.data, device=trainer.dev)
    ev.gen_forward(z, fs, fp)
    eng.g_batches_tracked += 1
    if normalize:
        fs = fs / 2 + 0.5
    return sample_genotypes_bernoulli(fs.clamp(0, 1))


def evaluate_2dswd(trainer, real_sites, normalize, n=500, seed=None, generated=None):
    """2DSWD of n generated against n input alignments (src/train_wgan_v2.py:723): SFS -> PCA(2) -> sliced
    Wasserstein of the first n genotyped alignments of the plotting forward (`generated`; made here when not given).
    Of the reference's --plot outputs only this number (w_dist_pca.txt) is produced."""
    from data_processing_v2 import plot_sfs_pca
    n = min(n, real_sites.shape[0])
    if generated is None:
        generated = plotting_forward(trainer, normalize, n)
    geno = generated[:n].cpu().numpy()
    pick = np.random.choice(range(real_sites.shape[0]), n, replace=False)
    real = real_sites[pick]
    real = real.cpu().numpy() if hasattr(real, "cpu") else np.asarray(real)
    return plot_sfs_pca(geno, 0, real, geno.shape[2], seed=seed)


def main(argv=None):
    args = parse_args(argv)
    if not args.use_cuda or not torch.cuda.is_available():
        raise SystemExit("this trainer runs on an MI355X GPU: pass --use_cuda on a machine with a HIP device")
    from pgan_hip import HipOps
    from pgan_trainer import FusedTrainer

    latent, batch = int(args.latent_size), int(args.batch_size)
    epochs, save_freq, critic_iter = int(args.epochs), int(args.save_freq), int(args.critic_iter)
    lam, num_in = int(args.gp_lambda), int(args.num_in)
    data_size = int(args.data_size)
    if data_size not in (64, 128):
        raise SystemExit("--data_size must be 64 or 128")
    act = "sigmoid" if args.gen == "sigGen" else "tanh"
    normalize = args.normalize or args.gen == "tanNorm"
    if args.plot:
        logging.warning("--plot: of the evaluation outputs only the 2DSWD number (w_dist_pca.txt) is produced")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    # several GPUs (python -m torch.distributed.run --nproc-per-node N train_wgan_v2.py ...): --batch_size stays
    # the global batch, split evenly; every rank reads the same files and takes its rows of every global batch
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        if batch % world:
            raise SystemExit(f"--batch_size {batch} is not a multiple of the {world} GPUs")
        batch //= world

    if world > 1:
        random.seed(1234)             # the ranks pick the same --num_in files
    import alignment_slab
    bits = None
    if data_size != 64:           # 128 x 128: CSV files through np.loadtxt; the packed slab format is for 64 x 64
        data = DataGeneratorDisk(args.idir, num_in=num_in, batch_size=batch, data_size=data_size)
        sites, pos = data.load_all()
    elif args.slab != "None" and os.path.exists(args.slab):
        bits, pos64 = alignment_slab.read_slab(args.slab)
        if bits.shape[0] < num_in:
            raise ValueError(f"{args.slab} holds {bits.shape[0]} alignments, --num_in asks for {num_in}")
        bits, pos64 = np.asarray(bits[:num_in]), np.asarray(pos64[:num_in])
    else:
        data = DataGeneratorDisk(args.idir, num_in=num_in, batch_size=batch)
        try:
            bits, pos64 = data.load_packed()
            if args.slab != "None" and rank == 0:
                alignment_slab.write_slab(args.slab, bits, pos64)
        except alignment_slab.NotPackable:      # sites other than 0 / 1: the reference's float loader
            sites, pos = data.load_all()
    if bits is not None:
        sites = torch.from_numpy(alignment_slab.unpack_numpy(bits)[:, None])
        pos = torch.from_numpy(pos64[:, None].copy())
    seed = random.randrange(1 << 30)      # drawn while the ranks' generators agree: one seed for all of them
    if world > 1:
        random.seed()
    trainer = FusedTrainer(HipOps(dev), batch, dev, loss=args.loss, act=act, lam=lam, critic_iter=critic_iter,
                           seed=seed, lr_d=float(args.disc_lr), lr_g=float(args.gen_lr),
                           latent=latent, rank=rank, world=world, sync_bn=(not args.local_bn) if world > 1 else None,
                           data=data_size)
    # --label_smooth / --mono_switch: drawn afresh for every critic batch, on the device, inside the recorded step
    if args.label_smooth or args.mono_switch:
        trainer.enable_augment(label_noise=float(args.label_noise) if args.label_smooth else None,
                               mono_switch=(0, 10) if args.mono_switch else None)
    if world > 1:      # after the trainer, whose streams then hold the hardware queues ahead of the collective library's
        import torch.distributed as dist
        backend = os.environ.get("PGAN_DIST_BACKEND", "nccl")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    if data_size == 64:
        import layers   # default PyTorch initialisation of the reference's layer types
        g0 = (layers.DC_Generator_Leaky_Sig if act == "sigmoid" else layers.DC_Generator_Leaky_Tan)(64, latent, 0.1, 1, 4)
        d0 = layers.DC_Discriminator(64, 0.1, 1, 4)
        trainer.eng.G.load({k: v for k, v in g0.state_dict().items() if k in trainer.eng.G.p})
        trainer.eng.D.load(d0.state_dict())
    else:
        default_init(trainer.eng.G)
        default_init(trainer.eng.D)
    if world > 1:                      # rank 0's initial weights everywhere
        import torch.distributed as dist
        for flat in (trainer.eng.G.flat, trainer.eng.D.flat):
            if dist.get_backend() == "nccl":
                dist.broadcast(flat, src=0)
            else:                          # other backends move host memory (see FusedTrainer.attach_process_group)
                host = flat.cpu()
                dist.broadcast(host, src=0)
                flat.copy_(host)
    trainer.weights_changed()
    if bits is not None:               # every rank holds the whole pool (the ranks take their rows of each global batch)
        trainer.set_pool_packed(bits, pos64, normalize=normalize)
    else:
        trainer.set_pool(sites, pos, normalize=normalize)
    if args.use_buffer:
        trainer.enable_fake_buffer(int(args.buffer_n) * batch)
    set_genotype_sampler(trainer.ops, seed + 6)     # the checkpoints' genotyping pass shares the trainer's backend
    trainer.use_tape = True      # first iteration recorded as a launch tape, the rest replayed (one or several GPUs)

    iters_per_epoch = max(1, (num_in // (batch * world)) // critic_iter)
    for epoch in range(1, epochs + 1):
        trainer.shuffle_pool(rows=args.permute or args.shuffle_inds)
        for j in range(iters_per_epoch):
            trainer.step()
            if j % (batch * world) == 1 and rank == 0:
                w, d_real, d_fake, pen, g_loss = trainer.losses()
                logging.info("Epoch:\t%d/%d Discriminator loss: %6.8f Generator loss: %6.8f d_real: %6.8f d_fake: %6.8f"
                             % (epoch, epochs, w, g_loss, d_real, d_fake))
        if rank == 0 and save_freq != 0 and (epoch == 1 or epoch % save_freq == 0 or epoch == epochs):
            generated = plotting_forward(trainer, normalize)      # before the save, as in the reference (:686-698)
            save_models(trainer, os.path.join(args.odir, "generator_model_{0:04d}.pt".format(epoch)),
                        os.path.join(args.odir, "discriminator_model_{0:04d}.pt".format(epoch)))
            if args.plot:
                w = evaluate_2dswd(trainer, sites, normalize, generated=generated)
                logging.info("Epoch:\t%d 2DSWD: %6.8f" % (epoch, w))
                with open(os.path.join(args.odir, "w_dist_pca.txt"), "a") as f:   # src/train_wgan_v2.py:724-726
                    f.write(str(epoch) + "\t" + str(w) + "\n")
    if world > 1:
        import torch.distributed as dist
        dump = os.environ.get("PGAN_DUMP_WEIGHT_SUMS")      # test aid: the replicas must stay identical
        if dump:
            with open(os.path.join(dump, f"weights_rank{rank}.txt"), "w") as f:
                f.write(repr([float(x.double().sum()) for x in (trainer.eng.G.flat, trainer.eng.D.flat)]) + "\n")
                f.write(repr([float(v.double().sum()) for v in trainer.eng.Gbuf.p.values()]) + "\n")
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
