"""The command-line trainer end to end on the GPU: CSV directory in, reference-named checkpoints out."""
import os

import numpy as np
import pytest
import torch

import dcwgan_oracle as orc

pytestmark = pytest.mark.gpu


def _write_sims(tmp, n, size=64):
    rs, rp = orc.synthetic_alignments(n, 5, size=size)
    for i in range(n):
        np.savetxt(os.path.join(tmp, f"{i}_sites.csv"), rs[i, 0].numpy().astype(int), fmt="%d", delimiter=",")
        np.savetxt(os.path.join(tmp, f"{i}_pos.csv"), rp[i, 0].numpy(), fmt="%.6f")


@pytest.mark.parametrize("extra", [[], ["--loss", "div", "--use_buffer", "--buffer_n", "3", "--permute"],
                                   ["--gen", "tanNorm", "--normalize"]])
def test_train_cli(tmp_path, extra):
    import train_wgan_v2 as tw
    idir, odir = tmp_path / "sims", tmp_path / "out"
    idir.mkdir()
    _write_sims(str(idir), 96)
    tw.main(["--idir", str(idir), "--odir", str(odir), "--use_cuda", "--num_in", "96", "--batch_size", "8",
             "--epochs", "2", "--save_freq", "1", "--critic_iter", "5"] + extra)
    g = torch.load(odir / "generator_model_0002.pt", weights_only=True)
    d = torch.load(odir / "discriminator_model_0002.pt", weights_only=True)
    want_g = dict(orc.gen_param_shapes())
    want_g.update(orc.gen_buffer_shapes())
    assert {k: tuple(v.shape) for k, v in g.items()} == want_g
    assert {k: tuple(v.shape) for k, v in d.items()} == orc.disc_param_shapes()
    iters = 2 * ((96 // 8) // 5)
    # every save is preceded by the 1000-sample plotting forward (src/train_wgan_v2.py:686-698): two saves so far
    assert int(g["twod_conv_layers.1.num_batches_tracked"]) == 6 * iters + 2
    assert all(bool(torch.isfinite(v.float()).all()) for v in list(g.values()) + list(d.values()))
    import layers
    G = layers.DC_Generator_Leaky_Sig(64, 128, 0.1, 1, 4)
    G.load_state_dict(g)            # interchangeable with the module front-end (and the reference's)


def test_train_cli_writes_2dswd(tmp_path):
    """--plot: the 2DSWD evaluation runs at every saved epoch and is appended to w_dist_pca.txt
    (src/train_wgan_v2.py:723-726) from the genotypes of the plotting forward made before the save."""
    import train_wgan_v2 as tw
    idir, odir = tmp_path / "sims", tmp_path / "out"
    idir.mkdir()
    _write_sims(str(idir), 96)
    tw.main(["--idir", str(idir), "--odir", str(odir), "--use_cuda", "--num_in", "96", "--batch_size", "8",
             "--epochs", "2", "--save_freq", "1", "--critic_iter", "5", "--plot"])
    lines = (odir / "w_dist_pca.txt").read_text().strip().splitlines()
    assert [ln.split("\t")[0] for ln in lines] == ["1", "2"]
    vals = [float(ln.split("\t")[1]) for ln in lines]
    assert all(np.isfinite(v) and v > 0 for v in vals)
    g = torch.load(odir / "generator_model_0002.pt", weights_only=True)
    iters = 2 * ((96 // 8) // 5)
    assert int(g["twod_conv_layers.1.num_batches_tracked"]) == 6 * iters + 2     # the two saves' plotting forwards



def test_train_cli_slab_cache(tmp_path):
    """--slab: the first run parses the CSV directory and writes the packed copy, the second run trains from it with
    the directory gone."""
    import shutil

    import alignment_slab
    import train_wgan_v2 as tw
    idir, odir, slab = tmp_path / "sims", tmp_path / "out", tmp_path / "cache.slab"
    idir.mkdir()
    _write_sims(str(idir), 48)
    argv = ["--idir", str(idir), "--odir", str(odir), "--use_cuda", "--num_in", "48", "--batch_size", "8",
            "--epochs", "1", "--save_freq", "1", "--critic_iter", "5", "--slab", str(slab)]
    tw.main(argv)
    bits, pos = alignment_slab.read_slab(str(slab))
    assert bits.shape == (48, 512) and pos.shape == (48, 64)
    rs, _ = orc.synthetic_alignments(48, 5)
    assert sorted(map(bytes, alignment_slab.pack_numpy(rs[:, 0].numpy()))) == sorted(map(bytes, np.asarray(bits)))
    shutil.rmtree(idir)
    tw.main(argv)
    g = torch.load(odir / "generator_model_0001.pt", weights_only=True)
    assert all(bool(torch.isfinite(v.float()).all()) for v in g.values())


def test_train_cli_scaled_128(tmp_path):
    """--data_size 128: the scaled networks (no counterpart in the reference) from 128 x 128 CSV pairs to checkpoints
    with one more stage per tower; the plotting forward and its genotyping run at that size too."""
    import train_wgan_v2 as tw
    idir, odir = tmp_path / "sims", tmp_path / "out"
    idir.mkdir()
    _write_sims(str(idir), 24, size=128)
    tw.main(["--idir", str(idir), "--odir", str(odir), "--use_cuda", "--num_in", "24", "--batch_size", "4",
             "--epochs", "1", "--save_freq", "1", "--critic_iter", "2", "--data_size", "128", "--plot"])
    g = torch.load(odir / "generator_model_0001.pt", weights_only=True)
    d = torch.load(odir / "discriminator_model_0001.pt", weights_only=True)
    want_g = dict(orc.gen_param_shapes(data=128))
    want_g.update(orc.gen_buffer_shapes(data=128))
    assert {k: tuple(v.shape) for k, v in g.items()} == want_g
    assert {k: tuple(v.shape) for k, v in d.items()} == orc.disc_param_shapes(128)
    assert all(bool(torch.isfinite(v.float()).all()) for v in list(g.values()) + list(d.values()))
    w = float((odir / "w_dist_pca.txt").read_text().split()[1])
    assert np.isfinite(w) and w > 0
