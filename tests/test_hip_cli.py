"""The command-line trainer end to end on the GPU: CSV directory in, reference-named checkpoints out."""
import os

import numpy as np
import pytest
import torch

import dcwgan_oracle as orc

pytestmark = pytest.mark.gpu


def _write_sims(tmp, n):
    rs, rp = orc.synthetic_alignments(n, 5)
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
    assert int(g["twod_conv_layers.1.num_batches_tracked"]) == 6 * iters
    assert all(bool(torch.isfinite(v.float()).all()) for v in list(g.values()) + list(d.values()))
    import layers
    G = layers.DC_Generator_Leaky_Sig(64, 128, 0.1, 1, 4)
    G.load_state_dict(g)            # interchangeable with the module front-end (and the reference's)
