"""End-to-end run of the command-line trainer on synthetic ms-style alignments (run on the GPU box): writes the CSV
pairs, trains with --plot and prints the 2DSWD of every saved epoch (w_dist_pca.txt) with the wall time."""
import os, sys, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pg-alignments-gan_amd")]
import numpy as np
from bench import synthetic_pool
import train_wgan_v2 as tw
n, epochs, every = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
idir, odir = "/tmp/curve_sims", "/tmp/curve_out"
shutil.rmtree(idir, ignore_errors=True); os.makedirs(idir)
rs, rp = synthetic_pool(n, 11)
for i in range(n):
    np.savetxt(f"{idir}/{i}_sites.csv", rs[i, 0].numpy().astype(int), fmt="%d", delimiter=",")
    np.savetxt(f"{idir}/{i}_pos.csv", rp[i, 0].numpy(), fmt="%.6f")
t0 = time.time()
tw.main(["--idir", idir, "--odir", odir, "--use_cuda", "--num_in", str(n), "--batch_size", "64", "--epochs", str(epochs),
         "--save_freq", str(every), "--plot", "--slab", "/tmp/curve.slab"] + sys.argv[4:])
dt = time.time() - t0
iters = epochs * max(1, (n // 64) // 5)
print(f"{epochs} epochs = {iters} iterations ({iters * 5 * 64} real alignments) in {dt:.1f} s wall, evaluation and checkpoints included")
print(open(os.path.join(odir, "w_dist_pca.txt")).read())
