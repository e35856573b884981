import sys, torch
sys.path[:0] = ["/root/repo/pg-alignments-gan_amd", "/root/repo/oracle", "/root/repo/tests"]
import dcwgan_oracle as orc
from pgan_hip import HipOps
from pgan_engine import Engine
dev = torch.device("cuda:0")
hip = HipOps(dev)
B, S = 4, 128
eng = Engine(hip, B, dev, act="sigmoid", data=S)
gp_ = orc.make_params(orc.gen_param_shapes(data=S), 8642); dp = orc.make_params(orc.disc_param_shapes(S), 8643)
eng.G.load(gp_); eng.D.load(dp)
z = torch.randn(B, 128, 1, 1)
eng.Z.copy_(z.reshape(B, 128).to(dev))
eng.side.enabled = eng.wside.enabled = False
eng._gen_forward_1d(eng.gXp); eng._gen_forward_2d(eng.gXs)
torch.cuda.synchronize()
acts = {}
xs, xp = orc.gen_forward(gp_, None, z, "sigmoid", acts)
for i in range(eng.nG):
    print("fwd", i, float((eng.gH[i + 1].cpu() - acts[f"g2d.act{i}"]).abs().max()), float((eng.hH[i + 1].cpu() - acts[f"g1d.act{i}"]).abs().max()))
print("out", float((eng.gXs.cpu() - xs).abs().max()), float((eng.gXp.cpu() - xp).abs().max()))
snap = [t.clone() for t in eng.gH[1:]]
eng.generator_grads_resident(forward_done=True)
torch.cuda.synchronize()
for i in range(eng.nG):
    print("after bwd", i, float((eng.gH[i + 1] - snap[i]).abs().max()))
