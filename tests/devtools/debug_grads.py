"""Dev tool: per-parameter gradient error of the HIP training path vs torch autograd through the CPU oracle."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import hdiff_amd
from hdiff_amd.DiffusionFreeGuidence import ModelCondition as MC, DiffusionCondition as DC
from oracle import cpu_path as O
G = os.path.join(ROOT, "tests", "golden")
u, d = np.load(os.path.join(G, "unet_small.npz")), np.load(os.path.join(G, "trainer_small.npz"))
c = json.loads(bytes(u["cfg_json"]).decode())
T = lambda a: torch.from_numpy(np.asarray(a))
sd = {k[3:]: T(u[k]) for k in u.files if k.startswith("sd/")}
m = MC.UNet(**c); m.load_state_dict(sd); m = m.to("cuda:0").train()
tr = DC.GaussianDiffusionTrainer(m, 1e-4, 0.028, c["T"]).to("cuda:0")
x0 = T(d["x_0"]); lab = T(d["labels"]); t = T(d["t"]); nz = T(d["noise"])
loss = tr(x0.cuda(), lab.cuda(), t=t.cuda(), noise=nz.cuda())
(loss.sum() / 16.).backward()
cfg = O.UNetConfig(T=c["T"], num_labels=c["num_labels"], ch=c["ch"], ch_mult=tuple(c["ch_mult"]), num_res_blocks=c["num_res_blocks"])
sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
lr = O.trainer_loss(sdr, cfg, O.trainer_schedule(1e-4, 0.028, c["T"]), x0, lab, t, nz)
(lr.sum() / 16.).backward()
for n, p in m.named_parameters():
    ref = sdr[n].grad
    if ref is None:
        print(f"{n:55s} ref None"); continue
    e = (p.grad.cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
    print(f"{n:55s} rel err {e:.2e}  ref max {ref.abs().max().item():.2e}")
