"""Dev tool: gradients of a tiny-image training step (HIP and CPU fp32) against float64 autograd through the oracle."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, hdiff_amd
from hdiff_amd.DiffusionFreeGuidence import DiffusionCondition as DC, ModelCondition as MC
from oracle import cpu_path as O
DEV="cuda:0"
for S in (8, 16):
    torch.manual_seed(21)
    cfgd = dict(T=10, num_labels=4, ch=32, ch_mult=[1, 2, 2, 2], num_res_blocks=1, dropout=0.0)
    m = MC.UNet(**cfgd).train()
    cfg = O.UNetConfig(T=10, num_labels=4, ch=32, ch_mult=(1, 2, 2, 2), num_res_blocks=1)
    g = torch.Generator().manual_seed(4)
    B=3
    x0 = torch.rand(B,3,S,S,generator=g)*2-1; lab=torch.tensor([1,0,3]); t=torch.tensor([0,5,9]); noise=torch.randn(B,3,S,S,generator=g)
    sd = {k: v.detach().clone().double().requires_grad_(True) for k,v in m.state_dict().items()}
    sched=O.trainer_schedule(1e-4,0.02,10)
    rl=O.trainer_loss(sd,cfg,sched,x0.double(),lab,t,noise.double()); (rl.sum()/B**2.).backward()
    sd32 = {k: v.detach().clone().requires_grad_(True) for k,v in m.state_dict().items()}
    rl32=O.trainer_loss(sd32,cfg,sched,x0,lab,t,noise); (rl32.sum()/B**2.).backward()
    md=m.to(DEV); tr=DC.GaussianDiffusionTrainer(md,1e-4,0.02,10).to(DEV)
    loss=tr(x0.to(DEV),lab.to(DEV),t=t.to(DEV),noise=noise.to(DEV)); (loss.sum()/B**2.).backward()
    worst=[]
    for n,p in md.named_parameters():
        gref=sd[n].grad; g32=sd32[n].grad.double(); gh=p.grad.detach().cpu().double()
        sc=gref.abs().max().item()+1e-30
        worst.append((( gh-gref).abs().max().item()/sc, (g32-gref).abs().max().item()/sc, n))
    worst.sort(reverse=True)
    print("S",S,"loss err", (loss.cpu().double()-rl).abs().max().item())
    names = ["head.weight", "downblocks.6.block1.2.weight", "middleblocks.0.attn.in_proj_weight", "upblocks.0.block2.3.weight",
             "upblocks.7.t.weight", "tail.2.bias", "time_embedding.timembedding.1.weight", "cond_embedding.condEmbedding.0.weight"]
    for w in worst:
        if w[2] in names: print("  hip rel err %.2e  cpu-fp32 rel err %.2e  %s"%w, "  max|g| %.3e" % sd[w[2]].grad.abs().max().item())
