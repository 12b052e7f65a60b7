"""Drop-in for the reference's ``DiffusionFreeGuidence/TrainCondition.py``: ``train(modelConfig)`` and ``eval(modelConfig)``
with the same config keys (MainCondition.py:5-29) and the same loop semantics:

  train: AdamW(lr, weight_decay=1e-4); CosineAnnealingLR(T_max=epoch) behind GradualWarmupScheduler(multiplier,
         warm_epoch=epoch//10), stepped per epoch; labels + 1; with probability 0.1 the WHOLE batch's labels are zeroed
         (host numpy RNG); loss = trainer(x_0, labels).sum() / b**2; clip_grad_norm_(grad_clip); one weights-only
         checkpoint ckpt_<e>_.pt per epoch (plain state_dict, loadable by the reference)            [reference :20-72]
  eval:  labels in batch_size//10 contiguous groups (+1); strict load of test_load_weight; sampler from x_T ~ N(0, I);
         noisy and sampled grids saved as PNG (x*0.5+0.5)                                             [reference :75-108]

Differences: the data set.  The reference hard-codes torchvision CIFAR10; here ``modelConfig["dataset"]`` selects
"folder" (``data_dir/<domain>/*.png``: clean image = x_0, domain index = label), "synthetic", or "cifar10" (only if
torchvision is installed).  Optional keys, all with reference-equivalent defaults: ``num_labels`` (10), ``num_workers`` (4),
``max_steps_per_epoch``.  Under ``torch.distributed.run`` (WORLD_SIZE > 1) training is data-parallel: replicated weights,
per-rank shard of every epoch, ONE mean all-reduce of the gradients per step (hdiff_amd.parallel), rank-0 checkpoints.
"""
import os
from typing import Dict

import numpy as np
import torch
from torch.utils.data import DataLoader, Subset

from .. import parallel
from ..Scheduler import GradualWarmupScheduler
from ..imageio import ImageDomainFolder, SyntheticDomains, save_image
from .DiffusionCondition import GaussianDiffusionSampler, GaussianDiffusionTrainer
from .ModelCondition import UNet

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(it, **_):
        return it


def _dataset(cfg: Dict):
    kind = cfg.get("dataset", "folder" if cfg.get("data_dir") else "synthetic")
    if kind == "folder":
        return ImageDomainFolder(cfg["data_dir"], cfg["img_size"])
    if kind == "synthetic":
        return SyntheticDomains(cfg.get("synthetic_size", 512), cfg["img_size"], cfg.get("num_labels", 10))
    if kind == "cifar10":
        from torchvision import transforms
        from torchvision.datasets import CIFAR10
        return CIFAR10(root='./CIFAR10', train=True, download=True,
                       transform=transforms.Compose([transforms.ToTensor(),
                                                     transforms.Normalize((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))]))
    raise ValueError(f"unknown dataset kind {kind!r}")


def _epoch_indices(n: int, epoch: int, rank: int, world: int, seed: int = 0):
    """DistributedSampler semantics: one permutation per epoch shared by all ranks, padded to a multiple of world, strided."""
    g = torch.Generator().manual_seed(seed + epoch)
    perm = torch.randperm(n, generator=g).tolist()
    if world > 1:
        total = (n + world - 1) // world * world
        perm = (perm + perm[: total - n])[rank:total:world]
    return perm


def train(modelConfig: Dict):
    rank, local, world = parallel.init_from_env()
    device = torch.device(modelConfig["device"]) if world == 1 else torch.device("cuda", local)
    dataset = _dataset(modelConfig)
    num_labels = modelConfig.get("num_labels", 10)

    net_model = UNet(T=modelConfig["T"], num_labels=num_labels, ch=modelConfig["channel"],
                     ch_mult=modelConfig["channel_mult"], num_res_blocks=modelConfig["num_res_blocks"],
                     dropout=modelConfig["dropout"]).to(device)
    if modelConfig["training_load_weight"] is not None:
        net_model.load_state_dict(torch.load(os.path.join(modelConfig["save_dir"], modelConfig["training_load_weight"]),
                                             map_location=device), strict=False)
        print("Model weight load down.")
    parallel.broadcast_parameters_(net_model.parameters())
    optimizer = torch.optim.AdamW(net_model.parameters(), lr=modelConfig["lr"], weight_decay=1e-4)
    cosine = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer=optimizer, T_max=modelConfig["epoch"], eta_min=0,
                                                        last_epoch=-1)
    warmup = GradualWarmupScheduler(optimizer=optimizer, multiplier=modelConfig["multiplier"],
                                    warm_epoch=modelConfig["epoch"] // 10, after_scheduler=cosine)
    trainer = GaussianDiffusionTrainer(net_model, modelConfig["beta_1"], modelConfig["beta_T"], modelConfig["T"]).to(device)
    params = [p for p in net_model.parameters()]
    os.makedirs(modelConfig["save_dir"], exist_ok=True)
    history = []

    for e in range(modelConfig["epoch"]):
        idx = _epoch_indices(len(dataset), e, rank, world)
        loader = DataLoader(Subset(dataset, idx), batch_size=modelConfig["batch_size"], shuffle=False,
                            num_workers=modelConfig.get("num_workers", 4), drop_last=True, pin_memory=True)
        bar = tqdm(loader, dynamic_ncols=True, disable=rank != 0)
        for step, (images, labels) in enumerate(bar):
            if step >= modelConfig.get("max_steps_per_epoch", 1 << 30):
                break
            b = images.shape[0]
            optimizer.zero_grad()
            x_0 = images.to(device)
            labels = torch.as_tensor(labels).to(device) + 1
            if np.random.rand() < 0.1:
                labels = torch.zeros_like(labels)
            loss = trainer(x_0, labels).sum() / b ** 2.
            loss.backward()
            parallel.allreduce_mean_grads_(params)
            torch.nn.utils.clip_grad_norm_(net_model.parameters(), modelConfig["grad_clip"])
            optimizer.step()
            lv = loss.item()
            history.append(lv)
            if hasattr(bar, "set_postfix"):
                bar.set_postfix(ordered_dict={"epoch": e, "loss: ": lv, "img shape: ": tuple(x_0.shape),
                                              "LR": optimizer.state_dict()['param_groups'][0]["lr"]})
        warmup.step()
        if rank == 0:
            torch.save(net_model.state_dict(), os.path.join(modelConfig["save_dir"], 'ckpt_' + str(e) + "_.pt"))
    return history


def eval(modelConfig: Dict):
    device = torch.device(modelConfig["device"])
    num_labels = modelConfig.get("num_labels", 10)
    with torch.no_grad():
        step = max(1, int(modelConfig["batch_size"] // 10))
        labelList, k = [], 0
        for i in range(1, modelConfig["batch_size"] + 1):
            labelList.append(torch.ones(size=[1]).long() * k)
            if i % step == 0 and k < min(10, num_labels) - 1:
                k += 1
        labels = torch.cat(labelList, dim=0).long().to(device) + 1
        print("labels: ", labels)
        model = UNet(T=modelConfig["T"], num_labels=num_labels, ch=modelConfig["channel"],
                     ch_mult=modelConfig["channel_mult"], num_res_blocks=modelConfig["num_res_blocks"],
                     dropout=modelConfig["dropout"]).to(device)
        ckpt = torch.load(os.path.join(modelConfig["save_dir"], modelConfig["test_load_weight"]), map_location=device)
        model.load_state_dict(ckpt)
        print("model load weight done.")
        model.eval()
        sampler = GaussianDiffusionSampler(model, modelConfig["beta_1"], modelConfig["beta_T"], modelConfig["T"],
                                           w=modelConfig["w"]).to(device)
        noisyImage = torch.randn(size=[modelConfig["batch_size"], 3, modelConfig["img_size"], modelConfig["img_size"]],
                                 device=device)
        saveNoisy = torch.clamp(noisyImage * 0.5 + 0.5, 0, 1)
        save_image(saveNoisy, os.path.join(modelConfig["sampled_dir"], modelConfig["sampledNoisyImgName"]),
                   nrow=modelConfig["nrow"])
        sampledImgs = sampler(noisyImage, labels)
        sampledImgs = sampledImgs * 0.5 + 0.5  # [0 ~ 1]
        save_image(sampledImgs, os.path.join(modelConfig["sampled_dir"], modelConfig["sampledImgName"]),
                   nrow=modelConfig["nrow"])
        return sampledImgs
