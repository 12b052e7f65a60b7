"""Minimal image I/O for the train / eval harness (the reference uses torchvision, which this image does not ship):
a folder dataset of (clean image, domain label) pairs and a ``save_image`` grid writer."""
from __future__ import annotations

import os
from typing import List, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".ppm")


class ImageDomainFolder(Dataset):
    """``root/<domain>/*.png``: x_0 = the image resized to img_size and normalised to [-1, 1] (the reference's
    ToTensor + Normalize(0.5, 0.5), TrainCondition.py:25-28); label = index of the domain folder (0-based; the
    train loop adds 1 because label 0 is the unconditional token, TrainCondition.py:56)."""

    def __init__(self, root: str, img_size: int):
        from PIL import Image  # noqa: F401  (import check)
        self.img_size = img_size
        self.domains = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
        self.items: List[Tuple[str, int]] = []
        for li, dname in enumerate(self.domains):
            for f in sorted(os.listdir(os.path.join(root, dname))):
                if f.lower().endswith(_EXT):
                    self.items.append((os.path.join(root, dname, f), li))
        if not self.items:
            raise FileNotFoundError(f"no images under {root}/<domain>/")

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        from PIL import Image
        path, label = self.items[i]
        img = Image.open(path).convert("RGB").resize((self.img_size, self.img_size), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
        return (x - 0.5) / 0.5, label


class SyntheticDomains(Dataset):
    """Deterministic synthetic (image, label) pairs in [-1, 1] -- smooth colour fields whose statistics depend on the label
    (stands in for underwater / atmospheric image sets when no data is on disk)."""

    def __init__(self, n: int, img_size: int, num_labels: int, seed: int = 0):
        self.n, self.img_size, self.num_labels, self.seed = n, img_size, num_labels, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        label = i % self.num_labels
        s = self.img_size
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, s), torch.linspace(-1, 1, s), indexing="ij")
        ph = torch.rand(3, 2, generator=g) * 6.283
        fr = 1.0 + label + torch.rand(3, generator=g)
        img = torch.stack([torch.sin(fr[c] * xx + ph[c, 0]) * torch.cos(fr[c] * yy + ph[c, 1]) for c in range(3)])
        tint = torch.tensor([0.2 * label, -0.1 * label, 0.1]).view(3, 1, 1)
        return (0.7 * img + tint).clamp(-1, 1), label


def save_image(tensor: torch.Tensor, path: str, nrow: int = 8, padding: int = 2) -> None:
    """Grid of a [N,3,H,W] batch in [0,1] written as PNG (what torchvision.utils.save_image does in TrainCondition.py:101-108)."""
    from PIL import Image
    t = tensor.detach().float().cpu().clamp(0, 1)
    n, c, h, w = t.shape
    ncol = min(nrow, n)
    nrows = (n + ncol - 1) // ncol
    grid = torch.zeros(c, nrows * (h + padding) + padding, ncol * (w + padding) + padding)
    for i in range(n):
        r, col = divmod(i, ncol)
        y0, x0 = padding + r * (h + padding), padding + col * (w + padding)
        grid[:, y0:y0 + h, x0:x0 + w] = t[i]
    arr = (grid * 255.0 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(arr).save(path)
