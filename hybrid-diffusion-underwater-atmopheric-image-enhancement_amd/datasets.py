"""The reference's on-disk data layout for its underwater / atmospheric image sets (reference
``utils/utils.py:41-296`` path loaders + split, ``:309-473`` ``Atmospheric_Dataset`` / ``Underwater_Dataset``).  CPU-side IO,
not part of the hot path; it lets a user of the reference point this package at the same ``data/`` tree.

Same directory conventions, same 70 / 10 / 20 unshuffled train / val / test split, same class names, constructor arguments
and item structure (``(degraded, reference)`` pairs of uint8 CHW tensors resized to 256 x 256 = ``A.Resize(256, 256)`` +
``ToTensorV2()``; unsupervised mode returns the degraded image twice; the underwater validation split also returns the file name).
Differences, on purpose: file lists are SORTED (the reference pairs the two lists by index in ``glob`` order, which is only
right when the file system happens to list both folders alike), images are decoded with Pillow instead of OpenCV (absent in
this image; bilinear resize -- last-bit differences, parity unpinned), and a custom ``transforms`` callable receives and
returns ``{"image": HWC uint8 array}`` like an albumentations pipeline.
"""
from __future__ import annotations

import glob
import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

Split = Tuple[List[str], List[str], List[str]]          # (train, test, val) -- the reference's return order


def split_data(data_list: Sequence[str], train_ratio: float = 0.7, val_ratio: float = 0.1, test_ratio: float = 0.2) -> Split:
    """First 70 % train, next 10 % validation, rest test; returned as (train, test, val) (reference utils.py:41-77)."""
    if abs(train_ratio + val_ratio + test_ratio - 1.0) >= 1e-6:
        raise ValueError("train, validation and test ratios must sum to 1")
    items = list(data_list)
    n_train, n_val = int(len(items) * train_ratio), int(len(items) * val_ratio)
    return items[:n_train], items[n_train + n_val:], items[n_train:n_train + n_val]


def _files(root: str, *pattern: str) -> List[str]:
    return sorted(glob.glob(os.path.join(root, *pattern)))


# name -> (default root, degraded-side loader, reference-side loader).  A loader maps the root to (train, test, val).
def _fixed(sub_train: str, sub_test: str, sub_val: str, ext: str) -> Callable[[str], Split]:
    return lambda root: (_files(root, sub_train, ext), _files(root, sub_test, ext), _files(root, sub_val, ext))


def _split_of(*pattern: str) -> Callable[[str], Split]:
    return lambda root: split_data(_files(root, *pattern))


UNDERWATER_SETS: Dict[str, Tuple[str, Callable[[str], Split], Callable[[str], Split]]] = {
    # utils.py:139-176: fixed Train / Test / Val folders, A = degraded, B = reference
    "HICRD": ("data/HICRD", _fixed("Train/trainA_paired", "Test/testA", "Val/valA", "*.png"),
              _fixed("Train/trainB_paired", "Test/testB", "Val/valB", "*.png")),
    # utils.py:179-193: input / GT, split 70 / 10 / 20
    "LSUI": ("data/LSUI", _split_of("input", "*.jpg"), _split_of("GT", "*.jpg")),
    # utils.py:202-208: no references: the degraded images stand on both sides
    "UIEB": ("data/UIEB", _split_of("train", "*.png"), _split_of("train", "*.png")),
    # utils.py:210-224
    "RUIE": ("data/RUIE", _split_of("*", "train", "*.jpg"), _split_of("*", "train", "*.jpg")),
}
ATMOSPHERIC_SETS: Dict[str, Tuple[str, Callable[[str], Split], Callable[[str], Split]]] = {
    # utils.py:107-137
    "HDR": ("data/HDR+ Burst_20171106_subset", _split_of("gallery_20171023", "*.jpg"), _split_of("results_20161014", "*", "*.jpg")),
    # utils.py:195-201: no references
    "TM-DIED": ("data/TM-DIED", _split_of("*.jpg"), _split_of("*.jpg")),
    # utils.py:226-282
    "LoLI": ("data/LoLI", _fixed("Train/low", "Test/low", "Val/low", "*.jpg"), _fixed("Train/high", "Test/high", "Val/high", "*.jpg")),
}


def load_image(image_path: str) -> np.ndarray:
    """RGB uint8 HWC array (reference utils.py:284-306, cv2.imread + BGR->RGB)."""
    from PIL import Image
    try:
        with Image.open(image_path) as im:
            return np.asarray(im.convert("RGB"), dtype=np.uint8)
    except (FileNotFoundError, OSError) as e:
        raise FileNotFoundError(f"could not load the image: {image_path}") from e


def _default_transform(image: np.ndarray) -> Dict[str, torch.Tensor]:
    """A.Compose([A.Resize(256, 256), ToTensorV2()]): bilinear resize, HWC uint8 -> CHW uint8 tensor (no scaling)."""
    from PIL import Image
    img = Image.fromarray(image).resize((256, 256), Image.BILINEAR)
    return {"image": torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1)}


class _PairedSet(Dataset):
    _SETS: Dict[str, Tuple[str, Callable[[str], Split], Callable[[str], Split]]] = {}
    _NAME_WITH_VAL = False

    def __init__(self, name: str, transforms=None, task: str = "train", supervised: bool = True, root: Optional[str] = None):
        if name not in self._SETS:
            raise ValueError(f"Dataset {name} not found. Choose between {sorted(self._SETS)}")
        default_root, load_a, load_b = self._SETS[name]
        self.root = root if root is not None else default_root
        self.task, self.supervised = task, supervised
        self.transform = (lambda image: transforms(image=image)) if transforms is not None else _default_transform
        a, b = load_a(self.root), load_b(self.root)
        which = {"train": 0, "test": 1}.get(task, 2)
        self.paths_a, self.paths_b = a[which], b[which]

    def __len__(self):
        return len(self.paths_a)

    def __getitem__(self, idx):
        img_a = self.transform(image=load_image(self.paths_a[idx]))["image"]
        if not self.supervised:
            return img_a, img_a
        img_b = self.transform(image=load_image(self.paths_b[idx]))["image"]
        if self._NAME_WITH_VAL and self.task not in ("train", "test"):
            return img_a, img_b, self.paths_a[idx].split("/")[-1]
        return img_a, img_b


class Underwater_Dataset(_PairedSet):
    """``Underwater_Dataset(underwater_dataset_name, transforms=None, task="train", supervised=True)`` (utils.py:394-473)."""
    _SETS = UNDERWATER_SETS
    _NAME_WITH_VAL = True

    def __init__(self, underwater_dataset_name: str, transforms=None, task: str = "train", supervised: bool = True,
                 root: Optional[str] = None):
        super().__init__(underwater_dataset_name, transforms, task, supervised, root)
        self.underwater_dataset_name = underwater_dataset_name


class Atmospheric_Dataset(_PairedSet):
    """``Atmospheric_Dataset(atmospheric_dataset_name, batch_size=8, transforms=None, task="train", supervised=True)``
    (utils.py:309-391)."""
    _SETS = ATMOSPHERIC_SETS

    def __init__(self, atmospheric_dataset_name: str, batch_size: int = 8, transforms=None, task: str = "train",
                 supervised: bool = True, root: Optional[str] = None):
        super().__init__(atmospheric_dataset_name, transforms, task, supervised, root)
        self.dataset_name, self.batch_size = atmospheric_dataset_name, batch_size


UNDERWATER_DOMAIN, ATMOSPHERIC_DOMAIN = 0, 1      # fixed class ids: a checkpoint trained on one domain samples with the same label later


class ReferenceImagesWithDomain(Dataset):
    """What the class-conditional trainer consumes (TrainCondition.py:27-30,55-56: an image in [-1, 1] and an integer
    label): the REFERENCE side of the paired sets, images resized to ``img_size``.  ``sets`` holds ``(set, domain_id)``
    pairs; a bare set gets the id of its kind (``Underwater_Dataset`` -> 0, ``Atmospheric_Dataset`` -> 1) -- never its position
    in the list, so an atmospheric-only run and a two-domain run agree on what label 1 (class 2 after the trainer's + 1) means."""

    def __init__(self, sets: Sequence, img_size: int):
        self.items = []
        for entry in sets:
            s, dom = entry if isinstance(entry, tuple) else (entry, None)
            if dom is None:
                dom = ATMOSPHERIC_DOMAIN if isinstance(s, Atmospheric_Dataset) else UNDERWATER_DOMAIN
            self.items += [(p, int(dom)) for p in s.paths_b]
        self.img_size = img_size

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        from PIL import Image
        path, label = self.items[i]
        img = Image.fromarray(load_image(path)).resize((self.img_size, self.img_size), Image.BILINEAR)
        x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
        return (x - 0.5) / 0.5, label
