"""The reference's data layout (utils/utils.py:41-473) rebuilt on synthetic image trees: folder conventions, the 70/10/20
split and its (train, test, val) return order, item structure, the trainer-facing wrapper."""
import os

import numpy as np
import pytest
import torch

import hdiff_amd  # noqa: F401
from hdiff_amd import datasets as D


def _write(path, seed, size=(20, 30), fmt=None):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rng = np.random.default_rng(seed)
    Image.fromarray(rng.integers(0, 256, size=size + (3,), dtype=np.uint8)).save(path, format=fmt)


def test_split_matches_the_reference_order_and_sizes():
    items = [f"f{i:03d}" for i in range(23)]
    train, test, val = D.split_data(items)
    assert (len(train), len(val), len(test)) == (16, 2, 5)               # int(23*.7), int(23*.1), rest
    assert train == items[:16] and val == items[16:18] and test == items[18:]
    with pytest.raises(ValueError):
        D.split_data(items, 0.5, 0.1, 0.1)


def test_underwater_and_atmospheric_layouts(tmp_path):
    root = str(tmp_path / "HICRD")
    for i in range(4):
        _write(f"{root}/Train/trainA_paired/im{i}.png", i)
        _write(f"{root}/Train/trainB_paired/im{i}.png", 100 + i)
    for i in range(2):
        _write(f"{root}/Test/testA/t{i}.png", 10 + i); _write(f"{root}/Test/testB/t{i}.png", 110 + i)
        _write(f"{root}/Val/valA/v{i}.png", 20 + i); _write(f"{root}/Val/valB/v{i}.png", 120 + i)
    tr = D.Underwater_Dataset("HICRD", task="train", root=root)
    assert len(tr) == 4 and len(D.Underwater_Dataset("HICRD", task="test", root=root)) == 2
    a, b = tr[1]
    assert a.dtype == torch.uint8 and tuple(a.shape) == (3, 256, 256) and tuple(b.shape) == (3, 256, 256)   # Resize(256) + ToTensorV2
    assert not torch.equal(a, b)
    va = D.Underwater_Dataset("HICRD", task="val", root=root)[0]
    assert len(va) == 3 and va[2] == "v0.png"                              # the validation split also names the file
    u = D.Underwater_Dataset("HICRD", task="train", supervised=False, root=root)[0]
    assert torch.equal(u[0], u[1])                                         # unsupervised: the degraded image on both sides
    with pytest.raises(ValueError):
        D.Underwater_Dataset("EUVP", root=root)
    # LSUI: input / GT with the 70/10/20 split; pairing by sorted file name
    lroot = str(tmp_path / "LSUI")
    for i in range(10):
        _write(f"{lroot}/input/{i:02d}.jpg", i, fmt="JPEG"); _write(f"{lroot}/GT/{i:02d}.jpg", 50 + i, fmt="JPEG")
    sizes = [len(D.Underwater_Dataset("LSUI", task=t, root=lroot)) for t in ("train", "val", "test")]
    assert sizes == [7, 1, 2]
    ds = D.Underwater_Dataset("LSUI", task="test", root=lroot)
    assert [os.path.basename(p) for p in ds.paths_a] == [os.path.basename(p) for p in ds.paths_b] == ["08.jpg", "09.jpg"]
    # atmospheric: LoLI low / high folders; a custom albumentations-style transform
    aroot = str(tmp_path / "LoLI")
    for i in range(3):
        _write(f"{aroot}/Train/low/{i}.jpg", i, fmt="JPEG"); _write(f"{aroot}/Train/high/{i}.jpg", 70 + i, fmt="JPEG")
    at = D.Atmospheric_Dataset("LoLI", batch_size=4, transforms=lambda image: {"image": torch.from_numpy(image.copy())},
                               task="train", root=aroot)
    assert len(at) == 3 and at.batch_size == 4 and tuple(at[0][0].shape) == (20, 30, 3)
    # the class-conditional trainer's view: reference images in [-1, 1] with the domain as the label
    both = D.ReferenceImagesWithDomain([tr, D.Atmospheric_Dataset("LoLI", task="train", root=aroot)], img_size=32)
    assert len(both) == 7
    x, lab = both[5]
    assert lab == 1 and tuple(x.shape) == (3, 32, 32) and x.dtype == torch.float32 and -1.0 <= float(x.min()) and float(x.max()) <= 1.0
    # the domain id belongs to the KIND of set, not to its position: an atmospheric-only run labels its images 1 too, and an
    # explicit (set, id) pair overrides
    alone = D.ReferenceImagesWithDomain([D.Atmospheric_Dataset("LoLI", task="train", root=aroot)], img_size=32)
    assert len(alone) == 3 and {alone[i][1] for i in range(3)} == {D.ATMOSPHERIC_DOMAIN} == {1}
    assert {both[i][1] for i in range(4)} == {D.UNDERWATER_DOMAIN} == {0}
    pinned = D.ReferenceImagesWithDomain([(tr, 5)], img_size=32)
    assert pinned[0][1] == 5
    with pytest.raises(FileNotFoundError):
        D.load_image(str(tmp_path / "missing.png"))
