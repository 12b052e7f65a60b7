"""Drop-in for the reference's ``MainCondition.py``: ``main(model_config=None)`` runs ``train`` or ``eval`` of
``DiffusionFreeGuidence/TrainCondition.py`` on a config dict; without an argument it uses the reference's defaults
(MainCondition.py:5-29), reproduced in ``DEFAULTS`` with what each key controls."""
from .DiffusionFreeGuidence.TrainCondition import eval, train

DEFAULTS = (
    # key, default, meaning
    ("state", "train", "which entry point runs: 'train' or 'eval'"),
    ("epoch", 70, "training epochs (also T_max of the cosine schedule; warm-up lasts epoch // 10)"),
    ("batch_size", 80, "per-process batch (eval: number of images sampled)"),
    ("T", 500, "diffusion steps"),
    ("channel", 128, "base width of the U-Net"),
    ("channel_mult", [1, 2, 2, 2], "width multiplier per resolution level"),
    ("num_res_blocks", 2, "residual blocks per level"),
    ("dropout", 0.15, "dropout inside the residual blocks"),
    ("lr", 1e-4, "AdamW base learning rate"),
    ("multiplier", 2.5, "warm-up target = lr * multiplier"),
    ("beta_1", 1e-4, "first noise-schedule value"),
    ("beta_T", 0.028, "last noise-schedule value"),
    ("img_size", 32, "image side"),
    ("grad_clip", 1., "max gradient norm"),
    ("device", "cuda:0", "single-process device"),
    ("w", 1.8, "classifier-free guidance strength"),
    ("save_dir", "./CheckpointsCondition/", "where checkpoints go / come from"),
    ("training_load_weight", None, "checkpoint to resume from (non-strict load)"),
    ("test_load_weight", "ckpt_63_.pt", "checkpoint that eval samples with"),
    ("sampled_dir", "./SampledImgs/", "where eval writes its PNG grids"),
    ("sampledNoisyImgName", "NoisyGuidenceImgs.png", "grid of the start noise"),
    ("sampledImgName", "SampledGuidenceImgs.png", "grid of the samples"),
    ("nrow", 8, "images per grid row"),
)


def default_config() -> dict:
    return {key: (list(value) if isinstance(value, list) else value) for key, value, _ in DEFAULTS}


def main(model_config=None):
    config = default_config() if model_config is None else model_config
    entry = train if config["state"] == "train" else eval
    return entry(config)


if __name__ == '__main__':
    main()
