"""Drop-in for the reference's ``MainCondition.py``: the same config dict and ``main(model_config=None)`` dispatch."""
from .DiffusionFreeGuidence.TrainCondition import eval, train


def main(model_config=None):
    modelConfig = {
        "state": "train",  # or eval
        "epoch": 70, "batch_size": 80, "T": 500, "channel": 128, "channel_mult": [1, 2, 2, 2], "num_res_blocks": 2,
        "dropout": 0.15, "lr": 1e-4, "multiplier": 2.5, "beta_1": 1e-4, "beta_T": 0.028, "img_size": 32, "grad_clip": 1.,
        "device": "cuda:0", "w": 1.8, "save_dir": "./CheckpointsCondition/", "training_load_weight": None,
        "test_load_weight": "ckpt_63_.pt", "sampled_dir": "./SampledImgs/", "sampledNoisyImgName": "NoisyGuidenceImgs.png",
        "sampledImgName": "SampledGuidenceImgs.png", "nrow": 8,
    }
    if model_config is not None:
        modelConfig = model_config
    if modelConfig["state"] == "train":
        return train(modelConfig)
    return eval(modelConfig)


if __name__ == '__main__':
    main()
