"""Fixture of the GeneralConditioner parity case. TEST INFRASTRUCTURE ONLY.

The reference's GeneralConditioner (vwm/modules/encoders/modules.py:70-180) is executed as is (oracle/make_golden_cond.py, through
oracle/ref_shim.py with empty `kornia` / `open_clip` stand-in modules, which its module imports but the classes used here never call) over:
its OWN ConcatTimestepEmbedderND for every scalar / action key of configs/inference/vista.yaml:57-140, and two tiny parameter-free
stand-ins for the embedders that need downloaded weights (image tower, first-stage encoder) -- so the key routing, the zero segments of absent
actions, force_zero_embeddings and get_unconditional_conditioning under test are the reference's own code.
"""
import torch

P = "vwm.modules.encoders.modules."
N = 5  # frames (num_samples)


def emb_models(stub_module):
    """The emb_models list of vista.yaml with the two weight-carrying embedders replaced by `stub_module`'s stand-ins."""
    cte = P + "ConcatTimestepEmbedderND"
    return [
        {"input_key": "cond_frames_without_noise", "is_trainable": False, "target": stub_module + ".StubImageEmbedder", "params": {"dim": 1024}},
        {"input_key": "fps_id", "is_trainable": False, "target": cte, "params": {"outdim": 256}},
        {"input_key": "motion_bucket_id", "is_trainable": False, "target": cte, "params": {"outdim": 256}},
        {"input_key": "cond_frames", "is_trainable": False, "target": stub_module + ".StubLatentEmbedder", "params": {}},
        {"input_key": "cond_aug", "is_trainable": False, "target": cte, "params": {"outdim": 256}},
        {"input_key": "command", "is_trainable": False, "target": cte, "params": {"outdim": 128, "num_features": 1, "add_sequence_dim": True}},
        {"input_key": "trajectory", "is_trainable": False, "target": cte, "params": {"outdim": 128, "num_features": 8, "add_sequence_dim": True}},
        {"input_key": "speed", "is_trainable": False, "target": cte, "params": {"outdim": 128, "num_features": 4, "add_sequence_dim": True}},
        {"input_key": "angle", "is_trainable": False, "target": cte, "params": {"outdim": 128, "num_features": 4, "add_sequence_dim": True}},
        {"input_key": "goal", "is_trainable": False, "target": cte, "params": {"outdim": 128, "num_features": 2, "add_sequence_dim": True}},
    ]


def stub_image_embed(img, dim):
    """(n, 3, H, W) -> (n, 1, dim): a fixed nonlinear function of the image (stands in for the OpenCLIP tower + PredictionEmbedder)."""
    pooled = torch.nn.functional.adaptive_avg_pool2d(img.float(), (2, 4)).flatten(1)                  # (n, 24)
    k = torch.arange(dim, dtype=torch.float32, device=img.device)[None] * 0.01 + 0.3
    return torch.sin(pooled.sum(1, keepdim=True) * k + pooled[:, :1] * 2.0)[:, None, :]


def value_dict(device="cpu"):
    g = torch.Generator().manual_seed(11)
    return {"cond_frames_without_noise": torch.tanh(torch.randn(1, 3, 32, 64, generator=g)).to(device),
            "cond_frames": torch.randn(1, 4, 4, 8, generator=g).to(device),
            "trajectory": torch.tensor([0.5, 0.0, 1.0, 0.0, 1.5, 0.1, 2.0, 0.2]).to(device), "speed": torch.tensor([3.0, 3.5, 4.0, 4.5]).to(device),
            "fps_id": 9.0, "motion_bucket_id": 127.0, "cond_aug": 0.02}


FORCE_UC_ZERO = ["cond_frames", "cond_frames_without_noise", "command", "trajectory", "speed", "angle", "goal"]  # sample.py:243
