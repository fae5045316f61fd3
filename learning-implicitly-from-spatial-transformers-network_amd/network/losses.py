"""SDF loss of the reference (network/losses.py:6-38)."""
import torch
import torch.nn as nn


class SDFLoss(nn.Module):
    def __init__(self, sdf_scale):
        super().__init__()
        self.sdf_threshold = 0.01
        self.sdf_near_surface_weight = 4.0
        self.sdf_scale = sdf_scale
        self.sdf_coefficient = 1000.0

    def forward(self, outputs, targets):
        err = targets * self.sdf_scale - outputs
        per_image = (err * err).sum(-1)                               # sum over points
        real = targets - outputs / self.sdf_scale
        same_side = torch.eq(targets > 0.5, outputs > 0.5).float()
        return {"sdf_loss": per_image.mean(),                          # mean over the batch
                "ignore_sdf_loss_realvalue": (real * real).mean() * 10000,
                "ignore_sdf_accuracy": same_side.mean()}
