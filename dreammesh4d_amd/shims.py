"""The two threestudio components the shipped configs name next to the hot path: the Gaussian
renderers are constructed with ``(geometry, material, background)`` and only read the background's colour
(threestudio/models/renderers/base.py:28-35), so these are restatements of the two small modules, not of threestudio."""
import random

import torch
import torch.nn as nn


class SolidColorBackground(nn.Module):
    """``solid-color-background`` (threestudio/models/background/solid_color_background.py:14-51)."""

    def __init__(self, n_output_dims=3, color=(1.0, 1.0, 1.0), learned=False, random_aug=False, random_aug_prob=0.5):
        super().__init__()
        self.n_output_dims, self.random_aug, self.random_aug_prob = n_output_dims, random_aug, random_aug_prob
        c = torch.as_tensor(color, dtype=torch.float32)
        if learned:
            self.env_color = nn.Parameter(c)
        else:
            self.register_buffer("env_color", c)

    def forward(self, dirs):
        """dirs [B,H,W,3] -> colours [B,H,W,Nc]."""
        color = torch.ones(*dirs.shape[:-1], self.n_output_dims).to(dirs) * self.env_color.to(dirs)
        if self.training and self.random_aug and random.random() < self.random_aug_prob:
            color = color * 0 + torch.rand(dirs.shape[0], 1, 1, self.n_output_dims).to(dirs).expand(*dirs.shape[:-1], -1)
        return color


class NoMaterial(nn.Module):
    """``no-material`` without a network (threestudio/models/materials/no_material.py:16-54): the activation of the
    features (the configs of the hot path use it as a placeholder; the Gaussian renderers never call it)."""

    _ACT = {"sigmoid": torch.sigmoid, "none": lambda x: x, "relu": torch.relu, "exp": torch.exp, "tanh": torch.tanh}

    def __init__(self, n_output_dims=3, color_activation="sigmoid", requires_normal=False):
        super().__init__()
        if color_activation not in self._ACT:
            raise ValueError(f"unsupported activation {color_activation!r}")
        self.n_output_dims, self.color_activation, self.requires_normal = n_output_dims, color_activation, requires_normal

    def forward(self, features, **kwargs):
        if features.shape[-1] != self.n_output_dims:
            raise AssertionError(f"Expected {self.n_output_dims} output dims, only got {features.shape[-1]} dims input.")
        return self._ACT[self.color_activation](features)
