"""Tiny deterministic stand-in backbone used ONLY to pin the scoring loop.

The golden capture (tools/capture_goldens.py) drives the reference's own `classify` with
this module; the tests drive the oracle's and the product's `classify` with the same
module and weights (stored in the fixture), so any difference is the loop's.
It follows the backbone call convention of reference diffusion_classifier.py:700-704.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn


class TinyBackbone(nn.Module):
    def __init__(self, ch=3, hid=8, n_classes=4, mode="nn"):
        super().__init__()
        self.mode = mode
        self.config = SimpleNamespace(encoder_hid_dim=hid)
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)
        self.tl = nn.Linear(1, ch)
        if mode == "nn":
            self.lin = nn.Linear(hid, ch)
        else:
            self.table = nn.Embedding(n_classes + 1, ch)

    def forward(self, x, noise_labels, encoder_hidden_states=None):
        if self.mode == "nn":
            cvec = self.lin(encoder_hidden_states[:, 0])
        else:
            cvec = self.table(encoder_hidden_states)
        tvec = self.tl(noise_labels[:, None].float())
        gain = 1.0 + 0.5 * torch.tanh(cvec + tvec)
        return self.conv(x) * gain[:, :, None, None] + torch.tanh(cvec)[:, :, None, None] * x
