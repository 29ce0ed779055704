"""Shared test helpers: rebuild the stand-in backbone / configs stored in a golden fixture."""
import os

import numpy as np
import torch

from standin import TinyBackbone

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["1stage_eps", "2stage_pruned", "fast", "v_shifted", "dit_labels"]


def load_case(name):
    g = dict(np.load(os.path.join(GOLDEN, f"classify_{name}.npz"), allow_pickle=False))
    cfg = {}
    for k in list(g):
        if k.startswith("cfg."):
            v = g[k]
            cfg[k[4:]] = v.tolist() if v.ndim else v.item()
    return g, cfg


def standin_from(g, cfg):
    mode = "nn" if cfg["encoder_type"] == "nn" else "DiT"
    ch = g["x"].shape[1]
    hid = g["encoder.weight"].shape[1] if "encoder.weight" in g else 8
    bb = TinyBackbone(ch=ch, hid=hid, n_classes=cfg["classes"], mode=mode)
    bb.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("bb.")})
    return bb
