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


def hip_preds(dc, T, BS):
    """Backbone outputs of the LAST classify call of `dc` on the HIP path, as [T, BS, class columns, C, H, W] (the layout of the
    oracle's `return_preds`): read from the score plan's prediction buffer.  Needs a single-stage grid that fitted ONE
    micro-batch (pairs trial-major, units = pair x class: diffusion_classifier.py _HipRunner.run_stage)."""
    (sp,) = list(dc._score_plans.values())
    assert sp["n_bj"] == T * BS, "the grid must fit one micro-batch"
    plan, k = sp["plan"], sp["k"]
    pv = plan.pred_view().float().cpu()                         # [U, H, W, ld]
    bb = dc.ema.ema_model
    p = int(getattr(bb.config, "patch_size", 0) or 0)
    if p > 1:                                                   # DiT: [U, g, g, (py*p+px)*C + c] -> [U, C, H, W]
        U, g, _, _ = pv.shape
        oc = bb.config.out_channels
        img = pv[..., :p * p * oc].reshape(U, g, g, p, p, oc).permute(0, 5, 1, 3, 2, 4).reshape(U, oc, g * p, g * p)
    else:
        img = pv[..., :bb.config.out_channels].permute(0, 3, 1, 2)
    return img.reshape(T, BS, k, *img.shape[1:])


def pred_rel_l2(got, ref):
    """Largest per-sample relative L2 error of the predictions (every (trial, image, class) sample on its own)."""
    g, r = got.double().flatten(3), ref.double().flatten(3)
    return float(((g - r).norm(dim=3) / r.norm(dim=3).clamp_min(1e-30)).max())
