"""Developer probe (round 4: found the launch-size dependence of the epilogue): classify errors of rank 0's share of a 3-rank deal (config.simulate_rank) against the single-process errors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import diffusion_classifier_amd as dca

def run(sim):
    cfg = dict(pred_param="eps", schedule="cosine", noise_d=32, image_size=32, cfg_w=0.0, ema_beta=0.999, ema_warmup=0,
               ema_update_freq=1, encoder_type="nn", classes=10, n_stages=1, evaluation_per_stage=[3], n_keep_per_stage=[1],
               n_fast_classes=2, compute_dtype="bf16", simulate_rank=sim)
    torch.manual_seed(5)
    m = dca.UNetCondition2D(**dca.cifar10_unet_kwargs())
    dc = dca.DiffusionClassifier(m, dca.Config(**cfg)).to("cuda:0")
    torch.manual_seed(6)
    BS, T = 3, 3
    x = (torch.rand(BS, 3, 32, 32) * 2 - 1).to("cuda:0")
    t, eps = torch.rand(7, BS)[:T], torch.randn(7, BS, 3, 32, 32).to("cuda:0")[:T]
    lab, err = dc.classify(x, t=t, eps=eps, return_errors=True)
    return err.numpy()

one = run(None)
for r in range(3):
    part = run((r, 3))
    fin = np.isfinite(part)
    d = (part[fin] != one[fin])
    print("rank", r, "cells", fin.sum(), "differing", d.sum(), "where", np.argwhere(fin & (part != one))[:5].tolist())
