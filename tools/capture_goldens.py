#!/usr/bin/env python3
"""Capture golden vectors from the REAL reference scoring loop.  Runs ONLY in the build
container (needs /root/reference); the GPU box never runs it.  Output: small .npz files
under tests/golden/ (data only — inputs, weights of the stand-in backbone, RNG draws,
expected outputs).  No reference source is copied.

Recipe: import torch+accelerate first, then register two
arithmetic-free stub modules for the absent `comet_ml` (logging SaaS) and `ema_pytorch`
(the EMA wrapper = deepcopy + forward-to-copy), then import the reference's
`diffusion.diffusion_classifier`.

DWT goldens come from pywt 1.1.1 under /opt/conda/bin/python3.9 (tools/capture_dwt_goldens.py).
"""
import copy
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import accelerate  # noqa: F401  (must precede the stubs, see Appendix B step 1)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

comet = types.ModuleType("comet_ml")
comet.Experiment = type("Experiment", (), {})
comet.ExistingExperiment = type("ExistingExperiment", (), {})
sys.modules["comet_ml"] = comet


class _EMA(nn.Module):
    def __init__(self, model, beta=None, update_after_step=None, update_every=None):
        super().__init__()
        self.ema_model = copy.deepcopy(model)

    def forward(self, *a, **k):
        return self.ema_model(*a, **k)


ema = types.ModuleType("ema_pytorch")
ema.EMA = _EMA
sys.modules["ema_pytorch"] = ema

sys.path.insert(0, "/root/reference")
from diffusion.diffusion_classifier import DiffusionClassifier  # noqa: E402  (the reference)
from standin import TinyBackbone  # noqa: E402


class Bag:
    def __init__(self, **kw):
        self.__dict__["d"] = kw

    def __getattr__(self, k):
        return self.__dict__["d"].get(k)


OUT = os.path.join(ROOT, "tests", "golden")


def base_cfg(**over):
    d = dict(pred_param="eps", schedule="cosine", noise_d=32, image_size=32, cfg_w=0.0,
             ema_beta=0.999, ema_warmup=0, ema_update_freq=1, encoder_type="nn", classes=4,
             n_stages=1, evaluation_per_stage=[6], n_keep_per_stage=[1], n_fast_classes=2)
    d.update(over)
    return d


def capture_case(name, cfg_over, fast=False, seed=7, BS=5, ch=3, hw=8, hid=8):
    cfgd = base_cfg(**cfg_over)
    mode = "nn" if cfgd["encoder_type"] == "nn" else "DiT"
    torch.manual_seed(100 + seed)
    bb = TinyBackbone(ch=ch, hid=hid, n_classes=cfgd["classes"], mode=mode)
    dc = DiffusionClassifier(bb, Bag(**cfgd))
    x = torch.rand(BS, ch, hw, hw) * 2 - 1
    labels = torch.randint(0, cfgd["classes"], (BS,))

    rec = {"t": [], "eps": [], "means": [], "errors": None, "sel": None}
    o_rand, o_randn_like, o_topk, o_full, o_randint = torch.rand, torch.randn_like, torch.topk, torch.full, torch.randint

    def rand(*a, **k):
        r = o_rand(*a, **k); rec["t"].append(r.clone()); return r

    def randn_like(*a, **k):
        r = o_randn_like(*a, **k); rec["eps"].append(r.clone()); return r

    def topk(inp, *a, **k):
        rec["means"].append(inp.clone()); return o_topk(inp, *a, **k)

    def full(*a, **k):
        r = o_full(*a, **k); rec["errors"] = r; return r

    def randint(*a, **k):
        r = o_randint(*a, **k); rec["sel"] = r.clone(); return r

    torch.manual_seed(seed)
    torch.rand, torch.randn_like, torch.topk, torch.full, torch.randint = rand, randn_like, topk, full, randint
    try:
        out = dc.classify(x, labels if fast else None, fast=fast)
    finally:
        torch.rand, torch.randn_like, torch.topk, torch.full, torch.randint = o_rand, o_randn_like, o_topk, o_full, o_randint

    arrs = {
        "x": x.numpy(), "labels": labels.numpy(), "out": out.numpy(),
        "t": torch.stack(rec["t"]).numpy(), "eps": torch.stack(rec["eps"]).numpy(),
        "errors": rec["errors"].numpy(), "seed": np.int64(seed), "fast": np.bool_(fast),
        "n_stage_means": np.int64(len(rec["means"])),
    }
    for i, m in enumerate(rec["means"]):
        arrs[f"mean{i}"] = m.numpy()
    if rec["sel"] is not None:
        arrs["fast_select"] = rec["sel"].numpy()
    for k, v in bb.state_dict().items():
        arrs["bb." + k] = v.numpy()
    if dc.encoder is not None:
        arrs["encoder.weight"] = dc.encoder.weight.detach().numpy()
    for k, v in cfgd.items():
        arrs["cfg." + k] = np.array(v)
    np.savez_compressed(os.path.join(OUT, f"classify_{name}.npz"), **arrs)
    print(name, "labels", out.tolist(), "calls", len(rec["eps"]), "errors", rec["errors"].shape)


def capture_schedules():
    arrs = {}
    t = torch.linspace(0, 1, 41)
    arrs["t"] = t.numpy()
    for nd, im in [(32, 32), (64, 32), (128, 256), (256, 256)]:
        dc = DiffusionClassifier(TinyBackbone(), Bag(**base_cfg(noise_d=nd, image_size=im)))
        arrs[f"cosine_{nd}_{im}"] = dc.logsnr_schedule_cosine(t).numpy()
        arrs[f"shifted_{nd}_{im}"] = dc.logsnr_schedule_cosine_shifted(t).numpy()
    np.savez_compressed(os.path.join(OUT, "schedules.npz"), **arrs)
    print("schedules", {k: v.shape for k, v in arrs.items() if k != "t"})


def capture_sample():
    """Generation path (reference sample(), :210-293) with the stand-in backbone of the 1stage_eps case."""
    g0 = dict(np.load(os.path.join(OUT, "classify_1stage_eps.npz")))
    arrs = {}
    for tag, pp, from_t, seed in [("eps", "eps", 1, 31), ("v_from_t", "v", 0.6, 37)]:
        cfgd = base_cfg(pred_param=pp, cfg_w=1.5, sampling_steps=4)
        bb = TinyBackbone(ch=3, hid=8, n_classes=cfgd["classes"], mode="nn")
        bb.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g0.items() if k.startswith("bb.")})
        dc = DiffusionClassifier(bb, Bag(**cfgd))
        dc.encoder.weight.data.copy_(torch.from_numpy(g0["encoder.weight"]))
        x = torch.from_numpy(g0["x"])
        labels = torch.tensor([0, 1, 2, 3, 1])
        torch.manual_seed(seed)
        out = dc.sample(x, labels, from_t=from_t)
        arrs.update({tag + ".x": x.numpy(), tag + ".labels": labels.numpy(), tag + ".out": out.numpy(),
                     tag + ".seed": np.int64(seed), tag + ".from_t": np.float64(from_t)})
        print("sample", tag, out.shape, float(out.abs().mean()))
    np.savez_compressed(os.path.join(OUT, "sample_cases.npz"), **arrs)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sample":
        capture_sample()
        sys.exit(0)
    os.makedirs(OUT, exist_ok=True)
    capture_schedules()
    capture_case("1stage_eps", {})
    capture_case("2stage_pruned", dict(classes=10, n_stages=2, evaluation_per_stage=[4, 10],
                                       n_keep_per_stage=[3, 1]), seed=11)
    capture_case("fast", dict(classes=6, n_fast_classes=3, evaluation_per_stage=[5]), fast=True, seed=13)
    capture_case("v_shifted", dict(pred_param="v", schedule="shifted_cosine", noise_d=64), seed=17)
    capture_case("dit_labels", dict(encoder_type="DiT", classes=3, evaluation_per_stage=[5]), seed=19)
