#!/usr/bin/env python3
"""BASELINE.md §3 cross-check (build container only: needs /root/reference): is the oracle's restated scoring loop — what bench.py
times as `cpu_baseline` ("port") — as fast as the REAL reference loop?  The reference's own `DiffusionClassifier.classify`
(diffusion/diffusion_classifier.py:657-725, imported with the arithmetic-free `comet_ml` / `ema_pytorch` stubs of
tools/capture_goldens.py) drives the build's CPU backbone (the oracle's UNetCondition2D restatement: the reference's own backbone is
diffusers 0.31.0, absent here) on the CIFAR-10 configuration; the oracle loop drives the same backbone object on the same inputs.
Prints the two timings, their ratio and whether labels / errors agree.  The GPU box never runs this."""
import copy
import json
import os
import sys
import time
import types

import torch
import torch.nn as nn
import accelerate  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
import oracle  # noqa: E402
import diffusion_classifier_amd as dca  # noqa: E402
sys.path.insert(0, "/root/reference")
from diffusion.diffusion_classifier import DiffusionClassifier as RefDC  # noqa: E402  (the reference)


class Bag:
    def __init__(self, **kw):
        self.__dict__["d"] = kw

    def __getattr__(self, k):
        return self.__dict__["d"].get(k)


def main():
    trials = int(os.environ.get("TRIALS", "2"))
    threads = int(os.environ.get("THREADS", str(os.cpu_count() or 8)))
    torch.set_num_threads(threads)
    kw = dca.cifar10_unet_kwargs()
    cfg = dict(pred_param="eps", schedule="cosine", noise_d=32, image_size=32, cfg_w=0.0, ema_beta=0.999, ema_warmup=0, ema_update_freq=1,
               encoder_type="nn", classes=10, n_stages=1, evaluation_per_stage=[trials], n_keep_per_stage=[1], n_fast_classes=2)
    torch.manual_seed(0)
    bb = oracle.OracleUNetCondition2D(**kw)
    ref = RefDC(bb, Bag(**cfg))
    ora = oracle.OracleDiffusionClassifier(bb, oracle.AttrBag(**cfg))
    ora.encoder.load_state_dict(ref.encoder.state_dict())
    x = torch.rand(2, 3, 32, 32) * 2 - 1
    out = {}
    for name, fn in (("reference_loop", lambda: ref.classify(x)), ("oracle_loop", lambda: ora.classify(x)),
                     ("reference_loop_again", lambda: ref.classify(x)), ("oracle_loop_again", lambda: ora.classify(x))):
        torch.manual_seed(1234)
        t0 = time.perf_counter()
        with torch.no_grad():
            lab = fn()
        out[name] = dict(seconds=round(time.perf_counter() - t0, 3), labels=lab.tolist())
    r = out["oracle_loop_again"]["seconds"] / out["reference_loop_again"]["seconds"]
    rec = dict(what="real reference classify() vs the oracle's restated loop, both driving the oracle CPU backbone (CIFAR-10 UNet, fp32)",
               images=2, trials=trials, classes=10, unit_forwards=2 * trials * 10, threads=threads, runs=out,
               oracle_over_reference=round(r, 3), labels_equal=out["reference_loop_again"]["labels"] == out["oracle_loop_again"]["labels"])
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
