#!/usr/bin/env python3
"""Write a checkpoint directory with the REFERENCE's own `save_checkpoint` (diffusion/diffusion_classifier.py:727-767,
i.e. accelerate's `save_state` over the modules prepared in the reference's order, :381-386) and score a batch with the
reference's own `classify` on those weights.  Runs ONLY in the build container (needs /root/reference + accelerate).

Output (data only): tests/golden/ckpt_tiny_unet/{model,model_1,model_2}.safetensors, optimizer.bin, scheduler.bin,
random_states_0.pkl, experiment_state.pth — exactly what the reference wrote — plus expected.npz (inputs, draws, the
errors / labels the reference's loop produced with the EMA weights of that checkpoint).

Stubs: `comet_ml` (never called: experiment=None) and `ema_pytorch.EMA`.  The EMA stub registers what
ema_pytorch 0.7.7 registers (from knowledge of that package: `online_model` and `ema_model` submodules, `initted` and
`step` buffers), so model_1.safetensors carries the key layout a real training run writes.  The backbone is the build's
CPU restatement of UNet2DConditionModel (diffusers key names) at a tiny size.
"""
import copy
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn
import accelerate  # noqa: F401  (before the stubs)
from accelerate import Accelerator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

comet = types.ModuleType("comet_ml")
comet.Experiment = type("Experiment", (), {})
comet.ExistingExperiment = type("ExistingExperiment", (), {})
sys.modules["comet_ml"] = comet


class _EMA(nn.Module):
    def __init__(self, model, beta=None, update_after_step=None, update_every=None):
        super().__init__()
        self.online_model = model
        self.ema_model = copy.deepcopy(model)
        self.register_buffer("initted", torch.tensor(True))
        self.register_buffer("step", torch.tensor(123))

    def forward(self, *a, **k):
        return self.ema_model(*a, **k)


ema = types.ModuleType("ema_pytorch")
ema.EMA = _EMA
sys.modules["ema_pytorch"] = ema

sys.path.insert(0, "/root/reference")
from diffusion.diffusion_classifier import DiffusionClassifier  # noqa: E402  (the reference)
import oracle  # noqa: E402


class Bag:
    def __init__(self, **kw):
        self.__dict__["d"] = kw

    def __getattr__(self, k):
        return self.__dict__["d"].get(k)


TINY = dict(sample_size=16, in_channels=3, out_channels=3, layers_per_block=1, block_out_channels=(32, 32),
            down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"), up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"),
            mid_block_type="UNetMidBlock2DCrossAttn", encoder_hid_dim=32, encoder_hid_dim_type="text_proj", cross_attention_dim=32,
            attention_head_dim=2)      # 2 heads x 16 (diffusers reads this keyword as the head COUNT; libdcamd's attention needs head dim >= 16)
CFG = dict(pred_param="eps", schedule="cosine", noise_d=16, image_size=16, cfg_w=0.0, ema_beta=0.999, ema_warmup=0,
           ema_update_freq=1, encoder_type="nn", classes=3, n_stages=1, evaluation_per_stage=[4], n_keep_per_stage=[1],
           n_fast_classes=2)


def main():
    out = os.path.join(ROOT, "tests", "golden", "ckpt_tiny_unet")
    tmp = tempfile.mkdtemp()
    torch.manual_seed(2024)
    bb = oracle.OracleUNetCondition2D(**TINY)
    with torch.no_grad():
        for p in bb.parameters():
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    dc = DiffusionClassifier(bb, Bag(**dict(CFG, experiment_path=tmp)))
    with torch.no_grad():                                    # EMA weights differ from the online weights, as after training
        for p in dc.ema.ema_model.parameters():
            p.add_(torch.randn_like(p) * 0.02)
    acc = Accelerator(cpu=True)
    opt = torch.optim.SGD(dc.model.parameters(), lr=0.1)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 1.0)
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.zeros(4, 3, 16, 16)), batch_size=2)
    # the reference's preparation order (:381-386): model, ema, optimizer, loaders, scheduler; then the encoder
    model, ema_, opt, tl, vl, sched = acc.prepare(dc.model, dc.ema, opt, loader, loader, sched)
    dc.encoder = acc.prepare(dc.encoder)
    dc.save_checkpoint(acc, epoch=4, experiment=None, checkpoint_tracker={"value": 0.75, "save_flag": False})
    src = os.path.join(tmp, "checkpoints")
    # score a batch with the reference's own loop on these weights
    BS, T = 3, CFG["evaluation_per_stage"][0]
    x = torch.rand(BS, 3, 16, 16) * 2 - 1
    rec = {"t": [], "eps": [], "errors": None}
    o_rand, o_randn_like, o_full = torch.rand, torch.randn_like, torch.full

    def rand(*a, **k):
        r = o_rand(*a, **k); rec["t"].append(r.clone()); return r

    def randn_like(*a, **k):
        r = o_randn_like(*a, **k); rec["eps"].append(r.clone()); return r

    def full(*a, **k):
        r = o_full(*a, **k); rec["errors"] = r; return r
    torch.manual_seed(99)
    torch.rand, torch.randn_like, torch.full = rand, randn_like, full
    try:
        with torch.no_grad():
            labels = dc.classify(x)
    finally:
        torch.rand, torch.randn_like, torch.full = o_rand, o_randn_like, o_full
    if os.path.isdir(out):
        shutil.rmtree(out)
    shutil.copytree(src, out)
    np.savez_compressed(os.path.join(out, "expected.npz"), x=x.numpy(), t=torch.stack(rec["t"]).numpy(),
                        eps=torch.stack(rec["eps"]).numpy(), errors=rec["errors"].numpy(), labels=labels.numpy(),
                        **{"cfg." + k: np.array(v) for k, v in CFG.items()},
                        **{"arch." + k: np.array(v) for k, v in TINY.items()})
    for f in sorted(os.listdir(out)):
        print(f, os.path.getsize(os.path.join(out, f)))
    print("labels", labels.tolist())
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
