"""Minimal EMA wrapper with the surface the reference uses from `ema_pytorch.EMA`
(diffusion/diffusion_classifier.py:51-56, :453, :700): constructor keywords, `.ema_model`,
`forward -> ema_model(...)`, `update()`.  Inference only ever calls forward."""
import copy

import torch
import torch.nn as nn


class EMA(nn.Module):
    def __init__(self, model, beta=0.9999, update_after_step=100, update_every=10):
        super().__init__()
        self.online_model = [model]           # not registered: the wrapper owns only the copy
        self.ema_model = copy.deepcopy(model)
        self.ema_model.requires_grad_(False)
        self.beta, self.update_after_step, self.update_every = beta, update_after_step, update_every
        self.register_buffer("initted", torch.tensor(False))
        self.register_buffer("step", torch.tensor(0))

    def forward(self, *a, **k):
        return self.ema_model(*a, **k)

    @torch.no_grad()
    def update(self):
        step = int(self.step.item())
        self.step += 1
        if step % (self.update_every or 1) != 0:
            return
        src, dst = self.online_model[0], self.ema_model
        if step <= (self.update_after_step or 0) or not bool(self.initted.item()):
            dst.load_state_dict(src.state_dict())
            self.initted.fill_(True)
            return
        for (_, p), (_, q) in zip(src.named_parameters(), dst.named_parameters()):
            q.lerp_(p.detach(), 1.0 - self.beta)
        if hasattr(dst, "invalidate_packed"):
            dst.invalidate_packed()
