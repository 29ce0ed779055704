"""Oracle: the scoring loop.  Test infrastructure only (see oracle/__init__.py).

A structure-for-structure CPU restatement of reference
`diffusion/diffusion_classifier.py:657-725` (`classify`), with `diffuse` (`:100-117`),
`encode_text_prompt` (`:83-98`) and the schedules (`:119-161`): sequential trials x
classes, one backbone forward per (trial, class column) at batch BS.  PINNED by
tests/golden/classify_*.npz, which were captured from the reference's own `classify`
(tools/capture_goldens.py).

The only additions are keyword-only hooks to inject the RNG draws (`t`, `eps`,
`fast_select`), to return the `errors[BS, classes, T]` tensor and (`return_preds`) the
backbone's raw outputs `preds[trial, image, class column]` — the quantity the per-cell
eps-MSE barely sees (it is dominated by ||eps||^2), so the parity tests compare it too.
"""
import copy

import torch
import torch.nn as nn

from .schedule import logsnr_schedule_cosine, logsnr_schedule_cosine_shifted


class AttrBag:
    """Config bag whose missing keys read as None (reference experiments/cifar10/inference.py:24-38)."""

    def __init__(self, **kw):
        self.__dict__["_d"] = dict(kw)

    def __getattr__(self, k):
        return self.__dict__["_d"].get(k)

    def __setattr__(self, k, v):
        self.__dict__["_d"][k] = v


class OracleDiffusionClassifier(nn.Module):
    def __init__(self, backbone, config):
        super().__init__()
        self.config = config
        assert config.pred_param in ("v", "eps")           # reference :29-31
        assert config.schedule in ("cosine", "shifted_cosine")  # :34-35
        self.pred_param = config.pred_param
        self.noise_d, self.image_d = config.noise_d, config.image_size  # :40-41
        self.model = backbone
        self.ema_model = copy.deepcopy(backbone)            # EMA wrapper forwards to its deep copy (:51-56)
        self.encoder_type = config.encoder_type
        if self.encoder_type == "nn":                       # :65-70
            self.encoder = nn.Embedding(config.classes + 1, backbone.config.encoder_hid_dim)
        else:
            assert self.encoder_type == "DiT"               # :71-74
            self.encoder = None
        self.null_token = config.classes

    def schedule(self, t):
        if self.config.schedule == "cosine":
            return logsnr_schedule_cosine(t, self.noise_d, self.image_d)
        return logsnr_schedule_cosine_shifted(t, self.noise_d, self.image_d)

    def encode_text_prompt(self, text):
        if self.encoder_type == "nn":
            return self.encoder(text).unsqueeze(1)
        return text

    @torch.no_grad()
    def classify(self, x, text=None, fast=False, *, t=None, eps=None, fast_select=None,
                 return_errors=False, return_preds=False):
        cfg = self.config
        assert len(cfg.evaluation_per_stage) == cfg.n_stages          # :660
        assert len(cfg.n_keep_per_stage) == cfg.n_stages              # :661
        assert cfg.n_keep_per_stage[-1] == 1                          # :662
        assert 2 <= cfg.n_fast_classes <= cfg.classes                 # :663
        ends = [0] + list(cfg.evaluation_per_stage)                   # :665
        BS = x.shape[0]
        errors = torch.full((BS, cfg.classes, ends[-1]), torch.inf)   # :669
        preds = {}                                                    # (trial, class column) -> backbone output [BS, C, H, W]
        if fast:                                                      # :671-677
            text = text.view(-1, 1)
            classes = torch.arange(cfg.classes).repeat(BS, 1)
            wrong = classes[(classes == text) == False].view(BS, -1)  # noqa: E712
            sel = fast_select if fast_select is not None else \
                torch.randint(0, wrong.shape[1], (BS, cfg.n_fast_classes - 1))
            classes = torch.cat((text, torch.gather(wrong, 1, sel)), dim=1)
        else:
            classes = torch.arange(cfg.classes).repeat(BS, 1)          # :679
        for i in range(cfg.n_stages):                                 # :681
            for j in range(ends[i], ends[i + 1]):                     # :686
                tj = t[j] if t is not None else torch.rand(BS)        # :688
                logsnr = self.schedule(tj)                            # :689
                alpha = torch.sqrt(torch.sigmoid(logsnr)).view(-1, 1, 1, 1)   # :690
                sigma = torch.sqrt(torch.sigmoid(-logsnr)).view(-1, 1, 1, 1)  # :691
                e = eps[j] if eps is not None else torch.randn_like(x)        # :113
                z = alpha * x + sigma * e                             # :115
                for c in range(classes.shape[1]):                     # :695
                    lab = classes[:, c]
                    emb = self.encode_text_prompt(lab)                # :697
                    pred = self.ema_model(x=z, noise_labels=logsnr, encoder_hidden_states=emb)  # :700-704
                    if return_preds:
                        preds[(j, c)] = pred
                    eps_pred = sigma * z + alpha * pred if self.pred_param == "v" else pred     # :706-709
                    err = torch.norm((eps_pred - e).view(BS, -1), dim=1, p=2) ** 2              # :711
                    errors[torch.arange(BS), lab, j] = err            # :713-714
            mean = errors[:, :, :ends[i + 1]].mean(dim=2)             # :719
            _, classes = torch.topk(mean, cfg.n_keep_per_stage[i], dim=1, largest=False)  # :720-721
        assert classes.shape[1] == 1                                  # :723
        out = classes[:, 0]                                           # :725
        if return_preds:       # single-stage grids only: [T, BS, class columns, C, H, W]
            assert cfg.n_stages == 1
            ncol = 1 + max(c for _, c in preds)
            pt = torch.stack([torch.stack([preds[(j, c)] for c in range(ncol)], dim=1) for j in range(ends[-1])])
            return out, errors, pt
        return (out, errors) if return_errors else out
