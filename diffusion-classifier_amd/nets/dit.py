"""MI355X-native drop-in for reference `nets/dit.py`.

`DiT(**kwargs)` keeps the reference constructor keywords (nets/dit.py:9-27) and
`forward(x, noise_labels, encoder_hidden_states=None)` (nets/dit.py:49-51), where — exactly as
in the reference, which passes its third argument positionally into diffusers'
`class_labels` slot — `encoder_hidden_states` carries the int64 class labels
(`encoder_type='DiT'`, diffusion_classifier.py:90-92).  Parameter names are diffusers'
`DiTTransformer2DModel`'s.  Modules only hold parameters; arithmetic runs in libdcamd.
"""
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from .. import _lib as L
from .. import engine as E
from .. import engine_dit as ED
from .unet import _Bag, _HipBackbone


def _sincos_1d(dim, pos):
    omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_pos_embed(dim, grid, base_size, interpolation_scale=1.0):
    """Fixed 2-D sin-cos table of diffusers' PatchEmbed (meshgrid puts the W coordinate first)."""
    c = np.arange(grid, dtype=np.float32) / (grid / base_size) / interpolation_scale
    g = np.stack(np.meshgrid(c, c), axis=0).reshape(2, 1, grid, grid)
    return np.concatenate([_sincos_1d(dim // 2, g[0]), _sincos_1d(dim // 2, g[1])], axis=1)


class DiT(_HipBackbone):
    def __init__(
        self,
        num_attention_heads: int = 16,
        attention_head_dim: int = 72,
        in_channels: int = 4,
        out_channels: Optional[int] = None,
        num_layers: int = 28,
        dropout: float = 0.0,
        norm_num_groups: int = 32,
        attention_bias: bool = True,
        sample_size: int = 32,
        patch_size: int = 2,
        activation_fn: str = "gelu-approximate",
        num_embeds_ada_norm: Optional[int] = 1000,
        upcast_attention: bool = False,
        norm_type: str = "ada_norm_zero",
        norm_elementwise_affine: bool = False,
        norm_eps: float = 1e-5,
    ):
        super().__init__()
        for k, (got, want) in dict(dropout=(dropout, 0.0), attention_bias=(attention_bias, True),
                                   activation_fn=(activation_fn, "gelu-approximate"), norm_type=(norm_type, "ada_norm_zero"),
                                   norm_elementwise_affine=(norm_elementwise_affine, False)).items():
            if got != want:
                raise NotImplementedError(f"DiT({k}={got!r}) is outside the scoring path built here (supported: {want!r})")
        D = num_attention_heads * attention_head_dim
        out_channels = in_channels if out_channels is None else out_channels
        self.D = D
        self.config = SimpleNamespace(
            num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim, in_channels=in_channels,
            out_channels=out_channels, num_layers=num_layers, sample_size=sample_size, patch_size=patch_size,
            num_embeds_ada_norm=num_embeds_ada_norm, norm_eps=norm_eps, encoder_hid_dim=None)
        self.pos_embed = _Bag()
        self.pos_embed.proj = nn.Conv2d(in_channels, D, patch_size, stride=patch_size)
        g = sample_size // patch_size
        self.pos_embed.register_buffer(
            "pos_embed", torch.from_numpy(sincos_pos_embed(D, g, base_size=g)).float().unsqueeze(0), persistent=False)
        self.transformer_blocks = nn.ModuleList()
        for _ in range(num_layers):
            b = _Bag()
            b.norm1 = _Bag()
            b.norm1.emb = _Bag()
            b.norm1.emb.timestep_embedder = _Bag()
            b.norm1.emb.timestep_embedder.linear_1 = nn.Linear(256, D)
            b.norm1.emb.timestep_embedder.linear_2 = nn.Linear(D, D)
            b.norm1.emb.class_embedder = _Bag()
            b.norm1.emb.class_embedder.embedding_table = nn.Embedding(num_embeds_ada_norm + 1, D)
            b.norm1.linear = nn.Linear(D, 6 * D)
            a = _Bag()
            a.to_q, a.to_k, a.to_v = nn.Linear(D, D), nn.Linear(D, D), nn.Linear(D, D)
            a.to_out = nn.ModuleList([nn.Linear(D, D), nn.Identity()])
            b.attn1 = a
            b.ff = _Bag()
            pr = _Bag()
            pr.proj = nn.Linear(D, 4 * D)
            b.ff.net = nn.ModuleList([pr, nn.Identity(), nn.Linear(4 * D, D)])
            self.transformer_blocks.append(b)
        self.proj_out_1 = nn.Linear(D, 2 * D)
        self.proj_out_2 = nn.Linear(D, patch_size * patch_size * out_channels)
        self._init_engine()

    def packed_weights(self, dt, device):
        key = (dt, str(device))
        if key not in self._packed:
            self._packed[key] = ED.DiTWeights(self, dt, device)
        return self._packed[key]

    def make_plan(self, n_bj, n_cls, n_ctx, device, score=None, share_trunk=None):
        dt = E.DT[self.compute_dtype]
        return ED.DiTPlan(self, self.packed_weights(dt, device), n_bj, n_cls, n_ctx, score=score, device=device)

    def _feed(self, plan, x, noise_labels):
        """lambda and the patch-embedding GEMM operand (p x p patches of x) of a plain forward."""
        lib = L.lib()
        dev = x.device
        N, Cin, H, W = x.shape
        lam = noise_labels if torch.is_tensor(noise_labels) else torch.tensor([noise_labels])
        lam = lam.to(dev, torch.float32).reshape(-1)
        plan.lam.copy_(lam.expand(N) if lam.numel() == 1 else lam)
        xf = x.detach().to(torch.float32).contiguous()
        ones = torch.ones(N, dtype=torch.float32, device=dev)
        zeros = torch.zeros(N, dtype=torch.float32, device=dev)
        q = L.QsampleParams(x=xf.data_ptr(), eps=xf.data_ptr(), alpha=ones.data_ptr(), sigma=zeros.data_ptr(), img_of_bj=None,
                            out=plan.a0_buf.data_ptr(), out_dtype=plan.dt, n_bj=N, C=Cin, H=H, W=W,
                            ld=plan.a0_buf.shape[-1], im2col=2, patch=self.config.patch_size)
        L.check(lib.dc_qsample(q, L.stream_ptr()), "dc_qsample")

    @torch.no_grad()
    def forward_pair(self, x, noise_labels, cond, null):
        """Classifier-free-guidance pair as ONE batch-2 plan launch: unit 2b = image b under label `cond[b]`, unit 2b+1 under the
        null label `null[b]` (reference `sample`, :255-266, calls the backbone twice per step).  Returns the plan's un-patchified
        projection [2N, H/p, W/p, ld] fp32 (the layout `dc_ddpm_step` reads with patch = p) — a view the next call overwrites."""
        L.require_gpu()
        if not x.is_cuda:
            raise L.DcamdError("DiT.forward_pair needs CUDA/HIP tensors (no CPU fallback)")
        dev = x.device
        N = x.shape[0]
        key = ("pair", N, str(dev), self.compute_dtype)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self.make_plan(N, 2, None, dev)
        self._feed(plan, x, noise_labels)
        plan.ctx_of_unit.copy_(torch.stack([cond.reshape(-1), null.reshape(-1)], dim=1).reshape(-1).to(dev, torch.int32))
        plan.run()
        return plan.pred_view()

    @torch.no_grad()
    def forward(self, x, noise_labels, encoder_hidden_states=None):
        L.require_gpu()
        if not x.is_cuda:
            raise L.DcamdError("DiT.forward needs CUDA/HIP tensors (no CPU fallback)")
        dev = x.device
        N, Cin, H, W = x.shape
        cfg = self.config
        p = cfg.patch_size
        key = ("fwd", N, str(dev), self.compute_dtype)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self.make_plan(N, 1, None, dev)
        self._feed(plan, x, noise_labels)
        plan.ctx_of_unit.copy_(encoder_hidden_states.reshape(-1).to(dev, torch.int32))
        plan.run()
        g, oc = H // p, cfg.out_channels
        out = plan.pred_view().reshape(N, g, g, p, p, oc)          # un-patchify: nhwpqc -> nchpwq
        return out.permute(0, 5, 1, 3, 2, 4).reshape(N, oc, g * p, g * p).contiguous().to(x.dtype)
