"""Oracle: DiTTransformer2DModel forward, PyTorch eager.  Test infrastructure only.

Restates the backbone behind reference `nets/dit.py:8-51` (`DiT.forward(x, noise_labels,
encoder_hidden_states)` passes its third argument positionally as diffusers' `class_labels`).
Arithmetic is diffusers 0.31.0's `DiTTransformer2DModel` (adaLN-Zero blocks), absent
here -> PARITY UNPINNED (see oracle/__init__.py); key names are diffusers'.
Architecture instance: reference `models/chexpert-256-dit-b4.py:4-21`.

`lowp=True` emulates the HIP fp16/bf16 pipeline's storage rounding (see oracle/unet.py).
"""
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .unet import sinusoid, _TimestepEmbedding, _Attention


def _sincos_1d(dim, pos):
    omega = np.arange(dim // 2, dtype=np.float64) / (dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(dim, grid, base_size, interpolation_scale=1.0):
    """diffusers get_2d_sincos_pos_embed (meshgrid puts W first)."""
    gh = np.arange(grid, dtype=np.float32) / (grid / base_size) / interpolation_scale
    gw = np.arange(grid, dtype=np.float32) / (grid / base_size) / interpolation_scale
    g = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid, grid)
    return np.concatenate([_sincos_1d(dim // 2, g[0]), _sincos_1d(dim // 2, g[1])], axis=1)


class _LabelEmb(nn.Module):
    def __init__(self, n, dim):
        super().__init__()
        self.embedding_table = nn.Embedding(n + 1, dim)  # +1: cfg/null embedding (dropout_prob>0)


class _CombinedEmb(nn.Module):
    def __init__(self, n, dim):
        super().__init__()
        self.timestep_embedder = _TimestepEmbedding(256, dim)
        self.class_embedder = _LabelEmb(n, dim)

    def forward(self, lam, labels):
        t = sinusoid(lam, 256, True, 1.0)
        t = self.timestep_embedder.linear_2(F.silu(self.timestep_embedder.linear_1(t)))
        return t + self.class_embedder.embedding_table(labels)


class _AdaLNZero(nn.Module):
    def __init__(self, dim, n):
        super().__init__()
        self.emb = _CombinedEmb(n, dim)
        self.linear = nn.Linear(dim, 6 * dim)


class _GELU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner)


class _FFT(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([_GELU(dim, 4 * dim), nn.Identity(), nn.Linear(4 * dim, dim)])


class _DiTBlock(nn.Module):
    def __init__(self, dim, heads, dh, n):
        super().__init__()
        self.norm1 = _AdaLNZero(dim, n)
        self.attn1 = _Attention(dim, None, heads, dh, bias=True)
        self.ff = _FFT(dim)


class _PatchEmbed(nn.Module):
    def __init__(self, cin, dim, p):
        super().__init__()
        self.proj = nn.Conv2d(cin, dim, p, stride=p)


class OracleDiT(nn.Module):
    def __init__(self, num_attention_heads=16, attention_head_dim=72, in_channels=4, out_channels=None,
                 num_layers=28, sample_size=32, patch_size=2, num_embeds_ada_norm=1000,
                 norm_eps=1e-5, lowp=False, lowp_dtype=torch.float16, **unused):
        super().__init__()
        D = num_attention_heads * attention_head_dim
        out_channels = in_channels if out_channels is None else out_channels
        self.config = SimpleNamespace(
            num_attention_heads=num_attention_heads, attention_head_dim=attention_head_dim,
            in_channels=in_channels, out_channels=out_channels, num_layers=num_layers,
            sample_size=sample_size, patch_size=patch_size, num_embeds_ada_norm=num_embeds_ada_norm,
            norm_eps=norm_eps, encoder_hid_dim=None)
        self.lowp, self.lowp_dtype = lowp, lowp_dtype
        self.D, self.heads = D, num_attention_heads
        self.pos_embed = _PatchEmbed(in_channels, D, patch_size)
        g = sample_size // patch_size
        pe = torch.from_numpy(sincos_2d(D, g, base_size=g)).float().unsqueeze(0)
        self.pos_embed.register_buffer("pos_embed", pe, persistent=False)
        self.transformer_blocks = nn.ModuleList(
            [_DiTBlock(D, num_attention_heads, attention_head_dim, num_embeds_ada_norm)
             for _ in range(num_layers)])
        self.proj_out_1 = nn.Linear(D, 2 * D)
        self.proj_out_2 = nn.Linear(D, patch_size * patch_size * out_channels)

    def _q(self, x):
        return x.to(self.lowp_dtype).to(torch.float32) if self.lowp else x

    def _lin(self, m, x):
        return F.linear(x, self._q(m.weight), None)

    def forward(self, x, noise_labels, encoder_hidden_states=None):
        q, c, D = self._q, self.config, self.D
        N = x.shape[0]
        lam = noise_labels.reshape(-1)
        lam = lam.expand(N) if lam.numel() == 1 else lam
        labels = encoder_hidden_states.reshape(-1).long()
        p, eps = c.patch_size, c.norm_eps
        pe = self.pos_embed
        h = F.conv2d(q(x), q(pe.proj.weight), None, stride=p) + pe.proj.bias[None, :, None, None]
        h = q(h.flatten(2).transpose(1, 2) + pe.pos_embed)
        for b in self.transformer_blocks:
            cond = b.norm1.emb(lam, labels)                       # fp32 side path
            mod = b.norm1.linear(F.silu(cond))
            sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
            hn = q(F.layer_norm(h, (D,), eps=1e-6) * (1 + sc_a[:, None]) + sh_a[:, None])
            a = b.attn1
            qq = q(self._lin(a.to_q, hn) + a.to_q.bias)
            kk = q(self._lin(a.to_k, hn) + a.to_k.bias)
            vv = q(self._lin(a.to_v, hn) + a.to_v.bias)
            d = D // self.heads
            sh = lambda z: z.view(N, -1, self.heads, d).transpose(1, 2)
            s = torch.matmul(sh(qq), sh(kk).transpose(-1, -2)) * (d ** -0.5)
            o = q(torch.matmul(torch.softmax(s, dim=-1), sh(vv)).transpose(1, 2).reshape(N, -1, D))
            h = q(g_a[:, None] * (self._lin(a.to_out[0], o) + a.to_out[0].bias) + h)
            hn = q(F.layer_norm(h, (D,), eps=eps) * (1 + sc_m[:, None]) + sh_m[:, None])
            f = q(F.gelu(self._lin(b.ff.net[0].proj, hn) + b.ff.net[0].proj.bias, approximate="tanh"))
            h = q(g_m[:, None] * (self._lin(b.ff.net[2], f) + b.ff.net[2].bias) + h)
        cond = self.transformer_blocks[0].norm1.emb(lam, labels)
        shift, scale = self.proj_out_1(F.silu(cond)).chunk(2, dim=1)
        h = q(F.layer_norm(h, (D,), eps=1e-6) * (1 + scale[:, None]) + shift[:, None])
        h = self._lin(self.proj_out_2, h) + self.proj_out_2.bias
        g = int(math.isqrt(h.shape[1]))
        oc = c.out_channels
        h = h.reshape(N, g, g, p, p, oc)
        h = torch.einsum("nhwpqc->nchpwq", h)
        return h.reshape(N, oc, g * p, g * p)
