"""Oracle: UNet2DConditionModel forward, PyTorch eager.  Test infrastructure only.

What it restates: the backbone the reference calls at
`diffusion/diffusion_classifier.py:700-704` through `nets/unet.py:186-195`
(`UNetCondition2D.forward` -> diffusers `UNet2DConditionModel.forward`).  The
arithmetic lives in third-party `diffusers==0.31.0` (reference requirements.txt:9),
which is absent here, so this file restates the published architecture
 for the subset of constructor options the reference uses
(`nets/unet.py:78-132`; instantiations `experiments/cifar10/inference.py:94-116`,
`models/chexpert-256-unet-dwt-healthysick.py:4-28`, `models/ipmsa-5-unet.py:4-30`).
PARITY UNPINNED for this file (no diffusers, no reference fixture); state-dict key
names are diffusers' so a real checkpoint can validate it later.

`lowp=True` emulates the storage rounding of the HIP bf16 pipeline (bf16 weights and
stored activations, fp32 accumulate, fp32 norms/softmax, fp32 time/class-vector side
path) so bf16 kernels can be checked tightly; `lowp=False` is the plain fp32 oracle.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


def _bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _Resnet(nn.Module):
    # diffusers ResnetBlock2D (time_scale_shift="default", output_scale_factor=1)
    def __init__(self, cin, cout, temb_ch, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_ch, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None


class _Attention(nn.Module):
    # diffusers Attention (bias on q/k/v per `bias`, out bias always)
    def __init__(self, qdim, ctx_dim, heads, dim_head, bias):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(qdim, inner, bias=bias)
        self.to_k = nn.Linear(ctx_dim if ctx_dim else qdim, inner, bias=bias)
        self.to_v = nn.Linear(ctx_dim if ctx_dim else qdim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, qdim, bias=True), nn.Identity()])


class _GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)


class _FF(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([_GEGLU(dim, 4 * dim), nn.Identity(), nn.Linear(4 * dim, dim)])


class _TBlock(nn.Module):
    # diffusers BasicTransformerBlock (layer_norm, geglu, attention_bias=False)
    def __init__(self, dim, heads, dim_head, xdim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = _Attention(dim, None, heads, dim_head, bias=False)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = _Attention(dim, xdim, heads, dim_head, bias=False)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = _FF(dim)


class _Transformer2D(nn.Module):
    # diffusers Transformer2DModel, continuous input, use_linear_projection=False
    def __init__(self, ch, heads, xdim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, ch, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(ch, ch, 1)
        self.transformer_blocks = nn.ModuleList([_TBlock(ch, heads, ch // heads, xdim)])
        self.proj_out = nn.Conv2d(ch, ch, 1)


class _Sampler(nn.Module):
    def __init__(self, ch, stride):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=stride, padding=1)


class _Block(nn.Module):
    def __init__(self):
        super().__init__()


class _TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, dim)


def sinusoid(lam, dim, flip_sin_to_cos=True, freq_shift=0.0, max_period=10000):
    """diffusers get_timestep_embedding (scale=1)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    arg = lam[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class OracleUNetCondition2D(nn.Module):
    def __init__(self, sample_size=None, in_channels=4, out_channels=4,
                 down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                 mid_block_type="UNetMidBlock2DCrossAttn",
                 up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3,
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=1280,
                 encoder_hid_dim=None, encoder_hid_dim_type=None, attention_head_dim=8,
                 flip_sin_to_cos=True, freq_shift=0, lowp=False, **unused):
        super().__init__()
        assert mid_block_type == "UNetMidBlock2DCrossAttn"
        assert encoder_hid_dim_type == "text_proj"
        boc = tuple(block_out_channels)
        nb = len(boc)
        lpb = (layers_per_block,) * nb if isinstance(layers_per_block, int) else tuple(layers_per_block)
        heads = attention_head_dim  # diffusers aliases num_attention_heads <- attention_head_dim
        G, eps, xdim = norm_num_groups, norm_eps, cross_attention_dim
        self.config = SimpleNamespace(
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
            block_out_channels=boc, layers_per_block=lpb, norm_num_groups=G, norm_eps=eps,
            cross_attention_dim=xdim, encoder_hid_dim=encoder_hid_dim,
            attention_head_dim=attention_head_dim, flip_sin_to_cos=flip_sin_to_cos,
            freq_shift=freq_shift)
        self.lowp = lowp
        temb = boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = _TimestepEmbedding(boc[0], temb)
        self.encoder_hid_proj = nn.Linear(encoder_hid_dim, xdim)

        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, kind in enumerate(down_block_types):
            cin, out = out, boc[i]
            blk = _Block()
            blk.resnets = nn.ModuleList(
                [_Resnet(cin if j == 0 else out, out, temb, G, eps) for j in range(lpb[i])])
            if kind == "CrossAttnDownBlock2D":
                blk.attentions = nn.ModuleList([_Transformer2D(out, heads, xdim, G) for _ in range(lpb[i])])
            else:
                assert kind == "DownBlock2D", kind
            if i != nb - 1:
                blk.downsamplers = nn.ModuleList([_Sampler(out, 2)])
            self.down_blocks.append(blk)

        self.mid_block = _Block()
        self.mid_block.resnets = nn.ModuleList([_Resnet(boc[-1], boc[-1], temb, G, eps) for _ in range(2)])
        self.mid_block.attentions = nn.ModuleList([_Transformer2D(boc[-1], heads, xdim, G)])

        self.up_blocks = nn.ModuleList()
        rboc, rlpb = boc[::-1], lpb[::-1]
        out = rboc[0]
        for i, kind in enumerate(up_block_types):
            prev, out = out, rboc[i]
            cin = rboc[min(i + 1, nb - 1)]
            n = rlpb[i] + 1
            blk = _Block()
            blk.resnets = nn.ModuleList()
            for j in range(n):
                skip = cin if j == n - 1 else out
                rin = prev if j == 0 else out
                blk.resnets.append(_Resnet(rin + skip, out, temb, G, eps))
            if kind == "CrossAttnUpBlock2D":
                blk.attentions = nn.ModuleList([_Transformer2D(out, heads, xdim, G) for _ in range(n)])
            else:
                assert kind == "UpBlock2D", kind
            if i != nb - 1:
                blk.upsamplers = nn.ModuleList([_Sampler(out, 1)])
            self.up_blocks.append(blk)

        self.conv_norm_out = nn.GroupNorm(G, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    # ---- rounding hooks (identity in fp32 mode) ----
    def _q(self, x):
        return _bf16_round(x) if self.lowp else x

    def _conv(self, m, x, stride=1):
        w = self._q(m.weight)
        return F.conv2d(x, w, None, stride=stride, padding=m.padding)

    def _lin(self, m, x):
        return F.linear(x, self._q(m.weight), None)

    def _resnet(self, r, x, temb_act):
        q = self._q
        h = q(F.silu(r.norm1(x)))
        tv = F.linear(temb_act, r.time_emb_proj.weight, r.time_emb_proj.bias)  # fp32 side path
        h = q(self._conv(r.conv1, h) + r.conv1.bias[None, :, None, None] + tv[:, :, None, None])
        h = q(F.silu(r.norm2(h)))
        if r.conv_shortcut is not None:
            x = q(self._conv(r.conv_shortcut, x) + r.conv_shortcut.bias[None, :, None, None])
        return q(self._conv(r.conv2, h) + r.conv2.bias[None, :, None, None] + x)

    def _transformer(self, t, x, ctx):
        q = self._q
        N, C, H, W = x.shape
        res = x
        h = q(t.norm(x))
        h = q(self._conv(t.proj_in, h) + t.proj_in.bias[None, :, None, None])
        h = h.permute(0, 2, 3, 1).reshape(N, H * W, C)
        b = t.transformer_blocks[0]
        # self attention
        a = b.attn1
        hn = q(b.norm1(h))
        qq, kk, vv = q(self._lin(a.to_q, hn)), q(self._lin(a.to_k, hn)), q(self._lin(a.to_v, hn))
        d = C // a.heads
        sh = lambda z: z.view(N, -1, a.heads, d).transpose(1, 2)
        s = torch.matmul(sh(qq), sh(kk).transpose(-1, -2)) * (d ** -0.5)
        o = q(torch.matmul(torch.softmax(s, dim=-1), sh(vv)).transpose(1, 2).reshape(N, -1, C))
        # cross attention over ONE context token: softmax over a single key == 1, so the
        # output is to_out(to_v(ctx)) for every query; computed in fp32.
        a2 = b.attn2
        assert ctx.shape[1] == 1
        cv = F.linear(F.linear(ctx[:, 0], a2.to_v.weight), a2.to_out[0].weight, a2.to_out[0].bias)
        h = q(self._lin(a.to_out[0], o) + a.to_out[0].bias + h + cv[:, None, :])
        # feed forward (GEGLU, erf gelu)
        hn = q(b.norm3(h))
        p = self._lin(b.ff.net[0].proj, hn) + b.ff.net[0].proj.bias
        u, g = p.chunk(2, dim=-1)
        f = q(u * F.gelu(g))
        h = q(self._lin(b.ff.net[2], f) + b.ff.net[2].bias + h)
        h = h.reshape(N, H, W, C).permute(0, 3, 1, 2)
        return q(self._conv(t.proj_out, h) + t.proj_out.bias[None, :, None, None] + res)

    def cross_attn_exact(self, t, h_tokens, ctx):
        """Full SDPA cross-attention (for the test that proves the 1-token shortcut)."""
        b = t.transformer_blocks[0]
        a2 = b.attn2
        N, L, C = h_tokens.shape
        d = C // a2.heads
        sh = lambda z: z.view(N, -1, a2.heads, d).transpose(1, 2)
        qq = a2.to_q(b.norm2(h_tokens))
        kk, vv = a2.to_k(ctx), a2.to_v(ctx)
        o = F.scaled_dot_product_attention(sh(qq), sh(kk), sh(vv)).transpose(1, 2).reshape(N, L, C)
        return a2.to_out[0](o)

    def forward(self, x, noise_labels, downblock_additional_residuals=None,
                midblock_additional_residuals=None, encoder_hidden_states=None):
        # signature: reference nets/unet.py:186
        q = self._q
        N = x.shape[0]
        lam = noise_labels
        if not torch.is_tensor(lam):
            lam = torch.tensor([lam], dtype=torch.float32)
        lam = lam.reshape(-1).expand(N) if lam.numel() == 1 else lam.reshape(-1)
        c = self.config
        temb = sinusoid(lam, c.block_out_channels[0], c.flip_sin_to_cos, c.freq_shift)
        te = self.time_embedding
        temb = te.linear_2(F.silu(te.linear_1(temb)))
        temb_act = F.silu(temb)
        ctx = self.encoder_hid_proj(encoder_hidden_states)  # [N,1,xdim], fp32 side path

        h = q(self._conv(self.conv_in, q(x)) + self.conv_in.bias[None, :, None, None])
        skips = [h]
        for blk in self.down_blocks:
            for j, r in enumerate(blk.resnets):
                h = self._resnet(r, h, temb_act)
                if hasattr(blk, "attentions"):
                    h = self._transformer(blk.attentions[j], h, ctx)
                skips.append(h)
            if hasattr(blk, "downsamplers"):
                d = blk.downsamplers[0].conv
                h = q(self._conv(d, h, stride=2) + d.bias[None, :, None, None])
                skips.append(h)
        m = self.mid_block
        h = self._resnet(m.resnets[0], h, temb_act)
        h = self._transformer(m.attentions[0], h, ctx)
        h = self._resnet(m.resnets[1], h, temb_act)
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                h = torch.cat([h, skips.pop()], dim=1)
                h = self._resnet(r, h, temb_act)
                if hasattr(blk, "attentions"):
                    h = self._transformer(blk.attentions[j], h, ctx)
            if hasattr(blk, "upsamplers"):
                u = blk.upsamplers[0].conv
                h = F.interpolate(h, scale_factor=2.0, mode="nearest")
                h = q(self._conv(u, h) + u.bias[None, :, None, None])
        h = q(F.silu(self.conv_norm_out(h)))
        return self._conv(self.conv_out, h) + self.conv_out.bias[None, :, None, None]
