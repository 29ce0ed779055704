"""MI355X-native drop-ins for reference `nets/unet.py`.

`UNetCondition2D` keeps the reference constructor keywords (nets/unet.py:78-132) and forward
signature (nets/unet.py:186 — `forward(x, noise_labels, downblock_additional_residuals=None,
midblock_additional_residuals=None, encoder_hidden_states=None) -> Tensor[B,C,H,W]`), exposes
`.config.encoder_hid_dim` (read at diffusion_classifier.py:67) and `.parameters()`, and names
its parameters exactly like diffusers' `UNet2DConditionModel` so reference checkpoints load
with `load_state_dict`.  The modules below only HOLD parameters; the arithmetic runs in
libdcamd (HIP, gfx950) through `engine.UNetPlan`.  There is no eager/CPU fallback: calling
forward without the HIP library or without a GPU raises.
"""
from types import SimpleNamespace
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from .. import _lib as L
from .. import engine as E


class _Bag(nn.Module):
    pass


def _resnet(cin, cout, temb, groups, eps):
    r = _Bag()
    r.norm1 = nn.GroupNorm(groups, cin, eps=eps)
    r.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
    r.time_emb_proj = nn.Linear(temb, cout)
    r.norm2 = nn.GroupNorm(groups, cout, eps=eps)
    r.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
    if cin != cout:
        r.conv_shortcut = nn.Conv2d(cin, cout, 1)
    return r


def _attention(qdim, ctx_dim, inner):
    a = _Bag()
    a.to_q = nn.Linear(qdim, inner, bias=False)
    a.to_k = nn.Linear(ctx_dim or qdim, inner, bias=False)
    a.to_v = nn.Linear(ctx_dim or qdim, inner, bias=False)
    a.to_out = nn.ModuleList([nn.Linear(inner, qdim), nn.Identity()])
    return a


def _transformer2d(ch, xdim, groups):
    t = _Bag()
    t.norm = nn.GroupNorm(groups, ch, eps=1e-6)
    t.proj_in = nn.Conv2d(ch, ch, 1)
    b = _Bag()
    b.norm1 = nn.LayerNorm(ch)
    b.attn1 = _attention(ch, None, ch)
    b.norm2 = nn.LayerNorm(ch)
    b.attn2 = _attention(ch, xdim, ch)
    b.norm3 = nn.LayerNorm(ch)
    b.ff = _Bag()
    proj = _Bag()
    proj.proj = nn.Linear(ch, 8 * ch)
    b.ff.net = nn.ModuleList([proj, nn.Identity(), nn.Linear(4 * ch, ch)])
    t.transformer_blocks = nn.ModuleList([b])
    t.proj_out = nn.Conv2d(ch, ch, 1)
    return t


def _sampler(ch, stride):
    s = _Bag()
    s.conv = nn.Conv2d(ch, ch, 3, stride=stride, padding=1)
    return s


class _HipBackbone(nn.Module):
    """Shared plumbing: compute dtype, packed-weight cache, plan cache."""

    def _init_engine(self):
        self.compute_dtype = "f32"
        self.share_trunk = True
        self._packed = {}
        self._plans = {}
        self._wver = 0               # bumped whenever packed weights go stale; plan caches outside this module key on it
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate_packed())

    def set_compute_dtype(self, name):
        assert name in E.DT, name
        self.compute_dtype = name
        return self

    def invalidate_packed(self):
        """Call after mutating parameters in place (load_state_dict does it automatically)."""
        self._packed.clear()
        self._plans.clear()
        self._wver += 1

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        if hasattr(self, "_packed"):
            self.invalidate_packed()
        return out

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_packed", "_plans"):
                new.__dict__[k] = {}
            elif k == "_wver":
                new.__dict__[k] = 0
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new


class UNetCondition2D(_HipBackbone):
    def __init__(
        self,
        sample_size: Optional[int] = None,
        in_channels: int = 4,
        out_channels: int = 4,
        center_input_sample: bool = False,
        flip_sin_to_cos: bool = True,
        freq_shift: int = 0,
        down_block_types: Tuple[str] = ("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
        mid_block_type: Optional[str] = "UNetMidBlock2DCrossAttn",
        up_block_types: Tuple[str] = ("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
        only_cross_attention: Union[bool, Tuple[bool]] = False,
        block_out_channels: Tuple[int] = (320, 640, 1280, 1280),
        layers_per_block: Union[int, Tuple[int]] = 2,
        downsample_padding: int = 1,
        mid_block_scale_factor: float = 1,
        dropout: float = 0.0,
        act_fn: str = "silu",
        norm_num_groups: Optional[int] = 32,
        norm_eps: float = 1e-5,
        cross_attention_dim: Union[int, Tuple[int]] = 1280,
        transformer_layers_per_block: Union[int, Tuple[int], Tuple[Tuple]] = 1,
        reverse_transformer_layers_per_block=None,
        encoder_hid_dim: Optional[int] = None,
        encoder_hid_dim_type: Optional[str] = None,
        attention_head_dim: Union[int, Tuple[int]] = 8,
        num_attention_heads=None,
        dual_cross_attention: bool = False,
        use_linear_projection: bool = False,
        class_embed_type: Optional[str] = None,
        addition_embed_type: Optional[str] = None,
        addition_time_embed_dim: Optional[int] = None,
        num_class_embeds: Optional[int] = None,
        upcast_attention: bool = False,
        resnet_time_scale_shift: str = "default",
        resnet_skip_time_act: bool = False,
        resnet_out_scale_factor: float = 1.0,
        time_embedding_type: str = "positional",
        time_embedding_dim: Optional[int] = None,
        time_embedding_act_fn: Optional[str] = None,
        timestep_post_act: Optional[str] = None,
        time_cond_proj_dim: Optional[int] = None,
        conv_in_kernel: int = 3,
        conv_out_kernel: int = 3,
        projection_class_embeddings_input_dim: Optional[int] = None,
        attention_type: str = "default",
        class_embeddings_concat: bool = False,
        mid_block_only_cross_attention: Optional[bool] = None,
        cross_attention_norm: Optional[str] = None,
        addition_embed_type_num_heads: int = 64,
    ):
        super().__init__()
        # The reference only ever instantiates this subset; anything else is refused loudly.
        unsupported = dict(
            center_input_sample=(center_input_sample, False), only_cross_attention=(only_cross_attention, False),
            downsample_padding=(downsample_padding, 1), mid_block_scale_factor=(mid_block_scale_factor, 1),
            dropout=(dropout, 0.0), act_fn=(act_fn, "silu"), transformer_layers_per_block=(transformer_layers_per_block, 1),
            num_attention_heads=(num_attention_heads, None), dual_cross_attention=(dual_cross_attention, False),
            use_linear_projection=(use_linear_projection, False), class_embed_type=(class_embed_type, None),
            addition_embed_type=(addition_embed_type, None), num_class_embeds=(num_class_embeds, None),
            resnet_time_scale_shift=(resnet_time_scale_shift, "default"), resnet_skip_time_act=(resnet_skip_time_act, False),
            resnet_out_scale_factor=(resnet_out_scale_factor, 1.0), time_embedding_type=(time_embedding_type, "positional"),
            time_embedding_dim=(time_embedding_dim, None), time_cond_proj_dim=(time_cond_proj_dim, None),
            conv_in_kernel=(conv_in_kernel, 3), conv_out_kernel=(conv_out_kernel, 3), attention_type=(attention_type, "default"),
            mid_block_type=(mid_block_type, "UNetMidBlock2DCrossAttn"), encoder_hid_dim_type=(encoder_hid_dim_type, "text_proj"),
            cross_attention_norm=(cross_attention_norm, None), timestep_post_act=(timestep_post_act, None),
            time_embedding_act_fn=(time_embedding_act_fn, None))
        for k, (got, want) in unsupported.items():
            if got != want:
                raise NotImplementedError(f"UNetCondition2D({k}={got!r}) is outside the scoring path built here (supported: {want!r})")
        if not isinstance(attention_head_dim, int) or not isinstance(cross_attention_dim, int):
            raise NotImplementedError("per-block attention_head_dim / cross_attention_dim tuples are not supported")
        boc = tuple(block_out_channels)
        nb = len(boc)
        lpb = (layers_per_block,) * nb if isinstance(layers_per_block, int) else tuple(layers_per_block)
        assert len(down_block_types) == nb and len(up_block_types) == nb and len(lpb) == nb
        G, eps, xdim = norm_num_groups, norm_eps, cross_attention_dim
        self.config = SimpleNamespace(
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types), mid_block_type=mid_block_type,
            block_out_channels=boc, layers_per_block=lpb, norm_num_groups=G, norm_eps=eps, cross_attention_dim=xdim,
            encoder_hid_dim=encoder_hid_dim, encoder_hid_dim_type=encoder_hid_dim_type,
            attention_head_dim=attention_head_dim, flip_sin_to_cos=flip_sin_to_cos, freq_shift=freq_shift)
        temb = boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_embedding = _Bag()
        self.time_embedding.linear_1 = nn.Linear(boc[0], temb)
        self.time_embedding.linear_2 = nn.Linear(temb, temb)
        self.encoder_hid_proj = nn.Linear(encoder_hid_dim, xdim)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i, kind in enumerate(down_block_types):
            if kind not in ("DownBlock2D", "CrossAttnDownBlock2D"):
                raise NotImplementedError(kind)
            cin, out = out, boc[i]
            blk = _Bag()
            blk.resnets = nn.ModuleList([_resnet(cin if j == 0 else out, out, temb, G, eps) for j in range(lpb[i])])
            if kind == "CrossAttnDownBlock2D":
                blk.attentions = nn.ModuleList([_transformer2d(out, xdim, G) for _ in range(lpb[i])])
            if i != nb - 1:
                blk.downsamplers = nn.ModuleList([_sampler(out, 2)])
            self.down_blocks.append(blk)
        self.mid_block = _Bag()
        self.mid_block.attentions = nn.ModuleList([_transformer2d(boc[-1], xdim, G)])
        self.mid_block.resnets = nn.ModuleList([_resnet(boc[-1], boc[-1], temb, G, eps) for _ in range(2)])
        self.up_blocks = nn.ModuleList()
        rboc, rlpb = boc[::-1], lpb[::-1]
        out = rboc[0]
        for i, kind in enumerate(up_block_types):
            if kind not in ("UpBlock2D", "CrossAttnUpBlock2D"):
                raise NotImplementedError(kind)
            prev, out = out, rboc[i]
            cin = rboc[min(i + 1, nb - 1)]
            n = rlpb[i] + 1
            blk = _Bag()
            blk.resnets = nn.ModuleList(
                [_resnet((prev if j == 0 else out) + (cin if j == n - 1 else out), out, temb, G, eps) for j in range(n)])
            if kind == "CrossAttnUpBlock2D":
                blk.attentions = nn.ModuleList([_transformer2d(out, xdim, G) for _ in range(n)])
            if i != nb - 1:
                blk.upsamplers = nn.ModuleList([_sampler(out, 1)])
            self.up_blocks.append(blk)
        self.conv_norm_out = nn.GroupNorm(G, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)
        self._init_engine()

    # ---- engine hooks -----------------------------------------------------------------
    def packed_weights(self, dt, device):
        key = (dt, str(device))
        if key not in self._packed:
            self._packed[key] = E.UNetWeights(self, dt, device)
        return self._packed[key]

    def make_plan(self, n_bj, n_cls, n_ctx, device, score=None, share_trunk=None):
        dt = E.DT[self.compute_dtype]
        w = self.packed_weights(dt, device)
        return E.UNetPlan(self, w, n_bj, n_cls, n_ctx, share_trunk=self.share_trunk if share_trunk is None else share_trunk,
                          score=score, device=device)

    @torch.no_grad()
    def forward(self, x, noise_labels, downblock_additional_residuals=None, midblock_additional_residuals=None,
                encoder_hidden_states=None):
        if downblock_additional_residuals is not None or midblock_additional_residuals is not None:
            raise NotImplementedError("ControlNet residuals are not on the scoring path")
        L.require_gpu()
        if not x.is_cuda:
            raise L.DcamdError("UNetCondition2D.forward needs CUDA/HIP tensors (no CPU fallback)")
        dev = x.device
        N, Cin, H, W = x.shape
        assert encoder_hidden_states is not None and encoder_hidden_states.shape[1] == 1, \
            "the scoring path conditions on exactly one class token [N,1,hid]"
        key = ("fwd", N, str(dev), self.compute_dtype, self.share_trunk)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self.make_plan(N, 1, N, dev)
        self._feed(plan, x, noise_labels)
        plan.ctx.copy_(encoder_hidden_states[:, 0].to(dev, torch.float32))
        plan.run_ctx()
        plan.run()
        return plan.pred_view().permute(0, 3, 1, 2).contiguous().to(x.dtype)


    @torch.no_grad()
    def forward_pair(self, x, noise_labels, cond, null):
        """Classifier-free-guidance pair as ONE batch-2 plan launch (reference `sample`, :255-266, calls the backbone twice per
        step): unit 2b scores image b under its class token `cond[b]`, unit 2b+1 under the null token `null[b]`; every layer in
        front of the first cross-attention runs once per image (class-shared trunk).  Returns the plan's prediction buffer
        [2N, H, W, ld] fp32 NHWC (the layout `dc_ddpm_step` reads) — a view that the next call overwrites."""
        L.require_gpu()
        if not x.is_cuda:
            raise L.DcamdError("UNetCondition2D.forward_pair needs CUDA/HIP tensors (no CPU fallback)")
        dev = x.device
        N, Cin, H, W = x.shape
        key = ("pair", N, str(dev), self.compute_dtype, self.share_trunk)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self.make_plan(N, 2, 2 * N, dev)        # ctx_of_unit = unit index: one context row per unit
        self._feed(plan, x, noise_labels)
        ctx = torch.stack([cond[:, 0], null[:, 0]], dim=1).reshape(2 * N, -1)
        plan.ctx.copy_(ctx.to(dev, torch.float32))
        plan.run_ctx()
        plan.run()
        return plan.pred_view()

    def _feed(self, plan, x, noise_labels):
        """lambda and the conv_in GEMM operand (3x3 patches of x) of a plain forward."""
        lib = L.lib()
        dev = x.device
        N, Cin, H, W = x.shape
        lam = noise_labels if torch.is_tensor(noise_labels) else torch.tensor([noise_labels])
        lam = lam.to(dev, torch.float32).reshape(-1)
        plan.lam.copy_(lam.expand(N) if lam.numel() == 1 else lam)
        xf = x.detach().to(torch.float32).contiguous()
        ones = torch.ones(N, dtype=torch.float32, device=dev)
        zeros = torch.zeros(N, dtype=torch.float32, device=dev)
        p = L.QsampleParams(x=xf.data_ptr(), eps=xf.data_ptr(), alpha=ones.data_ptr(), sigma=zeros.data_ptr(), img_of_bj=None,
                            out=plan.a0_buf.data_ptr(), out_dtype=plan.dt, n_bj=N, C=Cin, H=H, W=W,
                            ld=plan.a0_buf.shape[-1], im2col=1)
        L.check(lib.dc_qsample(p, L.stream_ptr()), "dc_qsample")


class UNet2D(nn.Module):
    """Signature of reference `nets/unet.py:8-71` (diffusers `UNet2DModel`).  No experiment or
    model config of the reference uses it, so only the constructor/forward
    signature is kept; the unconditional backbone has no class input to score against."""

    def __init__(self, sample_size=None, in_channels: int = 3, out_channels: int = 3, center_input_sample: bool = False,
                 time_embedding_type: str = "positional", freq_shift: int = 0, flip_sin_to_cos: bool = True,
                 down_block_types=("DownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D"),
                 up_block_types=("AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
                 block_out_channels=(224, 448, 672, 896), layers_per_block: int = 2, mid_block_scale_factor: float = 1,
                 downsample_padding: int = 1, downsample_type: str = "conv", upsample_type: str = "conv", dropout: float = 0.0,
                 act_fn: str = "silu", attention_head_dim: Optional[int] = 8, norm_num_groups: int = 32,
                 attn_norm_num_groups: Optional[int] = None, norm_eps: float = 1e-5, resnet_time_scale_shift: str = "default",
                 add_attention: bool = True, class_embed_type: Optional[str] = None, num_class_embeds: Optional[int] = None,
                 num_train_timesteps: Optional[int] = None):
        super().__init__()
        self.config = SimpleNamespace(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      encoder_hid_dim=None)

    def forward(self, x, noise_labels):
        raise NotImplementedError(
            "UNet2D (unconditional diffusers UNet2DModel) is not on the diffusion-classifier scoring path: it takes no "
            "class input. Use UNetCondition2D or DiT.")
