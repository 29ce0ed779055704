"""DiT (adaLN-Zero transformer) scoring plan for libdcamd — see engine.py for the plan model.

Backbone restated: diffusers 0.31.0 `DiTTransformer2DModel` behind reference nets/dit.py:8-51
(instance: models/chexpert-256-dit-b4.py).  Class conditioning enters every block through
adaLN, so only q_sample / patch tokens and the timestep MLPs are shared per (image, trial);
the per-class part of the conditioning is a table row added in a GEMM epilogue.
"""
import torch

from . import _lib as L
from .engine import PlanBuilder, TORCH_DT, bke, f32c, pack_matrix, round_up


class DiTWeights:
    def __init__(self, model, dt, device):
        self.dt, self.dev = dt, device
        cfg = model.config
        sd = model.state_dict()
        P = {}
        p, cin = cfg.patch_size, cfg.in_channels
        self.kin = round_up(cin * p * p, bke(dt))
        w = sd["pos_embed.proj.weight"].reshape(model.D, cin * p * p)        # k = c*p*p + py*p + px
        P["patch.w"] = pack_matrix(w, dt, device, kpad=self.kin)
        P["patch.b"] = f32c(sd["pos_embed.proj.bias"], device)
        P["pos"] = model.pos_embed.pos_embed.detach().to(device=device, dtype=TORCH_DT[dt]).reshape(-1, model.D).contiguous()
        nl = cfg.num_layers
        self.nl = nl
        t1w = torch.cat([sd[f"transformer_blocks.{i}.norm1.emb.timestep_embedder.linear_1.weight"] for i in range(nl)], 0)
        t1b = torch.cat([sd[f"transformer_blocks.{i}.norm1.emb.timestep_embedder.linear_1.bias"] for i in range(nl)], 0)
        P["t1.w"], P["t1.b"] = pack_matrix(t1w, L.DC_F32, device), f32c(t1b, device)
        for i in range(nl):
            k = f"transformer_blocks.{i}"
            e = k + ".norm1.emb"
            P[k + ".t2.w"] = pack_matrix(sd[e + ".timestep_embedder.linear_2.weight"], L.DC_F32, device)
            P[k + ".t2.b"] = f32c(sd[e + ".timestep_embedder.linear_2.bias"], device)
            P[k + ".table"] = f32c(sd[e + ".class_embedder.embedding_table.weight"], device)
            P[k + ".mod.w"] = pack_matrix(sd[k + ".norm1.linear.weight"], L.DC_F32, device)
            P[k + ".mod.b"] = f32c(sd[k + ".norm1.linear.bias"], device)
            qkv_w = torch.cat([sd[k + f".attn1.to_{n}.weight"] for n in "qkv"], 0)
            qkv_b = torch.cat([sd[k + f".attn1.to_{n}.bias"] for n in "qkv"], 0)
            P[k + ".qkv.w"], P[k + ".qkv.b"] = pack_matrix(qkv_w, dt, device), f32c(qkv_b, device)
            P[k + ".out.w"] = pack_matrix(sd[k + ".attn1.to_out.0.weight"], dt, device)
            P[k + ".out.b"] = f32c(sd[k + ".attn1.to_out.0.bias"], device)
            P[k + ".ff1.w"] = pack_matrix(sd[k + ".ff.net.0.proj.weight"], dt, device)
            P[k + ".ff1.b"] = f32c(sd[k + ".ff.net.0.proj.bias"], device)
            P[k + ".ff2.w"] = pack_matrix(sd[k + ".ff.net.2.weight"], dt, device)
            P[k + ".ff2.b"] = f32c(sd[k + ".ff.net.2.bias"], device)
        P["po1.w"], P["po1.b"] = pack_matrix(sd["proj_out_1.weight"], L.DC_F32, device), f32c(sd["proj_out_1.bias"], device)
        P["po2.w"], P["po2.b"] = pack_matrix(sd["proj_out_2.weight"], dt, device), f32c(sd["proj_out_2.bias"], device)
        self.P = P


class DiTPlan:
    """inputs: lam [n_bj]; ctx_of_unit [U] = class label per unit (row of the embedding tables);
    a0 = patch tokens [n_bj, g, g, kin].  output: pred [U, g, g, p*p*out_ch] f32 (un-patchify is
    folded into dc_eps_mse / done by the caller)."""

    def __init__(self, model, weights, n_bj, n_cls, n_ctx, *, score=None, device=None, share_trunk=True):
        cfg = model.config
        dev = device or weights.dev
        dt = weights.dt
        P = weights.P
        D, heads, p = model.D, cfg.num_attention_heads, cfg.patch_size
        H = W = cfg.sample_size
        g = H // p
        U = n_bj * n_cls
        n_tab = P["transformer_blocks.0.table"].shape[0]
        self.dt, self.n_bj, self.n_cls = dt, n_bj, n_cls
        pb = self.pb = PlanBuilder(dev, n_bj, n_cls, n_tab)
        i32 = dict(dtype=torch.int32, device=dev)
        self.lam = score["lam"] if score is not None and "lam" in score else torch.zeros(n_bj, dtype=torch.float32, device=dev)
        self.ctx = None
        self.bj_of_unit = (torch.arange(U, **i32) // n_cls).contiguous()
        self.ctx_of_unit = score["ctx_of_unit"] if score is not None and "ctx_of_unit" in score else torch.zeros(U, **i32)
        self.zero_map = torch.zeros(U, **i32)
        pb.set_map("bj", "unit", self.bj_of_unit)
        pb.set_map("ctx", "unit", self.ctx_of_unit)
        pb.set_map("pos", "unit", self.zero_map)
        pb.n["pos"] = 1
        lam = pb.external("lam", self.lam, "bj", 1, 1, 1, L.DC_F32)
        kin = weights.kin
        if score is not None:
            eps_t = pb.external("eps", score["eps"], "bj", 1, 1, 1, L.DC_F32)
            al = pb.external("alpha", score["alpha"], "bj", 1, 1, 1, L.DC_F32)
            sg = pb.external("sigma", score["sigma"], "bj", 1, 1, 1, L.DC_F32)
            a0 = pb.qsample("a0", pb.const(score["x"]), eps_t, al, sg, pb.const(score["img_of_bj"]),
                            cfg.in_channels, H, W, kin, dt, im2col=2, patch=p)
        else:
            self.a0_buf = torch.zeros(n_bj, g, g, kin, dtype=TORCH_DT[dt], device=dev)
            a0 = pb.external("a0", self.a0_buf, "bj", g, g, kin, dt)
        pos = pb.external("pos", P["pos"], "pos", g, g, D, dt)
        # fp32 side path
        tsin = pb.sinusoid("t.sin", lam, 256, True, 1.0)
        t1 = pb.igemm("t.l1", tsin, pb.const(P["t1.w"]), weights.nl * D, bias=pb.const(P["t1.b"]), act=L.ACT_SILU)
        conds, mods = [], []
        for i in range(weights.nl):
            k = f"transformer_blocks.{i}"
            table = pb.external(k + ".table", P[k + ".table"], "ctx", 1, 1, D, L.DC_F32)
            c = pb.igemm(k + ".cond", t1.view(i * D, D), pb.const(P[k + ".t2.w"]), D, bias=pb.const(P[k + ".t2.b"]),
                         rowvec=table, act=L.ACT_SILU)                       # SiLU(t_emb + class_emb), per unit
            conds.append(c)
            mods.append(pb.igemm(k + ".mod", c, pb.const(P[k + ".mod.w"]), 6 * D, bias=pb.const(P[k + ".mod.b"])))
        fin = pb.igemm("final.mod", conds[0], pb.const(P["po1.w"]), 2 * D, bias=pb.const(P["po1.b"]))
        # main path
        h = pb.igemm("patch", a0, pb.const(P["patch.w"]), D, bias=pb.const(P["patch.b"]), residual=pos, dom="unit")
        for i in range(weights.nl):
            k = f"transformer_blocks.{i}"
            m = mods[i]
            hn = pb.layernorm(k + ".ln1", h, None, None, 1e-6, scale=m.view(D, D), shift=m.view(0, D))
            qkv = pb.igemm(k + ".qkv", hn, pb.const(P[k + ".qkv.w"]), 3 * D, bias=pb.const(P[k + ".qkv.b"]))
            o = pb.attention(k + ".attn", qkv.view(0, D), qkv.view(D, D), qkv.view(2 * D, D), heads)
            h = pb.igemm(k + ".attn_out", o, pb.const(P[k + ".out.w"]), D, bias=pb.const(P[k + ".out.b"]),
                         gate=m.view(2 * D, D), residual=h)
            hn = pb.layernorm(k + ".ln2", h, None, None, cfg.norm_eps, scale=m.view(4 * D, D), shift=m.view(3 * D, D))
            f = pb.igemm(k + ".ff1", hn, pb.const(P[k + ".ff1.w"]), 4 * D, bias=pb.const(P[k + ".ff1.b"]), act=L.ACT_GELU_TANH)
            h = pb.igemm(k + ".ff2", f, pb.const(P[k + ".ff2.w"]), D, bias=pb.const(P[k + ".ff2.b"]),
                         gate=m.view(5 * D, D), residual=h)
        hn = pb.layernorm("final.ln", h, None, None, 1e-6, scale=fin.view(D, D), shift=fin.view(0, D))
        oc = cfg.out_channels
        pred = pb.igemm("proj_out_2", hn, pb.const(P["po2.w"]), p * p * oc, bias=pb.const(P["po2.b"]), out_dt=L.DC_F32,
                        tile_n=32 if p * p * oc <= 32 else 128)
        self.pred = pred
        if score is not None:
            pb.eps_mse(pred, eps_t, pb.const(score["x"]), al, sg, pb.const(self.bj_of_unit), pb.const(score["img_of_bj"]),
                       pb.const(score["out_index"]), pb.const(score["errors"]), cfg.in_channels, score["v_param"], patch=p)
        pb.finalize(keep_alive=[pred])

    def run(self):
        self.pb.run()

    def run_ctx(self):
        pass   # class conditioning is a table row picked inside the plan

    def pred_view(self):
        return self.pb.tensor_view(self.pred)

    @property
    def arena_bytes(self):
        return self.pb.arena_bytes
