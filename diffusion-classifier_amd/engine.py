"""Scoring-step engine: turns a backbone into a static launch plan for libdcamd.

A *plan* is a flat array of `dc_op` records (include/dcamd.h) over one arena allocation.
It is built once per (backbone, dtype, n_bj, n_cls) shape, then replayed with ONE native
call (`dc_run_plan`) per micro-batch — the replacement for the reference's Python double
loop body (diffusion/diffusion_classifier.py:695-714; ~600 eager launches per forward).

Work unit = (image b, trial j, class c).  Tensors live in one of three *domains*:
  'bj'   one sample per (image, trial) pair  — q_sample, time embedding, and every layer
         before the first cross-attention (the class-shared trunk: computed once per pair,
         never per class; exact, because class conditioning enters only through attn2);
  'unit' one sample per (pair, class);
  'ctx'  one row per class (class-token side path).
When a 'unit' op reads a 'bj'/'ctx' tensor the kernel indexes it through an int32 map
(bj_of_unit / ctx_of_unit), so nothing is ever broadcast-materialised and skip
connections are never concatenated in memory.

PyTorch is used for device memory and streams only.
"""
import ctypes as C

import torch

from . import _lib as L

DT = {"f32": L.DC_F32, "fp32": L.DC_F32, "float32": L.DC_F32, "bf16": L.DC_BF16, "bfloat16": L.DC_BF16,
      "f16": L.DC_F16, "fp16": L.DC_F16, "float16": L.DC_F16}
TORCH_DT = {L.DC_F32: torch.float32, L.DC_BF16: torch.bfloat16, L.DC_F16: torch.float16}
DT_SIZE = {L.DC_F32: 4, L.DC_BF16: 2, L.DC_F16: 2}


def bke(dt):
    """K granule (elements per 128-byte LDS row) of dc_igemm for a dtype."""
    return 128 // DT_SIZE[dt]


def round_up(a, b):
    return (a + b - 1) // b * b


class TRef:
    """Handle of a plan tensor [n(dom), H, W, C] (NHWC) or a view of one."""
    __slots__ = ("name", "dom", "H", "W", "C", "dt", "ld", "eoff", "base", "nbytes", "first", "last", "off", "ext", "qs", "prod")

    def __init__(self, name, dom, H, W, C, dt, nbytes=0, ext=None):
        self.name, self.dom, self.H, self.W, self.C, self.dt = name, dom, H, W, C, dt
        self.ld, self.eoff, self.base = C, 0, self
        self.nbytes, self.first, self.last, self.off, self.ext = nbytes, None, None, None, ext
        self.qs = None       # (quad-statistics tensor, parts) written by the producing conv (dc_igemm_params.qstats)
        self.prod = None     # index of the producing dc_igemm op while it can still be asked to normalise this tensor (pn_claim)

    def view(self, coff, C):
        v = TRef(self.name + f"[{coff}:{coff + C}]", self.dom, self.H, self.W, C, self.dt)
        v.ld, v.eoff, v.base = self.ld, self.eoff + coff, self.base
        return v


class PlanBuilder:
    def __init__(self, device, n_bj, n_cls, n_ctx):
        self.dev = device
        self.n = {"bj": n_bj, "unit": n_bj * n_cls, "ctx": n_ctx}
        self.ops = []        # (kind, struct_cls, fields{name: value | TRef | ("ptr", TRef)}, reads, writes)
        self.keep = []       # torch tensors kept alive (weights, maps, inputs)
        self.maps = {}       # (src_dom, dst_dom) -> external int32 TRef
        self.arena = None
        self.arena_bytes = 0
        self.meta = []       # per op: name, kernel family, algorithmic flops / HBM bytes (bench.py roofline)
        import os
        self.pn_off = os.environ.get("DCAMD_NO_PN") is not None      # A/B runs and tests: never ask a producer to normalise

    # ---- tensors -------------------------------------------------------------------
    def tensor(self, name, dom, H, W, Cc, dt):
        nbytes = self.n[dom] * H * W * Cc * DT_SIZE[dt]
        return TRef(name, dom, H, W, Cc, dt, nbytes=round_up(nbytes, 256))

    def external(self, name, t, dom, H, W, Cc, dt):
        assert t.is_contiguous() and t.device.type == "cuda", name
        self.keep.append(t)
        return TRef(name, dom, H, W, Cc, dt, ext=t)

    def const(self, t):
        """Device pointer of a constant (weight / bias / map) tensor kept alive by the plan."""
        if t is None:
            return None
        assert t.is_contiguous() and t.device.type == "cuda"
        self.keep.append(t)
        return t.data_ptr()

    def set_map(self, src_dom, dst_dom, t):
        self.maps[(src_dom, dst_dom)] = self.const(t)

    def _map(self, src, dst_dom):
        if src is None or src.dom == dst_dom:
            return None
        return self.maps[(src.dom, dst_dom)]

    @staticmethod
    def _dom(*ts):
        doms = {t.dom for t in ts if t is not None}
        if "unit" in doms or len(doms) > 1:   # bj x ctx -> one sample per (pair, class)
            return "unit"
        return doms.pop()

    def _emit(self, kind, cls, fields, reads, writes, meta=None):
        idx = len(self.ops)
        self.meta.append(dict(meta or {}, kind=kind))
        for t in reads + writes:
            if t is None:
                continue
            b = t.base
            if b.ext is None:
                if b.first is None:
                    b.first = idx
                b.last = idx
        self.ops.append((kind, cls, fields))

    # ---- ops -----------------------------------------------------------------------
    def igemm(self, name, src0, W, Cout, *, taps=1, stride=1, upsample=0, src1=None, bias=None, rowvec=None,
              act=L.ACT_NONE, gate=None, residual=None, out_dt=None, tile_n=128, wdt=None, dom=None, k_real=None, gn=None,
              side=None, ln_eps=0.0, qstats=False, up4=False):
        dom = dom or self._dom(src0, src1, rowvec, gate, residual)
        Hin, Win = (src0.H * 2, src0.W * 2) if upsample else (src0.H, src0.W)
        if taps == 9:
            Hout, Wout = (Hin + 2 - 3) // stride + 1, (Win + 2 - 3) // stride + 1
        else:
            Hout, Wout = Hin, Win
        cout_out = Cout // 2 if act == L.ACT_GEGLU else Cout
        out = self.tensor(name, dom, Hout, Wout, cout_out, src0.dt if out_dt is None else out_dt)
        f = dict(dtype=src0.dt if wdt is None else wdt, taps=taps, stride=stride, upsample=upsample,
                 n_img=self.n[dom], Hin=Hin, Win=Win, Hout=Hout, Wout=Wout,
                 src0=src0, map0=self._map(src0, dom), C0=src0.C, ld0=src0.ld,
                 src1=src1, map1=self._map(src1, dom), C1=src1.C if src1 is not None else 0,
                 ld1=src1.ld if src1 is not None else 0,
                 W=W, Cout=Cout, tile_n=tile_n, bias=bias,
                 rowvec=rowvec, rowvec_map=self._map(rowvec, dom), rowvec_ld=rowvec.ld if rowvec is not None else 0,
                 act=act, gate=gate, gate_map=self._map(gate, dom), gate_ld=gate.ld if gate is not None else 0,
                 residual=residual, res_map=self._map(residual, dom),
                 res_dtype=residual.dt if residual is not None else 0,
                 res_ld=residual.ld if residual is not None else 0,
                 out=out, out_dtype=out.dt, out_ld=out.ld)
        if up4:                  # W is the four-phase form of an upsample conv's weights (pack_up4, dc_igemm_up4_ok)
            assert upsample and taps == 9
            f.update(up4=1)
            k_real = 4 * src0.C
        if ln_eps:               # row LayerNorm (no affine) of the A operand inside the GEMM (dc_igemm_ln_ok)
            f.update(ln_eps=float(ln_eps))
        if side is not None:     # (src2, W2): 1x1 side source summed into the same output (conv_shortcut folded into conv2)
            s2, W2 = side
            assert (s2.H, s2.W) == (Hout, Wout) and s2.dt == src0.dt
            f.update(src2=s2, map2=self._map(s2, dom), W2=W2, C2=s2.C, ld2=s2.ld)
        if gn is not None:       # fused GroupNorm(+SiLU) prologue: (scale, shift, silu) from groupnorm_stats
            assert gn[0].dom == dom and gn[0].C == f["C0"] + f["C1"], "GroupNorm affine must live in the conv's domain"
            f.update(gn_scale=gn[0], gn_shift=gn[1], gn_silu=int(gn[2]))
        if src1 is not None:
            assert src1.dt == src0.dt and (src1.H, src1.W) == (src0.H, src0.W)
        qs = None
        if qstats and L.lib().dc_igemm_qstats_parts is not None:
            # the conv also writes (mean, M2) per (sample, part, channel quad) of its output: the GroupNorm that consumes the
            # tensor then streams it once (read + write) instead of sweeping it twice
            fake = 1 << 20
            probe = L.IgemmParams(**{k: (fake if isinstance(v, TRef) else v) for k, v in f.items() if v is not None})
            parts = int(L.lib().dc_igemm_qstats_parts(probe))
            if parts > 0:
                qs = TRef(name + ".qs", dom, 1, 1, 1, L.DC_F32, nbytes=round_up(self.n[dom] * parts * (Cout // 4) * 2 * 4, 256))
                f.update(qstats=qs)
                out.qs = (qs, parts)
        M = self.n[dom] * Hout * Wout
        kreal = taps * (src0.C + (src1.C if src1 is not None else 0)) if k_real is None else k_real
        es = DT_SIZE[f["dtype"]]
        nsrc = {"bj": self.n["bj"], "unit": self.n["unit"], "ctx": self.n["ctx"]}
        in_bytes = sum(nsrc.get(t.dom, 1) * t.H * t.W * t.C * DT_SIZE[t.dt] for t in (src0, src1, residual) if t is not None)
        meta = dict(name=name, family=f"igemm_{'f32' if f['dtype'] == L.DC_F32 else 'bf16' if f['dtype'] == L.DC_BF16 else 'f16'}_n{tile_n}",
                    flops=2.0 * M * kreal * Cout, bytes=float(in_bytes + kreal * Cout * es + M * cout_out * DT_SIZE[out.dt]),
                    M=M, N=Cout, K=kreal, taps=taps)
        if side is not None:
            meta["flops"] += 2.0 * M * side[0].C * Cout
            meta["K"] = kreal + side[0].C
            meta["bytes"] += float(nsrc.get(side[0].dom, 1) * side[0].H * side[0].W * side[0].C * DT_SIZE[side[0].dt] + side[0].C * Cout * es)
        self._emit(L.OP_IGEMM, L.IgemmParams, f, [src0, src1, rowvec, gate, residual] + (list(gn[:2]) if gn else []) + ([side[0]] if side else []), [out] + ([qs] if qs else []), meta)
        if qstats and gn is None and (qs is not None or (taps == 9 and stride == 1 and not upsample and Hin == 4 and Win == 4)):
            out.prod = len(self.ops) - 1       # (4x4 images: no quad records, but the conv can still normalise its output inside the wave)
        return out

    def pn_capable(self, x, Cout, dom):
        """Would a plain 3x3 stride-1 conv of x (single source) be able to normalise its own output (dc_igemm_pn_ok)?"""
        if self.pn_off or L.lib().dc_igemm_pn_ok is None or Cout % 32:
            return False
        fake = 1 << 20
        p = L.IgemmParams(dtype=x.dt, taps=9, stride=1, upsample=0, n_img=self.n[dom], Hin=x.H, Win=x.W, Hout=x.H, Wout=x.W,
                          src0=fake, C0=x.C, ld0=x.ld, W=fake, Cout=Cout, tile_n=128, out=fake, out_dtype=x.dt, out_ld=Cout,
                          pn_groups=32, pn_eps=1e-5)
        return bool(L.lib().dc_igemm_pn_ok(p))

    def pn_claim(self, x, gamma, beta, groups, eps, silu):
        """Producer-side GroupNorm (dc_igemm_params.pn_*; csrc/epi_pn.h): ask the 3x3 conv that produced x to ALSO store
        act(GroupNorm(x)) — its accumulators hold every value in fp32 and it writes the statistics anyway — so that the consumer of
        this GroupNorm reads a finished tensor with a plain conv / GEMM and no GroupNorm pass (nor a normalising loader) runs.
        Returns the normalised tensor, or None when the producer cannot (then the caller takes the GroupNorm pass / the fused loader).
        One claim per tensor; whether the raw x is still stored is decided in finalize() (only if something reads it)."""
        if x is None or x.prod is None or x.base is not x or self.pn_off or L.lib().dc_igemm_pn_ok is None:
            return None
        idx = x.prod
        kind, cls, f = self.ops[idx]
        if f.get("pn_out") is not None or f["Cout"] % groups:
            return None
        fake = 1 << 20
        probe = L.IgemmParams(**{k: (fake if isinstance(v, TRef) else v) for k, v in f.items() if v is not None}, pn_groups=groups, pn_eps=float(eps))
        if not L.lib().dc_igemm_pn_ok(probe):
            return None
        dom = x.dom
        y = self.tensor(x.name + ".pn", dom, x.H, x.W, x.C, x.dt)
        # arrival counters of this call site: monotonic, zeroed once here, never part of the (reused) arena
        cnt = self.const(torch.zeros(self.n[dom] * ((f["Cout"] + 127) // 128), dtype=torch.int32, device=self.dev))
        y.first = y.last = idx
        f.update(pn_out=y, pn_gamma=gamma, pn_beta=beta, pn_cnt=cnt, pn_ld=y.ld, pn_groups=groups, pn_silu=int(silu), pn_eps=float(eps))
        self.meta[idx]["bytes"] += float(self.n[dom] * x.H * x.W * x.C * DT_SIZE[x.dt])
        self.meta[idx]["pn"] = True
        x.prod = None
        return y

    def up4_ok(self, src0, Cout):
        """Can the upsample conv of src0 run as four 2x2-tap phases on the low-resolution image (dc_igemm_up4_ok)?"""
        fake = 1 << 20
        p = L.IgemmParams(dtype=src0.dt, taps=9, stride=1, upsample=1, n_img=self.n[src0.dom], Hin=2 * src0.H, Win=2 * src0.W,
                          Hout=2 * src0.H, Wout=2 * src0.W, src0=fake, C0=src0.C, ld0=src0.ld, W=fake, Cout=Cout, tile_n=128,
                          bias=fake, out=fake, out_dtype=src0.dt, out_ld=Cout, up4=1)
        return bool(L.lib().dc_igemm_up4_ok(p))

    def ln_ok(self, src0, Cout, act=L.ACT_NONE):
        """Can this 1-tap GEMM normalise its input rows itself (dc_igemm_ln_ok)?"""
        fake = 1 << 20
        cout_out = Cout // 2 if act == L.ACT_GEGLU else Cout
        p = L.IgemmParams(dtype=src0.dt, taps=1, stride=1, upsample=0, n_img=self.n[src0.dom], Hin=src0.H, Win=src0.W,
                          Hout=src0.H, Wout=src0.W, src0=fake, C0=src0.C, ld0=src0.ld, W=fake, Cout=Cout, tile_n=128, act=act,
                          out=fake, out_dtype=src0.dt, out_ld=cout_out, ln_eps=1e-5)
        return bool(L.lib().dc_igemm_ln_ok(p))

    def side_ok(self, src0, s2, Cout, residual=None):
        """Can a 3x3 stride-1 conv of src0 take the 1x1 side source s2 (dc_igemm_side_ok)?"""
        fake = 1 << 20
        dom = self._dom(src0, s2, residual)
        p = L.IgemmParams(dtype=src0.dt, taps=9, stride=1, upsample=0, n_img=self.n[dom], Hin=src0.H, Win=src0.W,
                          Hout=src0.H, Wout=src0.W, src0=fake, C0=src0.C, ld0=src0.ld, W=fake, Cout=Cout, tile_n=128,
                          residual=fake if residual is not None else None, res_dtype=src0.dt, res_ld=Cout,
                          out=fake, out_dtype=src0.dt, out_ld=Cout, src2=fake, W2=fake, C2=s2.C, ld2=s2.ld)
        return bool(L.lib().dc_igemm_side_ok(p))

    def gn_fusable(self, src0, src1, Cout, tile_n=128, out_dt=None):
        """Can a 3x3 stride-1 conv of these sources take the GroupNorm prologue (dc_igemm_gn_fusable)?"""
        if L.lib().dc_igemm_gn_fusable is None:
            return False
        fake = 1 << 20
        p = L.IgemmParams(dtype=src0.dt, taps=9, stride=1, upsample=0, n_img=self.n[self._dom(src0, src1)], Hin=src0.H, Win=src0.W,
                          Hout=src0.H, Wout=src0.W, src0=fake, C0=src0.C, ld0=src0.ld, src1=fake if src1 is not None else None,
                          C1=src1.C if src1 is not None else 0, ld1=src1.ld if src1 is not None else 0, W=fake, Cout=Cout,
                          tile_n=tile_n, out=fake, out_dtype=src0.dt if out_dt is None else out_dt, out_ld=Cout)
        return bool(L.lib().dc_igemm_gn_fusable(p))

    def gn_ws_ok(self, x, Cout):
        """Would a 3x3 stride-1 conv of x with a fused GroupNorm prologue run on the wave-specialised halo kernel (conv3_ws.hip: the
        transform is done by loader waves, not in the MFMA waves' stream)?  Asked through dc_igemm_variant."""
        fake = 1 << 20
        p = L.IgemmParams(dtype=x.dt, taps=9, stride=1, upsample=0, n_img=self.n[x.dom], Hin=x.H, Win=x.W, Hout=x.H, Wout=x.W,
                          src0=fake, C0=x.C, ld0=x.ld, W=fake, Cout=Cout, tile_n=128, out=fake, out_dtype=x.dt, out_ld=Cout,
                          gn_scale=fake, gn_shift=fake, gn_silu=1)
        return L.lib().dc_igemm_variant(p).decode().startswith("conv3_ws")

    def groupnorm_stats(self, name, x0, gamma, beta, groups, eps, x1=None):
        """Statistics-only GroupNorm: returns the per-(sample, channel) scale / shift tensors for a fused conv prologue."""
        dom = self._dom(x0, x1)
        Cc = x0.C + (x1.C if x1 is not None else 0)
        assert x0.ld == x0.C and (x1 is None or x1.ld == x1.C)
        n, HW = self.n[dom], x0.H * x0.W
        sc = self.tensor(name + ".scale", dom, 1, 1, Cc, L.DC_F32)
        sh = self.tensor(name + ".shift", dom, 1, 1, Cc, L.DC_F32)
        splits = L.lib().dc_groupnorm_splits(n, HW, Cc)
        ws = TRef(name + ".ws", dom, 1, 1, 1, L.DC_F32, nbytes=round_up(L.lib().dc_groupnorm_ws_floats(n, groups, splits) * 4, 256))
        f = dict(x=x0, map0=self._map(x0, dom), x1=x1, map1=self._map(x1, dom), y=None, dtype=x0.dt, out_dtype=x0.dt,
                 n=n, HW=HW, C=x0.C, C1=x1.C if x1 is not None else 0, groups=groups, silu=0, splits=splits, eps=eps,
                 gamma=gamma, beta=beta, ws=ws, out_scale=sc, out_shift=sh)
        qs = None
        if self.qstats_ok(x0, x1, groups, dom):    # the producer's quad records: the affine is formed without reading the tensor
            qs = x0.qs[0]
            f.update(qstats=qs, qparts=x0.qs[1])
        self._emit(L.OP_GROUPNORM, L.GroupnormParams, f, [x0, x1, qs], [sc, sh, ws],
                   dict(name=name, family="groupnorm_stats", flops=0.0, bytes=(0.0 if qs is not None else 1.0 * n * HW * Cc * DT_SIZE[x0.dt])))
        return sc, sh

    @staticmethod
    def qstats_ok(x0, x1, groups, dom):
        """Does x0 carry quad statistics its GroupNorm can use (single source, same domain, groups made of whole quads)?"""
        return x1 is None and x0.qs is not None and ((x0.C // groups) % 4 == 0) and x0.dom == dom

    def groupnorm(self, name, x0, gamma, beta, groups, eps, silu, x1=None):
        dom = self._dom(x0, x1)
        Cc = x0.C + (x1.C if x1 is not None else 0)
        assert x0.ld == x0.C and (x1 is None or x1.ld == x1.C)
        out = self.tensor(name, dom, x0.H, x0.W, Cc, x0.dt)
        n, HW = self.n[dom], x0.H * x0.W
        splits = L.lib().dc_groupnorm_splits(n, HW, Cc)
        ws_floats = L.lib().dc_groupnorm_ws_floats(n, groups, splits)
        ws = TRef(name + ".ws", dom, 1, 1, 1, L.DC_F32, nbytes=round_up(ws_floats * 4, 256))
        f = dict(x=x0, map0=self._map(x0, dom), x1=x1, map1=self._map(x1, dom), y=out, dtype=x0.dt, out_dtype=x0.dt,
                 n=n, HW=HW, C=x0.C, C1=x1.C if x1 is not None else 0, groups=groups, silu=int(silu),
                 splits=splits, eps=eps, gamma=gamma, beta=beta, ws=ws)
        qs = None
        if self.qstats_ok(x0, x1, groups, dom):
            qs = x0.qs[0]
            f.update(qstats=qs, qparts=x0.qs[1])
        self._emit(L.OP_GROUPNORM, L.GroupnormParams, f, [x0, x1, qs], [out, ws],
                   dict(name=name, family="groupnorm", flops=0.0, bytes=2.0 * n * HW * Cc * DT_SIZE[x0.dt]))
        return out

    def layernorm(self, name, x, gamma, beta, eps, scale=None, shift=None):
        assert x.ld == x.C
        assert scale is None or x.dom == "unit" or scale.dom == x.dom, "LayerNorm input must already be per-unit"
        out = self.tensor(name, x.dom, x.H, x.W, x.C, x.dt)
        f = dict(x=x, y=out, dtype=x.dt, out_dtype=x.dt, rows=self.n[x.dom] * x.H * x.W, C=x.C,
                 rows_per_sample=x.H * x.W, mod_ld=scale.ld if scale is not None else 0, eps=eps,
                 gamma=gamma, beta=beta, scale=scale, shift=shift, mod_map=self._map(scale, x.dom))
        self._emit(L.OP_LAYERNORM, L.LayernormParams, f, [x, scale, shift], [out],
                   dict(name=name, family="layernorm", flops=0.0, bytes=2.0 * f["rows"] * x.C * DT_SIZE[x.dt]))
        return out

    def attention(self, name, q, k, v, heads):
        out = self.tensor(name, q.dom, q.H, q.W, q.C, q.dt)
        d = q.C // heads
        f = dict(q=q, k=k, v=v, out=out, dtype=q.dt, n=self.n[q.dom], L=q.H * q.W, heads=heads, d=d,
                 ld_qkv=q.ld, ld_out=out.ld, scale=float(d) ** -0.5)
        Lq = q.H * q.W
        self._emit(L.OP_ATTENTION, L.AttentionParams, f, [q, k, v], [out],
                   dict(name=name, family="attention", flops=4.0 * self.n[q.dom] * heads * Lq * Lq * d,
                        bytes=4.0 * self.n[q.dom] * Lq * q.C * DT_SIZE[q.dt]))
        return out

    def tblock_front_ok(self, x, heads):
        """Can ONE launch take proj_in -> LayerNorm -> q/k/v -> attention -> to_out of this block input (dc_tblock_front_ok)?"""
        fn = getattr(L.lib(), "dc_tblock_front_ok", None)
        if fn is None:
            return False
        p = L.TblockFrontParams(dtype=x.dt, n=self.n[x.dom], L=x.H * x.W, C=x.C, heads=heads, ldx=x.ld, ld_out=x.C)
        return int(fn(p)) == 1

    def tblock_front(self, name, x, Wp, bp, ln_g, ln_b, Wqkv, Wo, bo, rowvec, heads, eps):
        """out = to_out(attention(LayerNorm(h))) + bo + rowvec + h with h = proj_in(x) + bp, one workgroup per sample (dc_tblock_front)."""
        assert rowvec is None or self._dom(x, rowvec) == x.dom, "the class vector must not widen the block's domain"
        dom, Lq, Cc = x.dom, x.H * x.W, x.C
        d = Cc // heads
        out = self.tensor(name, dom, x.H, x.W, Cc, x.dt)
        f = dict(x=x, Wp=Wp, bp=bp, ln_g=ln_g, ln_b=ln_b, Wqkv=Wqkv, Wo=Wo, bo=bo, rowvec=rowvec, rowvec_map=self._map(rowvec, dom),
                 rowvec_ld=rowvec.ld if rowvec is not None else 0, out=out, dtype=x.dt, n=self.n[dom], L=Lq, C=Cc, heads=heads,
                 ldx=x.ld, ld_out=out.ld, ln_eps=float(eps), scale=float(d) ** -0.5)
        M, es = self.n[dom] * Lq, DT_SIZE[x.dt]
        self._emit(L.OP_TBLOCK_FRONT, L.TblockFrontParams, f, [x, rowvec], [out],
                   dict(name=name, family="tblock_front", flops=2.0 * M * Cc * 5 * Cc + 4.0 * self.n[dom] * heads * Lq * Lq * d,
                        bytes=float(2 * M * Cc * es + 5 * Cc * Cc * es), M=M, N=Cc, K=Cc))
        return out

    def sinusoid(self, name, lam, dim, flip, shift):
        out = self.tensor(name, lam.dom, 1, 1, dim, L.DC_F32)
        f = dict(lam=lam, out=out, n=self.n[lam.dom], dim=dim, flip_sin_to_cos=int(flip), freq_shift=float(shift))
        self._emit(L.OP_SINUSOID, L.SinusoidParams, f, [lam], [out], dict(name=name, family="sinusoid", flops=0.0, bytes=0.0))
        return out

    def qsample(self, name, x_ptr, eps, alpha, sigma, img_of_bj, Cin, H, W, ld, dt, im2col, patch=0):
        g = patch if int(im2col) == 2 else 1
        out = self.tensor(name, "bj", H // g, W // g, ld, dt)
        f = dict(x=x_ptr, eps=eps, alpha=alpha, sigma=sigma, img_of_bj=img_of_bj, out=out, out_dtype=dt,
                 n_bj=self.n["bj"], C=Cin, H=H, W=W, ld=ld, im2col=int(im2col), patch=int(patch))
        self._emit(L.OP_QSAMPLE, L.QsampleParams, f, [eps, alpha, sigma], [out],
                   dict(name=name, family="qsample", flops=0.0,
                        bytes=float(self.n["bj"] * (2 * Cin * H * W * 4 + (H // g) * (W // g) * ld * DT_SIZE[dt]))))
        return out

    def eps_mse(self, pred, eps, x_ptr, alpha, sigma, bj_of_unit, img_of_bj, out_index, out_ptr, Cin, v_param, patch=0):
        g = patch if patch > 1 else 1
        f = dict(pred=pred, eps=eps, x=x_ptr, alpha=alpha, sigma=sigma, bj_of_unit=bj_of_unit, img_of_bj=img_of_bj,
                 out_index=out_index, out=out_ptr, n_units=self.n["unit"], C=Cin, H=pred.H * g, W=pred.W * g, ld=pred.ld,
                 v_param=int(v_param), patch=int(patch))
        self._emit(L.OP_EPS_MSE, L.EpsMseParams, f, [pred, eps, alpha, sigma], [],
                   dict(name="eps_mse", family="eps_mse", flops=0.0, bytes=8.0 * self.n["unit"] * Cin * pred.H * g * pred.W * g))

    # ---- finalize: liveness-based arena + ctypes records ----------------------------
    def finalize(self, keep_alive=()):
        """keep_alive: tensors that must survive to the end of the plan (outputs)."""
        nops = len(self.ops)
        for t in keep_alive:
            t.base.last = nops
        # a producer-normalised conv whose raw output nobody reads stores only the normalised tensor
        for i, (kind, _, f) in enumerate(self.ops):
            if kind == L.OP_IGEMM and f.get("pn_out") is not None and isinstance(f.get("out"), TRef):
                b = f["out"].base
                if b.ext is None and b.first == i and b.last == i:
                    self.meta[i]["bytes"] -= float(self.n[b.dom] * b.H * b.W * b.C * DT_SIZE[b.dt])
                    f["out"] = None
        bases = {}
        for _, _, f in self.ops:
            for v in f.values():
                if isinstance(v, TRef) and v.base.ext is None:
                    bases[id(v.base)] = v.base
        order = sorted(bases.values(), key=lambda b: b.first)
        by_last = {}
        for b in order:
            by_last.setdefault(b.last, []).append(b)
        free, top = [], 0   # free: list of (off, size)
        pending = iter(order)
        nxt = next(pending, None)
        for idx in range(nops + 1):
            while nxt is not None and nxt.first == idx:
                need = nxt.nbytes
                best = None
                for i, (o, s) in enumerate(free):
                    if s >= need and (best is None or s < free[best][1]):
                        best = i
                if best is not None:
                    o, s = free.pop(best)
                    nxt.off = o
                    if s > need:
                        free.append((o + need, s - need))
                else:
                    nxt.off = top
                    top += need
                nxt = next(pending, None)
            for b in by_last.get(idx, []):
                free.append((b.off, b.nbytes))
                free.sort()
                merged = []
                for o, s in free:
                    if merged and merged[-1][0] + merged[-1][1] == o:
                        merged[-1] = (merged[-1][0], merged[-1][1] + s)
                    else:
                        merged.append((o, s))
                free = merged
        self.arena_bytes = top
        self.arena = torch.empty(max(top, 256), dtype=torch.uint8, device=self.dev)
        base_ptr = self.arena.data_ptr()
        assert base_ptr % 256 == 0

        def ptr(v):
            if v is None:
                return None
            if isinstance(v, TRef):
                b = v.base
                p0 = b.ext.data_ptr() if b.ext is not None else base_ptr + b.off
                return p0 + v.eoff * DT_SIZE[v.dt]
            return v

        self.structs = []
        arr = (L.Op * nops)()
        for i, (kind, cls, f) in enumerate(self.ops):
            s = cls()
            for name, ctype in cls._fields_:
                if name not in f:
                    continue
                v = f[name]
                if ctype is L.vp:
                    setattr(s, name, ptr(v))
                else:
                    setattr(s, name, v)
            self.structs.append(s)
            if kind == L.OP_IGEMM:      # the kernel libdcamd picks for this shape (bench.py groups timings by it)
                self.meta[i]["family"] = L.lib().dc_igemm_variant(s).decode()
            arr[i].kind = kind
            arr[i].params = C.cast(C.pointer(s), C.c_void_p)
        self.op_array = arr
        self.nops = nops
        self.ptr_of = ptr
        return self

    def run(self):
        L.check(L.lib().dc_run_plan(self.op_array, self.nops, L.stream_ptr()), "dc_run_plan")

    def run_timed(self):
        """Same launches with a HIP event pair around every op (on the launch stream); returns ms per op."""
        ms = (C.c_float * self.nops)()
        L.check(L.lib().dc_run_plan_timed(self.op_array, self.nops, L.stream_ptr(), ms), "dc_run_plan_timed")
        return list(ms)

    def tensor_view(self, t):
        """torch view of an arena/external tensor (tests and the plain forward read results through it)."""
        n = self.n[t.dom]
        assert t.ld == t.C and t.eoff == 0
        if t.base.ext is not None:
            return t.base.ext
        nel = n * t.H * t.W * t.C
        off = t.base.off
        return self.arena[off:off + nel * DT_SIZE[t.dt]].view(TORCH_DT[t.dt]).view(n, t.H, t.W, t.C)


# ======================================================================================
#                                   weight packing
# ======================================================================================
# The packed layouts belong to the C-ABI (include/dcamd.h, "weight packing"): these wrappers only move the fp32 parameter tensor to
# the device and call libdcamd's packers — nothing here knows how a packed row is laid out.
def _dev32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _packed(rows, cols, dt, device, lead=()):
    return torch.empty(tuple(lead) + (rows, cols), dtype=TORCH_DT[dt], device=device)


def pack_matrix(w2d, dt, device, tile_n=128, kpad=None, col_scale=None):
    """[Cout, K] fp32 (Linear / Conv2d 1x1 weight) -> packed [Cout_pad, kpad or K] in dt (dc_pack_weights_matrix).
    col_scale [K]: a LayerNorm gamma folded into the columns."""
    w = _dev32(w2d.reshape(w2d.shape[0], -1), device)
    cout, K = w.shape
    kp = K if kpad is None else kpad
    out = _packed(L.lib().dc_igemm_cout_pad(cout, tile_n), kp, dt, device)
    cs = None if col_scale is None else _dev32(col_scale, device)
    L.check(L.lib().dc_pack_weights_matrix(w.data_ptr(), cout, K, kp, None, None if cs is None else cs.data_ptr(), out.data_ptr(), dt, tile_n,
                                           L.stream_ptr()), "dc_pack_weights_matrix")
    return out


def pack_conv3x3(w, dt, device, tile_n=128, kpad=None, c_lo=0, c_hi=None):
    """Conv2d weight [Cout, Cin, 3, 3], input channels [c_lo, c_hi) -> packed [Cout_pad, 9*C (padded to kpad)] (dc_pack_weights_conv3x3)."""
    wd = _dev32(w, device)
    cout, cin = wd.shape[0], wd.shape[1]
    c_hi = cin if c_hi is None else c_hi
    kp = 9 * (c_hi - c_lo) if kpad is None else kpad
    out = _packed(L.lib().dc_igemm_cout_pad(cout, tile_n), kp, dt, device)
    L.check(L.lib().dc_pack_weights_conv3x3(wd.data_ptr(), cout, cin, c_lo, c_hi, kp, out.data_ptr(), dt, tile_n, L.stream_ptr()),
            "dc_pack_weights_conv3x3")
    return out


def pack_up4(w, dt, device, tile_n=128):
    """Conv2d weight [Cout, Cin, 3, 3] -> [4, Cout_pad, 4*Cin]: the four-phase form of "nearest-2x upsample, then 3x3 conv"
    (dc_pack_weights_up4 / dc_igemm_params.up4)."""
    wd = _dev32(w, device)
    cout, cin = wd.shape[0], wd.shape[1]
    out = _packed(L.lib().dc_igemm_cout_pad(cout, tile_n), 4 * cin, dt, device, lead=(4,))
    L.check(L.lib().dc_pack_weights_up4(wd.data_ptr(), cout, cin, out.data_ptr(), dt, tile_n, L.stream_ptr()), "dc_pack_weights_up4")
    return out


def pack_geglu(w, b, dt, device, ln_gamma=None, ln_beta=None):
    """GEGLU projection [2*n_half, K] + bias -> (packed weight, packed fp32 bias) in the value / gate interleaved row order
    dc_igemm's GEGLU epilogue expects, optionally with a LayerNorm (gamma, beta) folded in (dc_pack_weights_geglu)."""
    wd, bd = _dev32(w, device), _dev32(b, device)
    cout, K = wd.shape
    out = _packed(L.lib().dc_igemm_cout_pad(cout, 128), K, dt, device)
    ob = torch.empty(cout, dtype=torch.float32, device=device)
    ws = torch.empty(cout, dtype=torch.int32, device=device)
    g = None if ln_gamma is None else _dev32(ln_gamma, device)
    be = None if ln_beta is None else _dev32(ln_beta, device)
    L.check(L.lib().dc_pack_weights_geglu(wd.data_ptr(), bd.data_ptr(), cout // 2, K, None if g is None else g.data_ptr(),
                                          None if be is None else be.data_ptr(), out.data_ptr(), ob.data_ptr(), ws.data_ptr(), dt,
                                          L.stream_ptr()), "dc_pack_weights_geglu")
    return out, ob


def fold_layernorm_bias(w, bias, ln_beta, device):
    """W beta (+ bias): the bias of a GEMM that absorbed a LayerNorm's beta (dc_fold_layernorm_bias)."""
    wd, be = _dev32(w, device), _dev32(ln_beta, device)
    bd = None if bias is None else _dev32(bias, device)
    out = torch.empty(wd.shape[0], dtype=torch.float32, device=device)
    L.check(L.lib().dc_fold_layernorm_bias(wd.data_ptr(), None if bd is None else bd.data_ptr(), be.data_ptr(), wd.shape[0], wd.shape[1],
                                           out.data_ptr(), L.stream_ptr()), "dc_fold_layernorm_bias")
    return out


def geglu_perm(n_half):
    """Row order of the packed GEGLU projection (what dc_pack_weights_geglu applies): 16-row blocks alternate value / gate halves.
    Kept as the executable statement of that layout for the tests."""
    assert n_half % 16 == 0
    idx = []
    for b in range(n_half // 16):
        idx += list(range(16 * b, 16 * b + 16))
        idx += list(range(n_half + 16 * b, n_half + 16 * b + 16))
    return torch.tensor(idx, dtype=torch.long)


def f32c(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def pad_vec(v, rows):
    if v.shape[0] == rows:
        return v
    out = torch.zeros(rows, dtype=v.dtype, device=v.device)
    out[: v.shape[0]] = v
    return out


class UNetWeights:
    """Device-resident packed weights of a UNetCondition2D for one compute dtype."""

    def __init__(self, model, dt, device):
        self.dt, self.dev = dt, device
        cfg = model.config
        self.kin = round_up(9 * cfg.in_channels, bke(dt))
        P = {}
        sd = {k: v for k, v in model.state_dict().items()}

        def conv3(key, tile_n=128, kpad=None):
            P[key + ".w"] = pack_conv3x3(sd[key + ".weight"], dt, device, tile_n, kpad)
            P[key + ".b"] = f32c(sd[key + ".bias"], device)

        def conv1(key):
            w = sd[key + ".weight"]
            P[key + ".w"] = pack_matrix(w, dt, device)
            P[key + ".b"] = f32c(sd[key + ".bias"], device)

        def lin(key, dtype=None, bias=True):
            P[key + ".w"] = pack_matrix(sd[key + ".weight"], dt if dtype is None else dtype, device)
            if bias and key + ".bias" in sd:
                P[key + ".b"] = f32c(sd[key + ".bias"], device)

        def norm(key):
            P[key + ".g"] = f32c(sd[key + ".weight"], device)
            P[key + ".b"] = f32c(sd[key + ".bias"], device)

        conv3("conv_in", kpad=self.kin)
        lin("time_embedding.linear_1", L.DC_F32)
        lin("time_embedding.linear_2", L.DC_F32)
        lin("encoder_hid_proj", L.DC_F32)
        norm("conv_norm_out")
        conv3("conv_out", tile_n=32 if cfg.out_channels <= 32 else 128)
        self.resnets, self.attns = [], []
        for k in sd:
            if k.endswith("samplers.0.conv.weight"):
                conv3(k[: -len(".weight")])
        for key in sorted({k.rsplit(".", 2)[0] for k in sd if k.endswith(".time_emb_proj.weight")}):
            self.resnets.append(key)
            norm(key + ".norm1"); conv3(key + ".conv1"); norm(key + ".norm2"); conv3(key + ".conv2")
            if key + ".conv_shortcut.weight" in sd:
                conv1(key + ".conv_shortcut")
        # all time_emb_proj stacked into one [sum Cout, 4*C0] fp32 GEMM
        tw = torch.cat([sd[k + ".time_emb_proj.weight"] for k in self.resnets], 0)
        tb = torch.cat([sd[k + ".time_emb_proj.bias"] for k in self.resnets], 0)
        self.tproj_off, o = {}, 0
        for k in self.resnets:
            self.tproj_off[k] = o
            o += sd[k + ".time_emb_proj.weight"].shape[0]
        self.tproj_total = o
        P["tproj.w"] = pack_matrix(tw, L.DC_F32, device)
        P["tproj.b"] = f32c(tb, device)
        for key in sorted({k[: -len(".proj_in.weight")] for k in sd if k.endswith(".proj_in.weight")}):
            self.attns.append(key)
            norm(key + ".norm"); conv1(key + ".proj_in"); conv1(key + ".proj_out")
            tb_ = key + ".transformer_blocks.0"
            norm(tb_ + ".norm1"); norm(tb_ + ".norm3")
            qkv = torch.cat([sd[tb_ + f".attn1.to_{n}.weight"] for n in "qkv"], 0)
            P[tb_ + ".qkv.w"] = pack_matrix(qkv, dt, device)
            lin(tb_ + ".attn1.to_out.0")
            P[tb_ + ".ff.net.0.proj.w"], P[tb_ + ".ff.net.0.proj.b"] = pack_geglu(
                sd[tb_ + ".ff.net.0.proj.weight"], sd[tb_ + ".ff.net.0.proj.bias"], dt, device)
            lin(tb_ + ".ff.net.2")
            # class-token side path (fp32): to_v then to_out of attn2 (to_q/to_k never matter for 1 key)
            lin(tb_ + ".attn2.to_out.0", L.DC_F32)
        vw = torch.cat([sd[k + ".transformer_blocks.0.attn2.to_v.weight"] for k in self.attns], 0)
        self.cv_off, o = {}, 0
        for k in self.attns:
            self.cv_off[k] = o
            o += sd[k + ".transformer_blocks.0.attn2.to_v.weight"].shape[0]
        self.cv_total = o
        P["attn2v.w"] = pack_matrix(vw, L.DC_F32, device)
        self.P = P
        self._sd = sd              # references only (no copy): split packing of skip-connection convs is lazy

    def split_resnet(self, key, C0):
        """Packed halves of a skip-connection ResNet's input convs: `.conv1.wa/.wb` ([Cout, 9*C0] / [Cout, 9*C1], same
        k = tap*C + c order) and `.conv_shortcut.wa/.wb`, for channels [0, C0) and [C0, C0+C1) of the concatenated
        input.  conv(cat(a, b)) = conv_a(a) + conv_b(b): when b is class-independent its half runs once per (image, trial)
        pair instead of once per class (UNetPlan.resnet)."""
        P, sd = self.P, self._sd
        if key + ".conv1.wa" not in P:
            w = sd[key + ".conv1.weight"]
            P[key + ".conv1.wa"] = pack_conv3x3(w, self.dt, self.dev, c_lo=0, c_hi=C0)
            P[key + ".conv1.wb"] = pack_conv3x3(w, self.dt, self.dev, c_lo=C0)
        return self.split_shortcut(key, C0)

    def split_shortcut(self, key, C0):
        """`.conv_shortcut.wa/.wb`: the 1x1 shortcut's columns for channels [0, C0) / [C0, C0+C1) of the concatenated input."""
        P, sd = self.P, self._sd
        if key + ".conv_shortcut.weight" in sd and key + ".conv_shortcut.wa" not in P:
            ws = sd[key + ".conv_shortcut.weight"]
            ws = ws.reshape(ws.shape[0], -1)
            P[key + ".conv_shortcut.wa"] = pack_matrix(ws[:, :C0], self.dt, self.dev)
            P[key + ".conv_shortcut.wb"] = pack_matrix(ws[:, C0:], self.dt, self.dev)
        return P

    def fold_layernorms(self, tb_):
        """LayerNorm folded into the GEMM that consumes it: y = W (x_hat * g + b) + c = (W diag(g)) x_hat + (W b + c).  Packs
        `<tb>.qkv.wf/.bf` (norm1 -> q/k/v) and `<tb>.ff.net.0.proj.wf/.bf` (norm3 -> GEGLU projection); the kernel then only
        standardises the rows (dc_igemm ln_eps)."""
        P, sd = self.P, self._sd
        if tb_ + ".qkv.wf" in P:
            return P
        qkv = torch.cat([sd[tb_ + f".attn1.to_{n}.weight"] for n in "qkv"], 0).float()
        P[tb_ + ".qkv.wf"] = pack_matrix(qkv, self.dt, self.dev, col_scale=sd[tb_ + ".norm1.weight"])
        P[tb_ + ".qkv.bf"] = fold_layernorm_bias(qkv, None, sd[tb_ + ".norm1.bias"], self.dev)
        P[tb_ + ".ff.net.0.proj.wf"], P[tb_ + ".ff.net.0.proj.bf"] = pack_geglu(
            sd[tb_ + ".ff.net.0.proj.weight"], sd[tb_ + ".ff.net.0.proj.bias"], self.dt, self.dev,
            ln_gamma=sd[tb_ + ".norm3.weight"], ln_beta=sd[tb_ + ".norm3.bias"])
        return P

    def fold_proj_out(self, key):
        """proj_out folded into the feed-forward's second linear: with h3 = W2 f + b2 + h2 (ff.net.2 + residual) the block's output
        x + Wpo h3 + bpo equals x + [Wpo W2 | Wpo] [f ; h2] + (Wpo b2 + bpo) — ONE GEMM over K = 4C + C with x as its residual instead of
        two (h3 never exists).  Exact algebra; the product matrix is formed in fp64 and rounded once to the compute dtype.
        Packs `<key>.ffpo.w` [C, 5C] and `<key>.ffpo.b`."""
        P, sd = self.P, self._sd
        if key + ".ffpo.w" not in P:
            tb_ = key + ".transformer_blocks.0"
            w2, b2 = sd[tb_ + ".ff.net.2.weight"].double(), sd[tb_ + ".ff.net.2.bias"].double()
            wpo = sd[key + ".proj_out.weight"].double()
            wpo = wpo.reshape(wpo.shape[0], -1)
            P[key + ".ffpo.w"] = pack_matrix(torch.cat([wpo @ w2, wpo], 1).float(), self.dt, self.dev)
            P[key + ".ffpo.b"] = f32c((sd[key + ".proj_out.bias"].double() + wpo @ b2).float(), self.dev)
        return P

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.P.values())


# ======================================================================================
#                                   UNet plan
# ======================================================================================
class UNetPlan:
    """Static launch plan of one UNetCondition2D scoring step.

    inputs (plan-owned device buffers, refreshed per micro-batch by the caller):
      lam [n_bj] f32;  ctx [n_ctx, hid] f32;  a0 = conv_in GEMM operand [n_bj, H, W, kin]
      (written by the q_sample op when `score=True`, else by the caller through `a0_view`);
      maps bj_of_unit / ctx_of_unit [U] int32.
    output: pred [U, H, W, out_channels] f32 (NHWC); with score=True also err -> errors buffer.
    """

    def __init__(self, model, weights, n_bj, n_cls, n_ctx, *, share_trunk=True, score=None, device=None):
        import os
        fuse_gn_out = os.environ.get("DCAMD_NO_GN_OUT_FUSION") is None
        split_skips = os.environ.get("DCAMD_NO_SKIP_SPLIT") is None
        fold_short = os.environ.get("DCAMD_NO_SHORT_FOLD") is None
        # 3x3 convs also emit the (mean, M2) quad statistics of their output, so the GroupNorm that follows streams the
        # tensor once (read + write) instead of twice + write: GroupNorm 8.3 -> ~6 ms per cfg2 step
        use_qs = os.environ.get("DCAMD_NO_QSTATS") is None
        use_up4 = os.environ.get("DCAMD_NO_UP4") is None
        fold_ln = os.environ.get("DCAMD_NO_LN_FOLD") is None
        # the attention half of a transformer block (proj_in ... to_out) as one launch where libdcamd serves the shape (tblock.hip)
        fuse_tb = os.environ.get("DCAMD_NO_TBLOCK") is None
        fold_po = os.environ.get("DCAMD_NO_PO_FOLD") is None
        # GroupNorm(+SiLU) applied by the consuming 3x3 conv's loader waves (conv3_ws.hip) from the producer's quad records: no
        # GroupNorm launch, the normalised tensor never exists (DCAMD_NO_GN_WS: the GroupNorm pass + the plain conv, for A/B runs)
        fuse_ws = os.environ.get("DCAMD_NO_GN_WS") is None and use_qs
        cfg = model.config
        dev = device or weights.dev
        dt = weights.dt
        self.dt, self.n_bj, self.n_cls, self.n_ctx = dt, n_bj, n_cls, n_ctx
        U = n_bj * n_cls
        pb = PlanBuilder(dev, n_bj, n_cls, n_ctx)
        self.pb = pb
        P = weights.P
        H = W = cfg.sample_size
        C0 = cfg.block_out_channels[0]
        G, eps = cfg.norm_num_groups, cfg.norm_eps
        heads = cfg.attention_head_dim
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        # ---- inputs ----
        self.lam = score["lam"] if score is not None and "lam" in score else torch.zeros(n_bj, **f32)
        self.ctx = torch.zeros(n_ctx, cfg.encoder_hid_dim, **f32)
        self.bj_of_unit = (torch.arange(U, **i32) // n_cls).contiguous()
        if score is not None and "ctx_of_unit" in score:
            self.ctx_of_unit = score["ctx_of_unit"]
        else:
            self.ctx_of_unit = (torch.arange(U, **i32) % n_ctx).contiguous()
        pb.set_map("bj", "unit", self.bj_of_unit)
        pb.set_map("ctx", "unit", self.ctx_of_unit)
        lam = pb.external("lam", self.lam, "bj", 1, 1, 1, L.DC_F32)
        ctx = pb.external("ctx", self.ctx, "ctx", 1, 1, cfg.encoder_hid_dim, L.DC_F32)
        kin = weights.kin
        if score is not None:
            # score = dict(x=[B,C,H,W] f32 buffer, eps=[n_bj,C,H,W], alpha, sigma, img_of_bj, out_index, errors, v_param)
            self.score = score
            eps_t = pb.external("eps", score["eps"], "bj", 1, 1, 1, L.DC_F32)
            al = pb.external("alpha", score["alpha"], "bj", 1, 1, 1, L.DC_F32)
            sg = pb.external("sigma", score["sigma"], "bj", 1, 1, 1, L.DC_F32)
            a0 = pb.qsample("a0", pb.const(score["x"]), eps_t, al, sg, pb.const(score["img_of_bj"]),
                            cfg.in_channels, H, W, kin, dt, im2col=True)
        else:
            self.a0_buf = torch.zeros(n_bj, H, W, kin, dtype=TORCH_DT[dt], device=dev)
            a0 = pb.external("a0", self.a0_buf, "bj", H, W, kin, dt)
        # ---- fp32 side path: time embedding, stacked time_emb_proj, class-token vectors ----
        te = pb.sinusoid("temb.sin", lam, C0, cfg.flip_sin_to_cos, cfg.freq_shift)
        te = pb.igemm("temb.l1", te, pb.const(P["time_embedding.linear_1.w"]), 4 * C0,
                      bias=pb.const(P["time_embedding.linear_1.b"]), act=L.ACT_SILU)
        te = pb.igemm("temb.l2", te, pb.const(P["time_embedding.linear_2.w"]), 4 * C0,
                      bias=pb.const(P["time_embedding.linear_2.b"]), act=L.ACT_SILU)   # = SiLU(temb)
        tproj = pb.igemm("tproj", te, pb.const(P["tproj.w"]), weights.tproj_total, bias=pb.const(P["tproj.b"]))
        # class-token vectors: their own small plan (run_ctx), executed once per classify call / forward and
        # NOT once per micro-batch: they depend only on the weights and on `ctx`
        pc = self.ctx_pb = PlanBuilder(dev, 1, 1, n_ctx)
        cctx = pc.external("ctx", self.ctx, "ctx", 1, 1, cfg.encoder_hid_dim, L.DC_F32)
        hp = pc.igemm("ctx.hid_proj", cctx, pc.const(P["encoder_hid_proj.w"]), cfg.cross_attention_dim,
                      bias=pc.const(P["encoder_hid_proj.b"]))
        vall = pc.igemm("ctx.to_v", hp, pc.const(P["attn2v.w"]), weights.cv_total)
        cv_t = {}
        for k in weights.attns:
            tbk = k + ".transformer_blocks.0"
            Ck = P[k + ".norm.g"].shape[0]
            cv_t[k] = pc.igemm(k + ".cvec", vall.view(weights.cv_off[k], Ck), pc.const(P[tbk + ".attn2.to_out.0.w"]), Ck,
                               bias=pc.const(P[tbk + ".attn2.to_out.0.b"]))
        pc.finalize(keep_alive=list(cv_t.values()))
        cvec = {k: pb.external(k + ".cvec", pc.tensor_view(t).view(n_ctx, -1), "ctx", 1, 1, t.C, L.DC_F32) for k, t in cv_t.items()}

        def gn_conv3(gname, cname, x, gamma, beta, groups, Wp, Cout, **kw):
            """GroupNorm + SiLU of the single-source tensor x, then a 3x3 conv of it (bias / row vector / residual / side source
            in kw).  One launch where the wave-specialised conv can normalise its own input (affine from the quad records x's
            producer wrote: no pass over x at all); else the GroupNorm pass and the plain conv."""
            dom = pb._dom(x, kw.get("rowvec"), kw.get("residual"), kw["side"][0] if kw.get("side") else None)
            # first choice: x's producer stores the normalised tensor itself (csrc/epi_pn.h) and this conv is the plain halo conv
            y = pb.pn_claim(x, gamma, beta, groups, eps, True)
            if y is not None:
                return pb.igemm(cname, y, Wp, Cout, taps=9, **kw)
            # second: where this conv could in turn normalise its own output for the NEXT GroupNorm, keep it the plain halo conv behind a
            # GroupNorm pass (one pass at the head of a chain of producer-normalised convs) rather than the wave-specialised conv, whose
            # epilogue cannot (conv3_ws.hip: one workgroup per CU, 0.75 against 1.0 PF)
            chain = kw.get("qstats") and pb.pn_capable(x, Cout, dom)
            if not chain and fuse_ws and dom == x.dom and pb.qstats_ok(x, None, groups, dom) and pb.gn_ws_ok(x, Cout):
                aff = pb.groupnorm_stats(gname, x, gamma, beta, groups, eps)
                return pb.igemm(cname, x, Wp, Cout, taps=9, gn=(aff[0], aff[1], True), **kw)
            y = pb.groupnorm(gname, x, gamma, beta, groups, eps, True)
            return pb.igemm(cname, y, Wp, Cout, taps=9, **kw)

        def resnet(key, x0, x1=None):
            Cout = P[key + ".conv1.b"].shape[0]
            tvec = tproj.view(weights.tproj_off[key], Cout)
            # skip connection from the class-shared trunk into a per-class layer: conv(cat(h, skip)) = conv_a(h) + conv_b(skip)
            # and GroupNorm never mixes the two halves when its groups do not straddle the seam — the skip half is then
            # computed once per (image, trial) pair, not once per class.  Exact algebra; only the summation order differs.
            split = (split_skips and x1 is not None and x1.dom == "bj" and x0.dom == "unit"
                     and x0.C % ((x0.C + x1.C) // G) == 0 and key + ".conv_shortcut.w" in P)
            if split:
                # class-independent skip half, once per (image, trial) pair: its GroupNorm groups are its own, its conv
                # partial sum enters the per-unit conv as a residual read through bj_of_unit
                Ca, Cb = x0.C, x1.C
                cpg = (Ca + Cb) // G
                weights.split_resnet(key, Ca)
                g1, b1 = P[key + ".norm1.g"], P[key + ".norm1.b"]
                ts = gn_conv3(key + ".gn1s", key + ".conv1s", x1, pb.const(g1[Ca:]), pb.const(b1[Ca:]), Cb // cpg,
                              pb.const(P[key + ".conv1.wb"]), Cout)
                h = gn_conv3(key + ".gn1", key + ".conv1", x0, pb.const(g1[:Ca]), pb.const(b1[:Ca]), Ca // cpg,
                             pb.const(P[key + ".conv1.wa"]), Cout, bias=pb.const(P[key + ".conv1.b"]), rowvec=tvec, residual=ts, qstats=use_qs)
            elif x1 is None:
                h = gn_conv3(key + ".gn1", key + ".conv1", x0, pb.const(P[key + ".norm1.g"]), pb.const(P[key + ".norm1.b"]), G,
                             pb.const(P[key + ".conv1.w"]), Cout, bias=pb.const(P[key + ".conv1.b"]), rowvec=tvec, qstats=use_qs)
            else:
                h = pb.groupnorm(key + ".gn1", x0, pb.const(P[key + ".norm1.g"]), pb.const(P[key + ".norm1.b"]), G, eps, True, x1=x1)
                h = pb.igemm(key + ".conv1", h, pb.const(P[key + ".conv1.w"]), Cout, taps=9, bias=pb.const(P[key + ".conv1.b"]),
                             rowvec=tvec, qstats=use_qs)
            h_raw = h                  # conv1's output: conv2 normalises it itself where gn_conv3 can (norm2 + SiLU)
            n2g, n2b = pb.const(P[key + ".norm2.g"]), pb.const(P[key + ".norm2.b"])
            # conv_shortcut folded into conv2: the 1x1 over the raw input becomes extra K chunks of conv2's own MFMA loop
            # (conv3_halo side source) — no shortcut launch, no shortcut tensor written and read back
            # (the SHORTCUT of a class-shared skip splits even where a GroupNorm group straddles the seam and norm1 / conv1 cannot:
            #  shortcut(cat(x0, x1)) = Wa x0 + Wb x1 is linear — the x1 half once per pair, the x0 half as conv2's side source)
            split_sc = split or (split_skips and x1 is not None and x1.dom == "bj" and x0.dom == "unit" and key + ".conv_shortcut.w" in P)
            fold = (fold_short and key + ".conv_shortcut.w" in P and (x1 is None or split_sc)
                    and pb.side_ok(h, x0, Cout, residual=x1 if split_sc else None))
            if fold:
                if key + ".conv2.bs" not in P:
                    P[key + ".conv2.bs"] = (P[key + ".conv2.b"] + P[key + ".conv_shortcut.b"]).contiguous()
                if split_sc:
                    weights.split_shortcut(key, x0.C)
                ss = pb.igemm(key + ".shorts", x1, pb.const(P[key + ".conv_shortcut.wb"]), Cout) if split_sc else None
                w2 = P[key + ".conv_shortcut.wa"] if split_sc else P[key + ".conv_shortcut.w"]
                return gn_conv3(key + ".gn2", key + ".conv2", h_raw, n2g, n2b, G, pb.const(P[key + ".conv2.w"]), Cout,
                                bias=pb.const(P[key + ".conv2.bs"]), residual=ss, side=(x0, pb.const(w2)), qstats=use_qs)
            if split and key + ".conv_shortcut.w" in P:
                ss = pb.igemm(key + ".shorts", x1, pb.const(P[key + ".conv_shortcut.wb"]), Cout)
                sc = pb.igemm(key + ".short", x0, pb.const(P[key + ".conv_shortcut.wa"]), Cout,
                              bias=pb.const(P[key + ".conv_shortcut.b"]), residual=ss)
            elif key + ".conv_shortcut.w" in P:
                sc = pb.igemm(key + ".short", x0, pb.const(P[key + ".conv_shortcut.w"]), Cout, src1=x1,
                              bias=pb.const(P[key + ".conv_shortcut.b"]))
            else:
                assert x1 is None
                sc = x0
            return gn_conv3(key + ".gn2", key + ".conv2", h_raw, n2g, n2b, G, pb.const(P[key + ".conv2.w"]), Cout,
                            bias=pb.const(P[key + ".conv2.b"]), residual=sc, qstats=use_qs)

        def transformer(key, x):
            Cc = x.C
            tbk = key + ".transformer_blocks.0"
            h = pb.pn_claim(x, pb.const(P[key + ".norm.g"]), pb.const(P[key + ".norm.b"]), G, 1e-6, False)
            if h is None:
                h = pb.groupnorm(key + ".gn", x, pb.const(P[key + ".norm.g"]), pb.const(P[key + ".norm.b"]), G, 1e-6, False)
            if fuse_tb and h.dom == pb._dom(h, cvec[key]) and pb.tblock_front_ok(h, heads):
                # proj_in -> LayerNorm -> q/k/v -> attention -> to_out + class vector + residual in ONE launch, the sample on chip
                h = pb.tblock_front(tbk + ".front", h, pb.const(P[key + ".proj_in.w"]), pb.const(P[key + ".proj_in.b"]),
                                    pb.const(P[tbk + ".norm1.g"]), pb.const(P[tbk + ".norm1.b"]), pb.const(P[tbk + ".qkv.w"]),
                                    pb.const(P[tbk + ".attn1.to_out.0.w"]), pb.const(P[tbk + ".attn1.to_out.0.b"]), cvec[key], heads, 1e-5)
                return transformer_back(key, tbk, x, h, Cc)
            h = pb.igemm(key + ".proj_in", h, pb.const(P[key + ".proj_in.w"]), Cc, bias=pb.const(P[key + ".proj_in.b"]))
            # LayerNorm folded into the consuming GEMM where the activation-stationary kernel can standardise the rows itself
            # (gamma into W's columns, beta into the bias: UNetWeights.fold_layernorms): no LayerNorm launch, no normalised tensor
            # (q/k/v keeps its LayerNorm launch: measured faster on the 256x256 tile + LayerNorm than on the row-standardising GEMM)
            hn = pb.layernorm(tbk + ".ln1", h, pb.const(P[tbk + ".norm1.g"]), pb.const(P[tbk + ".norm1.b"]), 1e-5)
            qkv = pb.igemm(tbk + ".qkv", hn, pb.const(P[tbk + ".qkv.w"]), 3 * Cc)
            o = pb.attention(tbk + ".attn1", qkv.view(0, Cc), qkv.view(Cc, Cc), qkv.view(2 * Cc, Cc), heads)
            h = pb.igemm(tbk + ".attn_out", o, pb.const(P[tbk + ".attn1.to_out.0.w"]), Cc,
                         bias=pb.const(P[tbk + ".attn1.to_out.0.b"]), rowvec=cvec[key], residual=h)
            return transformer_back(key, tbk, x, h, Cc)

        def transformer_back(key, tbk, x, h, Cc):
            if fold_ln and pb.ln_ok(h, 8 * Cc, L.ACT_GEGLU):
                weights.fold_layernorms(tbk)
                f = pb.igemm(tbk + ".geglu", h, pb.const(P[tbk + ".ff.net.0.proj.wf"]), 8 * Cc,
                             bias=pb.const(P[tbk + ".ff.net.0.proj.bf"]), act=L.ACT_GEGLU, ln_eps=1e-5)
            else:
                hn = pb.layernorm(tbk + ".ln3", h, pb.const(P[tbk + ".norm3.g"]), pb.const(P[tbk + ".norm3.b"]), 1e-5)
                f = pb.igemm(tbk + ".geglu", hn, pb.const(P[tbk + ".ff.net.0.proj.w"]), 8 * Cc,
                             bias=pb.const(P[tbk + ".ff.net.0.proj.b"]), act=L.ACT_GEGLU)
            if fold_po:
                # ff.net.2 and proj_out as one GEMM over [f | h] (UNetWeights.fold_proj_out): no ff_out tensor, one launch fewer
                weights.fold_proj_out(key)
                return pb.igemm(key + ".ff_proj_out", f, pb.const(P[key + ".ffpo.w"]), Cc, src1=h, bias=pb.const(P[key + ".ffpo.b"]), residual=x)
            h = pb.igemm(tbk + ".ff_out", f, pb.const(P[tbk + ".ff.net.2.w"]), Cc, bias=pb.const(P[tbk + ".ff.net.2.b"]),
                         residual=h)
            return pb.igemm(key + ".proj_out", h, pb.const(P[key + ".proj_out.w"]), Cc, bias=pb.const(P[key + ".proj_out.b"]),
                            residual=x)

        # ---- main path ----
        # share_trunk=False recomputes the class-independent layers per unit (reference-equivalent
        # executed FLOPs; used for A/B tests): conv_in then reads its operand through bj_of_unit.
        h = pb.igemm("conv_in", a0, pb.const(P["conv_in.w"]), C0, bias=pb.const(P["conv_in.b"]),
                     dom=None if share_trunk else "unit", k_real=9 * cfg.in_channels)
        skips = [h]
        boc = cfg.block_out_channels
        for i, kind in enumerate(cfg.down_block_types):
            for j in range(cfg.layers_per_block[i]):
                h = resnet(f"down_blocks.{i}.resnets.{j}", h)
                if kind == "CrossAttnDownBlock2D":
                    h = transformer(f"down_blocks.{i}.attentions.{j}", h)
                skips.append(h)
            if i != len(boc) - 1:
                key = f"down_blocks.{i}.downsamplers.0.conv"
                h = pb.igemm(key, h, pb.const(P[key + ".w"]), boc[i], taps=9, stride=2, bias=pb.const(P[key + ".b"]))
                skips.append(h)
        h = resnet("mid_block.resnets.0", h)
        h = transformer("mid_block.attentions.0", h)
        h = resnet("mid_block.resnets.1", h)
        nb = len(boc)
        rlpb = cfg.layers_per_block[::-1]
        for i, kind in enumerate(cfg.up_block_types):
            for j in range(rlpb[i] + 1):
                h = resnet(f"up_blocks.{i}.resnets.{j}", h, skips.pop())
                if kind == "CrossAttnUpBlock2D":
                    h = transformer(f"up_blocks.{i}.attentions.{j}", h)
            if i != nb - 1:
                key = f"up_blocks.{i}.upsamplers.0.conv"
                if use_up4 and pb.up4_ok(h, h.C):
                    # nearest-2x upsample + 3x3 conv == four 2x2-tap convs of the low-resolution tensor (phase-summed weights): 4/9 of the MACs
                    if key + ".w4" not in P:
                        P[key + ".w4"] = pack_up4(weights._sd[key + ".weight"], weights.dt, weights.dev)
                    h = pb.igemm(key, h, pb.const(P[key + ".w4"]), h.C, taps=9, upsample=1, bias=pb.const(P[key + ".b"]), qstats=use_qs, up4=True)
                else:
                    h = pb.igemm(key, h, pb.const(P[key + ".w"]), h.C, taps=9, upsample=1, bias=pb.const(P[key + ".b"]), qstats=use_qs)
        tn_out = 32 if cfg.out_channels <= 32 else 128
        hn = pb.pn_claim(h, pb.const(P["conv_norm_out.g"]), pb.const(P["conv_norm_out.b"]), G, eps, True)
        if hn is not None:      # the last ResNet's conv2 stored conv_norm_out + SiLU of its output: conv_out is a plain thin conv
            pred = pb.igemm("conv_out", hn, pb.const(P["conv_out.w"]), cfg.out_channels, taps=9, bias=pb.const(P["conv_out.b"]),
                            out_dt=L.DC_F32, tile_n=tn_out)
        elif (fuse_gn_out and pb.qstats_ok(h, None, G, h.dom) and pb.gn_fusable(h, None, cfg.out_channels, tile_n=tn_out, out_dt=L.DC_F32)):
            # conv_norm_out + SiLU inside conv_out's halo load (the thin-output conv reads ~1 GB for 3 channels and its VALU is idle):
            # the affine comes from the last conv's quad records, the normalised tensor never exists, one launch instead of two
            aff = pb.groupnorm_stats("conv_norm_out", h, pb.const(P["conv_norm_out.g"]), pb.const(P["conv_norm_out.b"]), G, eps)
            pred = pb.igemm("conv_out", h, pb.const(P["conv_out.w"]), cfg.out_channels, taps=9, bias=pb.const(P["conv_out.b"]),
                            out_dt=L.DC_F32, tile_n=tn_out, gn=(aff[0], aff[1], True))
        else:
            h = pb.groupnorm("conv_norm_out", h, pb.const(P["conv_norm_out.g"]), pb.const(P["conv_norm_out.b"]), G, eps, True)
            pred = pb.igemm("conv_out", h, pb.const(P["conv_out.w"]), cfg.out_channels, taps=9, bias=pb.const(P["conv_out.b"]),
                            out_dt=L.DC_F32, tile_n=tn_out)
        if pred.dom != "unit":
            raise L.DcamdError("backbone has no class-conditioned layer: nothing to score per class")
        self.pred = pred
        if score is not None:
            pb.eps_mse(pred, eps_t, pb.const(score["x"]), al, sg, pb.const(self.bj_of_unit), pb.const(score["img_of_bj"]),
                       pb.const(score["out_index"]), pb.const(score["errors"]), cfg.in_channels, score["v_param"])
        pb.finalize(keep_alive=[pred])

    def run(self):
        self.pb.run()

    def run_ctx(self):
        """Refresh the per-class cross-attention vectors from `self.ctx` (call after changing ctx / weights)."""
        self.ctx_pb.run()

    def pred_view(self):
        return self.pb.tensor_view(self.pred)

    @property
    def arena_bytes(self):
        return self.pb.arena_bytes
