"""Developer tool: dc_tblock_front (one launch) against the chain of launches it replaces (proj_in, LayerNorm, q/k/v, attention, to_out),
at the cfg2 shape: n samples x 64 tokens x 256 channels x 8 heads, bf16.  usage: python3 tools/bench_tblock.py [n] [heads]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import diffusion_classifier_amd as dca  # noqa: F401
from diffusion_classifier_amd import _lib as L
from diffusion_classifier_amd import engine as E

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
heads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dt, td, DEV, Lq, Cc = L.DC_BF16, torch.bfloat16, "cuda:0", 64, 256
lib = L.lib()
torch.manual_seed(0)
ptr = lambda t: t.data_ptr()
x = torch.randn(n, Lq, Cc, device=DEV).to(td)
Wp, Wq, Wo = (E.pack_matrix(torch.randn(r, Cc) / Cc ** 0.5, dt, DEV) for r in (Cc, 3 * Cc, Cc))
bp, bo, g, b = (torch.randn(Cc, device=DEV) * 0.1 for _ in range(4))
cv = torch.randn(10, Cc, device=DEV)
cm = (torch.arange(n, device=DEV, dtype=torch.int32) % 10).contiguous()
out = torch.empty(n, Lq, Cc, dtype=td, device=DEV)
tp = L.TblockFrontParams(x=ptr(x), Wp=ptr(Wp), bp=ptr(bp), ln_g=ptr(g), ln_b=ptr(b), Wqkv=ptr(Wq), Wo=ptr(Wo), bo=ptr(bo), rowvec=ptr(cv),
                         rowvec_map=ptr(cm), out=ptr(out), dtype=dt, n=n, L=Lq, C=Cc, heads=heads, ldx=Cc, ld_out=Cc, rowvec_ld=Cc, ln_eps=1e-5,
                         scale=(Cc // heads) ** -0.5)
M = n * Lq
hd, hnd, od, ch = (torch.empty(M, Cc, dtype=td, device=DEV) for _ in range(4))
qkv = torch.empty(M, 3 * Cc, dtype=td, device=DEV)
ig = lambda src, W, Cout, **kw: L.IgemmParams(dtype=dt, taps=1, stride=1, upsample=0, n_img=n, Hin=8, Win=8, Hout=8, Wout=8, src0=ptr(src), C0=Cc, ld0=Cc,
                                              W=ptr(W), Cout=Cout, tile_n=128, out_dtype=dt, out_ld=Cout, **kw)
p1 = ig(x, Wp, Cc, bias=ptr(bp), out=ptr(hd))
p2 = L.LayernormParams(x=ptr(hd), y=ptr(hnd), dtype=dt, out_dtype=dt, rows=M, C=Cc, rows_per_sample=Lq, eps=1e-5, gamma=ptr(g), beta=ptr(b))
p3 = ig(hnd, Wq, 3 * Cc, out=ptr(qkv))
p4 = L.AttentionParams(q=ptr(qkv), k=ptr(qkv) + 2 * Cc, v=ptr(qkv) + 4 * Cc, out=ptr(od), dtype=dt, n=n, L=Lq, heads=heads, d=Cc // heads, ld_qkv=3 * Cc,
                       ld_out=Cc, scale=(Cc // heads) ** -0.5)
p5 = ig(od, Wo, Cc, bias=ptr(bo), rowvec=ptr(cv), rowvec_map=ptr(cm), rowvec_ld=Cc, residual=ptr(hd), res_dtype=dt, res_ld=Cc, out=ptr(ch))
st = L.stream_ptr()


def chain():
    L.check(lib.dc_igemm(p1, st)); L.check(lib.dc_layernorm(p2, st)); L.check(lib.dc_igemm(p3, st)); L.check(lib.dc_attention(p4, st)); L.check(lib.dc_igemm(p5, st))


def fused():
    L.check(lib.dc_tblock_front(tp, st), "tblock")


def timeit(f, it=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


tc, tf = timeit(chain), timeit(fused)
flops = 2.0 * M * Cc * 5 * Cc + 4.0 * n * heads * Lq * Lq * (Cc // heads)
err = (out.float() - ch.float().view(n, Lq, Cc)).abs().max().item()
print(f"n={n} heads={heads}: chain {tc:.3f} ms, fused {tf:.3f} ms ({flops / tf / 1e9:.0f} TFLOP/s, {2 * M * Cc * 2 / tf / 1e6:.0f} GB/s of sample bytes), max |fused - chain| {err:.3e}")
