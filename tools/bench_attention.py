#!/usr/bin/env python3
"""Micro-benchmark of dc_attention on the DiT-B/4 shape (1024 tokens, 12 heads x 64, f16 / bf16) and a 4096-token one (developer tool;
DCAMD_LIB selects the library, so A/B builds can be timed in one gpurun session).

  python tools/bench_attention.py [n_samples]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_classifier_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for dt, td in ((L.DC_F16, torch.float16), (L.DC_BF16, torch.bfloat16)):
    for Lq, heads, d, nn in ((1024, 12, 64, n), (4096, 12, 64, max(1, n // 16))):
        Cc = heads * d
        torch.manual_seed(1)
        qkv = (torch.randn(nn, Lq, 3 * Cc, device="cuda") * 1.2).to(td)
        out = torch.empty(nn, Lq, Cc, dtype=td, device="cuda")
        p = L.AttentionParams(q=qkv.data_ptr(), k=qkv.data_ptr() + Cc * 2, v=qkv.data_ptr() + 2 * Cc * 2, out=out.data_ptr(), dtype=dt,
                              n=nn, L=Lq, heads=heads, d=d, ld_qkv=3 * Cc, ld_out=Cc, scale=d ** -0.5)
        for _ in range(3):
            L.check(L.lib().dc_attention(p, L.stream_ptr()), "attn")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            L.check(L.lib().dc_attention(p, L.stream_ptr()), "attn")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 4.0 * nn * heads * Lq * Lq * d
        print(f"{'f16' if dt == L.DC_F16 else 'bf16'} L={Lq} n={nn}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s  checksum {out.float().abs().mean().item():.6f}")
