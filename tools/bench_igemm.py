#!/usr/bin/env python3
"""Micro-benchmark of dc_igemm on the shapes that dominate the cfg2 scoring step (developer tool;
also the target of the rocprofv3 --pmc passes whose summaries are committed under profiles/).
  python tools/bench_igemm.py [--reps 20] [--dtype bf16] [--shapes name,...]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusion_classifier_amd as dca  # noqa: E402
from diffusion_classifier_amd import _lib as L  # noqa: E402
from diffusion_classifier_amd import engine as E  # noqa: E402

# name: (n_img, H, W, Cin, Cout, taps, residual)
SHAPES = {
    "c32_128_128": (1020, 32, 32, 128, 128, 9, True),
    "c32_256_128": (1020, 32, 32, 256, 128, 9, False),
    "c16_256_128": (1020, 16, 16, 256, 128, 9, False),
    "c16_128_128": (1020, 16, 16, 128, 128, 9, True),
    "c8_512_256": (1020, 8, 8, 512, 256, 9, False),
    "c4_1024_512": (1020, 4, 4, 1024, 512, 9, False),
    "p32_256_128": (1020, 32, 32, 256, 128, 1, False),
    "g64_256_2048": (1020, 64, 1, 256, 2048, 1, False),
    # transformer-block GEMMs of cfg2 at the production micro-batch (4000 units): name -> (..., act)
    "geglu16": (4000, 16, 16, 256, 2048, 1, False, "geglu"),
    "lin16_2048": (4000, 16, 16, 256, 2048, 1, False),
    "lin16_1024": (4000, 16, 16, 256, 1024, 1, False),
    "qkv16": (4000, 16, 16, 256, 768, 1, False),
    "out16": (4000, 16, 16, 256, 256, 1, True),
    "ffo16": (4000, 16, 16, 1024, 256, 1, True),
    "geglu8": (4000, 8, 8, 512, 4096, 1, False, "geglu"),
    "qkv8": (4000, 8, 8, 512, 1536, 1, False),
    "ffo8": (4000, 8, 8, 2048, 512, 1, True),
    "short32": (4000, 32, 32, 256, 128, 1, False),
    # cfg2 transformer blocks as they really run: 8x8 (C=256) and 4x4 (C=512) tokens per unit, 4000 units
    "t8_geglu": (4000, 8, 8, 256, 2048, 1, False, "geglu"),
    "t8_qkv": (4000, 8, 8, 256, 768, 1, False),
    "t8_out": (4000, 8, 8, 256, 256, 1, True),
    "t8_ffo": (4000, 8, 8, 1024, 256, 1, True),
    "t4_geglu": (4000, 4, 4, 512, 4096, 1, False, "geglu"),
    "t4_qkv": (4000, 4, 4, 512, 1536, 1, False),
    "t4_out": (4000, 4, 4, 512, 512, 1, True),
    "t4_ffo": (4000, 4, 4, 2048, 512, 1, True),
    # DiT-B/4 GEMMs (1000 units x 1024 tokens)
    "dit_qkv": (1000, 32, 32, 768, 2304, 1, False),
    "dit_proj": (1000, 32, 32, 768, 768, 1, True),
    "dit_fc1": (1000, 32, 32, 768, 3072, 1, False, "gelu_tanh"),
    "dit_fc2": (1000, 32, 32, 3072, 768, 1, True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--cus", type=int, default=0, help="run on a stream masked to this many CUs (hipExtStreamCreateWithCUMask)")
    ap.add_argument("--cu-pattern", default="low", choices=["low", "spread"], help="which mask bits: the low N, or N spread evenly over 256")
    args = ap.parse_args()
    dt = E.DT[args.dtype]
    td = E.TORCH_DT[dt]
    lib = L.require_gpu()
    dev = "cuda:0"
    masked = None
    if args.cus:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        bits = list(range(args.cus)) if args.cu_pattern == "low" else [int(i * 256 / args.cus) for i in range(args.cus)]
        words = (ctypes.c_uint32 * 8)()
        for b_ in bits:
            words[b_ // 32] |= 1 << (b_ % 32)
        sp = ctypes.c_void_p()
        torch.zeros(1, device=dev)
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), 8, words)
        assert rc == 0, rc
        masked = torch.cuda.ExternalStream(sp.value)
        print(f"# stream masked to {args.cus} CUs ({args.cu_pattern})", flush=True)
    ctx = torch.cuda.stream(masked) if masked is not None else torch.cuda.stream(torch.cuda.current_stream())
    ctx.__enter__()
    for name in args.shapes.split(","):
        n, H, W, Ci, Co, taps, res = SHAPES[name][:7]
        act = SHAPES[name][7] if len(SHAPES[name]) > 7 else None
        Cout_out = Co // 2 if act == "geglu" else Co
        x = torch.randn(n, H, W, Ci, device=dev).to(td)
        w = torch.randn(Co, Ci, 3, 3) / (3 * Ci ** 0.5) if taps == 9 else torch.randn(Co, Ci) / Ci ** 0.5
        Wp = E.pack_conv3x3(w, dt, dev) if taps == 9 else E.pack_matrix(w, dt, dev)
        b = torch.randn(Co, device=dev)
        r = torch.randn(n, H, W, Cout_out, device=dev).to(td) if res else None
        out = torch.empty(n, H, W, Cout_out, device=dev, dtype=td)
        p = L.IgemmParams(dtype=dt, taps=taps, stride=1, upsample=0, n_img=n, Hin=H, Win=W, Hout=H, Wout=W,
                          src0=x.data_ptr(), C0=Ci, W=Wp.data_ptr(), Cout=Co, tile_n=128, bias=b.data_ptr(),
                          residual=r.data_ptr() if res else None, res_dtype=dt, res_ld=Cout_out,
                          act={"geglu": L.ACT_GEGLU, "gelu_tanh": L.ACT_GELU_TANH}.get(act, L.ACT_NONE),
                          out=out.data_ptr(), out_dtype=dt, out_ld=Cout_out)
        for _ in range(3):
            L.check(lib.dc_igemm(p, L.stream_ptr()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            L.check(lib.dc_igemm(p, L.stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        fl = 2.0 * n * H * W * Ci * taps * Co
        gb = (n * H * W * (Ci + Cout_out * (2 if res else 1))) * x.element_size() / 1e9
        var = lib.dc_igemm_variant(p).decode() if hasattr(lib, "dc_igemm_variant") else ""
        print(f"{name:14s} M={n * H * W:8d} N={Co:5d} K={Ci * taps:5d}  {ms:8.4f} ms  {fl / ms / 1e9:7.1f} TF  {gb / ms * 1e3:7.0f} GB/s  {var}", flush=True)


if __name__ == "__main__":
    main()
