#!/usr/bin/env python3
"""Developer diagnostic: do an MFMA-bound conv and an HBM-bound GroupNorm overlap when they run on two streams with DISJOINT
CU masks (hipExtStreamCreateWithCUMask)?  Prints the loop times alone and together.
  python tools/dev/overlap_probe.py [--split 192]"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import diffusion_classifier_amd as dca  # noqa: E402,F401
from diffusion_classifier_amd import _lib as L  # noqa: E402
from diffusion_classifier_amd import engine as E  # noqa: E402


def masked_stream(hip, bits):
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    sp = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(sp.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--split", type=int, default=192)
    ap.add_argument("--n", type=int, default=4000)
    args = ap.parse_args()
    lib = L.require_gpu()
    dev = "cuda:0"
    torch.zeros(1, device=dev)
    hip = ctypes.CDLL("libamdhip64.so")
    dt, td = L.DC_BF16, torch.bfloat16
    n, H, W, Ci, Co = args.n, 32, 32, 128, 128
    x = torch.randn(n, H, W, Ci, device=dev).to(td)
    Wp = E.pack_conv3x3(torch.randn(Co, Ci, 3, 3) / 30, dt, dev)
    b = torch.randn(Co, device=dev)
    out = torch.empty(n, H, W, Co, device=dev, dtype=td)
    pc = L.IgemmParams(dtype=dt, taps=9, stride=1, upsample=0, n_img=n, Hin=H, Win=W, Hout=H, Wout=W, src0=x.data_ptr(), C0=Ci,
                       W=Wp.data_ptr(), Cout=Co, tile_n=128, bias=b.data_ptr(), out=out.data_ptr(), out_dtype=dt, out_ld=Co)
    parts = lib.dc_igemm_qstats_parts(pc)
    qs = torch.zeros(n * parts * (Co // 4) * 2, device=dev)
    pc.qstats = qs.data_ptr()
    L.check(lib.dc_igemm(pc, L.stream_ptr()))
    gx = out.clone()
    gy = torch.empty_like(gx)
    gam, bet = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    ws = torch.zeros(n * 32 * 4 * 2, device=dev)
    pg = L.GroupnormParams(x=gx.data_ptr(), y=gy.data_ptr(), dtype=dt, out_dtype=dt, n=n, HW=H * W, C=Co, C1=0, groups=32, silu=1,
                           splits=4, eps=1e-5, gamma=gam.data_ptr(), beta=bet.data_ptr(), ws=ws.data_ptr(), qstats=qs.data_ptr(), qparts=parts)
    torch.cuda.synchronize()

    def loop(fn, p, stream, reps):
        with torch.cuda.stream(stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                L.check(fn(p, ctypes.c_void_p(stream.cuda_stream)))
            e1.record()
        return e0, e1

    full = torch.cuda.current_stream()
    sa = masked_stream(hip, range(args.split))
    sb = masked_stream(hip, range(args.split, 256))
    RC, RG = 10, 30
    for name, s in (("full chip", full), (f"{args.split} CUs", sa)):
        for _ in range(2):
            e = loop(lib.dc_igemm, pc, s, RC); torch.cuda.synchronize()
        print(f"conv alone on {name}: {e[0].elapsed_time(e[1]) / RC:.3f} ms", flush=True)
    for name, s in (("full chip", full), (f"{256 - args.split} CUs", sb)):
        for _ in range(2):
            e = loop(lib.dc_groupnorm, pg, s, RG); torch.cuda.synchronize()
        ms = e[0].elapsed_time(e[1]) / RG
        print(f"groupnorm alone on {name}: {ms:.3f} ms  {2 * gx.numel() * 2 / ms / 1e6:.0f} GB/s", flush=True)
    for _ in range(2):
        ec = loop(lib.dc_igemm, pc, sa, RC)
        eg = loop(lib.dc_groupnorm, pg, sb, RG)
        torch.cuda.synchronize()
    mc, mg = ec[0].elapsed_time(ec[1]) / RC, eg[0].elapsed_time(eg[1]) / RG
    print(f"together: conv {mc:.3f} ms on {args.split} CUs, groupnorm {mg:.3f} ms ({2 * gx.numel() * 2 / mg / 1e6:.0f} GB/s) on {256 - args.split} CUs", flush=True)
    # the same pair without masks, two plain streams
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(2):
        ec = loop(lib.dc_igemm, pc, s1, RC)
        eg = loop(lib.dc_groupnorm, pg, s2, RG)
        torch.cuda.synchronize()
    print(f"two unmasked streams: conv {ec[0].elapsed_time(ec[1]) / RC:.3f} ms, groupnorm {eg[0].elapsed_time(eg[1]) / RG:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
