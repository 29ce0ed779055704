#!/usr/bin/env python3
"""Developer diagnostic: socket power and shader clock (rocm-smi) while ONE kernel family runs back to back for a few seconds.
  python tools/dev/power_probe.py conv|gn|geglu|idle [seconds]"""
import ctypes
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import diffusion_classifier_amd as dca  # noqa: E402,F401
from diffusion_classifier_amd import _lib as L  # noqa: E402
from diffusion_classifier_amd import engine as E  # noqa: E402

which = sys.argv[1]
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
lib = L.require_gpu()
dev, dt, td = "cuda:0", L.DC_BF16, torch.bfloat16
n, H, W, Ci, Co = 4000, 32, 32, 128, 128
x = torch.randn(n, H, W, Ci, device=dev).to(td)
out = torch.empty(n, H, W, Co, device=dev, dtype=td)
b = torch.randn(2048, device=dev)
if which in ("dit_qkv", "dit_fc2"):
    K, N = (768, 2304) if which == "dit_qkv" else (3072, 768)
    td16 = torch.float16
    Wp = E.pack_matrix(torch.randn(N, K) / K ** 0.5, L.DC_F16, dev)
    xg = torch.randn(1000, 32, 32, K, device=dev).to(td16)
    og = torch.empty(1000, 32, 32, N, device=dev, dtype=td16)
    p = L.IgemmParams(dtype=L.DC_F16, taps=1, stride=1, upsample=0, n_img=1000, Hin=32, Win=32, Hout=32, Wout=32, src0=xg.data_ptr(), C0=K,
                      W=Wp.data_ptr(), Cout=N, tile_n=128, bias=b.data_ptr() if N <= 2048 else None, out=og.data_ptr(), out_dtype=L.DC_F16, out_ld=N)
    fn, flops, nbytes = lib.dc_igemm, 2.0 * 1000 * 1024 * K * N, 0
elif which == "geglu":
    Wp = E.pack_matrix(torch.randn(2048, 256) / 16, dt, dev)
    xg = torch.randn(8000, 8, 8, 256, device=dev).to(td)
    og = torch.empty(8000, 8, 8, 1024, device=dev, dtype=td)
    p = L.IgemmParams(dtype=dt, taps=1, stride=1, upsample=0, n_img=8000, Hin=8, Win=8, Hout=8, Wout=8, src0=xg.data_ptr(), C0=256,
                      W=Wp.data_ptr(), Cout=2048, tile_n=128, bias=b.data_ptr(), act=L.ACT_GEGLU, out=og.data_ptr(), out_dtype=dt, out_ld=1024)
    fn, flops, nbytes = lib.dc_igemm, 2.0 * 8000 * 64 * 256 * 2048, 0
else:
    Wp = E.pack_conv3x3(torch.randn(Co, Ci, 3, 3) / 30, dt, dev)
    p = L.IgemmParams(dtype=dt, taps=9, stride=1, upsample=0, n_img=n, Hin=H, Win=W, Hout=H, Wout=W, src0=x.data_ptr(), C0=Ci,
                      W=Wp.data_ptr(), Cout=Co, tile_n=128, bias=b.data_ptr(), out=out.data_ptr(), out_dtype=dt, out_ld=Co)
    parts = lib.dc_igemm_qstats_parts(p)
    qs = torch.zeros(n * parts * (Co // 4) * 2, device=dev)
    p.qstats = qs.data_ptr()
    fn, flops, nbytes = lib.dc_igemm, 2.0 * n * H * W * Ci * 9 * Co, 0
    L.check(fn(p, L.stream_ptr()))
    if which == "gn":
        gy = torch.empty_like(out)
        gam, bet = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
        ws = torch.zeros(n * 32 * 4 * 2, device=dev)
        p = L.GroupnormParams(x=out.data_ptr(), y=gy.data_ptr(), dtype=dt, out_dtype=dt, n=n, HW=H * W, C=Co, C1=0, groups=32, silu=1, splits=4,
                              eps=1e-5, gamma=gam.data_ptr(), beta=bet.data_ptr(), ws=ws.data_ptr(), qstats=qs.data_ptr(), qparts=parts)
        fn, flops, nbytes = lib.dc_groupnorm, 0, 2.0 * out.numel() * 2
torch.cuda.synchronize()


def smi():
    o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    pw = [l.split(":")[-1].strip() for l in o.splitlines() if "Power (W)" in l]
    sc = [l.split("(")[-1].rstrip(")") for l in o.splitlines() if "sclk" in l]
    return (pw[0] if pw else "?"), (sc[0] if sc else "?")


t0 = time.time()
samples, launches = [], 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    if which != "idle":
        for _ in range(200 if not which.startswith("dit") else 40):
            L.check(fn(p, L.stream_ptr()))
        launches += 200 if not which.startswith("dit") else 40
    if time.time() - t0 > 1.0:
        samples.append(smi())
    if which == "idle":
        time.sleep(0.3)
    else:
        torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / max(launches, 1)
print(which, f"{ms:.4f} ms/launch (incl. sampling gaps)", f"{flops / ms / 1e9:.0f} TF" if flops else "", f"{nbytes / ms / 1e6:.0f} GB/s" if nbytes else "", samples[:10], flush=True)
