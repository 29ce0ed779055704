#!/usr/bin/env python3
"""In-kernel clock of the hot kernels (MI355X_MICROARCH.md "DVFS give-back" item 6; replaces the rocm-smi sclk / socket-power samples
DESIGN 6c used to rest on).  Builds a DIAGNOSTIC copy of libdcamd with -DDC_CLOCK_STAMPS into gpurun_out/ (thread 0 of every
workgroup stamps s_memtime and s_memrealtime once before and once after the kernel's main loop, into a buffer of its own), runs each
kernel back to back on random data for >= 2 s, and reports the median over workgroups of
    clock = (delta s_memtime) / (delta s_memrealtime) x 100 MHz
together with the TFLOP/s of those launches.  Never a timing claim for the shipped library (no stamp executes there).
  python tools/clock_probe.py [--seconds 2.5] [--only name,...] > profiles/r03_inkernel_clock.json
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libdcamd_clock.so")
src = os.path.join(ROOT, "diffusion-classifier_amd", "csrc")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [f for f in sorted(os.listdir(src)) if f.endswith(".hip")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DDC_CLOCK_STAMPS", "-shared", "-Wno-unused-function",
                f"-I{ROOT}/include", "-o", out] + [os.path.join(src, f) for f in srcs], check=True)
os.environ["DCAMD_LIB"] = out
import torch  # noqa: E402
from diffusion_classifier_amd import _lib as L, engine as E  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    lib = L.lib()
    dev = "cuda"
    bf, f16 = L.DC_BF16, L.DC_F16
    tdt = {bf: torch.bfloat16, f16: torch.float16}

    def conv_case(n, H, Ci, Co, dt, gn=False, res=True):
        x = torch.randn(n, H, H, Ci, device=dev).to(tdt[dt])
        Wp = E.pack_conv3x3(torch.randn(Co, Ci, 3, 3) / (3 * Ci ** 0.5), dt, dev)
        b = torch.randn(Co, device=dev)
        r = torch.randn(n, H, H, Co, device=dev).to(tdt[dt]) if res else None
        o = torch.empty(n, H, H, Co, device=dev, dtype=tdt[dt])
        kw = dict(dtype=dt, taps=9, stride=1, upsample=0, n_img=n, Hin=H, Win=H, Hout=H, Wout=H, src0=x.data_ptr(), C0=Ci, W=Wp.data_ptr(), Cout=Co,
                  tile_n=128, bias=b.data_ptr(), residual=r.data_ptr() if res else None, res_dtype=dt, res_ld=Co, out=o.data_ptr(), out_dtype=dt, out_ld=Co)
        keep = [x, Wp, b, r, o]
        if gn:
            sc, sh = torch.rand(n, Ci, device=dev) + 0.5, torch.randn(n, Ci, device=dev) * 0.3
            keep += [sc, sh]
            kw.update(gn_scale=sc.data_ptr(), gn_shift=sh.data_ptr(), gn_silu=1)
        return L.IgemmParams(**kw), keep, 2.0 * n * H * H * 9 * Ci * Co

    def gemm_case(M, K, N, dt, act=L.ACT_NONE, res=False):
        x = torch.randn(M, K, device=dev).to(tdt[dt])
        w = torch.randn(N, K) / K ** 0.5
        if act == L.ACT_GEGLU:
            Wp, b = E.pack_geglu(w, torch.randn(N), dt, dev)
        else:
            Wp, b = E.pack_matrix(w, dt, dev), torch.randn(N, device=dev)
        No = N // 2 if act == L.ACT_GEGLU else N
        r = torch.randn(M, No, device=dev).to(tdt[dt]) if res else None
        o = torch.empty(M, No, device=dev, dtype=tdt[dt])
        p = L.IgemmParams(dtype=dt, taps=1, stride=1, upsample=0, n_img=M // 64, Hin=8, Win=8, Hout=8, Wout=8, src0=x.data_ptr(), C0=K, W=Wp.data_ptr(),
                          Cout=N, tile_n=128, bias=b.data_ptr(), residual=r.data_ptr() if res else None, res_dtype=dt, res_ld=No, act=act,
                          out=o.data_ptr(), out_dtype=dt, out_ld=No)
        return p, [x, Wp, b, r, o], 2.0 * M * K * N

    cases = {
        # name: (builder, stamp-buffer setter of the kernel's translation unit, workgroups)
        "conv3_halo<bf16,4w> 32x32x128 K=1152": (lambda: conv_case(4000, 32, 128, 128, bf), "conv3_halo"),
        "conv3_halo<bf16,4w> 32x32 K=3456": (lambda: conv_case(2000, 32, 384, 128, bf, res=False), "conv3_halo"),
        "conv3_halo<bf16,8w> 8x8x512 K=4608": (lambda: conv_case(2000, 8, 512, 512, bf), "conv3_halo"),
        "conv3_ws<bf16,gn> 32x32x128 K=1152": (lambda: conv_case(4000, 32, 128, 128, bf, gn=True), "conv3_ws"),
        "igemm_wide8<f16> DiT qkv K=768": (lambda: gemm_case(512000, 768, 2304, f16), "igemm_wide"),
        "igemm_wide8<f16> DiT fc2 K=3072": (lambda: gemm_case(512000, 3072, 768, f16, res=True), "igemm_wide"),
        "igemm_wide8<bf16> 8192^3": (lambda: gemm_case(8192, 8192, 8192, bf), "igemm_wide"),
        "igemm_xreg<bf16> GEGLU K=256": (lambda: gemm_case(512000, 256, 2048, bf, act=L.ACT_GEGLU), "igemm_xreg"),
    }
    rec = {}
    for name, (build, tu) in cases.items():
        if args.only and not any(k in name for k in args.only.split(",")):
            continue
        p, keep, flops = build()
        variant = lib.dc_igemm_variant(p).decode()
        nbuf = 1 << 20
        st = torch.zeros(nbuf * 4, dtype=torch.int64, device=dev)
        setter = getattr(lib, "dc_debug_set_clk_" + tu)
        setter.argtypes = [ctypes.c_void_p]
        for _ in range(3):
            L.check(lib.dc_igemm(p, L.stream_ptr()))
        torch.cuda.synchronize()
        setter(st.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        launches = 0
        e0.record()
        while True:
            for _ in range(50):
                L.check(lib.dc_igemm(p, L.stream_ptr()))
            launches += 50
            torch.cuda.synchronize()
            if time.perf_counter() - t0 >= args.seconds:
                break
        e1.record()
        torch.cuda.synchronize()
        setter(None)
        ms = e0.elapsed_time(e1) / launches
        s = st.view(nbuf, 4).cpu()
        s = s[(s[:, 0] != 0) & (s[:, 2] != 0)].double()
        dcyc, dreal = s[:, 2] - s[:, 0], s[:, 3] - s[:, 1]
        ok = dreal > 0
        clk = (dcyc[ok] / dreal[ok]) * 100.0          # MHz
        q = torch.quantile(clk, torch.tensor([0.1, 0.5, 0.9], dtype=torch.float64))
        tf = flops / ms / 1e9
        rec[name] = dict(kernel=variant, workgroups=int(ok.sum()), launches=launches, seconds=round(time.perf_counter() - t0, 2), ms_per_launch=round(ms, 4),
                         tflops=round(tf, 1), loop_cycles_median=float(dcyc[ok].median()), clock_mhz_p10=round(float(q[0]), 1),
                         clock_mhz_median=round(float(q[1]), 1), clock_mhz_p90=round(float(q[2]), 1),
                         frac_of_2500_at_2400=round(tf / 2500.0, 4), frac_of_peak_at_held_clock=round(tf / (2500.0 * float(q[1]) / 2400.0), 4))
        print(f"# {name}: {variant} {tf:.0f} TF, in-kernel clock median {float(q[1]):.0f} MHz (p10 {float(q[0]):.0f}, p90 {float(q[2]):.0f})", file=sys.stderr, flush=True)
        del keep, st
        torch.cuda.empty_cache()
    print(json.dumps(dict(method="delta s_memtime / delta s_memrealtime x 100 MHz around the main loop, median over workgroups of the last launch, "
                                 "after >= %.1f s of back-to-back launches on random data (diagnostic -DDC_CLOCK_STAMPS build)" % args.seconds,
                          kernels=rec), indent=1))


if __name__ == "__main__":
    main()
