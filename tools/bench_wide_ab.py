#!/usr/bin/env python3
"""A/B of the two main loops of the 256 x 256 GEMM tile inside ONE process, interleaved rounds, random operands
(cdna_hip_programming.md section 5.4 rules 24 / 25): `igemm_wide8` (8-phase loop, igemm_wide.hip) against the 2-stage loop it
replaces (`DCAMD_WIDE_OLD`, igemm_pipe.hip).  Prints per shape the median and the best TFLOP/s of each arm.
  python tools/bench_wide_ab.py [--dtype f16] [--rounds 7] [--reps 10] [--shapes dit_qkv,...]
"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_classifier_amd import _lib as L  # noqa: E402
from diffusion_classifier_amd import engine as E  # noqa: E402

# name: (rows M, K, N, residual, act)
SHAPES = {
    "dit_qkv": (512000, 768, 2304, False, None), "dit_proj": (512000, 768, 768, True, None),
    "dit_fc1": (512000, 768, 3072, False, "gelu_tanh"), "dit_fc2": (512000, 3072, 768, True, None),
    "sq4k": (4096, 4096, 4096, False, None), "sq8k": (8192, 8192, 8192, False, None),
    "t8_qkv": (512000, 256, 768, False, None), "t8_out": (512000, 256, 256, True, None), "t8_ffo": (512000, 1024, 256, True, None),
    "t4_qkv": (128000, 512, 1536, False, None), "t4_ffo": (128000, 2048, 512, True, None),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--order", action="store_true",
                    help="A/B the TILE ORDER of the new loop instead: 'old' = M fastest (DCAMD_NFAST_GEMM_BYTES=0), 'new' = N fastest (default)")
    args = ap.parse_args()
    dt = E.DT[args.dtype]
    td = E.TORCH_DT[dt]
    lib = L.require_gpu()
    dev = "cuda:0"
    for name in args.shapes.split(","):
        M, K, N, res, act = SHAPES[name]
        x = torch.randn(M, K, device=dev).to(td)
        Wp = E.pack_matrix(torch.randn(N, K) / K ** 0.5, dt, dev)
        b = torch.randn(N, device=dev)
        r = torch.randn(M, N, device=dev).to(td) if res else None
        out = torch.empty(M, N, device=dev, dtype=td)
        p = L.IgemmParams(dtype=dt, taps=1, stride=1, upsample=0, n_img=M // 64, Hin=8, Win=8, Hout=8, Wout=8, src0=x.data_ptr(), C0=K,
                          W=Wp.data_ptr(), Cout=N, tile_n=128, bias=b.data_ptr(), residual=r.data_ptr() if res else None, res_dtype=dt,
                          res_ld=N, act={"gelu_tanh": L.ACT_GELU_TANH}.get(act, L.ACT_NONE), out=out.data_ptr(), out_dtype=dt, out_ld=N)
        tf = {"old": [], "new": []}
        names = {}
        outs = {}
        for rnd in range(args.rounds + 1):                 # round 0 = warm-up
            for arm in ("old", "new"):
                if args.order:
                    if arm == "old":
                        os.environ["DCAMD_NFAST_GEMM_BYTES"] = "0"
                    else:
                        os.environ.pop("DCAMD_NFAST_GEMM_BYTES", None)
                elif arm == "old":
                    os.environ["DCAMD_WIDE_OLD"] = "1"
                else:
                    os.environ.pop("DCAMD_WIDE_OLD", None)
                names[arm] = lib.dc_igemm_variant(p).decode()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    L.check(lib.dc_igemm(p, L.stream_ptr()))
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    tf[arm].append(2.0 * M * K * N / (e0.elapsed_time(e1) / args.reps) / 1e9)
                else:
                    outs[arm] = out.float().clone() if M * N <= (1 << 27) else out[:4096].float().clone()
        same = bool(torch.equal(outs["old"], outs["new"]))
        print(f"{name:9s} M={M:7d} K={K:5d} N={N:5d}  old {statistics.median(tf['old']):7.1f} (best {max(tf['old']):7.1f}) TF  "
              f"new {statistics.median(tf['new']):7.1f} (best {max(tf['new']):7.1f}) TF  x{statistics.median(tf['new']) / statistics.median(tf['old']):.3f}"
              f"  bit-identical={same}  [{names['old']} | {names['new']}]", flush=True)
        del x, Wp, r, out
    os.environ.pop("DCAMD_WIDE_OLD", None)
    os.environ.pop("DCAMD_NFAST_GEMM_BYTES", None)


if __name__ == "__main__":
    main()
