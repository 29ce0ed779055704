#!/usr/bin/env python3
"""Diagnostic (developer tool): build libdcamd with -DDC_STAMPS into gpurun_out/ and print where a producer-normalising conv3_halo
workgroup (csrc/epi_pn.h) spends its cycles (s_memtime stamps per workgroup, medians; shares only, never a timing claim), next to the
plain conv of the same shape.  env: HW (32), N (2000), CI / CO (128), RES (1), RAW (1: also store the raw output), ABLS (0,16,32,48:
16 = do not wait for the sample's other tiles, 32 = no SiLU; results wrong on purpose), PLAIN_ABLS (0), STAMP_LIB (a prebuilt -DDC_STAMPS library)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libdcamd_stamps.so")
src = os.path.join(ROOT, "diffusion-classifier_amd", "csrc")
srcs = [f for f in sorted(os.listdir(src)) if f.endswith(".hip")]
if os.environ.get("STAMP_LIB"):        # a -DDC_STAMPS library built beforehand (e.g. into tools/dev/_build/, which travels to the GPU box)
    out = os.path.abspath(os.environ["STAMP_LIB"])
else:
  subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DDC_STAMPS", "-shared", "-Wno-unused-function",
                  f"-I{ROOT}/include", "-o", out] + [os.path.join(src, f) for f in srcs], check=True)
os.environ["DCAMD_LIB"] = out
import torch
import diffusion_classifier_amd as dca
from diffusion_classifier_amd import _lib as L, engine as E
lib = L.lib()
HWS = int(os.environ.get("HW", "32"))
n, H, W, Ci, Co = int(os.environ.get("N", "2000")), HWS, HWS, int(os.environ.get("CI", "128")), int(os.environ.get("CO", "128"))
dt = L.DC_BF16
x = torch.randn(n, H, W, Ci, device="cuda").to(torch.bfloat16)
Wp = E.pack_conv3x3(torch.randn(Co, Ci, 3, 3) / 30, dt, "cuda")
b = torch.randn(Co, device="cuda")
r = torch.randn(n, H, W, Co, device="cuda").to(torch.bfloat16) if os.environ.get("RES", "1") == "1" else None
o = torch.empty(n, H, W, Co, device="cuda", dtype=torch.bfloat16)
y = torch.empty(n, H, W, Co, device="cuda", dtype=torch.bfloat16)
gamma, beta = torch.rand(Co, device="cuda") + 0.5, torch.randn(Co, device="cuda")
base = dict(dtype=dt, taps=9, stride=1, upsample=0, n_img=n, Hin=H, Win=W, Hout=H, Wout=W, src0=x.data_ptr(), C0=Ci, W=Wp.data_ptr(), Cout=Co, tile_n=128,
            bias=b.data_ptr(), residual=r.data_ptr() if r is not None else None, res_dtype=dt, res_ld=Co, out_dtype=dt, out_ld=Co)
qs = torch.zeros(n * (H * W // 128) * (Co // 4) * 2, device="cuda")
cnt = torch.zeros(n * ((Co + 127) // 128), dtype=torch.int32, device="cuda")
plain = L.IgemmParams(out=o.data_ptr(), qstats=qs.data_ptr(), **base)
pn = L.IgemmParams(out=o.data_ptr() if os.environ.get("RAW", "1") == "1" else None, qstats=qs.data_ptr(), pn_out=y.data_ptr(), pn_gamma=gamma.data_ptr(),
                   pn_beta=beta.data_ptr(), pn_cnt=cnt.data_ptr(), pn_ld=Co, pn_groups=32, pn_silu=1, pn_eps=1e-5, **base)
T_ = H * W // 256
nblk = ((n * ((Co + 127) // 128) + 7) // 8 * 8) * T_          # the producer-normalising launch pads its grid to 8 whole groups
st = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
lib.dc_debug_set_stamps.argtypes = [ctypes.c_void_p]
lib.dc_debug_set_halo_abl.argtypes = [ctypes.c_int]
for name, p, names in (("plain", plain, ["setup", "mainloop", "epi:bias", "epi:loads0", "epi:math", "epi:stats+stores"]),
                       ("pn", pn, ["setup", "mainloop", "epi:bias+residual", "stats+publish", "wait", "load+fold", "stores"])):
    print("kernel:", lib.dc_igemm_variant(p).decode())
    for abl in ([int(v) for v in os.environ.get("PLAIN_ABLS", "0").split(",")] if name == "plain" else [int(v) for v in os.environ.get("ABLS", "0,16,32,48").split(",")]):
        lib.dc_debug_set_halo_abl(abl)
        lib.dc_debug_set_stamps(None)
        for _ in range(2):
            L.check(lib.dc_igemm(p, L.stream_ptr()))
        torch.cuda.synchronize()
        lib.dc_debug_set_stamps(st.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.dc_igemm(p, L.stream_ptr()))
        e1.record()
        torch.cuda.synchronize()
        s = st.view(nblk, 8).cpu().double()
        s = s[s[:, 7] > 0]
        st.zero_()
        cols = [0, 1, 2, 3, 4, 5, 7] if name == "plain" else [0, 1, 2, 3, 4, 5, 6, 7]
        s = s[:, cols]
        d = s[:, 1:] - s[:, :-1]
        print(f"--- {name} abl {abl}: launch {e0.elapsed_time(e1):.3f} ms; cycles per workgroup, median: " +
              "  ".join(f"{nm} {d[:, i].median().item():.0f}" for i, nm in enumerate(names)) + f"  total {(s[:, -1] - s[:, 0]).median().item():.0f}")
    lib.dc_debug_set_halo_abl(0)
