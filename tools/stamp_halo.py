#!/usr/bin/env python3
"""Diagnostic (developer tool): build libdcamd with -DDC_STAMPS into gpurun_out/ and print where a
conv3_halo block spends its cycles (s_memtime stamps; shares only, never a timing claim)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libdcamd_stamps.so")
src = os.path.join(ROOT, "diffusion-classifier_amd", "csrc")
srcs = [f for f in sorted(os.listdir(src)) if f.endswith(".hip")]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DDC_STAMPS", "-shared", "-Wno-unused-function",
                f"-I{ROOT}/include", "-o", out] + [os.path.join(src, f) for f in srcs], check=True)
os.environ["DCAMD_LIB"] = out
os.environ["DCAMD_NO_WS"] = "1"        # this tool stamps conv3_halo_kernel (tools/stamp_ws.py: the wave-specialised kernel)
import torch
import diffusion_classifier_amd as dca
from diffusion_classifier_amd import _lib as L, engine as E
lib = L.lib()
n, H, W, Ci, Co = 1020, 32, 32, int(os.environ.get("CI", "128")), 128
dt = L.DC_BF16
x = torch.randn(n, H, W, Ci, device="cuda").to(torch.bfloat16)
Wp = E.pack_conv3x3(torch.randn(Co, Ci, 3, 3) / 30, dt, "cuda")
b = torch.randn(Co, device="cuda")
r = torch.randn(n, H, W, Co, device="cuda").to(torch.bfloat16) if os.environ.get("RES", "1") == "1" else None
o = torch.empty(n, H, W, Co, device="cuda", dtype=torch.bfloat16)
p = L.IgemmParams(dtype=dt, taps=9, stride=1, upsample=0, n_img=n, Hin=H, Win=W, Hout=H, Wout=W, src0=x.data_ptr(), C0=Ci,
                  W=Wp.data_ptr(), Cout=Co, tile_n=128, bias=b.data_ptr(), residual=r.data_ptr() if r is not None else None,
                  res_dtype=dt, res_ld=Co, out=o.data_ptr(), out_dtype=dt, out_ld=Co)
if os.environ.get("QS") == "1":        # also form the quad statistics (they sit inside the "epi:stores" interval)
    qs = torch.zeros(n * lib.dc_igemm_qstats_parts(p) * (Co // 4) * 2, device="cuda")
    p.qstats = qs.data_ptr()
nblk = n * H * W // (256 if H > 8 else 512)
st = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
lib.dc_debug_set_stamps.argtypes = [ctypes.c_void_p]
for _ in range(2):
    L.check(lib.dc_igemm(p, L.stream_ptr()))
torch.cuda.synchronize()
lib.dc_debug_set_stamps(st.data_ptr())
L.check(lib.dc_igemm(p, L.stream_ptr()))
torch.cuda.synchronize()
s = st.view(nblk, 8).cpu().double()
names = ["setup", "mainloop", "epi:bias", "epi:loads0", "epi:math+batch1", "epi:stores"]
s = s[:, [0, 1, 2, 3, 4, 5, 7]]
d = s[:, 1:] - s[:, :-1]
print("cycles (100MHz s_memtime ticks) per block, median:")
for i, nm in enumerate(names):
    print(f"  {nm:10s} {d[:, i].median().item():10.0f}")
print("  total      ", (s[:, 6] - s[:, 0]).median().item())
t0 = s[:, 0].min().item()
print("  kernel span", (s[:, 6].max().item() - t0), "ticks;  blocks", nblk, " blocks per CU-slot (512):", nblk / 512)
