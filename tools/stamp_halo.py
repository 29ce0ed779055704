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
if os.environ.get("NW8"):
    os.environ["DCAMD_HALO_NW"] = "8"
import torch
import diffusion_classifier_amd as dca
from diffusion_classifier_amd import _lib as L, engine as E
lib = L.lib()
HWS = int(os.environ.get("HW", "32"))
n, H, W, Ci, Co = int(os.environ.get("N", "1020")), HWS, HWS, int(os.environ.get("CI", "128")), int(os.environ.get("CO", "128"))
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
nblk = (n * H * W // (512 if (H <= 8 or os.environ.get("NW8")) else 256)) * ((Co + 127) // 128)
st = torch.zeros(nblk * 8, dtype=torch.int64, device="cuda")
lib.dc_debug_set_stamps.argtypes = [ctypes.c_void_p]
print("kernel:", lib.dc_igemm_variant(p).decode(), " staggered:", not os.environ.get("DCAMD_HALO_NO_STAG"))
lib.dc_debug_set_halo_abl.argtypes = [ctypes.c_int]
ABL = {0: "as shipped", 1: "no MFMAs", 2: "no fragment reads", 3: "barriers and LDS-DMA only", 4: "no LDS-DMA instructions", 5: "reads and barriers only", 6: "MFMAs and barriers only", 7: "barriers only", 8: "no halo LDS-DMA (W tiles still fetched)"}
for abl in [int(v) for v in os.environ.get("ABLS", "0").split(",")]:
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
    names = ["setup", "mainloop", "epi:bias", "epi:loads0", "epi:math+batch1", "epi:stores"]
    s = s[:, [0, 1, 2, 3, 4, 5, 7]]
    d = s[:, 1:] - s[:, :-1]
    print(f"--- ablation {abl} ({ABL.get(abl, '?')}): launch {e0.elapsed_time(e1):.3f} ms; cycles per block, median: " + "  ".join(f"{nm} {d[:, i].median().item():.0f}" for i, nm in enumerate(names)) + f"  total {(s[:, 6] - s[:, 0]).median().item():.0f}")
