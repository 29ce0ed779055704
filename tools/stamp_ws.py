#!/usr/bin/env python3
"""Diagnostic (developer tool): where a conv3_ws workgroup spends its cycles, per team (s_memtime stamps of a -DDC_STAMPS build into
gpurun_out/; shares only, never a timing claim).  GN=0 for the plain variant (DCAMD_WS_PLAIN)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libdcamd_stamps_ws.so")
src = os.path.join(ROOT, "diffusion-classifier_amd", "csrc")
srcs = [f for f in sorted(os.listdir(src)) if f.endswith(".hip")]
extra = ["-DDC_WS_WR=" + os.environ["WR"]] if os.environ.get("WR") else []      # deeper W ring (plain variant only)
extra += ["-D" + d for d in os.environ.get("DEFS", "").split(",") if d]           # e.g. DEFS=DC_WR_CONTIG
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DDC_STAMPS", "-shared", "-Wno-unused-function"] + extra + [
                f"-I{ROOT}/include", "-o", out] + [os.path.join(src, f) for f in srcs], check=True)
os.environ["DCAMD_LIB"] = out
gn = os.environ.get("GN", "1") == "1"
if not gn:
    os.environ["DCAMD_WS_PLAIN"] = "1"
if os.environ.get("PERSIST"):
    os.environ["DCAMD_WS_PERSIST"] = "1"
if os.environ.get("WRK"):
    os.environ["DCAMD_WS_WR"] = "1"
import torch
from diffusion_classifier_amd import _lib as L, engine as E
lib = L.lib()
lib.dc_debug_set_ws_abl.argtypes = [ctypes.c_int]
n, H, W, Ci, Co = 2040, 32, 32, int(os.environ.get("CI", "128")), 128
dt = L.DC_BF16
x = torch.randn(n, H, W, Ci, device="cuda").to(torch.bfloat16)
Wp = E.pack_conv3x3(torch.randn(Co, Ci, 3, 3) / 30, dt, "cuda")
b = torch.randn(Co, device="cuda")
r = torch.randn(n, H, W, Co, device="cuda").to(torch.bfloat16)
o = torch.empty(n, H, W, Co, device="cuda", dtype=torch.bfloat16)
sc, sh = torch.rand(n, Ci, device="cuda") + 0.5, torch.randn(n, Ci, device="cuda") * 0.3
p = L.IgemmParams(dtype=dt, taps=9, stride=1, upsample=0, n_img=n, Hin=H, Win=W, Hout=H, Wout=W, src0=x.data_ptr(), C0=Ci,
                  W=Wp.data_ptr(), Cout=Co, tile_n=128, bias=b.data_ptr(), residual=r.data_ptr() if os.environ.get('RES', '1') == '1' else None, res_dtype=dt, res_ld=Co, out=o.data_ptr(),
                  out_dtype=dt, out_ld=Co, gn_scale=sc.data_ptr() if gn else None, gn_shift=sh.data_ptr() if gn else None, gn_silu=1)
qs = torch.zeros(n * lib.dc_igemm_qstats_parts(p) * (Co // 4) * 2, device="cuda")
p.qstats = qs.data_ptr()
print("kernel:", lib.dc_igemm_variant(p).decode())
one_tile = not (os.environ.get("PERSIST") or os.environ.get("WRK"))      # the default kernel: one tile per workgroup
nblk = n * H * W // 256 if one_tile else 256
st = torch.zeros(nblk * 2 * 8, dtype=torch.int64, device="cuda")
lib.dc_debug_set_ws_stamps.argtypes = [ctypes.c_void_p]
ABL = {0: "as shipped", 1: "no MFMAs", 2: "W fetch dropped", 4: "halo fetch dropped", 6: "no fetch at all", 8: "no transform",
       14: "no fetch, no transform", 15: "barriers, LDS-DMA instructions and fragment reads only", 16: "no LDS-DMA instructions", 24: "no LDS-DMA instructions, no transform",
       25: "fragment reads and barriers only", 31: "fragment reads and barriers only", 48: "barriers and transform only", 56: "barriers only", 40: "no fragment reads, no MFMAs, no transform",
       64: "W fragment loads contiguous (fragment-major image)", 72: "contiguous W fragment loads, no transform"}
for abl in [int(v) for v in os.environ.get("ABLS", "0").split(",")]:
    lib.dc_debug_set_ws_abl(abl)
    lib.dc_debug_set_ws_stamps(None)
    for _ in range(3):
        L.check(lib.dc_igemm(p, L.stream_ptr()))
    torch.cuda.synchronize()
    lib.dc_debug_set_ws_stamps(st.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.dc_igemm(p, L.stream_ptr()))
    e1.record()
    torch.cuda.synchronize()
    s = st.view(nblk, 2, 8).cpu().double()
    print(f"--- ablation {abl} ({ABL.get(abl, '?')}): launch {e0.elapsed_time(e1):.3f} ms")
    for team, nm in ((0, "MFMA team"), (1, "loader team"))[: 2 if one_tile else 1]:
        t = s[:, team]
        if team == 0:
            print(f"{nm}: setup {(t[:,1]-t[:,0]).median():.0f}  main loop {(t[:,2]-t[:,1]).median():.0f}  epilogue {(t[:,7]-t[:,2]).median():.0f}"
                  f" [tables {(t[:,3]-t[:,2]).median():.0f}, residual loads issued {(t[:,4]-t[:,3]).median():.0f}, residual math {(t[:,5]-t[:,4]).median():.0f},"
                  f" statistics + stores {(t[:,7]-t[:,5]).median():.0f}]  total {(t[:,7]-t[:,0]).median():.0f}")
        else:
            print(f"{nm}: setup {(t[:,1]-t[:,0]).median():.0f}  main loop {(t[:,2]-t[:,1]).median():.0f}  total {(t[:,2]-t[:,0]).median():.0f}")
    if os.environ.get("PERSIST"):
        m, l = s[:, 0], s[:, 1]
        print(f"MFMA team (second tile): loop {(m[:,2]-m[:,1]).median():.0f} cycles, of which at barriers {m[:,3].median():.0f}")
        print(f"loader team (second tile): loop {(l[:,2]-l[:,1]).median():.0f} cycles, of which at barriers {l[:,3].median():.0f}, in counted vmcnt waits {l[:,4].median():.0f}")
print("workgroups stamped", nblk, "(persistent kernels: each workgroup's second tile)")
