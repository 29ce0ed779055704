#!/usr/bin/env python3
"""Diagnostic (developer tool): how long a CU sits between two conv3_ws workgroups.  A -DDC_STAMPS build records, per workgroup, the
wall clock (s_memrealtime, 100 MHz) at its first and last instruction and HW_ID / XCC_ID; workgroups are grouped by (XCC, SE, SH, CU),
sorted by start, and the gap start[i+1] - end[i] is the turnover the in-kernel cycle stamps cannot see.  GN=0: plain variant."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "libdcamd_stamps_ws.so")
src = os.path.join(ROOT, "diffusion-classifier_amd", "csrc")
srcs = [f for f in sorted(os.listdir(src)) if f.endswith(".hip")]
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DDC_STAMPS", "-shared", "-Wno-unused-function",
                f"-I{ROOT}/include", "-o", out] + [os.path.join(src, f) for f in srcs], check=True)
os.environ["DCAMD_LIB"] = out
gn = os.environ.get("GN", "1") == "1"
if not gn:
    os.environ["DCAMD_WS_PLAIN"] = "1"
import torch
from diffusion_classifier_amd import _lib as L, engine as E
lib = L.lib()
n, H, W, Ci, Co = 2040, 32, 32, 128, 128
dt = L.DC_BF16
x = torch.randn(n, H, W, Ci, device="cuda").to(torch.bfloat16)
Wp = E.pack_conv3x3(torch.randn(Co, Ci, 3, 3) / 30, dt, "cuda")
b = torch.randn(Co, device="cuda")
r = torch.randn(n, H, W, Co, device="cuda").to(torch.bfloat16)
o = torch.empty(n, H, W, Co, device="cuda", dtype=torch.bfloat16)
sc, sh = torch.rand(n, Ci, device="cuda") + 0.5, torch.randn(n, Ci, device="cuda") * 0.3
p = L.IgemmParams(dtype=dt, taps=9, stride=1, upsample=0, n_img=n, Hin=H, Win=W, Hout=H, Wout=W, src0=x.data_ptr(), C0=Ci,
                  W=Wp.data_ptr(), Cout=Co, tile_n=128, bias=b.data_ptr(), residual=r.data_ptr(), res_dtype=dt, res_ld=Co, out=o.data_ptr(),
                  out_dtype=dt, out_ld=Co, gn_scale=sc.data_ptr() if gn else None, gn_shift=sh.data_ptr() if gn else None, gn_silu=1)
qs = torch.zeros(n * lib.dc_igemm_qstats_parts(p) * (Co // 4) * 2, device="cuda")
p.qstats = qs.data_ptr()
print("kernel:", lib.dc_igemm_variant(p).decode())
nblk = n * H * W // 256
st = torch.zeros(nblk * 2 * 8, dtype=torch.int64, device="cuda")
lib.dc_debug_set_ws_stamps.argtypes = [ctypes.c_void_p]
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
s = st.view(nblk, 2, 8).cpu()
start, hw, end = s[:, 1, 3], s[:, 1, 4], s[:, 0, 6]
cyc = (s[:, 0, 7] - s[:, 0, 0]).double()
hwid, xcc = hw & 0xFFFFFFFF, (hw >> 32) & 0xF
cu, sh_, se = (hwid >> 8) & 0xF, (hwid >> 12) & 1, (hwid >> 13) & 0x7
key = (((xcc * 8 + se) * 2 + sh_) * 16 + cu).tolist()
from collections import defaultdict
by = defaultdict(list)
for i, k in enumerate(key):
    by[k].append((int(start[i]), int(end[i])))
gaps, busy, span = [], [], []
for k, v in by.items():
    v.sort()
    for (s0, e0_), (s1, _) in zip(v, v[1:]):
        gaps.append((s1 - e0_) * 10.0)      # ns
    busy.append(sum(e - s_ for s_, e in v) * 10.0)
    span.append((v[-1][1] - v[0][0]) * 10.0)
g = torch.tensor(gaps)
print(f"launch {e0.elapsed_time(e1):.3f} ms, {nblk} workgroups on {len(by)} CU slots, {nblk / len(by):.1f} per slot")
print(f"in-kernel cycles per workgroup (s_memtime, thread 0): median {cyc.median():.0f}")
print(f"workgroup wall time (s_memrealtime): median {((end - start).double() * 10).median():.0f} ns")
print(f"gap between consecutive workgroups on one CU slot: median {g.median():.0f} ns, mean {g.mean():.0f} ns, p10 {g.quantile(0.1):.0f}, p90 {g.quantile(0.9):.0f}")
print(f"per CU slot: busy {sum(busy) / len(busy) / 1e3:.1f} us of a {sum(span) / len(span) / 1e3:.1f} us span ({sum(busy) / sum(span):.3f})")
