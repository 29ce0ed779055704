#!/usr/bin/env python3
"""Diagnostic (developer tool): where a dc_tblock_front workgroup spends its cycles (s_memtime stamps of wave 0, medians over workgroups; a
-DDC_STAMPS build of csrc/tblock.hip linked into STAMP_LIB — tools/dev/build_tb_alt.sh tbstamps -DDC_STAMPS).  usage: STAMP_LIB=... python3 tools/stamp_tblock.py [n] [heads]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DCAMD_LIB"] = os.path.abspath(os.environ.get("STAMP_LIB", os.path.join(ROOT, "tools/dev/_build/libdcamd_tbstamps.so")))
import torch
from diffusion_classifier_amd import _lib as L, engine as E
lib = L.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
heads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dt, td, Lq, Cc = L.DC_BF16, torch.bfloat16, 64, 256
ptr = lambda t: t.data_ptr()
x = torch.randn(n, Lq, Cc, device="cuda").to(td)
Wp, Wq, Wo = (E.pack_matrix(torch.randn(r, Cc) / Cc ** 0.5, dt, "cuda") for r in (Cc, 3 * Cc, Cc))
bp, bo, g, b = (torch.randn(Cc, device="cuda") * 0.1 for _ in range(4))
cv = torch.randn(10, Cc, device="cuda")
cm = (torch.arange(n, device="cuda", dtype=torch.int32) % 10).contiguous()
out = torch.empty(n, Lq, Cc, dtype=td, device="cuda")
tp = L.TblockFrontParams(x=ptr(x), Wp=ptr(Wp), bp=ptr(bp), ln_g=ptr(g), ln_b=ptr(b), Wqkv=ptr(Wq), Wo=ptr(Wo), bo=ptr(bo), rowvec=ptr(cv),
                         rowvec_map=ptr(cm), out=ptr(out), dtype=dt, n=n, L=Lq, C=Cc, heads=heads, ldx=Cc, ld_out=Cc, rowvec_ld=Cc, ln_eps=1e-5,
                         scale=(Cc // heads) ** -0.5)
st = torch.zeros(n * 16, dtype=torch.int64, device="cuda")
lib.dc_debug_set_tb_stamps.argtypes = [ctypes.c_void_p]
lib.dc_debug_set_tb_stamps(None)
for _ in range(3):
    L.check(lib.dc_tblock_front(tp, L.stream_ptr()))
torch.cuda.synchronize()
lib.dc_debug_set_tb_stamps(st.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
L.check(lib.dc_tblock_front(tp, L.stream_ptr()))
e1.record()
torch.cuda.synchronize()
s = st.view(n, 16).cpu().double()
s = s[s[:, 12] > 0]          # (two samples per workgroup: half as many workgroups)
n = s.shape[0]
names = [(0, 1, "x -> LDS (+ first weight fragments, barrier)"), (1, 2, "proj_in MFMAs (8 k-steps)"), (2, 3, "h -> LDS, LayerNorm, 2 barriers"),
         (3, 4, "K MFMAs"), (4, 6, "V MFMAs (+ K pack)"), (6, 8, "Q MFMAs (+ V pack)"), (8, 9, "Q pack, barrier, attention, barrier"),
         (9, 10, "to_out MFMAs"), (10, 11, "epilogue -> LDS, barrier"), (11, 12, "row stores issued"), (0, 12, "workgroup")]
print(f"dc_tblock_front n={n} heads={heads}: launch {e0.elapsed_time(e1):.3f} ms (stamped build); cycles of wave 0, median over workgroups")
for a, b_, nm in names:
    d = (s[:, b_] - s[:, a])
    print(f"  {nm:48s} {d.median().item():9.0f}   (10% {d.kthvalue(max(1, n // 10)).values.item():.0f}, 90% {d.kthvalue(max(1, n * 9 // 10)).values.item():.0f})")
