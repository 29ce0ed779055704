#!/usr/bin/env python3
"""Diagnostic (developer tool): cycles wave 0 of an igemm_xreg workgroup spends in its K loops (of which: waiting for weight slices and at
the slice barrier) and in its epilogues, for the GEGLU projection of a cfg2 8x8-level block (M = 512000, K = 256, N = 2048) or the 4x4-level
one (M = 128000, K = 512, N = 4096).  STAMP_LIB: a -DDC_XR_STAMPS build (tools/dev/build_one_alt.sh xrstamps igemm_xreg.hip -DDC_XR_STAMPS)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DCAMD_LIB"] = os.path.abspath(os.environ.get("STAMP_LIB", os.path.join(ROOT, "tools/dev/_build/libdcamd_xrstamps.so")))
import torch
from diffusion_classifier_amd import _lib as L, engine as E
lib = L.lib()
lib.dc_debug_set_xr_stamps.argtypes = [ctypes.c_void_p]
dt, td = L.DC_BF16, torch.bfloat16
for M, K, NH, HW in ((512000, 256, 1024, 64), (128000, 512, 2048, 16)):
    x = torch.randn(M, K, device="cuda").to(td)
    w, b = torch.randn(2 * NH, K) / K ** 0.5, torch.randn(2 * NH) * 0.1
    Wp, bp = E.pack_geglu(w, b, dt, "cuda", ln_gamma=torch.ones(K), ln_beta=torch.zeros(K))
    out = torch.empty(M, NH, dtype=td, device="cuda")
    side = int(HW ** 0.5)
    p = L.IgemmParams(dtype=dt, taps=1, stride=1, upsample=0, n_img=M // HW, Hin=side, Win=side, Hout=side, Wout=side, src0=x.data_ptr(), C0=K, ld0=K,
                      W=Wp.data_ptr(), Cout=2 * NH, tile_n=128, bias=bp.data_ptr(), act=L.ACT_GEGLU, out=out.data_ptr(), out_dtype=dt, out_ld=NH, ln_eps=1e-5)
    print("kernel:", lib.dc_igemm_variant(p).decode(), f"M={M} K={K} N={2 * NH}")
    rows = 96 if K <= 256 else 64
    nb = (M + rows - 1) // rows
    st = torch.zeros(nb * 4, dtype=torch.int64, device="cuda")
    lib.dc_debug_set_xr_stamps(None)
    for _ in range(3):
        L.check(lib.dc_igemm(p, L.stream_ptr()))
    torch.cuda.synchronize()
    lib.dc_debug_set_xr_stamps(st.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); L.check(lib.dc_igemm(p, L.stream_ptr())); e1.record()
    torch.cuda.synchronize()
    s = st.view(nb, 4).cpu().double()
    ntile = 2 * NH // 128
    med = lambda v: v.median().item()
    print(f"  launch {e0.elapsed_time(e1):.3f} ms (stamped); per workgroup (median cycles of wave 0): K loops {med(s[:,0]):.0f} (of which waiting for slices + barrier "
          f"{med(s[:,3]):.0f}), epilogues {med(s[:,1]):.0f}, N-tile loop {med(s[:,2]):.0f}; per N tile ({ntile}): K {med(s[:,0]) / ntile:.0f} (wait {med(s[:,3]) / ntile:.0f}), epilogue {med(s[:,1]) / ntile:.0f}; "
          f"MFMA issue per N tile and wave: {(3 if K <= 256 else 2) * 4 * (K // 32) * 16} cycles")
