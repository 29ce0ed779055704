#!/opt/conda/bin/python3.9
"""Capture Haar DWT goldens from pywt (the arithmetic behind reference utils/wavelet.py:27,62).
Runs ONLY in the build container under /opt/conda/bin/python3.9 (pywt 1.1.1, no torch there);
channel stacking follows reference utils/wavelet.py:22-33.  Output: tests/golden/dwt_*.npz."""
import os
import numpy as np
import pywt

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def dec(img):
    C, H, W = img.shape
    out = np.zeros((4 * C, H // 2, W // 2), dtype=np.float32)
    for i in range(C):
        cA, (cH, cV, cD) = pywt.dwt2(img[i], "haar")
        out[4 * i], out[4 * i + 1], out[4 * i + 2], out[4 * i + 3] = cA, cH, cV, cD
    return out


def enc(w):
    C = w.shape[0] // 4
    out = np.zeros((C, w.shape[1] * 2, w.shape[2] * 2), dtype=np.float32)
    for i in range(C):
        out[i] = pywt.idwt2((w[4 * i], (w[4 * i + 1], w[4 * i + 2], w[4 * i + 3])), "haar")
    return out


rng = np.random.default_rng(0)
cases = {
    "rand_3x64x64": (rng.random((3, 64, 64), dtype=np.float32) * 2 - 1),
    "rand_10x32x48": (rng.random((10, 32, 48), dtype=np.float32) * 2 - 1),
    "ramp_1x4x4": np.arange(16, dtype=np.float32).reshape(1, 4, 4),
}
arrs = {}
for k, x in cases.items():
    w = dec(x)
    assert w.dtype == np.float32
    arrs[k + ".x"], arrs[k + ".dec"], arrs[k + ".enc_of_dec"] = x, w, enc(w)
np.savez_compressed(os.path.join(OUT, "dwt_pywt.npz"), **arrs)
print({k: v.shape for k, v in arrs.items()}, "pywt", pywt.__version__)
