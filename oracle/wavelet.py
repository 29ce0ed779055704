"""Oracle: one-level 2-D Haar DWT / inverse.  Test infrastructure only.

Restates reference `utils/wavelet.py:4-35` (`wavelet_dec_2`) and `:37-68`
(`wavelet_enc_2`), whose arithmetic is `pywt.dwt2 / idwt2(…, 'haar')` (pywt is a
third-party dependency, undeclared in requirements.txt; v1.1.1 is what the build
container offers).  For a 2x2 block [[a, b], [c, d]]:
    cA=(a+b+c+d)/2  cH=(a+b-c-d)/2  cV=(a-b+c-d)/2  cD=(a-b-c+d)/2
output channel order 4i+{0,1,2,3} = (cA, cH, cV, cD) of input channel i.
PINNED by tests/golden/dwt_*.npz (pywt 1.1.1 outputs; tolerance 1e-6 abs because pywt
does two separable 1/sqrt(2) passes while this closed form does one /2).
"""
import numpy as np


def haar_dwt2(img):
    """img [C,H,W] float32 (H, W even) -> [4C, H/2, W/2]."""
    img = np.asarray(img, dtype=np.float32)
    a, b = img[:, 0::2, 0::2], img[:, 0::2, 1::2]
    c, d = img[:, 1::2, 0::2], img[:, 1::2, 1::2]
    C, h, w = a.shape
    out = np.empty((4 * C, h, w), dtype=np.float32)
    out[0::4] = (a + b + c + d) * np.float32(0.5)
    out[1::4] = (a + b - c - d) * np.float32(0.5)
    out[2::4] = (a - b + c - d) * np.float32(0.5)
    out[3::4] = (a - b - c + d) * np.float32(0.5)
    return out


def haar_idwt2(wav):
    """wav [4C,h,w] float32 -> [C, 2h, 2w]."""
    wav = np.asarray(wav, dtype=np.float32)
    cA, cH, cV, cD = wav[0::4], wav[1::4], wav[2::4], wav[3::4]
    C, h, w = cA.shape
    out = np.empty((C, 2 * h, 2 * w), dtype=np.float32)
    out[:, 0::2, 0::2] = (cA + cH + cV + cD) * np.float32(0.5)
    out[:, 0::2, 1::2] = (cA + cH - cV - cD) * np.float32(0.5)
    out[:, 1::2, 0::2] = (cA - cH + cV - cD) * np.float32(0.5)
    out[:, 1::2, 1::2] = (cA - cH - cV + cD) * np.float32(0.5)
    return out
