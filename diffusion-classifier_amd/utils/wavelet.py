"""MI355X-native drop-in for reference `utils/wavelet.py`.

`wavelet_dec_2(images[C,H,W]) -> [4C,H/2,W/2]` (reference :4-35) and
`wavelet_enc_2(wavelet_images[4C,h,w]) -> [C,2h,2w]` (reference :37-68): one-level 2-D Haar
DWT / inverse per channel, output channel order 4i+{0,1,2,3} = (cA, cH, cV, cD).  The
reference loops over channels through pywt on the host; here ONE coalesced HIP kernel
(`dc_haar_dwt2` / `dc_haar_idwt2`) does all channels (and, with a 4-D input, a whole batch).
Like the reference, the result lives on `images.device`; CPU tensors are moved to the GPU,
transformed there and moved back.  No CPU fallback.
"""
import torch

from .. import _lib as L


def _prep(t):
    lib = L.require_gpu()
    dev_in = t.device
    x = t.detach().to("cuda" if not t.is_cuda else t.device, torch.float32).contiguous()
    squeeze = x.dim() == 3
    if squeeze:
        x = x.unsqueeze(0)
    assert x.dim() == 4, "expected [C,H,W] or [N,C,H,W]"
    return lib, dev_in, x, squeeze


def wavelet_dec_2(images, scale: float = 1.0):
    lib, dev_in, x, squeeze = _prep(images)
    N, Cc, H, W = x.shape
    out = torch.empty(N, 4 * Cc, H // 2, W // 2, dtype=torch.float32, device=x.device)
    L.check(lib.dc_haar_dwt2(x.data_ptr(), out.data_ptr(), N, Cc, H, W, float(scale), L.stream_ptr()), "dc_haar_dwt2")
    out = out[0] if squeeze else out
    return out.to(dev_in)


def wavelet_enc_2(wavelet_images, scale: float = 1.0):
    lib, dev_in, x, squeeze = _prep(wavelet_images)
    N, C4, h, w = x.shape
    assert C4 % 4 == 0
    out = torch.empty(N, C4 // 4, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
    L.check(lib.dc_haar_idwt2(x.data_ptr(), out.data_ptr(), N, C4 // 4, h, w, float(scale), L.stream_ptr()), "dc_haar_idwt2")
    out = out[0] if squeeze else out
    return out.to(dev_in)
