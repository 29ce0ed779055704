"""CPU oracle for the diffusion-classifier scoring path.  TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-eager fp32 restatement of the reference hot path
(`diffusion/diffusion_classifier.py:657-725` + the diffusers backbones that
`nets/unet.py` / `nets/dit.py` wrap + `utils/wavelet.py`).  It exists to CHECK the
HIP path; it is never the thing shipped or measured:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
    import it;
  * nothing under `diffusion-classifier_amd/` imports it, and the product path raises
    when the HIP library is missing instead of falling back to this code.

Pinning status
  * scoring loop (`classify`, schedules, `diffuse`, encoders): PINNED — golden vectors
    captured in the build container by importing the reference's own `classify`
    (tools/capture_goldens.py -> tests/golden/classify_*.npz).
  * Haar DWT: PINNED — golden vectors from pywt 1.1.1 (tests/golden/dwt_*.npz).
  * backbone arithmetic (UNet2DConditionModel / DiTTransformer2DModel): **parity
    unpinned** — it lives in third-party `diffusers==0.31.0` (reference
    requirements.txt:9) which is neither under /root/reference nor installed, and the
    reference holds no test/fixture at that boundary.  The restatement follows the
    published diffusers 0.31.0 architecture and keeps its state-dict key names so a
    real checkpoint can close the gap later.
"""
from .schedule import logsnr_schedule_cosine, logsnr_schedule_cosine_shifted  # noqa: F401
from .unet import OracleUNetCondition2D  # noqa: F401
from .dit import OracleDiT  # noqa: F401
from .classifier import OracleDiffusionClassifier, AttrBag  # noqa: F401
from .wavelet import haar_dwt2, haar_idwt2  # noqa: F401
