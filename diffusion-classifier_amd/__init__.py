"""diffusion-classifier_amd — MI355X (gfx950) native scoring path of faverogian/diffusion-classifier.

Import as `diffusion_classifier_amd` (the repo-root shim maps that name onto this directory,
whose on-disk name carries the reference's hyphen).  Layout mirrors the reference so that
    from nets.unet import UNetCondition2D            ->  from diffusion_classifier_amd.nets.unet import UNetCondition2D
    from nets.dit import DiT                         ->  from diffusion_classifier_amd.nets.dit import DiT
    from diffusion.diffusion_classifier import DiffusionClassifier
                                                     ->  from diffusion_classifier_amd.diffusion.diffusion_classifier import DiffusionClassifier
    from utils.wavelet import wavelet_dec_2          ->  from diffusion_classifier_amd.utils.wavelet import wavelet_dec_2
All arithmetic of the scoring path runs in `libdcamd.so` (HIP kernels + C-ABI, csrc/, include/dcamd.h).
"""
from . import _lib  # noqa: F401
from .nets.unet import UNetCondition2D, UNet2D  # noqa: F401
from .nets.dit import DiT  # noqa: F401
from .diffusion.diffusion_classifier import DiffusionClassifier  # noqa: F401
from .utils.wavelet import wavelet_dec_2, wavelet_enc_2  # noqa: F401


class Config:
    """Attribute bag whose missing keys read as None (reference experiments/cifar10/inference.py:24-38)."""

    def __init__(self, **kw):
        self.__dict__["_d"] = dict(kw)

    def __getattr__(self, k):
        return self.__dict__["_d"].get(k)

    def __setattr__(self, k, v):
        self.__dict__["_d"][k] = v


# architectures BASELINE.json names (reference file:line in each comment)
def cifar10_unet_kwargs():
    # experiments/cifar10/inference.py:94-116
    return dict(sample_size=32, in_channels=3, out_channels=3, layers_per_block=2, block_out_channels=(128, 128, 256, 512),
                down_block_types=("DownBlock2D", "DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
                up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D", "UpBlock2D"),
                mid_block_type="UNetMidBlock2DCrossAttn", encoder_hid_dim=128, encoder_hid_dim_type="text_proj",
                cross_attention_dim=128)


def small_unet_kwargs():
    # BASELINE config 1 "small": not defined by the reference; channel counts are multiples of 64 so
    # the bf16 MFMA path (K granule 64) accepts it as well as the f32 one.
    return dict(sample_size=32, in_channels=3, out_channels=3, layers_per_block=1, block_out_channels=(64, 128),
                down_block_types=("DownBlock2D", "CrossAttnDownBlock2D"), up_block_types=("CrossAttnUpBlock2D", "UpBlock2D"),
                mid_block_type="UNetMidBlock2DCrossAttn", encoder_hid_dim=64, encoder_hid_dim_type="text_proj",
                cross_attention_dim=64)


def chexpert_dwt_unet_kwargs():
    # models/chexpert-256-unet-dwt-healthysick.py:4-28
    return dict(sample_size=128, in_channels=12, out_channels=12, layers_per_block=2,
                block_out_channels=(128, 128, 256, 512, 1024),
                down_block_types=("DownBlock2D", "DownBlock2D", "DownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D", "UpBlock2D", "UpBlock2D"),
                mid_block_type="UNetMidBlock2DCrossAttn", encoder_hid_dim=512, encoder_hid_dim_type="text_proj",
                cross_attention_dim=512)


def ipmsa5_unet_kwargs():
    # models/ipmsa-5-unet.py:4-30
    return dict(sample_size=256, in_channels=10, out_channels=10, layers_per_block=(2, 2, 2, 2, 4, 2),
                block_out_channels=(128, 128, 256, 512, 512, 1024),
                down_block_types=("DownBlock2D",) * 4 + ("CrossAttnDownBlock2D",) * 2,
                up_block_types=("CrossAttnUpBlock2D",) * 2 + ("UpBlock2D",) * 4,
                mid_block_type="UNetMidBlock2DCrossAttn", encoder_hid_dim=512, encoder_hid_dim_type="text_proj",
                cross_attention_dim=512)


def chexpert_dit_b4_kwargs(wavelet_transform=True):
    # models/chexpert-256-dit-b4.py:4-21 (image_size 256, 3 channels, patch 4)
    ch, size = (12, 128) if wavelet_transform else (3, 256)
    return dict(num_attention_heads=12, attention_head_dim=64, in_channels=ch, out_channels=ch, num_layers=12,
                sample_size=size, patch_size=4, num_embeds_ada_norm=1000, norm_eps=1e-5)
