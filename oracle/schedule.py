"""Oracle: log-SNR schedules.  Test infrastructure only (see oracle/__init__.py).

Restates reference `diffusion/diffusion_classifier.py:14-15` (log helper),
`:119-144` (cosine) and `:146-161` (shifted cosine).  Constants are fp64 Python
scalars, the tensor arithmetic is fp32 torch, exactly like the reference.
"""
import math
import torch


def _log(t, eps=1e-20):
    # reference diffusion_classifier.py:14-15
    return torch.log(t.clamp(min=eps))


def logsnr_schedule_cosine(t, noise_d, image_d, logsnr_min=-15, logsnr_max=15):
    # reference diffusion_classifier.py:137-142
    logsnr_max = logsnr_max + math.log(noise_d / image_d)
    logsnr_min = logsnr_min + math.log(noise_d / image_d)
    t_min = math.atan(math.exp(-0.5 * logsnr_max))
    t_max = math.atan(math.exp(-0.5 * logsnr_min))
    return -2 * _log(torch.tan(t_min + t * (t_max - t_min)))


def logsnr_schedule_cosine_shifted(t, noise_d, image_d):
    # reference diffusion_classifier.py:158-159
    return logsnr_schedule_cosine(t, noise_d, image_d) + 2 * math.log(noise_d / image_d)
