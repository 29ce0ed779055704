from .unet import UNetCondition2D, UNet2D  # noqa: F401
from .dit import DiT  # noqa: F401
