"""models/unet_multi_filters/Unet_singleFrame.py: the image generator under the reference's class name."""
from ..generator import UNet  # noqa: F401
