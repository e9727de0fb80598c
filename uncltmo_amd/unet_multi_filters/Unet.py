"""models/unet_multi_filters/Unet.py: the recurrent (video) generator under the reference's class name."""
from ..generator import UNetVideo as UNet  # noqa: F401
