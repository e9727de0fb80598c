"""Import paths of the reference's generator package (models/unet_multi_filters): `Unet.UNet` is the video generator,
`Unet_singleFrame.UNet` the image generator -- the same class names the reference's factories import
(utils/model_save_util.py:66-118)."""
