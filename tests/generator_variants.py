"""The generator configurations outside the published one that the HIP path runs on the published kernels (tests/golden/make_golden.py
capture_generator_variants): (tag, con_operator, bilinear, up_mode)."""
GENERATOR_VARIANTS = [("original_unet", "original_unet", 0, 0), ("square", "square", 0, 0), ("square_root", "square_root", 0, 0),
                      ("bilinear", "square_and_square_root", 1, 0), ("square_bilinear", "square", 1, 0),
                      ("up_mode", "square_and_square_root", 0, 1)]
