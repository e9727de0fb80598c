"""Constants of the tone-mapping hot path.

Values follow the reference's `utils/params.py` (epsilon :48, epsilon2 :49, input_size :37,
con-operator names :70-81, manualSeed :66).  Only the handful the hot path reads are kept.
"""

epsilon = 1e-08          # added before sqrt in the skip-concat "ssr" operator (unet_parts.py:320)
epsilon2 = 1e-05         # StructLoss variance / divisor floor (struct_loss.py:71-85)
input_size = 256         # the only spatial size the generator accepts (12x12 pos_embed, Unet.py:66)
manualSeed = 999

unet_network = "unet"
torus_network = "torus"

original_unet = "original_unet"
square = "square"
square_root = "square_root"
square_and_square_root = "square_and_square_root"
gamma = "gamma"
square_and_square_root_manual_d = "square_and_square_root_manual_d"

layer_factor_2_operators = [original_unet]
layer_factor_3_operators = [square, square_root, gamma]
layer_factor_4_operators = [square_and_square_root, square_and_square_root_manual_d]


def get_layer_factor(con_operator):
    """Number of concatenated copies of the skip width (model_save_util.py:145-153)."""
    if con_operator in layer_factor_2_operators:
        return 2
    if con_operator in layer_factor_3_operators:
        return 3
    if con_operator in layer_factor_4_operators:
        return 4
    assert 0, "Unsupported con_operator request: {}".format(con_operator)


# keys of the loader dictionaries the trainers read (params.py:83-91)
gray_input_image_key = "input_im"
color_image_key = "color_im"
original_gray_norm_key = "original_gray_norm"
original_gray_key = "original_gray"
gamma_factor = "gamma_factor"
