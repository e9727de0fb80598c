"""Factories with the reference's signatures (utils/model_save_util.py:41-118, 145-153, 177-201) that build the
MI355X-native modules.  `weights_init_xavier` is folded into the modules' reset_parameters()."""
import torch

from . import params
from .discriminator import NLayerDiscriminator, SimpleDiscriminator
from .generator import UNet, UNetVideo


def get_layer_factor(con_operator):
    return params.get_layer_factor(con_operator)


def _create_g(cls, model, device_, is_checkpoint, input_dim_, last_layer, filters, con_operator, unet_depth_, add_frame,
              unet_norm, stretch_g, activation, use_xaviar, output_dim, g_doubleConvTranspose, bilinear, padding,
              convtranspose_kernel, up_mode, compute_dtype):
    layer_factor = get_layer_factor(con_operator)
    if model != params.unet_network:
        assert 0, "Unsupported g model request: {}".format(model)
    net = cls(input_dim_, output_dim, last_layer, depth=unet_depth_, layer_factor=layer_factor, con_operator=con_operator,
              filters=filters, bilinear=bilinear, network=model, dilation=0, to_crop=add_frame, unet_norm=unet_norm,
              stretch_g=stretch_g, activation=activation, doubleConvTranspose=g_doubleConvTranspose, padding_mode=padding,
              convtranspose_kernel=convtranspose_kernel, up_mode=up_mode, compute_dtype=compute_dtype)
    return net.to(device_)


def create_G_net(model, device_, is_checkpoint, input_dim_, last_layer, filters, con_operator, unet_depth_, add_frame,
                 unet_norm, stretch_g, activation, use_xaviar, output_dim, g_doubleConvTranspose, bilinear, padding,
                 convtranspose_kernel, up_mode, compute_dtype="bf16"):
    """Video generator (reference: model_save_util.py:66-81)."""
    return _create_g(UNetVideo, model, device_, is_checkpoint, input_dim_, last_layer, filters, con_operator, unet_depth_,
                     add_frame, unet_norm, stretch_g, activation, use_xaviar, output_dim, g_doubleConvTranspose, bilinear,
                     padding, convtranspose_kernel, up_mode, compute_dtype)


def create_G_net2(model, device_, is_checkpoint, input_dim_, last_layer, filters, con_operator, unet_depth_, add_frame,
                  unet_norm, stretch_g, activation, use_xaviar, output_dim, g_doubleConvTranspose, bilinear, padding,
                  convtranspose_kernel, up_mode, compute_dtype="bf16"):
    """Image generator (reference: model_save_util.py:83-98)."""
    return _create_g(UNet, model, device_, is_checkpoint, input_dim_, last_layer, filters, con_operator, unet_depth_,
                     add_frame, unet_norm, stretch_g, activation, use_xaviar, output_dim, g_doubleConvTranspose, bilinear,
                     padding, convtranspose_kernel, up_mode, compute_dtype)


def create_D_net(input_dim_, down_dim, device_, is_checkpoint, norm, use_xaviar, d_model, d_nlayers, last_activation, num_D,
                 d_fully_connected, simpleD_maxpool, d_padding):
    """reference: model_save_util.py:101-118; `simpleD` (the published trainer's D, forward + backward) and `patchD`
    (forward-parity PatchGAN) are on the HIP path."""
    if d_model == "patchD":
        net = NLayerDiscriminator(input_dim_, ndf=down_dim, n_layers=d_nlayers, norm_layer=norm, last_activation=last_activation)
    elif d_model == "simpleD":
        net = SimpleDiscriminator(params.input_size, input_dim_, down_dim, norm, last_activation, simpleD_maxpool, d_padding)
    else:
        assert 0, "Unsupported d model request: {}".format(d_model)
    if use_xaviar:
        for m in net.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear)):
                torch.nn.init.xavier_normal_(m.weight, gain=2.0 ** 0.5)
                if m.bias is not None:
                    torch.nn.init.constant_(m.bias, 0)
    return net.to(device_)


def load_g_model(model_params, device, net_path, compute_dtype="bf16"):
    """reference: model_save_util.py:177-201 (strips a DataParallel 'module.' prefix, eval mode)."""
    g = create_G_net2(model_params["model"], device, True, model_params["input_dim"], model_params["last_layer"],
                      model_params["filters"], model_params["con_operator"], model_params["depth"], model_params["add_frame"],
                      model_params["unet_norm"], model_params["stretch_g"], "relu", False, 1,
                      model_params["g_doubleConvTranspose"], model_params["bilinear"], model_params["padding"],
                      model_params["convtranspose_kernel"], model_params["up_mode"], compute_dtype)
    sd = torch.load(net_path, map_location=device)["modelG_state_dict"]
    if "module" in list(sd.keys())[0]:
        sd = {k[7:]: v for k, v in sd.items()}
    g.load_state_dict(sd)
    g.to(device)
    g.eval()
    return g


def save_model(path, epoch, epoch_iter, output_dir, netG, optimizerG, netD, optimizerD):
    """reference: model_save_util.py:121-131 -- same file name, same five keys, so a checkpoint written here loads in the
    reference (`load_g_model`, :177-201) and the other way round (state_dict keys / shapes are the reference's)."""
    import os
    path = os.path.join(output_dir, path, "net_epoch" + str(epoch) + "_iter" + str(epoch_iter) + ".pth")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({"epoch": epoch,
                "modelD_state_dict": netD.state_dict(), "modelG_state_dict": netG.state_dict(),
                "optimizerD_state_dict": optimizerD.state_dict(), "optimizerG_state_dict": optimizerG.state_dict()}, path)
    return path


def load_checkpoint(net_path, device, netG, optimizerG=None, netD=None, optimizerD=None):
    """Resume: the inverse of save_model (the reference's trainers restore the same keys, GanTrainerImg.py:484-493, where
    the epoch counter restarts from the checkpoint's `epoch`).  Returns the stored epoch."""
    ck = torch.load(net_path, map_location=device)
    strip = lambda sd: {k[7:]: v for k, v in sd.items()} if sd and "module" in list(sd.keys())[0] else sd
    netG.load_state_dict(strip(ck["modelG_state_dict"]))
    if netD is not None and "modelD_state_dict" in ck:
        netD.load_state_dict(strip(ck["modelD_state_dict"]))
    if optimizerG is not None and "optimizerG_state_dict" in ck:
        optimizerG.load_state_dict(ck["optimizerG_state_dict"])
    if optimizerD is not None and "optimizerD_state_dict" in ck:
        optimizerD.load_state_dict(ck["optimizerD_state_dict"])
    return ck.get("epoch", 0)
