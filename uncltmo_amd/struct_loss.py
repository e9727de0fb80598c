"""Structural loss, MI355X-native.  Same constructor / forward signature as the reference's
`models/struct_loss.py:StructLoss` (:8-40); forward and backward are one HIP call each way
(csrc/struct_loss.hip) instead of `unfold`-materialised (N,1,H-4,W-4,25) window stacks."""
import torch

from . import _hip


def crop_input_hdr_batch(input_hdr_batch, diffY, diffX):
    """Centre crop (utils/data_loader_util.py:165-172)."""
    b, c, h, w = input_hdr_batch.shape
    th, tw = h - diffY, w - diffX
    i = int(round((h - th) / 2.))
    j = int(round((w - tw) / 2.))
    return input_hdr_batch[:, :, i:i + th, j:j + tw]


class _StructLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fake, hdr, weights):
        lib = _hip.lib()
        n, c, h, w = fake.shape
        f = fake.detach().reshape(n * c, h, w).float().contiguous()
        g = hdr.detach().reshape(n * c, h, w).float().contiguous()
        levels = len(weights)
        wbuf = (_hip.C.c_float * levels)(*[float(x) for x in weights])
        ws = torch.empty(lib.uncl_struct_loss_workspace_bytes(n * c, h, w, levels), dtype=torch.uint8, device=f.device)
        loss = torch.empty(1, dtype=torch.float32, device=f.device)
        need_grad = fake.requires_grad
        grad = torch.empty_like(f) if need_grad else None
        _hip.check(lib.uncl_struct_loss(_hip.ptr(f), _hip.ptr(g), wbuf, levels, loss.data_ptr(),
                                        grad.data_ptr() if need_grad else None, None, n * c, h, w, ws.data_ptr(),
                                        _hip.stream_ptr()), "uncl_struct_loss")
        ctx.grad = grad
        ctx.shape = fake.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        g = ctx.grad
        if g is None:
            return None, None, None
        return (g * gout).reshape(ctx.shape), None, None


class StructLoss(torch.nn.Module):
    def __init__(self, pyramid_weight_list, window_size=5, pyramid_pow=False, use_c3=False,
                 struct_method="gamma_struct", crop_input=True, final_shape_addition=0):
        super().__init__()
        if window_size != 5:
            raise NotImplementedError("the HIP struct loss is built for the published 5x5 window")
        self.window_size = window_size
        self.final_shape_addition = final_shape_addition
        self.crop_input = crop_input
        self.pyramid_weight_list = pyramid_weight_list
        self.pyramid_pow = pyramid_pow
        self.use_c3 = use_c3
        self.struct_method = struct_method

    def forward(self, fake, hdr_input_original_gray_norm, hdr_input, pyramid_weight_list):
        # like the reference, `hdr_input_original_gray_norm` is accepted and ignored (struct_loss.py:23-40)
        if not fake.is_cuda:
            raise _hip.HipError("StructLoss needs CUDA(HIP) tensors; there is no CPU path")
        if self.crop_input:
            hdr_input = crop_input_hdr_batch(hdr_input, self.final_shape_addition, self.final_shape_addition)
        weights = [float(w) for w in pyramid_weight_list]
        return _StructLossFn.apply(fake, hdr_input, weights)
