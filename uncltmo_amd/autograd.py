"""Autograd bridge of the generator (backward kernels).  Placeholder until the HIP backward path lands:
training through the generator raises instead of silently falling back to an eager path."""


def generator_image_apply(module, x):
    raise NotImplementedError("uncltmo_amd: the generator's HIP backward kernels are not built yet; run the "
                              "forward under torch.no_grad() (inference / tiler) for now")
