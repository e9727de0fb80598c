"""ORACLE helper (test infrastructure): plain-tensor state dicts built from the deterministic generator."""
from uncltmo_amd import state_spec, synth

from .generator import sincos_relative_pos


def synth_state(spec, salt):
    sd = {}
    import torch
    for k, shape, kind in spec:
        if kind == "bn_count":                      # num_batches_tracked: fill_state_dict leaves the module's zero
            sd[k] = torch.zeros((), dtype=torch.long)
        else:
            sd[k] = sincos_relative_pos() if kind == "buffer" else synth.synth_tensor(k, shape, salt)
    return sd


def generator_state(salt="g0"):
    return synth_state(state_spec.generator_spec(), salt)


def discriminator_state(salt="d0"):
    return synth_state(state_spec.simple_d_spec(), salt)
