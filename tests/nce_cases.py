"""Inputs of the nce() cases with several positives / negatives (GanTrainerImg.py:410-439), shared by tests/golden/make_golden.py
(which runs them through the reference) and by the oracle / GPU tests."""
import numpy as np
import torch

from uncltmo_amd import synth

# tag, anchor shape, positives, negatives, negatives are single rows repeated over the batch (as infoNCE2 builds its own, :401-402), k, c
NCE_LISTS_CASES = [("nl_map", (3, 32, 8, 8), 2, 3, False, 1, 1e-2), ("nl_d", (4, 2, 1, 1), 3, 2, False, 1e3, 2),
                   ("nl_one_pos", (3, 16, 8, 8), 1, 4, True, 1, 1e-2), ("nl_one_neg", (2, 8, 8, 8), 3, 1, False, 1, 1e-2)]


def nce_lists_inputs(tag, shape, n_pos, n_neg, shared_neg=False):
    n = int(np.prod(shape))
    an = torch.tensor(synth.hash_uniform(tag + "a", n)).reshape(shape)
    pos = [torch.tensor(synth.hash_uniform(tag + "p%d" % i, n)).reshape(shape) for i in range(n_pos)]
    if shared_neg:
        m = n // shape[0]
        neg = [torch.tensor(synth.hash_uniform(tag + "n%d" % i, m)).reshape((1,) + tuple(shape[1:])) for i in range(n_neg)]
    else:
        neg = [torch.tensor(synth.hash_uniform(tag + "n%d" % i, n)).reshape(shape) for i in range(n_neg)]
    return an, pos, neg
