"""Shared helpers for the test-suite (fixtures loader, seeded weights)."""
import os

import numpy as np
import torch

from oracle import dit_oracle as mo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: d[k] for k in d.files}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def shape_from(fx) -> mo.DitShape:
    depth, hidden, heads, ncls = (int(v) for v in fx["shape"])
    return mo.DitShape(depth=depth, hidden=hidden, heads=heads, num_classes=ncls)


def weights_for(fx):
    """Rebuild the fixture's seeded weights and check the pinned checksum."""
    shape = shape_from(fx)
    rough = bool(fx["rough"]) if "rough" in fx else False
    sd = (mo.seeded_state_dict(shape, int(fx["wseed"]), pos_gain=1.0, mod_std=0.2) if rough
          else mo.seeded_state_dict(shape, int(fx["wseed"])))
    wsum = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(wsum - float(fx["wsum"])) <= 1e-9 * abs(wsum), "seeded weights differ from the fixture's"
    return shape, sd


def maxdiff(a, b):
    return (torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max().item()
