"""nn.Module skeletons that carry the reference's parameter tree (same state_dict keys and shapes,
conan_amd/specs.py) so reference checkpoints load unchanged; compute is delegated to libconan_hip.so."""
import numpy as np
import torch
from torch import nn

from .. import synth


class ParamTree(nn.Module):
    """Nested module hierarchy generated from dotted state_dict keys."""

    def __init__(self):
        super().__init__()

    def _add(self, dotted, tensor, buffer=False):
        parts = dotted.split(".")
        node = self
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, ParamTree())
            node = node._modules[p]
        if buffer:
            node.register_buffer(parts[-1], tensor)
        else:
            node.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


def build_tree(module, spec, buffers=(), seed=0):
    """Registers every entry of `spec` on `module` with deterministic synthetic ("random-init") values."""
    sd = synth.state_dict(spec, seed)
    for key, shape in spec.items():
        t = torch.from_numpy(np.ascontiguousarray(sd[key])).reshape(tuple(shape))
        ParamTree._add(module, key, t, buffer=key in buffers)


def host_state_dict(module):
    return {k: v.detach().cpu().float().numpy() for k, v in module.state_dict().items() if v.is_floating_point()}
