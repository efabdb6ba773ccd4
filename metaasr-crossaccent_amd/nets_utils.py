"""Host-side helpers mirroring src/nets_utils.py (masks, state_dict cloning)."""
from collections import OrderedDict

import torch


def generate_square_subsequent_mask(sz):
    """0 on/below the diagonal, -inf above (src/nets_utils.py:9-15).  The HIP attention kernel applies this
    mask analytically (kj > qi); the tensor form exists for API parity."""
    return torch.triu(torch.full((sz, sz), float('-inf')), diagonal=1)


def make_bool_pad_mask(lengths):
    """mask[b, t] = t >= lengths[b] (src/nets_utils.py:85-94)."""
    lengths = torch.as_tensor(lengths)
    return torch.arange(int(lengths.max())).unsqueeze(0) >= lengths.view(-1, 1)


def clone(tensor):
    """detach + clone keeping requires_grad and a cloned .grad (src/nets_utils.py:17-27)."""
    out = tensor.detach().clone()
    out.requires_grad = tensor.requires_grad
    if tensor.grad is not None:
        out.grad = clone(tensor.grad)
    return out


def clone_state_dict(state_dict):
    """src/nets_utils.py:62-68"""
    return OrderedDict((k, clone(v)) for k, v in state_dict.items())


def to_device(m, x):
    """src/nets_utils.py:70-83: move x to the device of module/engine m."""
    return x.to(m.device)
