"""Shared helpers of the CPU oracle (test infrastructure; see oracle/__init__.py)."""
import numpy as np
import torch
import torch.nn.functional as F


def to_torch_sd(sd):
    """numpy / torch state_dict -> dict of float32 CPU tensors."""
    out = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v.detach().cpu()
        out[k] = t.float() if t.is_floating_point() else t
    return out


def causal_conv1d(x, w, b, dilation=1, st=None, key=None):
    """Left-padded ("causal") Conv1d on x[B,C,T].

    st is None  -> zero left padding, exactly F.pad(x,(left_pad,0)) + conv
                   (hifigan_causal.py:51-54, diff/net.py:44-47, conv.py:150-152).
    st is a dict -> streaming: the left context is the tail of the previous
                   calls' inputs (zeros at stream start), which is the same
                   arithmetic because every layer on the path is causal
                   (SURVEY.md §0.5)."""
    k = w.shape[-1]
    pad = (k - 1) * dilation
    if st is None:
        xp = F.pad(x, (pad, 0))
    else:
        hist = st.get(key)
        if hist is None:
            hist = x.new_zeros(x.shape[0], x.shape[1], pad)
        xp = torch.cat([hist, x], dim=2)
        st[key] = xp[:, :, xp.shape[2] - pad:].clone() if pad > 0 else hist
    return F.conv1d(xp, w, b, dilation=dilation)


def layer_norm_c(x, w, b, eps=1e-5):
    """modules/commons/layers.py:5-24 with dim=1: LayerNorm over the channel axis of x[B,C,T]."""
    return F.layer_norm(x.transpose(1, -1), (x.shape[1],), w, b, eps).transpose(1, -1)
