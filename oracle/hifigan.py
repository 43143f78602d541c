"""Oracle: causal pixel-shuffle HiFi-GAN generator.

Restates modules/vocoder/hifigan/hifigan_causal.py:
  CausalConv1d :30-58, CausalUpsampleBlock1 :60-145, CausalUpsampleBlock2 :151-165, CausalPixelShuffle1d :171-189, CausalUpsampleBlock3 :191-212,
  ResBlock1 :217-244, ResBlock2 :246-267, HifiGanGenerator.forward :314-333,
and the numpy wrapper tasks/tts/vocoder_infer/hifigan.py:23-31 (spec2wav).
Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .common import causal_conv1d, to_torch_sd

LRELU_SLOPE = 0.1  # hifigan_causal.py:20


def folded_weight(sd, prefix):
    """weight_norm(Conv1d) weight: g * v / ||v|| per output channel (dim=0), i.e.
    torch._weight_norm(v, g, 0) as applied by hifigan_causal.py:45-46.  Checkpoints with
    weight-norm already removed carry '<prefix>.weight' instead."""
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"]
    return torch._weight_norm(sd[prefix + ".weight_v"], sd[prefix + ".weight_g"], 0)


def _cconv(sd, prefix, x, dilation=1, st=None):
    return causal_conv1d(x, folded_weight(sd, prefix), sd[prefix + ".bias"], dilation, st, prefix)


def pixel_shuffle_1d(x, r):
    """hifigan_causal.py:179-189: (B, C*r, T) -> (B, C, T*r); channel c*r+j at t -> (c, t*r+j)."""
    B, Cr, T = x.shape
    C = Cr // r
    return x.view(B, C, r, T).permute(0, 1, 3, 2).reshape(B, C, T * r)


def resblock1(sd, idx, x, dilations, st=None):
    """hifigan_causal.py:230-238."""
    for d_i, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = _cconv(sd, f"resblocks.{idx}.convs1.{d_i}.conv", xt, d, st)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = _cconv(sd, f"resblocks.{idx}.convs2.{d_i}.conv", xt, 1, st)
        x = x + xt
    return x


def resblock2(sd, idx, x, dilations, st=None):
    """ResBlock2.forward, hifigan_causal.py:255-261."""
    for d_i, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = _cconv(sd, f"resblocks.{idx}.convs.{d_i}.conv", xt, d, st)
        x = x + xt
    return x


def zero_insert(x, stride):
    """CausalUpsampleBlock2.forward, hifigan_causal.py:158-161: x[..., t] lands on sample t*stride, zeros between."""
    B, C, T = x.shape
    up = x.new_zeros(B, C, T * stride)
    up[:, :, ::stride] = x
    return up


def transposed_upsample(sd, prefix, x, stride, k):
    """CausalUpsampleBlock1.forward, hifigan_causal.py:118-141: left-pad k/2 - 1 zero frames, ConvTranspose1d (padding 0,
    output_padding stride - 1; weight_norm over dim 0 = the INPUT channel of a ConvTranspose1d weight [Cin, Cout, k]),
    drop the first (k/2 - 1) * stride + k - 1 samples.  T*stride samples remain.  Despite its name the block looks
    ahead: sample t depends on input frames ceil(t/stride) .. floor((t + k - 1)/stride), none of them in the past."""
    if prefix + ".weight" in sd:
        w = sd[prefix + ".weight"]
    else:
        w = torch._weight_norm(sd[prefix + ".weight_v"], sd[prefix + ".weight_g"], 0)
    pad = k // 2 - 1
    y = F.conv_transpose1d(F.pad(x, (pad, 0)), w, sd[prefix + ".bias"], stride=stride, padding=0, output_padding=stride - 1)
    y = y[:, :, pad * stride + k - 1:]
    assert y.shape[2] == x.shape[2] * stride
    return y


@torch.no_grad()
def generator_forward(sd, hp, mel, st=None, taps=None):
    """HifiGanGenerator.forward (hifigan_causal.py:314-333). mel[B,80,T] -> wav[B,1,T*prod(rates)].
    `st` (dict) switches to stateful streaming (new frames only); `taps` (dict) collects
    per-stage pre-activation tensors for the parity tests."""
    mode = hp.get("upsample", "shuffle")
    assert mode in ("shuffle", "zero", "nn")
    assert mode != "nn" or st is None      # 'nn' looks ahead: whole-utterance (or whole-window) forward only
    rb = resblock1 if str(hp.get("resblock", "1")) == "1" else resblock2
    x = _cconv(sd, "conv_pre.conv", mel, 1, st)
    if taps is not None:
        taps["conv_pre"] = x
    n_rb = len(hp["resblock_kernel_sizes"])
    ridx = 0
    for i, (u, k) in enumerate(zip(hp["upsample_rates"], hp["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, LRELU_SLOPE)
        if mode == "shuffle":
            x = _cconv(sd, f"ups.{i}.conv.conv", x, 1, st)
            x = pixel_shuffle_1d(x, u)
        elif mode == "nn":
            x = transposed_upsample(sd, f"ups.{i}.deconv", x, u, k)
        else:
            x = _cconv(sd, f"ups.{i}.conv.conv", zero_insert(x, u), 1, st)
        if taps is not None:
            taps[f"ups.{i}"] = x
        xs = 0
        for j in range(n_rb):
            xs = xs + rb(sd, ridx, x, hp["resblock_dilation_sizes"][j], st)
            ridx += 1
        x = xs / n_rb
        if taps is not None:
            taps[f"stage.{i}"] = x
    x = F.leaky_relu(x, LRELU_SLOPE)
    x = _cconv(sd, "conv_post.conv", x, 1, st)
    if taps is not None:
        taps["pre_tanh"] = x
    return torch.tanh(x)


def spec2wav(sd, hp, mel_np, st=None):
    """tasks/tts/vocoder_infer/hifigan.py:23-31: numpy [T,80] -> numpy [T*hop]."""
    c = torch.FloatTensor(mel_np).unsqueeze(0).transpose(2, 1)
    y = generator_forward(sd, hp, c, st).view(-1)
    return y.cpu().numpy()


class Generator:
    """Convenience holder: state_dict (numpy or torch) + hparams."""

    def __init__(self, sd, hp):
        self.sd = to_torch_sd(sd)
        self.hp = hp

    def __call__(self, mel, st=None, taps=None):
        if isinstance(mel, np.ndarray):
            mel = torch.from_numpy(mel)
        return generator_forward(self.sd, self.hp, mel.float(), st, taps)
