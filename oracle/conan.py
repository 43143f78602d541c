"""Oracle: Conan style-adaptive causal mel decoder (infer=True path of
egs/conan_emformer.yaml: style=true, f0_gen='orig', decoder_type='conv').

Restates modules/Conan/Conan.py:115-351,584-589 and its building blocks
  modules/commons/conv.py:49-125 (ConvBlocks/ResidualBlock), :127-264 (Causal*),
  modules/Conan/prosody_util.py:17-94 (VQ), :96-161 (aligner), :173-200 (LocalStyleAdaptor),
  :299-336 (local ConvBlocks), modules/commons/wavenet.py:14-97 (WN),
  modules/commons/nar_tts_modules.py:103-146 (PitchPredictor),
  modules/commons/transformer.py:13-72 (sinusoidal positions),
  utils/nn/seq_utils.py:6-18, :307-325, utils/audio/pitch/utils.py:17-28, :71-82.
Test infrastructure only (see oracle/__init__.py).

Two call shapes share one arithmetic:
  conan_forward(...)                  -- whole-prefix forward, like the reference loop
  style_pass(...) + decode_frames(...) -- per-utterance style cache + stateful frames
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .common import causal_conv1d, layer_norm_c, to_torch_sd

# ----------------------------------------------------------------------------- pitch utils


def denorm_f0(f0, uv):
    """utils/audio/pitch/utils.py:71-82 (pitch_norm='log', min=50, max=900)."""
    f0 = 2 ** f0
    f0 = f0.clamp(min=50, max=900)
    if uv is not None:
        f0[uv > 0] = 0
    return f0


def f0_to_coarse(f0, f0_bin=256, f0_max=900.0, f0_min=50.0):
    """utils/audio/pitch/utils.py:17-28 (torch branch)."""
    f0_mel_min = 1127 * np.log(1 + f0_min / 700)
    f0_mel_max = 1127 * np.log(1 + f0_max / 700)
    f0_mel = 1127 * (1 + f0 / 700).log()
    f0_mel[f0_mel > 0] = (f0_mel[f0_mel > 0] - f0_mel_min) * (f0_bin - 2) / (f0_mel_max - f0_mel_min) + 1
    f0_mel[f0_mel <= 1] = 1
    f0_mel[f0_mel > f0_bin - 1] = f0_bin - 1
    f0_coarse = (f0_mel + 0.5).long()
    assert f0_coarse.max() <= 255 and f0_coarse.min() >= 1
    return f0_coarse


# ----------------------------------------------------------------------------- conv blocks


def _gelu(x):
    return F.gelu(x)  # nn.GELU() default: exact erf form (conv.py:38)


def conv_blocks(sd, prefix, x, nonpadding, k, n_blocks, n_in_block, post_k):
    """Non-causal ConvBlocks on x[B,C,T] (conv.py:84-125 / prosody_util.py:299-336).
    Sequential = [LN(dim=1,eps 1e-5), Conv(k,'same'), *k^-0.5, GELU, Conv1x1]."""
    for b in range(n_blocks):
        np_blk = (x.abs().sum(1) > 0).float()[:, None, :]          # conv.py:73
        for j in range(n_in_block):
            p = f"{prefix}.res_blocks.{b}.blocks.{j}"
            h = layer_norm_c(x, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], 1e-5)
            h = F.conv1d(h, sd[f"{p}.1.weight"], sd[f"{p}.1.bias"], padding=(k - 1) // 2)
            h = h * k ** -0.5
            h = _gelu(h)
            h = F.conv1d(h, sd[f"{p}.4.weight"], sd[f"{p}.4.bias"])
            x = x + h
            x = x * np_blk
    x = x * nonpadding
    x = layer_norm_c(x, sd[f"{prefix}.last_norm.weight"], sd[f"{prefix}.last_norm.bias"], 1e-5) * nonpadding
    x = F.conv1d(x, sd[f"{prefix}.post_net1.weight"], sd[f"{prefix}.post_net1.bias"], padding=post_k // 2) * nonpadding
    return x


def causal_conv_blocks(sd, prefix, x, k, dilations, n_in_block, post_k, st=None):
    """CausalConvBlocks on x[B,C,T] (conv.py:127-264).
    Sequential = [LN, left-pad d*(k-1), Conv(k), *k^-0.5, GELU, Conv1x1]; the pad comes AFTER the
    norm, so streamed history is post-LN activations with exact zeros before stream start."""
    nonpadding = (x.abs().sum(1) > 0).float()[:, None, :]          # conv.py:254
    for b, d in enumerate(dilations):
        np_blk = (x.abs().sum(1) > 0).float()[:, None, :]          # conv.py:172
        for j in range(n_in_block):
            p = f"{prefix}.res_blocks.{b}.blocks.{j}"
            h = layer_norm_c(x, sd[f"{p}.0.weight"], sd[f"{p}.0.bias"], 1e-5)
            h = causal_conv1d(h, sd[f"{p}.2.weight"], sd[f"{p}.2.bias"], d, st, p + ".2")
            h = h * k ** -0.5
            h = _gelu(h)
            h = F.conv1d(h, sd[f"{p}.5.weight"], sd[f"{p}.5.bias"])
            x = (x + h) * np_blk
    x = x * nonpadding
    x = layer_norm_c(x, sd[f"{prefix}.last_norm.weight"], sd[f"{prefix}.last_norm.bias"], 1e-5) * nonpadding
    x = causal_conv1d(x, sd[f"{prefix}.post_net1.1.weight"], sd[f"{prefix}.post_net1.1.bias"], 1, st,
                      prefix + ".post_net1.1") * nonpadding
    return x


# ----------------------------------------------------------------------------- style side


def encode_spk_embed(sd, hp, x):
    """Conan.encode_spk_embed + temporal_avg_pool (Conan.py:200-219). x[B,80,Tr] -> [B,C,1]."""
    in_nonpadding = (x.abs().sum(dim=-2) > 0).float()[:, None, :]
    xg = F.conv1d(x, sd["global_conv_in.weight"], sd["global_conv_in.bias"]) * in_nonpadding
    z = conv_blocks(sd, "global_encoder", xg, in_nonpadding, 31, 5, 2, 3) * in_nonpadding
    mask = in_nonpadding == 0
    len_ = (~mask).sum(dim=-1).unsqueeze(-1)
    z = z.masked_fill(mask, 0)
    z = z.sum(dim=-1).unsqueeze(-1)
    return torch.div(z, len_)


def _wn_weight(sd, p):
    if p + ".weight" in sd:
        return sd[p + ".weight"]
    return torch._weight_norm(sd[p + ".weight_v"], sd[p + ".weight_g"], 0)


def wavenet(sd, prefix, x, nonpadding, hidden=80, n_layers=4):
    """WN.forward (wavenet.py:56-89), kernel 3, dilation_rate 1, no conditioning."""
    output = torch.zeros_like(x)
    for i in range(n_layers):
        x_in = F.conv1d(x, _wn_weight(sd, f"{prefix}.in_layers.{i}"), sd[f"{prefix}.in_layers.{i}.bias"], padding=1)
        acts = torch.tanh(x_in[:, :hidden]) * torch.sigmoid(x_in[:, hidden:])   # wavenet.py:5-11
        rs = F.conv1d(acts, _wn_weight(sd, f"{prefix}.res_skip_layers.{i}"), sd[f"{prefix}.res_skip_layers.{i}.bias"])
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * nonpadding
            output = output + rs[:, hidden:]
        else:
            output = output + rs
    return output * nonpadding


def group_hidden_by_segs(h, seg_ids, max_len):
    """utils/nn/seq_utils.py:307-325."""
    B, T, H = h.shape
    h_g = h.new_zeros([B, max_len + 1, H]).scatter_add_(1, seg_ids[:, :, None].repeat([1, 1, H]), h)
    cnt = h.new_zeros([B, max_len + 1]).scatter_add_(1, seg_ids, h.new_ones(h.shape[:2])).contiguous()
    h_g, cnt = h_g[:, 1:], cnt[:, 1:]
    return h_g / torch.clamp(cnt[:, :, None], min=1), cnt


def vq_encode(embedding, x):
    """VQEmbeddingEMA.encode/forward in eval (prosody_util.py:34-46, :88). Returns the
    straight-through value x + (q - x) (NOT bit-identical to q) and the indices."""
    B, T, D = x.shape
    x_flat = x.reshape(-1, D)
    distances = torch.addmm(torch.sum(embedding ** 2, dim=1) + torch.sum(x_flat ** 2, dim=1, keepdim=True),
                            x_flat, embedding.t(), alpha=-2.0, beta=1.0)
    indices = torch.argmin(distances.float(), dim=-1)
    quantized = F.embedding(indices, embedding).view_as(x)
    return x + (quantized - x), indices.reshape(B, T), distances


def sinusoid_table(n, dim, padding_idx=0):
    """SinusoidalPositionalEmbedding.get_embedding (transformer.py:30-47)."""
    half = dim // 2
    e = math.log(10000) / (half - 1)
    e = torch.exp(torch.arange(half, dtype=torch.float) * -e)
    e = torch.arange(n, dtype=torch.float).unsqueeze(1) * e.unsqueeze(0)
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=1).view(n, -1)
    if dim % 2 == 1:
        e = torch.cat([e, torch.zeros(n, 1)], dim=1)
    e[padding_idx, :] = 0
    return e


def make_positions(t, padding_idx=0):
    """utils/nn/seq_utils.py:6-18."""
    mask = t.ne(padding_idx).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + padding_idx


@torch.no_grad()
def style_pass(sd, hp, ref):
    """Everything of Conan.forward that depends only on the reference mel ref[B,Tr,80]
    (Conan.py:157-159 and the ref side of get_prosody :221-245).  Returns a cache dict."""
    H = hp["hidden_size"]
    B, Tr, _ = ref.shape
    style_embed = encode_spk_embed(sd, hp, ref.transpose(1, 2)).transpose(1, 2)      # [B,1,H]
    ids = (torch.arange(Tr) // 4 + 1).unsqueeze(0).expand(B, -1)                      # Conan.py:227-230
    # LocalStyleAdaptor.forward (prosody_util.py:183-200)
    padding_mask = ref[:, :, 0].eq(0)
    wn_np = (~padding_mask).unsqueeze(1).repeat([1, 80, 1]).float()
    r = wavenet(sd, "prosody_extractor.wavenet", ref.transpose(1, 2), wn_np).transpose(1, 2)
    ref_ph, _ = group_hidden_by_segs(r, ids, int(torch.max(ids)))
    xe = ref_ph.transpose(1, 2)
    np_e = (xe.abs().sum(1) > 0).float()[:, None, :]
    enc = conv_blocks(sd, "prosody_extractor.encoder", xe, np_e, 5, 5, 2, 3).transpose(1, 2)   # [B,S,H]
    z, vq_ids, dist = vq_encode(sd["prosody_extractor.vqvae.embedding"], enc)
    # positions + l1 (Conan.py:244-245)
    pos = make_positions(z[:, :, 0], 0)
    table = sinusoid_table(max(2000 + 1, int(pos.max()) + 1), H, 0)
    positions = table.index_select(0, pos.view(-1)).view(B, pos.shape[1], -1)
    tokens = F.linear(torch.cat([z, positions], dim=-1), sd["l1.weight"], sd["l1.bias"])
    key_padding_mask = tokens[:, :, 0].eq(0)                                          # Conan.py:249
    return {"style_embed": style_embed, "tokens": tokens, "key_padding_mask": key_padding_mask,
            "vq_ids": vq_ids, "ref_upsample": ids, "enc": enc, "vq_dist": dist, "wn_out": r,
            "ref_ph": ref_ph, "z": z}


def cross_atten_layer(sd, p, src, mem, key_padding_mask, nhead=2):
    """CrossAttenLayer.forward with forcing=False (prosody_util.py:119-126); sequence-first."""
    H = src.shape[-1]
    src2, attn = F.multi_head_attention_forward(
        src, mem, mem, H, nhead,
        sd[f"{p}.multihead_attn.in_proj_weight"], sd[f"{p}.multihead_attn.in_proj_bias"],
        None, None, False, 0.0,
        sd[f"{p}.multihead_attn.out_proj.weight"], sd[f"{p}.multihead_attn.out_proj.bias"],
        training=False, key_padding_mask=key_padding_mask, need_weights=True)
    src = src + src2
    src = F.layer_norm(src, (H,), sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"], 1e-5)
    src2 = F.linear(F.relu(F.linear(src, sd[f"{p}.linear1.weight"], sd[f"{p}.linear1.bias"])),
                    sd[f"{p}.linear2.weight"], sd[f"{p}.linear2.bias"])
    src = src + src2
    src = F.layer_norm(src, (H,), sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"], 1e-5)
    return src, attn


def pitch_predictor(sd, prefix, x, n_layers=5, st=None):
    """PitchPredictor.forward (nar_tts_modules.py:130-146): x[B,T,H] -> [B,T,2]."""
    h = x.transpose(1, 2)
    for i in range(n_layers):
        p = f"{prefix}.conv.{i}.0.conv"
        h = F.relu(causal_conv1d(h, sd[p + ".weight"], sd[p + ".bias"], 1, st, p))
    h = h.transpose(1, 2)
    h = F.layer_norm(h, (h.shape[-1],), sd[f"{prefix}.post_ln.weight"], sd[f"{prefix}.post_ln.bias"], 1e-5)
    return F.linear(h, sd[f"{prefix}.linear.weight"], sd[f"{prefix}.linear.bias"])


@torch.no_grad()
def decode_frames(sd, hp, content, cache, st=None):
    """The content-dependent part of Conan.forward(infer=True) for content[B,T] int64 given the
    style cache; st=None -> whole sequence with zero left context, st=dict -> stateful frames."""
    ret = {"content": content}
    emb = F.embedding(content, sd["content_embedding.weight"])                         # Conan.py:140
    ce = causal_conv1d(emb.transpose(1, 2), sd["content_proj.0.conv.weight"], sd["content_proj.0.conv.bias"],
                       1, st, "content_proj.0.conv")
    ce = F.leaky_relu(ce, 0.01).transpose(1, 2)                                         # Conan.py:57-60
    ret["content_embed_proj"] = ce
    ret["style_embed"] = style_embed = cache["style_embed"]
    pitch_inp = ce + style_embed                                                        # Conan.py:162
    # ProsodyAligner (prosody_util.py:139-161), forcing=False since global_steps >= hp['forcing']
    out = pitch_inp.transpose(0, 1)
    mem = cache["tokens"].transpose(0, 1)
    attns = []
    for l in range(2):
        out, a = cross_atten_layer(sd, f"align.layers.{l}", out, mem, cache["key_padding_mask"])
        attns.append(a.unsqueeze(1))
    prosody = out.transpose(0, 1)
    ret["attn"] = attns
    ret["pitch_embed"] = pitch_inp = pitch_inp + prosody                                # Conan.py:168
    # add_orig_pitch (Conan.py:324-351)
    ret["uv_pred"] = uv_pred = pitch_predictor(sd, "uv_predictor", pitch_inp, 5, st)
    uv = uv_pred[:, :, 0] > 0
    uv[content == hp["silent_token"]] = 1
    f0 = uv_pred[:, :, 1]
    ret["fdiff"] = 0.0
    f0_denorm = denorm_f0(f0, uv)                                                       # Conan.py:298
    pitch = f0_to_coarse(f0_denorm)
    ret["f0_denorm_pred"] = f0_denorm
    ret["uv"] = uv
    ret["pitch_bins"] = pitch
    pitch_embed = F.embedding(pitch, sd["pitch_embed.weight"], padding_idx=0)
    ret["decoder_inp"] = decoder_inp = pitch_inp + pitch_embed                          # Conan.py:181
    x = causal_conv_blocks(sd, "decoder", decoder_inp.transpose(1, 2), hp["dec_kernel_size"],
                           hp["dec_dilations"], hp["layers_in_block"], hp.get("dec_post_net_kernel", 3), st)
    ret["decoder_out"] = x.transpose(1, 2)
    ret["mel_out"] = F.linear(x.transpose(1, 2), sd["mel_out.weight"], sd["mel_out.bias"])   # Conan.py:586-589
    ret["tgt_nonpadding"] = (content != -1).float()[:, :, None]
    return ret


@torch.no_grad()
def conan_forward(sd, hp, content, ref):
    """Conan.forward(content, ref=ref, infer=True, global_steps=200000) (Conan.py:115-198)."""
    cache = style_pass(sd, hp, ref)
    ret = decode_frames(sd, hp, content, cache, None)
    ret["ref_upsample"] = cache["ref_upsample"]
    ret["vq_ids"] = cache["vq_ids"]
    ret["vq_loss"] = None
    ret["ppl"] = None
    ret["gloss"] = None
    return ret


class Model:
    def __init__(self, sd, hp):
        self.sd = to_torch_sd(sd)
        self.hp = hp

    def __call__(self, content, ref):
        return conan_forward(self.sd, self.hp, torch.as_tensor(content).long(), torch.as_tensor(ref).float())

    def style_pass(self, ref):
        return style_pass(self.sd, self.hp, torch.as_tensor(ref).float())

    def decode_frames(self, content, cache, st=None):
        return decode_frames(self.sd, self.hp, torch.as_tensor(content).long(), cache, st)
