"""Oracle: streaming Emformer content encoder.

PARITY UNPINNED (see oracle/__init__.py): the arithmetic belongs to the third-party
`torchaudio==2.5.1` (reference requirements.txt:3; torchaudio/models/emformer.py classes
Emformer / _EmformerImpl.infer / _EmformerLayer.infer / _EmformerAttention.infer), which is not
vendored in /root/reference and not installed here.  This file restates that published
algorithm at the reference's call sites:
  ctor   modules/Emformer/emformer.py:14-22  (input_dim 80, 8 heads, ffn 2048, L layers,
         segment_length = chunk_size//20, left_context_length 50, right_context_length rc,
         max_memory_size 0 -> no memory bank, activation relu, negative_inf -1e8).
         The memory bank (torchaudio's max_memory_size > 0, tanh_on_mem) is restated too: the reference never
         enables it, the build exposes it through hparams 'emformer_max_memory_size' / 'emformer_tanh_on_mem'.
  infer  inference/Conan.py:115, modules/Emformer/emformer.py:88
  head   modules/Emformer/emformer.py:25 (proj) + argmax inference/Conan.py:123-124
Test infrastructure only.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .common import to_torch_sd


class EmformerCfg:
    def __init__(self, hp, input_dim=80, num_heads=8, ffn_dim=2048, left_context_length=50,
                 max_memory_size=None, tanh_on_mem=None):
        self.input_dim = input_dim
        self.num_heads = num_heads
        self.ffn_dim = ffn_dim
        self.num_layers = hp["emformer_layers"]
        self.segment_length = hp["chunk_size"] // 20
        self.left_context_length = left_context_length
        self.right_context_length = hp["right_context"]
        self.max_memory_size = int(hp.get("emformer_max_memory_size", 0)) if max_memory_size is None else max_memory_size
        self.tanh_on_mem = bool(hp.get("emformer_tanh_on_mem", False)) if tanh_on_mem is None else tanh_on_mem


def init_state(cfg, batch):
    """_EmformerLayer._init_state: [memory(M,B,D), lc_key(L,B,D), lc_val(L,B,D), past_length(1,B) int32]."""
    D, L = cfg.input_dim, cfg.left_context_length
    return [[torch.zeros(cfg.max_memory_size, batch, D), torch.zeros(L, batch, D), torch.zeros(L, batch, D),
             torch.zeros(1, batch, dtype=torch.int32)] for _ in range(cfg.num_layers)]


def _attention_infer(sd, p, cfg, utt, rc, lc_key, lc_val, summary=None, mems=None):
    """_EmformerAttention.infer/_forward_impl.  utt[U,B,D], rc[R,B,D] are layer-normed inputs.
    Queries [rc | utt | summary]; keys [mems | rc | left-context | utt]; the summary query does not see the memory
    columns (attention_mask[-1, :mems] = True, filled with negative_inf = -1e8 before the softmax).
    Returns (rows of rc|utt after out_proj, output_mems[S,B,D] (clamped / tanh), next_k, next_v)."""
    D, Hh = cfg.input_dim, cfg.num_heads
    B = utt.shape[1]
    if summary is None:
        summary = utt.new_zeros(0, B, D)
    if mems is None:
        mems = utt.new_zeros(0, B, D)
    S, Mm = summary.shape[0], mems.shape[0]
    T = rc.shape[0] + utt.shape[0] + S
    q = F.linear(torch.cat([rc, utt, summary]), sd[f"{p}.attention.emb_to_query.weight"], sd[f"{p}.attention.emb_to_query.bias"])
    kv = F.linear(torch.cat([mems, rc, utt]), sd[f"{p}.attention.emb_to_key_value.weight"], sd[f"{p}.attention.emb_to_key_value.bias"])
    k, v = kv.chunk(2, dim=2)
    R = rc.shape[0]
    k = torch.cat([k[:Mm + R], lc_key, k[Mm + R:]])
    v = torch.cat([v[:Mm + R], lc_val, v[Mm + R:]])
    scaling = (D // Hh) ** -0.5
    rq, rk, rv = [t.contiguous().view(-1, B * Hh, D // Hh).transpose(0, 1) for t in (q, k, v)]
    w = torch.bmm(rq * scaling, rk.transpose(1, 2)).float()
    if S > 0 and Mm > 0:
        mask = torch.zeros(T, k.shape[0], dtype=torch.bool)
        mask[-1, :Mm] = True
        w = w.masked_fill(mask.unsqueeze(0), -1e8)
    # padding mask None for equal lengths / B == 1
    probs = torch.softmax(w, dim=-1).type_as(q)
    att = torch.bmm(probs, rv).transpose(0, 1).contiguous().view(T, B, D)
    out = F.linear(att, sd[f"{p}.attention.out_proj.weight"], sd[f"{p}.attention.out_proj.bias"])
    out_rc_utt, out_mems = out[:T - S], out[T - S:]
    out_mems = torch.tanh(out_mems) if cfg.tanh_on_mem else torch.clamp(out_mems, min=-10, max=10)
    return out_rc_utt, out_mems, k[Mm + R:], v[Mm + R:]                   # next_k/v = left-context | utt


def _layer_infer(sd, p, cfg, utt, rc, state, mems):
    """_EmformerLayer.infer.  `mems` [1,B,D] (or [0,B,D]) is this step's memory INPUT of the layer: it is appended to the
    layer's bank (_pack_state) while the attention reads the bank as it was BEFORE this step (_unpack_state)."""
    D, M = cfg.input_dim, cfg.max_memory_size
    lnw, lnb = sd[f"{p}.layer_norm_input.weight"], sd[f"{p}.layer_norm_input.bias"]
    ln_utt = F.layer_norm(utt, (D,), lnw, lnb, 1e-5)
    ln_rc = F.layer_norm(rc, (D,), lnw, lnb, 1e-5)
    # _unpack_state: batch element 0 decides for the whole (lock-step) batch
    past_length = int(state[3][0][0].item())
    L = min(cfg.left_context_length, past_length)
    lc_key = state[1][cfg.left_context_length - L:]
    lc_val = state[2][cfg.left_context_length - L:]
    summary = pre_mems = None
    if M > 0:
        past_mem = min(M, math.ceil(past_length / cfg.segment_length))
        pre_mems = state[0][M - past_mem:]
        # memory_op = AvgPool1d(kernel = stride = segment_length, ceil_mode=True) over the normalised utterance, first row
        summary = F.avg_pool1d(ln_utt.permute(1, 2, 0), cfg.segment_length, cfg.segment_length, ceil_mode=True).permute(2, 0, 1)[:1]
    rc_out, out_mems, next_k, next_v = _attention_infer(sd, p, cfg, ln_utt, ln_rc, lc_key, lc_val, summary, pre_mems)
    # _pack_state
    new_k = torch.cat([state[1], next_k])
    new_v = torch.cat([state[2], next_v])
    new_mem = torch.cat([state[0], mems])[-M:] if M > 0 else state[0]
    state = [new_mem, new_k[new_k.shape[0] - cfg.left_context_length:],
             new_v[new_v.shape[0] - cfg.left_context_length:], state[3] + utt.shape[0]]
    # _process_attention_output
    res = rc_out + torch.cat([rc, utt])
    ff = F.layer_norm(res, (D,), sd[f"{p}.pos_ff.0.weight"], sd[f"{p}.pos_ff.0.bias"], 1e-5)
    ff = F.linear(F.relu(F.linear(ff, sd[f"{p}.pos_ff.1.weight"], sd[f"{p}.pos_ff.1.bias"])),
                  sd[f"{p}.pos_ff.4.weight"], sd[f"{p}.pos_ff.4.bias"])
    res = ff + res
    res = F.layer_norm(res, (D,), sd[f"{p}.layer_norm_output.weight"], sd[f"{p}.layer_norm_output.bias"], 1e-5)
    R = rc.shape[0]
    return res[R:], res[:R], state, out_mems


@torch.no_grad()
def emformer_infer(sd, cfg, inp, lengths, states=None):
    """Emformer.infer: inp[B, seg+rc, D] -> (out[B, seg, D], lengths - rc, states)."""
    if inp.shape[1] != cfg.segment_length + cfg.right_context_length:
        raise ValueError("Per configured segment_length and right_context_length, expected size of "
                         f"{cfg.segment_length + cfg.right_context_length} for dimension 1 of input, "
                         f"but got {inp.shape[1]}.")
    x = inp.permute(1, 0, 2)
    s = x.shape[0] - cfg.right_context_length
    rc, utt = x[s:], x[:s]
    out_lengths = torch.clamp(lengths - cfg.right_context_length, min=0)
    if states is None:
        states = init_state(cfg, inp.shape[0])
    new_states = []
    out = utt
    # _EmformerImpl.infer: the first layer's memory input is the average of the raw segment
    if cfg.max_memory_size > 0:
        mems = F.avg_pool1d(utt.permute(1, 2, 0), cfg.segment_length, cfg.segment_length, ceil_mode=True).permute(2, 0, 1)
    else:
        mems = utt.new_zeros(0, utt.shape[1], utt.shape[2])
    for i in range(cfg.num_layers):
        out, rc, st, mems = _layer_infer(sd, f"emformer.emformer_layers.{i}", cfg, out, rc, states[i], mems)
        new_states.append(st)
    return out.permute(1, 0, 2), out_lengths, new_states


@torch.no_grad()
def logits_and_codes(sd, chunk_out):
    """proj (emformer.py:25) + argmax (inference/Conan.py:123-124)."""
    if "proj1.weight" in sd:       # mode == 'both': inference/Conan.py:117-118 reads proj1
        logits = F.linear(chunk_out, sd["proj1.weight"], sd["proj1.bias"])
    else:
        logits = F.linear(chunk_out, sd["proj.weight"], sd["proj.bias"]) if "proj.weight" in sd else chunk_out
    return logits, torch.argmax(logits, dim=-1)


def chunk_iter(mel, seg, rc):
    """Chunking of inference/Conan.py:95-110 / emformer.py:66-83: yields (pos, emit, chunk[B,seg+rc,F])
    with the last real frame repeated as padding."""
    B, T, Fd = mel.shape
    pos = 0
    while pos < T:
        emit = min(seg, T - pos)
        look = min(rc, T - (pos + emit))
        real = emit + look
        chunk = mel[:, pos:pos + real, :]
        need = (seg + rc) - real
        if need > 0:
            chunk = torch.cat([chunk, chunk[:, -1:, :].expand(B, need, Fd)], dim=1)
        yield pos, emit, chunk
        pos += emit


@torch.no_grad()
def stream_codes(sd, cfg, mel):
    """EmformerDistillModel.inference (emformer.py:48-98) + argmax: mel[B,T,80] -> (logits[B,T,K], codes[B,T])."""
    state = None
    outs = []
    for pos, emit, chunk in chunk_iter(mel, cfg.segment_length, cfg.right_context_length):
        lengths = torch.full((mel.shape[0],), chunk.shape[1], dtype=torch.long)
        o, _, state = emformer_infer(sd, cfg, chunk, lengths, state)
        outs.append(o[:, :emit])
    out = torch.cat(outs, dim=1)
    return logits_and_codes(sd, out)


@torch.no_grad()
def dense_reference(sd, cfg, mel):
    """Independently formulated check of the streaming recursion (SURVEY.md §8c mitigation ii):
    processes the whole utterance layer by layer with explicit per-segment key sets
    {rc of segment s} | {last <=50 utterance keys before segment s} | {utterance keys of segment s},
    using only whole-sequence tensors (no rolling caches).  With a memory bank the keys of segment s are preceded by
    the memory inputs of the layer for segments s-M .. s-1 (layer 0: the segment means of the raw input; layer l > 0:
    the clamped summary outputs of layer l-1), and one summary query per segment (mean of the normalised segment) that
    sees everything but the memory columns yields the next layer's memory input.  Returns out[B,T,D] for T % seg == 0."""
    D, Hh = cfg.input_dim, cfg.num_heads
    seg, R, LC = cfg.segment_length, cfg.right_context_length, cfg.left_context_length
    B, T, _ = mel.shape
    assert T % seg == 0
    nseg = T // seg
    # layer-0 inputs: utterance frames and per-segment right-context frames (repeat-last padding)
    utt = mel.clone()                                            # [B,T,D]
    rcs = []
    for s in range(nseg):
        idx = [min((s + 1) * seg + r, T - 1) for r in range(R)]
        rcs.append(mel[:, idx])
    rcx = torch.stack(rcs, 1) if R > 0 else mel.new_zeros(B, nseg, 0, D)  # [B,nseg,R,D]
    dh = D // Hh
    M = cfg.max_memory_size
    mem_in = utt.view(B, nseg, seg, D).mean(2) if M > 0 else None          # [B,nseg,D] memory inputs of layer 0
    for i in range(cfg.num_layers):
        p = f"emformer.emformer_layers.{i}"
        ln = lambda t: F.layer_norm(t, (D,), sd[f"{p}.layer_norm_input.weight"], sd[f"{p}.layer_norm_input.bias"], 1e-5)
        lu, lr = ln(utt), ln(rcx)
        Wq, bq = sd[f"{p}.attention.emb_to_query.weight"], sd[f"{p}.attention.emb_to_query.bias"]
        Wkv, bkv = sd[f"{p}.attention.emb_to_key_value.weight"], sd[f"{p}.attention.emb_to_key_value.bias"]
        qu, qr = F.linear(lu, Wq, bq), F.linear(lr, Wq, bq)
        ku, vu = F.linear(lu, Wkv, bkv).chunk(2, -1)
        kr, vr = F.linear(lr, Wkv, bkv).chunk(2, -1)
        new_utt = torch.empty_like(utt)
        new_rc = torch.empty_like(rcx)
        if M > 0:
            km, vm = F.linear(mem_in, Wkv, bkv).chunk(2, -1)                 # K/V of every segment's memory input
            qsum = F.linear(lu.view(B, nseg, seg, D).mean(2), Wq, bq)        # summary queries [B,nseg,D]
            mem_out = torch.empty_like(mem_in)
        for s in range(nseg):
            lo = max(0, s * seg - LC)
            keys = torch.cat([kr[:, s], ku[:, lo:s * seg], ku[:, s * seg:(s + 1) * seg]], 1)   # [B,nk,D]
            vals = torch.cat([vr[:, s], vu[:, lo:s * seg], vu[:, s * seg:(s + 1) * seg]], 1)
            qs = torch.cat([qr[:, s], qu[:, s * seg:(s + 1) * seg]], 1)                          # [B,R+seg,D]
            if M > 0:
                # the summary query sees the segment's own keys only; its output is the next layer's memory input
                nk0 = keys.shape[1]
                qh = qsum[:, s].view(B, 1, Hh, dh).permute(0, 2, 1, 3) * dh ** -0.5
                kh = keys.view(B, nk0, Hh, dh).permute(0, 2, 1, 3)
                vh = vals.view(B, nk0, Hh, dh).permute(0, 2, 1, 3)
                am = (torch.softmax(qh @ kh.transpose(-1, -2), -1) @ vh).permute(0, 2, 1, 3).reshape(B, 1, D)
                am = F.linear(am, sd[f"{p}.attention.out_proj.weight"], sd[f"{p}.attention.out_proj.bias"])[:, 0]
                mem_out[:, s] = torch.tanh(am) if cfg.tanh_on_mem else am.clamp(-10, 10)
                m0 = max(0, s - M)
                keys = torch.cat([km[:, m0:s], keys], 1)
                vals = torch.cat([vm[:, m0:s], vals], 1)
            nq, nk = qs.shape[1], keys.shape[1]
            qh = qs.view(B, nq, Hh, dh).permute(0, 2, 1, 3) * dh ** -0.5
            kh = keys.view(B, nk, Hh, dh).permute(0, 2, 1, 3)
            vh = vals.view(B, nk, Hh, dh).permute(0, 2, 1, 3)
            a = torch.softmax(qh @ kh.transpose(-1, -2), -1) @ vh
            a = a.permute(0, 2, 1, 3).reshape(B, nq, D)
            a = F.linear(a, sd[f"{p}.attention.out_proj.weight"], sd[f"{p}.attention.out_proj.bias"])
            res = a + torch.cat([rcx[:, s], utt[:, s * seg:(s + 1) * seg]], 1)
            ff = F.layer_norm(res, (D,), sd[f"{p}.pos_ff.0.weight"], sd[f"{p}.pos_ff.0.bias"], 1e-5)
            ff = F.linear(F.relu(F.linear(ff, sd[f"{p}.pos_ff.1.weight"], sd[f"{p}.pos_ff.1.bias"])),
                          sd[f"{p}.pos_ff.4.weight"], sd[f"{p}.pos_ff.4.bias"])
            res = F.layer_norm(ff + res, (D,), sd[f"{p}.layer_norm_output.weight"], sd[f"{p}.layer_norm_output.bias"], 1e-5)
            new_rc[:, s] = res[:, :R]
            new_utt[:, s * seg:(s + 1) * seg] = res[:, R:]
        utt, rcx = new_utt, new_rc
        if M > 0:
            mem_in = mem_out
    return utt


class Model:
    def __init__(self, sd, hp):
        self.sd = to_torch_sd(sd)
        self.cfg = EmformerCfg(hp)

    def infer(self, chunk, lengths, states=None):
        return emformer_infer(self.sd, self.cfg, chunk, lengths, states)

    def stream_codes(self, mel):
        return stream_codes(self.sd, self.cfg, torch.as_tensor(mel).float())
