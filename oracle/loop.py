"""Oracle: the chunk loop of StreamingVoiceConversion.infer_once (inference/Conan.py:72-166),
operating on mel inputs (the librosa front-end, :57-70, is a "next" row).

Two variants with identical results up to fp32 reassociation:
  infer_once_ref   -- reference semantics: per chunk, re-run Conan on the whole code prefix
                      (incl. the style encoders, :132-141) and the vocoder on the whole mel
                      prefix through the numpy hop (:148-149), keep the new slice (:151-156).
  infer_once_stateful -- style pass once per utterance, conv state carried, only new frames.
Test infrastructure only (see oracle/__init__.py).
"""
import numpy as np
import torch

from . import conan as oconan
from . import emformer as oemf
from . import hifigan as ohifi


@torch.no_grad()
def infer_once_ref(emf_sd, emf_cfg, conan_sd, conan_hp, voc_sd, voc_hp, src_mel, ref_mel, codes_override=None,
                   on_chunk=None):
    """src_mel[T,80], ref_mel[Tr,80] numpy/torch -> (wav np[N], mel np[T,80], codes np[T])."""
    src = torch.as_tensor(src_mel).float().unsqueeze(0)
    ref = torch.as_tensor(ref_mel).float()
    seg, rc = emf_cfg.segment_length, emf_cfg.right_context_length
    hop = conan_hp["hop_size"]
    code_buf, mel_chunks, wav_chunks = [], [], []
    prev_len, state = 0, None
    for pos, emit, chunk in oemf.chunk_iter(src, seg, rc):
        lengths = torch.full((1,), chunk.size(1), dtype=torch.long)
        out, _, state = oemf.emformer_infer(emf_sd, emf_cfg, chunk, lengths, state)
        _, new_codes = oemf.logits_and_codes(emf_sd, out)
        new_codes = new_codes[:, :emit]
        if codes_override is not None:
            new_codes = torch.as_tensor(codes_override[pos:pos + emit]).long().unsqueeze(0)
        code_buf.append(new_codes.squeeze(0))
        all_codes = torch.cat(code_buf, 0).unsqueeze(0)
        mel_out = oconan.conan_forward(conan_sd, conan_hp, all_codes, ref.unsqueeze(0))["mel_out"][0]
        mel_chunks.append(mel_out[prev_len:])
        prev_len = mel_out.shape[0]
        p_end = pos + emit
        wav_all = ohifi.spec2wav(voc_sd, voc_hp, torch.cat(mel_chunks, 0).cpu().numpy())
        s0, s1 = max(0, (p_end - emit) * hop), min(len(wav_all), p_end * hop)
        if s1 > s0:
            wav_chunks.append(wav_all[s0:s1])
        if on_chunk is not None:
            on_chunk()
    mel_pred = torch.cat(mel_chunks, 0)
    return np.concatenate(wav_chunks, 0), mel_pred.cpu().numpy(), torch.cat(code_buf, 0).cpu().numpy()


@torch.no_grad()
def infer_once_stateful(emf_sd, emf_cfg, conan_sd, conan_hp, voc_sd, voc_hp, src_mel, ref_mel, codes_override=None,
                        on_chunk=None):
    src = torch.as_tensor(src_mel).float().unsqueeze(0)
    ref = torch.as_tensor(ref_mel).float().unsqueeze(0)
    seg, rc = emf_cfg.segment_length, emf_cfg.right_context_length
    cache = oconan.style_pass(conan_sd, conan_hp, ref)
    cst, vst, state = {}, {}, None
    codes, mels, wavs = [], [], []
    for pos, emit, chunk in oemf.chunk_iter(src, seg, rc):
        lengths = torch.full((1,), chunk.size(1), dtype=torch.long)
        out, _, state = oemf.emformer_infer(emf_sd, emf_cfg, chunk, lengths, state)
        _, new_codes = oemf.logits_and_codes(emf_sd, out)
        new_codes = new_codes[:, :emit]
        if codes_override is not None:
            new_codes = torch.as_tensor(codes_override[pos:pos + emit]).long().unsqueeze(0)
        codes.append(new_codes[0])
        mel_new = oconan.decode_frames(conan_sd, conan_hp, new_codes, cache, cst)["mel_out"]    # [1,emit,80]
        mels.append(mel_new[0])
        wav = ohifi.generator_forward(voc_sd, voc_hp, mel_new.transpose(1, 2), vst).view(-1)
        wavs.append(wav)
        if on_chunk is not None:
            on_chunk()
    return torch.cat(wavs).cpu().numpy(), torch.cat(mels, 0).cpu().numpy(), torch.cat(codes).cpu().numpy()
