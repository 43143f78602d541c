"""Emformer oracle self-checks (parity is UNPINNED by the reference: torchaudio 2.5.1 is a
third-party dependency absent from /root/reference; SURVEY.md §8c mitigations ii and iii)."""
import numpy as np
import pytest
import torch

from conan_amd import configs, synth
from oracle import emformer as oemf
from oracle.common import to_torch_sd


@pytest.fixture(scope="module")
def model():
    hp = configs.conan_hparams()
    return oemf.Model(synth.emformer_state_dict(hp, 0), hp)


@pytest.mark.parametrize("T", [4, 8, 60, 72])   # first chunk, second chunk, cache saturation (>50)
def test_streaming_equals_dense(model, T):
    mel = torch.from_numpy(synth.mel(T, 5, 2))
    state, outs = None, []
    for pos, emit, chunk in oemf.chunk_iter(mel, 4, 2):
        o, l, state = model.infer(chunk, torch.full((2,), 6, dtype=torch.long), state)
        assert l.tolist() == [4, 4]
        outs.append(o[:, :emit])
    stream = torch.cat(outs, 1)
    dense = oemf.dense_reference(model.sd, model.cfg, mel)
    np.testing.assert_allclose(stream.numpy(), dense.numpy(), atol=2e-5, rtol=1e-5)
    assert int(state[0][3][0][0]) == T


def test_state_shapes_and_errors(model):
    mel = torch.from_numpy(synth.mel(6, 1, 1))
    o, l, st = model.infer(mel, torch.tensor([6]))
    assert o.shape == (1, 4, 80) and len(st) == 6
    assert st[0][1].shape == (50, 1, 80) and st[0][3].dtype == torch.int32
    with pytest.raises(ValueError):
        model.infer(mel[:, :5], torch.tensor([5]))


def test_ragged_tail_chunking(model):
    """T not a multiple of seg: repeat-last padding (inference/Conan.py:104-110)."""
    mel = torch.from_numpy(synth.mel(10, 3, 1))
    logits, codes = model.stream_codes(mel)
    assert logits.shape == (1, 10, 100) and codes.shape == (1, 10)
    assert int(codes.min()) >= 0 and int(codes.max()) < 100


def test_against_torchaudio_if_installed(model):
    ta = pytest.importorskip("torchaudio")
    em = ta.models.Emformer(80, 8, 2048, 6, 4, left_context_length=50, right_context_length=2)
    sd = {k[len("emformer."):]: v for k, v in model.sd.items() if k.startswith("emformer.")}
    em.load_state_dict(sd, strict=True)
    em.eval()
    mel = torch.from_numpy(synth.mel(64, 5, 2))
    s1 = s2 = None
    for pos, emit, chunk in oemf.chunk_iter(mel, 4, 2):
        lengths = torch.full((2,), 6, dtype=torch.long)
        with torch.no_grad():
            o1, _, s1 = em.infer(chunk, lengths, s1)
        o2, _, s2 = model.infer(chunk, lengths, s2)
        np.testing.assert_allclose(o1.numpy(), o2.numpy(), atol=1e-5)


@pytest.mark.parametrize("M,tanh,T", [(4, False, 8), (4, False, 40), (3, True, 28), (1, False, 16)])
def test_memory_bank_streaming_equals_dense(M, tanh, T):
    """torchaudio's memory bank (max_memory_size > 0; the reference leaves it off, modules/Emformer/emformer.py:14-22):
    the rolling-state restatement against the whole-sequence formulation with explicit per-segment memory columns -
    bank ramp-up (fewer than M past segments), saturation and roll-over (> M segments), clamp and tanh variants."""
    hp = dict(configs.conan_hparams(), emformer_layers=3, emformer_max_memory_size=M, emformer_tanh_on_mem=tanh)
    m = oemf.Model(synth.emformer_state_dict(hp, 0), hp)
    assert m.cfg.max_memory_size == M and m.cfg.tanh_on_mem == tanh
    mel = torch.from_numpy(synth.mel(T, 9, 2))
    state, outs = None, []
    for pos, emit, chunk in oemf.chunk_iter(mel, 4, 2):
        o, _, state = m.infer(chunk, torch.full((2,), 6, dtype=torch.long), state)
        outs.append(o[:, :emit])
    stream = torch.cat(outs, 1)
    dense = oemf.dense_reference(m.sd, m.cfg, mel)
    np.testing.assert_allclose(stream.numpy(), dense.numpy(), atol=2e-5, rtol=1e-5)
    assert state[0][0].shape == (M, 2, 80)
    # the bank matters: the same weights without it give a different output once a past segment exists
    m0 = oemf.Model(synth.emformer_state_dict(hp, 0), dict(hp, emformer_max_memory_size=0))
    d0 = oemf.dense_reference(m0.sd, m0.cfg, mel)
    assert float((d0[:, 4:] - dense[:, 4:]).abs().max()) > 1e-3
    np.testing.assert_allclose(d0[:, :4].numpy(), dense[:, :4].numpy(), atol=2e-5)   # first segment: empty bank
