"""Generate tests/golden/*.npz by running the REFERENCE modules (imported from /root/reference
with non-arithmetic deps stubbed, tools/ref_import.py) on seeded inputs with the procedural
weights of conan_amd.synth.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py

The fixtures hold inputs + expected outputs only (no reference source).  Emformer has no
fixture: torchaudio is absent, see oracle/emformer.py ("parity unpinned").
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402
from conan_amd import configs, synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def t(x):
    return {k: torch.from_numpy(v) for k, v in x.items()}


def npf(x):
    return x.detach().cpu().numpy()


def build_ref_models(tiny):
    m, hp = ref_import.build_conan()
    g, vhp = ref_import.build_vocoder()
    if tiny:
        from utils.commons.hparams import hparams
        hp = dict(hp)
        hp.update(hidden_size=32, nVQ=16)
        hparams.update(hidden_size=32, nVQ=16)   # the reference reads the global dict in constructors
        from modules.Conan.Conan import Conan
        m = Conan(0, hp).eval()
        vhp = dict(vhp)
        vhp["upsample_initial_channel"] = configs.HIFIGAN_TINY["upsample_initial_channel"]
        from modules.vocoder.hifigan.hifigan_causal import HifiGanGenerator
        g = HifiGanGenerator(vhp).eval()
    chp, ghp = configs.conan_hparams(tiny), configs.hifigan_hparams(tiny)
    m.load_state_dict(t(synth.conan_state_dict(chp, 0)), strict=True)
    g.load_state_dict(t(synth.hifigan_state_dict(ghp, 0)), strict=True)
    return m, g, chp, ghp


@torch.no_grad()
def vocoder_goldens(g, tag, frames=(12, 150)):
    out = {}
    for T in frames:
        mel = synth.mel(T, 99 + T)[0].T[None]            # [1,80,T]
        feats = {}
        hooks = []
        hooks.append(g.conv_pre.register_forward_hook(lambda m, i, o: feats.__setitem__("conv_pre", o)))
        hooks.append(g.conv_post.register_forward_hook(lambda m, i, o: feats.__setitem__("pre_tanh", o)))
        for i, up in enumerate(g.ups):
            hooks.append(up.register_forward_hook(lambda m, inp, o, i=i: feats.__setitem__(f"ups.{i}", o)))
        wav = g(torch.from_numpy(mel))
        for h in hooks:
            h.remove()
        out[f"mel_{T}"] = mel
        out[f"wav_{T}"] = npf(wav)[0, 0]
        if T == 12:
            for k, v in feats.items():
                out[f"{k}_{T}"] = npf(v)[0]
    np.savez_compressed(os.path.join(OUT, f"hifigan_{tag}.npz"), **out)
    return out


@torch.no_grad()
def vocoder_variant_goldens():
    """`upsample: zero` + `resblock: "2"`, and `upsample: nn` (hifigan_causal.py:287-303): the reference generator
    built from those configs."""
    from modules.vocoder.hifigan.hifigan_causal import HifiGanGenerator
    for tag, ghp in (("zero_rb2_tiny", configs.HIFIGAN_ZERO_RB2_TINY), ("zero_rb2_full", configs.HIFIGAN_ZERO_RB2),
                     ("nn_tiny", configs.HIFIGAN_NN_TINY), ("nn_full", configs.HIFIGAN_NN)):
        g = HifiGanGenerator(dict(ghp)).eval()
        g.load_state_dict(t(synth.hifigan_state_dict(ghp, 0)), strict=True)
        vocoder_goldens(g, tag, frames=(12, 40))


@torch.no_grad()
def conan_goldens(m, tag, T=150, Tr=150):
    content = synth.codes(T, 1, seed=7)
    ref = synth.mel(Tr, 4321)
    ret = m(content=torch.from_numpy(content), spk_embed=None, target=None, ref=torch.from_numpy(ref),
            f0=None, uv=None, infer=True, global_steps=200000)
    # VQ ids are not in ret: recompute them through the reference's own sub-modules
    pe = m.prosody_extractor
    rm = torch.from_numpy(ref)
    pm = rm[:, :, 0].eq(0)
    wn = pe.wavenet(rm.transpose(1, 2), nonpadding=(~pm).unsqueeze(1).repeat([1, 80, 1])).transpose(1, 2)
    from utils.nn.seq_utils import group_hidden_by_segs
    ph, _ = group_hidden_by_segs(wn, ret["ref_upsample"], torch.max(ret["ref_upsample"]))
    enc = pe.encoder(ph)
    _, _, vq_ids, _ = pe.vqvae(enc)
    from utils.audio.pitch.utils import f0_to_coarse
    out = {
        "content": content, "ref": ref,
        "style_embed": npf(ret["style_embed"]), "vq_ids": npf(vq_ids),
        "prosody_enc": npf(enc),
        "content_embed_proj": npf(ret["content_embed_proj"]),
        "pitch_embed": npf(ret["pitch_embed"]),
        "uv_pred": npf(ret["uv_pred"]), "f0_denorm_pred": npf(ret["f0_denorm_pred"]),
        "pitch_bins": npf(f0_to_coarse(ret["f0_denorm_pred"].clone())),
        "decoder_inp": npf(ret["decoder_inp"]), "mel_out": npf(ret["mel_out"]),
        "attn0": npf(ret["attn"][0]),
        "keys": np.array(sorted(ret.keys())),
    }
    # windowed cases of SURVEY.md §0.6: last `c+4` codes only
    for c in (8, 16, 32):
        r2 = m(content=torch.from_numpy(content[:, T - c - 4:]), ref=torch.from_numpy(ref), infer=True,
               global_steps=200000)
        out[f"mel_out_win{c}"] = npf(r2["mel_out"])[:, -4:]
    np.savez_compressed(os.path.join(OUT, f"conan_{tag}.npz"), **out)
    return out


@torch.no_grad()
def loop_goldens(m, g, chp, tag, T=24, Tr=40):
    """Reference-semantics loop of inference/Conan.py:95-156 driven with GIVEN codes (Emformer
    bypassed: torchaudio absent), built from the reference's own modules."""
    codes = synth.codes(T, 1, seed=11)[0]
    ref = torch.from_numpy(synth.mel(Tr, 4321))
    hop, seg = 320, 4
    mel_chunks, wav_chunks, prev = [], [], 0
    pos = 0
    while pos < T:
        emit = min(seg, T - pos)
        all_codes = torch.from_numpy(codes[:pos + emit])[None]
        mel_out = m(content=all_codes, ref=ref, infer=True, global_steps=200000)["mel_out"][0]
        mel_chunks.append(mel_out[prev:])
        prev = mel_out.shape[0]
        pos += emit
        allmel = torch.cat(mel_chunks, 0).cpu().numpy()
        c = torch.FloatTensor(allmel).unsqueeze(0).transpose(2, 1)
        wav = g(c).view(-1).cpu().numpy()
        wav_chunks.append(wav[(pos - emit) * hop: pos * hop])
    np.savez_compressed(os.path.join(OUT, f"loop_{tag}.npz"), codes=codes, ref=npf(ref),
                        mel=npf(torch.cat(mel_chunks, 0)), wav=np.concatenate(wav_chunks))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_import.install()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    vocoder_variant_goldens()
    print("wrote goldens: vocoder variants (zero-insert upsampler + ResBlock2; transposed-conv upsampler)")
    if len(sys.argv) > 1 and sys.argv[1] == "variants":
        return
    for tiny in (False, True):
        tag = "tiny" if tiny else "full"
        m, g, chp, ghp = build_ref_models(tiny)
        vocoder_goldens(g, tag)
        conan_goldens(m, tag)
        loop_goldens(m, g, chp, tag, T=(24 if not tiny else 40))
        print("wrote goldens:", tag)


if __name__ == "__main__":
    main()
