"""Conan with the reference's constructor / forward / state_dict contract (modules/Conan/Conan.py:45-198),
inference path only (infer=True, style=true, f0_gen='orig', decoder_type='conv'), computed by the HIP path:
per-utterance style pass (conan_set_reference) + stateful decoder steps over the content codes."""
import torch
from torch import nn

from .. import _tree
from ... import specs
from ...runtime import Context


class Conan(_tree.ParamTree):
    STEP_FRAMES = 16

    def __init__(self, dict_size, hparams, out_dims=None):
        super().__init__()
        self.hparams = dict(hparams)
        self.hidden_size = hparams["hidden_size"]
        self.out_dims = hparams["audio_num_mel_bins"] if out_dims is None else out_dims
        self.padding_idx = 0
        _tree.build_tree(self, specs.conan_spec(hparams), buffers=specs.CONAN_BUFFERS)
        self._ctx = None
        self._streams = None

    def refresh(self):
        self._drop()
        ctx = Context(self.hparams, None, torch.cuda.current_device(), emformer=False, conan=True, hifigan=False)
        ctx.load_state_dict("conan", _tree.host_state_dict(self))
        ctx.finalize()
        self._ctx = ctx

    def _drop(self):
        if self._streams is not None:
            self._streams.close()
            self._streams = None
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._drop()
        return out

    def _get_streams(self, B, Tr):
        if self._ctx is None:
            self.refresh()
        s = self._streams
        if s is None or s.max_slots < B or s.max_ref_frames < Tr:
            if s is not None:
                s.close()
            self._streams = self._ctx.streams(B, max_frames=self.STEP_FRAMES, max_ref_frames=max(Tr, 64))
        return self._streams

    @torch.no_grad()
    def forward(self, content, spk_embed=None, target=None, ref=None, f0=None, uv=None, infer=False, global_steps=0, **kwargs):
        if spk_embed is None and ref is None:
            raise ValueError("When spk_embed is None, need target tensor to extract speaker embedding.")   # Conan.py:152-155
        if not infer:
            raise NotImplementedError("the HIP hot path covers Conan.forward(..., infer=True) (inference/Conan.py:132-141)")
        if ref is None:
            # the reference reaches get_prosody(pitch_inp, ref, ...) with ref=None and fails there (Conan.py:166, style: true)
            raise ValueError("ref is required: the prosody tokens are extracted from the reference mel (Conan.py:166)")
        if not content.is_cuda:
            raise RuntimeError("conan_amd.Conan runs on a HIP device only (no CPU fallback)")
        B, T = content.shape
        Tr = ref.shape[1]
        st = self._get_streams(B, Tr)
        slots = list(range(B))
        st.reset(slots, which=2)
        st.set_reference(slots, ref.float().contiguous())
        if spk_embed is not None:   # Conan.py:146-149: style_embed = spk_embed ([B,1,H]); the prosody still comes from ref
            st.set_style(slots, spk_embed)
        codes = content.to(torch.int32).contiguous()
        mels, taps = [], {"uv_pred": [], "f0_denorm_pred": [], "pitch_bins": [], "decoder_inp": [], "content_embed_proj": []}
        attn = [[], []]
        for p in range(0, T, self.STEP_FRAMES):
            m, tp = st.decoder_step(slots, codes[:, p:p + self.STEP_FRAMES], taps=True)
            mels.append(m)
            for k in taps:
                taps[k].append(tp[k])
            for l in range(2):
                attn[l].append(tp["attn"][l])
        ret = {"content": content, "mel_out": torch.cat(mels, 1), "tgt_nonpadding": (content != -1).float()[:, :, None], "fdiff": 0.0,
               "vq_loss": None, "ppl": None, "gloss": None}
        for k in taps:
            ret[k] = torch.cat(taps[k], 1)
        # style_embed [B,1,H] (Conan.py:158) and the ProsodyAligner attention list, [B,1,T,S] per layer (prosody_util.py:119-126)
        ret["style_embed"] = st.style_embed(slots).unsqueeze(1)
        S = (Tr + 3) // 4
        ret["attn"] = [torch.cat(a, 1)[:, :, :S].unsqueeze(1) for a in attn]
        ret["ref_upsample"] = (torch.arange(Tr, device=content.device) // 4 + 1).unsqueeze(0).expand(B, -1)
        return ret
