"""HifiGanGenerator with the reference's constructor / forward / state_dict contract
(modules/vocoder/hifigan/hifigan_causal.py:269-341), computed by the HIP path.

forward(x[B, 80, T]) -> [B, 1, T*prod(upsample_rates)].  Because every layer is causal, the whole-utterance
forward equals a reset followed by stateful steps; it is run as steps of <= 16 frames so the
per-stream rings stay small.  `upsample: nn` (CausalUpsampleBlock1, :60-145) is the exception: the transposed
convolution looks two input frames ahead at every stage, so its forward is one step over the whole input (rings sized
for T frames) and the library refuses a second step without a reset."""
import torch
from torch import nn

from ... import _tree
from .... import specs
from ....runtime import Context


class HifiGanGenerator(_tree.ParamTree):
    STEP_FRAMES = 16

    def __init__(self, hparams):
        super().__init__()
        self.h = hparams
        if hparams.get("upsample", "shuffle") not in ("shuffle", "zero", "nn"):
            raise NotImplementedError("upsample=%r: 'shuffle', 'zero' or 'nn' (hifigan_causal.py:287-293)" % (hparams.get("upsample"),))
        self.one_shot = hparams.get("upsample", "shuffle") == "nn"
        spec = specs.hifigan_spec(hparams)
        _tree.build_tree(self, spec, buffers=[k for k in spec if k.endswith("._cache")])
        self._frames = self.STEP_FRAMES
        self._ctx = None
        self._streams = None

    # -- HIP context management
    def refresh(self):
        """Re-pack the current parameters (call after load_state_dict)."""
        self._drop()
        vhp = dict(self.h)
        vhp.setdefault("audio_num_mel_bins", vhp.get("num_mels", 80))
        ctx = Context(None, vhp, torch.cuda.current_device(), emformer=False, conan=False, hifigan=True)
        ctx.load_state_dict("hifigan", _tree.host_state_dict(self))
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
        # checkpoints saved after remove_weight_norm carry '<conv>.weight' instead of weight_g / weight_v
        sd = dict(state_dict)
        own = self.state_dict()
        for k in list(sd.keys()):
            if k.endswith(".weight") and k not in own and k[:-len("weight")] + "weight_v" in own:
                w = sd.pop(k)
                sd[k[:-len("weight")] + "weight_v"] = w
                sd[k[:-len("weight")] + "weight_g"] = w.flatten(1).norm(dim=1).view(-1, 1, 1)
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._drop()
        return out

    def _get_streams(self, B, frames):
        if self._ctx is None:
            self.refresh()
        if self._streams is None or self._streams.max_slots < B or self._frames < frames:
            if self._streams is not None:
                self._streams.close()
            self._frames = max(self._frames, frames)
            self._streams = self._ctx.streams(B, max_frames=self._frames, max_ref_frames=4)
        return self._streams

    @torch.no_grad()
    def forward(self, x, f0=None):
        if not x.is_cuda:
            raise RuntimeError("conan_amd.HifiGanGenerator runs on a HIP device only (no CPU fallback)")
        B, C, T = x.shape
        step = T if self.one_shot else self.STEP_FRAMES
        st = self._get_streams(B, step)
        slots = list(range(B))
        st.reset(slots, which=4)
        mel = x.transpose(1, 2).contiguous().float()
        outs = []
        for p in range(0, T, step):
            outs.append(st.hifigan_step(slots, mel[:, p:p + step]))
        return torch.cat(outs, 1).unsqueeze(1)

    def remove_weight_norm(self):
        """No-op for compute (the fold g*v/||v|| happens at pack time, ctx.hip pack_weightnorm); kept for API parity."""
        print("Removing weight_norm from Generator...")
        print("Generator weight_norm removal completed.")
