"""EmformerDistillModel with the reference's constructor / attribute / state_dict contract
(modules/Emformer/emformer.py:6-98).  `.emformer.infer(input, lengths, states)` is the streaming step of
torchaudio.models.Emformer (used at inference/Conan.py:115); states are opaque handles onto the per-slot K/V
rings of a conan_streams object instead of lists of tensors."""
import torch
from torch import nn

from .. import _tree
from ... import specs
from ...runtime import Context


class EmformerState:
    """Opaque streaming state returned by Emformer.infer (replaces torchaudio's List[List[Tensor]])."""

    def __init__(self, streams, slots):
        self.streams, self.slots = streams, slots


class _Proj:
    """An output head of EmformerDistillModel (emformer.py:25 `proj`, :29-30 `proj1` / `proj2`).  The streaming step
    (conan_emformer_step) already projects with `proj` / `proj1`: applied to the very tensor infer() just returned
    (inference/Conan.py:115-120) the head hands back those logits.  Any other input (a slice, the concatenated chunks of
    `inference()`, or the second head) goes through conan_emformer_project - a k = 1 conv on the same MFMA kernel, with the
    weights packed at finalize; there is no torch / rocBLAS arithmetic behind these modules."""

    def __init__(self, owner, name, fused):
        self.owner, self.name, self.fused = owner, name, fused

    def __call__(self, x):
        last = self.owner._last
        if self.fused and last is not None and x is last[0]:
            return last[1]
        if not x.is_cuda:
            raise RuntimeError("conan_amd.Emformer runs on a HIP device only (no CPU fallback)")
        return self.owner._get_streams(1).emformer_project(self.name, x)


class _Emformer:
    def __init__(self, owner):
        self.owner = owner

    @torch.no_grad()
    def infer(self, input, lengths, states=None):
        o = self.owner
        seg, rc = o.segment_length, o.right_context_len
        if input.size(1) != seg + rc:
            raise ValueError(f"Per configured segment_length and right_context_length, expected size of {seg + rc} for "
                             f"dimension 1 of input, but got {input.size(1)}.")
        if not input.is_cuda:
            raise RuntimeError("conan_amd.Emformer runs on a HIP device only (no CPU fallback)")
        B = input.shape[0]
        if states is None:
            st = o._get_streams(B)
            slots = list(range(B))
            st.reset(slots, which=1)
            states = EmformerState(st, slots)
        out, logits, _ = states.streams.emformer_step(states.slots, input.float().contiguous(), want_codes=False)
        o._last = (out, logits)
        return out, torch.clamp(lengths - rc, min=0), states


class EmformerDistillModel(_tree.ParamTree):
    def __init__(self, hparams, input_dim=80, output_dim=None):
        super().__init__()
        if output_dim is None:
            output_dim = hparams.get("emformer_output_dim", 768)
        self.hp = dict(hparams, emformer_output_dim=output_dim)
        self.segment_length = hparams["chunk_size"] // 20
        self.right_context_len = hparams["right_context"]
        self.mode = hparams.get("mode", None)
        _tree.build_tree(self, specs.emformer_spec(hparams, input_dim, output_dim))
        # `.emformer` must stay the container of the parameter tree (state_dict keys 'emformer.emformer_layers...'),
        # so the streaming entry point is attached to that sub-module; `.proj` keeps its parameters likewise.
        self._modules["emformer"].infer = _Emformer(self).infer
        # the head the streaming step projects with (proj1 in 'both' mode, inference/Conan.py:117-118) returns the step's own
        # logits; every head can also project arbitrary features (conan_emformer_project)
        step_head = "proj1" if self.mode == "both" else "proj"
        for name in ("proj", "proj1", "proj2"):
            if name in self._modules and isinstance(self._modules[name], nn.Module) and hasattr(self._modules[name], "weight"):
                self._modules[name].forward = _Proj(self, name, fused=name == step_head)
        self._last = None
        self._ctx = None
        self._streams = None

    def refresh(self):
        self._drop()
        ctx = Context(self.hp, None, torch.cuda.current_device(), emformer=True, conan=False, hifigan=False)
        ctx.load_state_dict("emformer", _tree.host_state_dict(self))
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

    def _get_streams(self, B):
        if self._ctx is None:
            self.refresh()
        if self._streams is None or self._streams.max_slots < B:
            if self._streams is not None:
                self._streams.close()
            self._streams = self._ctx.streams(B, max_frames=self.segment_length, max_ref_frames=4)
        return self._streams

    @torch.inference_mode()
    def inference_rtf(self, mel_input, verbose=False):
        """emformer.py:99-156: `inference` with a clock around every streaming step.  Returns what the reference returns:
        (proj(features), latency_list, rtf_list), or (proj1(features), proj2(features), latency_list) when mode == 'both'; latencies in
        seconds per chunk step (host submit -> the step's outputs complete on the device: the reference reads time.time() around an
        asynchronous CUDA call, i.e. the submit time - here the device is synchronised so that the figure is the step's latency),
        rtf = latency / (segment_length * 20 ms)."""
        import time
        B, T, F = mel_input.shape
        seg, rc = self.segment_length, self.right_context_len
        pos, state, outs, logits, lat, rtf = 0, None, [], [], [], []
        while pos < T:
            emit = min(seg, T - pos)
            look = min(rc, T - (pos + emit))
            real = emit + look
            chunk = mel_input[:, pos:pos + real, :]
            need = (seg + rc) - real
            if need > 0:
                chunk = torch.cat([chunk, chunk[:, -1:, :].expand(B, need, F)], dim=1)
            lengths = torch.full((B,), chunk.size(1), dtype=torch.long, device=mel_input.device)
            torch.cuda.synchronize()
            t0 = time.time()
            out, _, state = self.emformer.infer(chunk, lengths, state)
            torch.cuda.synchronize()
            lat.append(time.time() - t0)
            rtf.append(lat[-1] / (seg * 0.02))
            if verbose:
                print("latency: {:.4f}, rtf: {:.4f}".format(lat[-1], rtf[-1]))
            outs.append(out[:, :emit, :])
            logits.append(self._last[1][:, :emit, :])
            pos += emit
        first = torch.cat(logits, dim=1)
        if self.mode == "both":
            return first, self.proj2(torch.cat(outs, dim=1)), lat
        if "proj" not in self._modules or not hasattr(self._modules["proj"], "weight"):
            return torch.cat(outs, dim=1), lat, rtf
        return first, lat, rtf

    @torch.inference_mode()
    def inference(self, mel_input):
        """emformer.py:48-98: chunked streaming over mel_input[B, T, F] -> proj(features) [B, T, out_dim], or the tuple
        (proj1(features), proj2(features)) when mode == 'both' (emformer.py:95-97)."""
        B, T, F = mel_input.shape
        seg, rc = self.segment_length, self.right_context_len
        pos, state, outs, logits = 0, None, [], []
        while pos < T:
            emit = min(seg, T - pos)
            look = min(rc, T - (pos + emit))
            real = emit + look
            chunk = mel_input[:, pos:pos + real, :]
            need = (seg + rc) - real
            if need > 0:
                chunk = torch.cat([chunk, chunk[:, -1:, :].expand(B, need, F)], dim=1)
            lengths = torch.full((B,), chunk.size(1), dtype=torch.long, device=mel_input.device)
            out, _, state = self.emformer.infer(chunk, lengths, state)
            outs.append(out[:, :emit, :])
            logits.append(self._last[1][:, :emit, :])      # the step projected these rows already (proj / proj1)
            pos += emit
        first = torch.cat(logits, dim=1)
        if self.mode == "both":
            return first, self.proj2(torch.cat(outs, dim=1))
        if "proj" not in self._modules or not hasattr(self._modules["proj"], "weight"):    # nn.Identity: input_dim == output_dim
            return torch.cat(outs, dim=1)
        return first

    def forward(self, mel_input, lengths):
        raise NotImplementedError("non-streaming Emformer.forward (training) is outside the hot path")
