"""Thin Python handles over the C-ABI: Context (weights) and Streams (per-slot state).

PyTorch-ROCm is used only as the device-memory / stream plumbing: every tensor argument is a
CUDA(=HIP) torch tensor whose data_ptr() is handed to libconan_hip.so, and work is enqueued on
torch's current HIP stream.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("conan_amd needs a HIP device (MI355X); there is no CPU fallback")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _i32(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int32))
    return a, a.ctypes.data_as(C.c_void_p)


class Context:
    """conan_ctx: packed weights of up to three models on one device."""

    def __init__(self, conan_hp=None, hifigan_hp=None, device=0, emformer=True, conan=True, hifigan=True):
        _require_gpu()
        self.lib = _lib.lib()
        self.cfg = _lib.make_cfg(conan_hp, hifigan_hp, emformer, conan, hifigan)
        self.device = int(device)
        self.conan_hp, self.hifigan_hp = conan_hp, hifigan_hp
        h = C.c_void_p()
        _lib.check(self.lib.conan_ctx_create(self.device, C.byref(self.cfg), C.byref(h)))
        self.h = h
        self.finalized = False

    def load_state_dict(self, model, sd):
        """model in {'emformer','conan','hifigan'}; sd maps the reference's state_dict keys to
        numpy arrays / torch tensors (utils/commons/ckpt_utils.py:26-66)."""
        for k, v in sd.items():
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if a.dtype.kind != "f":
                continue
            a = np.ascontiguousarray(a, dtype=np.float32)
            shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
            _lib.check(self.lib.conan_ctx_load_tensor(self.h, f"{model}.{k}".encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        return self

    def finalize(self):
        _lib.check(self.lib.conan_ctx_finalize(self.h))
        self.finalized = True
        return self

    @property
    def hop(self):
        return self.lib.conan_hop_size(self.h)

    @property
    def weight_bytes(self):
        return self.lib.conan_ctx_weight_bytes(self.h)

    def wav2mel(self, wav, fft_size=1024, hop_size=320, win_length=1024, num_mels=80, fmin=80, fmax=7600, sample_rate=16000,
                eps=1e-6, mel_vmin=-6.0, mel_vmax=1.5, framing=0, natural_log=False, mag_eps=0.0):
        """Mel front-end on the GPU (conan_wav2mel): wav cuda float32 [n, samples] -> mel [n, frames, num_mels].
        Defaults: clip(librosa_wav2spec(wav)['mel'], mel_vmin, mel_vmax) of inference/Conan.py:57-70 (loud_norm off),
        frames = 1 + samples // hop.  framing=1, natural_log=True, mag_eps=1e-9: the torch.stft front-end of
        inference/Conan_previous.py:100-121 (reflect padding, center=False), frames = samples // hop."""
        wav = wav.to(torch.device("cuda", self.device), torch.float32).contiguous()
        if wav.dim() == 1:
            wav = wav[None]
        n, samples = wav.shape
        mc = _lib.MelCfg(fft_size, hop_size, win_length, num_mels, sample_rate, float(fmin), float(fmax), eps, mel_vmin, mel_vmax,
                         int(framing), int(bool(natural_log)), float(mag_eps))
        if framing == 0:
            frames = 1 + samples // hop_size
        else:
            frames = max(0, (samples + 2 * ((fft_size - hop_size) // 2) - fft_size) // hop_size + 1)
        mel = torch.empty(n, frames, num_mels, device=wav.device)
        got = C.c_int32(0)
        _lib.check(self.lib.conan_wav2mel(self.h, C.byref(mc), _ptr(wav), n, samples, _ptr(mel), C.byref(got), _stream()))
        assert got.value == frames
        return mel

    def streams(self, max_slots, max_frames=4, max_ref_frames=256, arith="auto", flags=0, dev_plan=None):
        """A stream-set.  arith: 'auto' (library default), 'f32' (f32-input MFMA everywhere) or 'limb' (fp32 products of the
        vocoder's matrix kernels as bf16 limb products) - conan_streams_opts.arith; flags: _lib.STREAMS_*; dev_plan: developer /
        test switches of the launch plan, "NAME=value;..." (conan_streams_opts.dev_plan; None in deployments)."""
        return Streams(self, max_slots, max_frames, max_ref_frames, arith, flags, dev_plan)

    def close(self):
        if getattr(self, "h", None):
            self.lib.conan_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Streams:
    """conan_streams: per-slot streaming state + the step functions."""

    def __init__(self, ctx, max_slots, max_frames=4, max_ref_frames=256, arith="auto", flags=0, dev_plan=None):
        self.ctx, self.lib = ctx, ctx.lib
        self.max_slots, self.max_frames, self.max_ref_frames = max_slots, max_frames, max_ref_frames
        if arith not in _lib.ARITH_NAMES:
            raise ValueError(f"arith must be one of {sorted(_lib.ARITH_NAMES)}, got {arith!r}")
        opts = _lib.StreamsOpts(_lib.ABI_VERSION, _lib.ARITH_NAMES[arith], int(flags), 0, dev_plan.encode() if dev_plan else None)
        h = C.c_void_p()
        _lib.check(self.lib.conan_streams_create_opts(ctx.h, max_slots, max_frames, max_ref_frames, C.byref(opts), C.byref(h)))
        self._keep = []   # buffers of pipelined steps in flight (released by join())
        self.h = h
        self.dev = torch.device("cuda", ctx.device)
        c = ctx.cfg
        self.seg, self.rc = c.emf_segment, c.emf_right_context

    @property
    def state_bytes(self):
        return self.lib.conan_streams_state_bytes(self.h)

    @property
    def arith(self):
        """'f32' or 'limb': the arithmetic this stream-set's vocoder launches use where both forms exist ('auto' resolved)."""
        return {_lib.ARITH_F32: "f32", _lib.ARITH_LIMB: "limb"}[_lib.check(self.lib.conan_streams_arith(self.h))]

    def _release(self):
        """Buffers of pipelined steps may be dropped once the current torch stream waits for the library's internal
        streams (every stream-ordered entry point joins them in C).  Those streams are invisible to torch's caching
        allocator, so each buffer is first marked as in use on the current stream: its block then returns to the
        pool only after work enqueued here - which sits behind the join - has completed."""
        if self._keep:
            cur = torch.cuda.current_stream()
            for group in self._keep:
                for t in group:
                    if t is not None and t.is_cuda:
                        t.record_stream(cur)
            self._keep.clear()

    def reset(self, slots, which=7):
        a, p = _i32(slots)
        _lib.check(self.lib.conan_streams_reset(self.h, p, len(a), which, _stream()))
        self._release()

    def set_reference(self, slots, ref_mel, ref_len=None):
        """ref_mel: cuda float32 [n, Tr, 80]."""
        a, p = _i32(slots)
        ref_mel = ref_mel.to(self.dev, torch.float32).contiguous()
        n, tr = ref_mel.shape[0], ref_mel.shape[1]
        if ref_len is None:
            ref_len = [tr] * n
        l, lp = _i32(ref_len)
        _lib.check(self.lib.conan_set_reference(self.h, p, len(a), _ptr(ref_mel), lp, tr, _stream()))
        self._release()

    def emformer_step(self, slots, chunk, want_out=True, want_logits=True, want_codes=True):
        a, p = _i32(slots)
        n = len(a)
        c = self.ctx.cfg
        chunk = chunk.to(self.dev, torch.float32).contiguous()
        assert chunk.shape == (n, self.seg + self.rc, c.emf_input_dim), chunk.shape
        out = torch.empty(n, self.seg, c.emf_input_dim, device=self.dev) if want_out else None
        logits = torch.empty(n, self.seg, c.emf_output_dim, device=self.dev) if want_logits else None
        codes = torch.empty(n, self.seg, dtype=torch.int32, device=self.dev) if want_codes else None
        _lib.check(self.lib.conan_emformer_step(self.h, p, n, _ptr(chunk), _ptr(out), _ptr(logits), _ptr(codes), _stream()))
        self._release()
        return out, logits, codes

    def emformer_project(self, head, x):
        """x [..., D] (cuda float32) through the checkpoint's output head `head` ('proj' / 'proj1' / 'proj2'):
        conan_emformer_project, a k = 1 conv on the MFMA path."""
        D = self.ctx.cfg.emf_input_dim
        x2 = x.to(self.dev, torch.float32).reshape(-1, D).contiguous()
        K = int(self.lib.conan_emformer_head_dim(self.h, head.encode()))
        if K <= 0:
            raise _lib.ConanError(_lib.ERR_MISSING, f"no Emformer output head '{head}'")
        y = torch.empty(x2.shape[0], K, device=self.dev)
        _lib.check(self.lib.conan_emformer_project(self.h, head.encode(), _ptr(x2), x2.shape[0], _ptr(y), _stream()))
        self._release()
        return y.reshape(*x.shape[:-1], K)

    def decoder_step(self, slots, codes, taps=False):
        a, p = _i32(slots)
        n = len(a)
        c = self.ctx.cfg
        codes = codes.to(self.dev, torch.int32).contiguous()
        T = codes.shape[1]
        mel = torch.empty(n, T, c.num_mels, device=self.dev)
        if not taps:
            _lib.check(self.lib.conan_decoder_step(self.h, p, n, T, _ptr(codes), _ptr(mel), None, None, None, None, _stream()))
            self._release()
            return mel
        S = (self.max_ref_frames + 3) // 4 if self.max_ref_frames >= 4 else 1
        out = {"uv_pred": torch.empty(n, T, 2, device=self.dev), "f0_denorm_pred": torch.empty(n, T, device=self.dev),
               "pitch_bins": torch.empty(n, T, dtype=torch.int32, device=self.dev),
               "decoder_inp": torch.empty(n, T, c.hidden_size, device=self.dev),
               "content_embed_proj": torch.empty(n, T, c.hidden_size, device=self.dev),
               "attn": [torch.empty(n, T, S, device=self.dev) for _ in range(2)]}
        t = _lib.DecoderTaps()
        t.uv_pred, t.f0_denorm_pred, t.pitch_bins = out["uv_pred"].data_ptr(), out["f0_denorm_pred"].data_ptr(), out["pitch_bins"].data_ptr()
        t.decoder_inp, t.content_embed_proj = out["decoder_inp"].data_ptr(), out["content_embed_proj"].data_ptr()
        t.attn[0], t.attn[1] = out["attn"][0].data_ptr(), out["attn"][1].data_ptr()
        _lib.check(self.lib.conan_decoder_step_taps(self.h, p, n, T, _ptr(codes), _ptr(mel), C.byref(t), _stream()))
        self._release()
        return mel, out

    def style_embed(self, slots):
        """style_embed [n, H] of the slots' current reference (Conan.encode_spk_embed, cached by set_reference)."""
        a, p = _i32(slots)
        out = torch.empty(len(a), self.ctx.cfg.hidden_size, device=self.dev)
        _lib.check(self.lib.conan_get_style(self.h, p, len(a), _ptr(out), None, _stream()))
        self._release()
        return out

    def set_style(self, slots, style):
        """Conan.forward(spk_embed=...): override the cached global style vector, style [n, H] (cuda)."""
        a, p = _i32(slots)
        style = style.to(self.dev, torch.float32).reshape(len(a), self.ctx.cfg.hidden_size).contiguous()
        _lib.check(self.lib.conan_set_style(self.h, p, len(a), _ptr(style), _stream()))
        self._release()

    def prosody_ids(self, slots):
        """VQ indices of the slots' prosody tokens: (ids int32 [n, max_tokens] (-1 padded), counts int32 [n])."""
        a, p = _i32(slots)
        S = (self.max_ref_frames + 3) // 4 if self.max_ref_frames >= 4 else 1
        ids = torch.empty(len(a), S, dtype=torch.int32, device=self.dev)
        cnt = torch.empty(len(a), dtype=torch.int32, device=self.dev)
        _lib.check(self.lib.conan_get_prosody_ids(self.h, p, len(a), _ptr(ids), _ptr(cnt), _stream()))
        self._release()
        return ids, cnt

    def hifigan_step_taps(self, slots, mel, stage_out=False):
        """hifigan_step plus the generator's intermediate tensors: (wav, pre_tanh, conv_pre_act [n,T,C0], [ups_i [n,T*rate_i,C_i]]);
        stage_out=True appends [stage_out_i [n,T*rate_i,C_i]] = leaky_relu(mean of the stage's ResBlocks)."""
        a, p = _i32(slots)
        n = len(a)
        c = self.ctx.cfg
        mel = mel.to(self.dev, torch.float32).contiguous()
        T = mel.shape[1]
        hop = self.ctx.hop
        wav = torch.empty(n, T * hop, device=self.dev)
        pre = torch.empty(n, T * hop, device=self.dev)
        cpre = torch.empty(n, T, c.voc_initial_channel, device=self.dev)
        ups, outs, ch_, rate = [], [], c.voc_initial_channel, 1
        t = _lib.HifiganTaps()
        t.conv_pre_act = cpre.data_ptr()
        for i in range(c.voc_num_ups):
            ch_ //= 2
            rate *= c.voc_up_rates[i]
            ups.append(torch.empty(n, T * rate, ch_, device=self.dev))
            t.ups[i] = ups[-1].data_ptr()
            if stage_out:
                outs.append(torch.empty(n, T * rate, ch_, device=self.dev))
                t.stage_out[i] = outs[-1].data_ptr()
        _lib.check(self.lib.conan_hifigan_step_taps(self.h, p, n, T, _ptr(mel), _ptr(wav), _ptr(pre), C.byref(t), _stream()))
        self._release()
        return (wav, pre, cpre, ups, outs) if stage_out else (wav, pre, cpre, ups)

    def hifigan_step(self, slots, mel, want_pre_tanh=False, out=None):
        """mel: cuda float32 [n, frames, 80] -> wav [n, frames*hop]."""
        a, p = _i32(slots)
        n = len(a)
        mel = mel.to(self.dev, torch.float32).contiguous()
        T = mel.shape[1]
        hop = self.ctx.hop
        wav = out if out is not None else torch.empty(n, T * hop, device=self.dev)
        pre = torch.empty(n, T * hop, device=self.dev) if want_pre_tanh else None
        _lib.check(self.lib.conan_hifigan_step(self.h, p, n, T, _ptr(mel), _ptr(wav), _ptr(pre), _stream()))
        self._release()
        return (wav, pre) if want_pre_tanh else wav

    def step(self, slots, mel_chunk, emit=None, codes=None, mel_out=None, wav_out=None):
        """Fused chunk step (one iteration of inference/Conan.py:95-156 for all slots)."""
        a, p = _i32(slots)
        n = len(a)
        emit = self.seg if emit is None else emit
        hop = self.ctx.hop
        if codes is None:
            codes = torch.empty(n, self.seg, dtype=torch.int32, device=self.dev)
        if mel_out is None:
            mel_out = torch.empty(n, emit, self.ctx.cfg.num_mels, device=self.dev)
        if wav_out is None:
            wav_out = torch.empty(n, emit * hop, device=self.dev)
        _lib.check(self.lib.conan_step(self.h, p, n, emit, _ptr(mel_chunk), _ptr(codes), _ptr(mel_out), _ptr(wav_out), _stream()))
        self._release()
        return codes, mel_out, wav_out

    def step_async(self, slots, mel_chunk, wav_out, emit=None, codes=None, mel_out=None, out_fence=None):
        """Pipelined chunk step (conan_step_async): returns at once; the front-end of the next call overlaps this
        call's vocoder.  `wav_out` (and the optional outputs) must be caller-owned tensors kept alive until join().
        out_fence: a torch.cuda.Stream whose work enqueued so far - or a recorded torch.cuda.Event that - must finish before this
        step's vocoder writes `wav_out` (conan_streams_output_fence / _event: e.g. the collective that still reads the buffer)."""
        a, p = _i32(slots)
        n = len(a)
        emit = self.seg if emit is None else int(emit)
        mel_chunk = mel_chunk.to(self.dev, torch.float32).contiguous()
        assert mel_chunk.shape == (n, self.seg + self.rc, self.ctx.cfg.emf_input_dim), mel_chunk.shape
        assert wav_out.is_cuda and wav_out.is_contiguous() and wav_out.numel() >= n * emit * self.ctx.hop
        self._keep.append((mel_chunk, wav_out, codes, mel_out))
        if out_fence is not None:
            if isinstance(out_fence, torch.cuda.Event):      # an event recorded behind the one operation that read wav_out
                _lib.check(self.lib.conan_streams_output_fence_event(self.h, C.c_void_p(out_fence.cuda_event)))
            else:
                _lib.check(self.lib.conan_streams_output_fence(self.h, C.c_void_p(out_fence.cuda_stream)))
        _lib.check(self.lib.conan_step_async(self.h, p, n, emit, _ptr(mel_chunk), _ptr(codes), _ptr(mel_out), _ptr(wav_out), _stream()))

    def join(self):
        """Make the current torch stream wait for every pipelined step enqueued so far."""
        _lib.check(self.lib.conan_streams_join(self.h, _stream()))
        self._release()

    def profile_mark(self):
        """One cnk::profile_mark_kernel dispatch on the current stream (marker for rocprofv3 post-processing)."""
        _lib.check(self.lib.conan_profile_mark(self.h, _stream()))

    def step_clock(self, capacity):
        """Record a completion stamp per pipelined step on the internal vocoder stream (conan_step_clock); 0 = off."""
        _lib.check(self.lib.conan_step_clock(self.h, int(capacity)))

    def step_clock_read(self, cap=4096):
        """Intervals (ms) between the completions of consecutive pipelined steps since step_clock()."""
        buf = (C.c_double * cap)()
        n = _lib.check(self.lib.conan_step_clock_read(self.h, buf, cap))
        return [buf[i] for i in range(n)]

    def step_timeline(self, capacity):
        """Record start / end events of the three stages of every following pipelined step (conan_step_timeline); 0 = off."""
        _lib.check(self.lib.conan_step_timeline(self.h, int(capacity)))

    def step_timeline_read(self, cap=1024):
        """Per recorded step [emf start, emf end, dec start, dec end, voc start, voc end] in ms since the first event."""
        buf = (C.c_double * (cap * 6))()
        n = _lib.check(self.lib.conan_step_timeline_read(self.h, buf, cap))
        return [[buf[i * 6 + e] for e in range(6)] for i in range(n)]

    def profile_begin(self):
        _lib.check(self.lib.conan_profile_begin(self.h))

    def profile_end(self):
        """-> (conv kernel ms, algorithmic conv FLOPs, conv launches) since profile_begin()."""
        ms, fl, nl = C.c_double(), C.c_double(), C.c_int64()
        _lib.check(self.lib.conan_profile_end(self.h, C.byref(ms), C.byref(fl), C.byref(nl)))
        return ms.value, fl.value, nl.value

    def profile_kernels(self):
        """Per kernel instantiation after profile_end(): list of (name, ms, flops, launches)."""
        out, i = [], 0
        while True:
            buf = C.create_string_buffer(128)
            ms, fl, nl = C.c_double(), C.c_double(), C.c_int64()
            rc = _lib.check(self.lib.conan_profile_kernel(self.h, i, buf, 128, C.byref(ms), C.byref(fl), C.byref(nl)))
            if rc == 0:
                return out
            out.append((buf.value.decode(), ms.value, fl.value, nl.value))
            i += 1

    def close(self):
        if getattr(self, "h", None):
            self.lib.conan_streams_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
