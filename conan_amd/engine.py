"""Batched multi-stream serving engine: B independent utterance streams on one GPU, sharded by
slot range over the ranks of a torch.distributed job (one process per GPU).

Streams are independent units (private conv/KV state, no cross-stream arithmetic),
so the compute path has no collective.  The only exchange is the gather of finished audio to
rank 0 (RCCL `gather` over xGMI on GPUs, gloo in the CPU tests)."""
import os

import numpy as np
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous slot range [lo, hi) of `rank`: GPU g owns streams [g*B/W, (g+1)*B/W)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_distributed(backend=None):
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def gather_audio(wav_local, world, rank, dst=0):
    """Gather per-rank audio [b_r, samples] to `dst` (ragged b_r allowed); returns the concatenated
    [B, samples] tensor on dst, None elsewhere."""
    if world == 1:
        return wav_local
    counts = [torch.zeros(1, dtype=torch.int64, device=wav_local.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([wav_local.shape[0]], dtype=torch.int64, device=wav_local.device))
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    pad = wav_local
    if wav_local.shape[0] < mx:
        pad = torch.cat([wav_local, wav_local.new_zeros(mx - wav_local.shape[0], wav_local.shape[1])])
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)])


def gather_audio_equal(wav_local, world, rank, bufs=None, dst=0):
    """Fast path when every rank holds the same number of streams (the benchmark's weak scaling)."""
    if world == 1:
        return wav_local
    if rank == dst and bufs is None:
        bufs = [torch.empty_like(wav_local) for _ in range(world)]
    dist.gather(wav_local, bufs if rank == dst else None, dst=dst)
    return bufs


class _HostStream:
    """Stand-in for torch.cuda.Stream / Event on a host-only process group (gloo): everything is synchronous, so
    waits and records are no-ops.  Lets the CPU tests drive the very choreography the GPU benchmark runs."""

    def wait_event(self, ev):
        pass

    def wait_stream(self, s):
        pass

    def synchronize(self):
        pass

    def record(self, stream=None):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class AudioGatherRing:
    """Collection of finished audio on rank 0 (the path's only exchange) without serialising the chunk
    pipeline: audio of step j goes to buffer j % nb; gathers (RCCL on GPUs, gloo on CPU) are enqueued on a side stream
    behind `join()` (= "the audio of every step enqueued so far is complete"), and a buffer is handed out again only after
    the gather that read it has finished.

        ring = AudioGatherRing(lambda: torch.empty(B, samples, device=dev), world, rank, every=4)
        for j in range(steps):
            buf, fence = ring.acquire(j, fence=True)   # fence: the side stream, when a gather that read buf may be pending
            ... enqueue the step that writes buf (Streams.step_async(..., out_fence=fence)) ...
            ring.submit(j, join)                        # every `every`-th step: side stream: join(); gather(group); record
        ring.flush(steps - 1); ring.drain()

    `every` = steps per gather (SURVEY.md §8e allows per step or per utterance).  With every = 1 each step's buffer is gathered
    by itself (nb buffers).  With every = E > 1 the ring is 2 groups of E step buffers, contiguous in one allocation: the E
    buffers of a group travel in ONE collective after the group's last step while the next group's steps already write the
    other group, and the side stream is touched - one join, one gather, one event record, one output fence for the step that
    re-opens the group - once per E steps instead of once per step.  (Measured on one MI355X, world 1, where the gather itself
    is a no-op: the per-step form costs the 64-stream pipelined step 6.5 %)

    Pipelined steps should not take the wait on the current stream: it holds back the step's Emformer and decoder stages,
    which never touch the buffer, and drains the three-stage pipeline (measured: 1.81 -> 2.6 ms per step).  `acquire(j,
    fence=True)` returns the event recorded behind the gather that last read the buffer's group instead (None while no gather
    can be pending, and for every step but the first of a group) for `Streams.step_async(..., out_fence=fence)`: only the
    stage that writes the audio waits, and only for that gather.
    """

    def __init__(self, make_buffer, world, rank, nb=4, always=False, on_gathered=None, every=1):
        self.world, self.rank = world, rank
        self.every = max(1, int(every))
        self.nb = nb if self.every == 1 else 2 * self.every
        proto = make_buffer()
        self.pool = proto.new_zeros((self.nb,) + tuple(proto.shape))          # group g = pool[g * every : (g + 1) * every]
        self.bufs = [self.pool[i] for i in range(self.nb)]
        self.active = world > 1 or always
        self.cuda = proto.is_cuda
        self.on_gathered = on_gathered
        self._g = [proto.new_zeros((self.every,) + tuple(proto.shape)) for _ in range(world)] if (world > 1 and rank == 0) else None
        self.gbufs = None if self._g is None else [g[0] if self.every == 1 else g for g in self._g]   # what rank 0 last gathered, per rank
        ngroups = self.nb // self.every
        if self.active and self.cuda:
            self.comm = torch.cuda.Stream()
            self.done = [torch.cuda.Event() for _ in range(ngroups)]
        else:
            self.comm = _HostStream()
            self.done = [_HostStream() for _ in range(ngroups)]
        self.recorded = [False] * ngroups   # done[g] has been recorded at least once (an unrecorded event is no fence)
        self.submitted = 0            # gathers enqueued
        self.last_gathered = None     # last step whose audio has been handed to a gather
        self.last_sent = None         # the local tensor of that gather (a group of step buffers)
        self._next = 0                # first step not yet covered by a gather

    def acquire(self, j, fence=False):
        k = j % self.nb
        first = j % self.every == 0                       # the step that re-opens a group is the one that may collide with its gather
        # (a group whose gather was skipped - no submit() for it - has no recorded event: nothing to wait for)
        pending = self.active and j >= self.nb and first and self.recorded[k // self.every]
        if fence:       # the event recorded right behind the gather that read this group (NOT the side stream's tail: a later gather's
                        # join is already enqueued there, and it waits for the newest step)
            return self.bufs[k], (self.done[k // self.every] if (pending and self.cuda) else None)
        if pending:
            (torch.cuda.current_stream() if self.cuda else _HostStream()).wait_event(self.done[k // self.every])
        return self.bufs[k]

    def _gather(self, j, join, wait_current):
        """One collective for the steps self._next .. j (all in one group).  Only THEIR buffers travel: a flush() in the middle
        of a group must not read the buffers of the group's later steps, which may be written while the collective runs, and a
        group's remainder after a flush travels without the buffers the flush already sent."""
        g = (j % self.nb) // self.every
        lo, hi = max(0, self._next - (j - j % self.every)), j % self.every + 1
        grp = self.pool[g * self.every + lo:g * self.every + hi]
        dst = None if self._g is None else [b[lo:hi] for b in self._g]
        cur = torch.cuda.current_stream() if self.cuda else None
        ctxm = torch.cuda.stream(self.comm) if self.cuda else self.comm
        with ctxm:
            if wait_current and self.cuda:
                self.comm.wait_stream(cur)
            if join is not None:
                join()
            out = gather_audio_equal(grp, self.world, self.rank, dst)
            if self.on_gathered is not None and self.rank == 0:
                src = out if self.world > 1 else [grp]
                for js in range(self._next, j + 1):
                    self.on_gathered(js, [b[js % self.every - lo] for b in src])
            self.done[g].record(self.comm) if self.cuda else None
            self.recorded[g] = True
        if dst is not None:
            self.gbufs = [b[0] if self.every == 1 else b for b in dst]        # what rank 0 gathered last, per rank
        self.submitted += 1
        self.last_gathered, self.last_sent, self._next = j, grp, j + 1

    def submit(self, j, join=None, wait_current=False):
        """join: callable making the CURRENT stream wait for the audio of every step enqueued so far (Streams.join);
        wait_current: the audio was produced on the stream that is current now (blocking steps) - the side stream waits for
        it.  A gather is enqueued when step j completes a group."""
        if not self.active:
            self.last_gathered = j
            return
        self._join, self._wait_current = join, wait_current
        if (j + 1) % self.every == 0:
            self._gather(j, join, wait_current)

    def flush(self, j):
        """Gather the steps up to j that no gather has covered yet (a timed region that ends inside a group)."""
        if self.active and j >= self._next:
            self._gather(j, getattr(self, "_join", None), getattr(self, "_wait_current", False))

    def drain(self):
        if self.active:
            self.comm.synchronize()


class StreamingVoiceConversionEngine:
    """The chunk loop of StreamingVoiceConversion.infer_once (inference/Conan.py:72-166) for many
    streams at once: mel in -> (wav, mel, codes) out, state carried in a conan_streams handle."""

    def __init__(self, ctx, n_streams, max_ref_frames=256, max_frames=None, arith="auto", flags=0, dev_plan=None):
        self.ctx = ctx
        self.n = n_streams
        self.arith = arith          # conan_streams_opts.arith of the stream-set: 'auto' | 'f32' | 'limb'
        self.flags, self.dev_plan = flags, dev_plan      # conan_streams_opts.flags (_lib.STREAMS_*) / .dev_plan
        self.st = ctx.streams(n_streams, max_frames=max(ctx.cfg.emf_segment, max_frames or 0), max_ref_frames=max_ref_frames, arith=arith,
                              flags=flags, dev_plan=dev_plan)
        self.slots = list(range(n_streams))
        self.seg, self.rc = ctx.cfg.emf_segment, ctx.cfg.emf_right_context

    def start(self, ref_mel, ref_len=None):
        self.st.reset(self.slots)
        self.st.set_reference(self.slots, ref_mel, ref_len)

    def chunks(self, src_mel):
        """inference/Conan.py:95-110: (pos, emit, chunk[B, seg+rc, 80]) with repeat-last padding."""
        B, T, F = src_mel.shape
        pos = 0
        while pos < T:
            emit = min(self.seg, T - pos)
            look = min(self.rc, T - (pos + emit))
            real = emit + look
            chunk = src_mel[:, pos:pos + real]
            need = self.seg + self.rc - real
            if need > 0:
                chunk = torch.cat([chunk, chunk[:, -1:].expand(B, need, F)], 1)
            yield pos, emit, chunk.contiguous()
            pos += emit

    @torch.no_grad()
    def windowed_step(self, chunk, ctx_codes, return_mel=False):
        """One chunk in the bounded-window mode of BASELINE.json configs[1] / configs[4] ("80 ms chunk + 160 ms context",
        "40 ms chunk / 320 ms context window"): the Emformer stays stateful (its left context is its own
        K/V cache); the Conan decoder and the vocoder are reset and fed `ctx_codes` ([B, ctx] int32, the codes of the
        preceding frames, possibly fewer at the start of an utterance) + this chunk's codes in one step; only the last
        `seg` frames are kept.  The oracle of this mode is the reference module fed the same window.
        Returns (codes [B, seg], wav [B, seg*hop]) (+ mel [B, seg, 80])."""
        st, seg, hop = self.st, self.seg, self.ctx.hop
        _, _, codes = st.emformer_step(self.slots, chunk, want_out=False, want_logits=False)
        win = torch.cat([ctx_codes.to(codes.dtype), codes], 1) if ctx_codes is not None and ctx_codes.shape[1] else codes
        st.reset(self.slots, which=2 | 4)
        mel = st.decoder_step(self.slots, win)
        wav = st.hifigan_step(self.slots, mel)
        out = (codes, wav[:, -seg * hop:])
        return out + (mel[:, -seg:],) if return_mel else out

    @torch.no_grad()
    def infer(self, src_mel, ref_mel, ref_len=None, pipelined=True):
        """src_mel [B,T,80], ref_mel [B,Tr,80] (cuda) -> wav [B, T*hop], mel [B,T,80], codes [B,T].

        The whole source is available here, so by default the chunks are issued as pipelined steps
        (conan_step_async): the Emformer + decoder of chunk t+1 overlap the vocoder of chunk t.  The
        results are bit-identical to the blocking loop (pipelined=False)."""
        if self.ctx.cfg.voc_upsample == 2:
            return self._infer_prefix_vocoder(src_mel, ref_mel, ref_len)
        self.start(ref_mel, ref_len)
        B = src_mel.shape[0]
        hop, nm = self.ctx.hop, self.ctx.cfg.num_mels
        wavs, mels, codes = [], [], []
        for pos, emit, chunk in self.chunks(src_mel):
            if pipelined:
                c = torch.empty(B, self.seg, dtype=torch.int32, device=src_mel.device)
                m = torch.empty(B, emit, nm, device=src_mel.device)
                w = torch.empty(B, emit * hop, device=src_mel.device)
                self.st.step_async(self.slots, chunk, w, emit=emit, codes=c, mel_out=m)
            else:
                c, m, w = self.st.step(self.slots, chunk, emit=emit)
            wavs.append(w)
            mels.append(m)
            codes.append(c[:, :emit])
        if pipelined:
            self.st.join()
        return torch.cat(wavs, 1), torch.cat(mels, 1), torch.cat(codes, 1)

    @torch.no_grad()
    def _infer_prefix_vocoder(self, src_mel, ref_mel, ref_len=None):
        """Vocoders that look ahead (`upsample: nn`, CausalUpsampleBlock1) cannot carry state from chunk to chunk; the
        reference loop does not need them to: it runs the vocoder on ALL mel frames so far and keeps the samples of the
        current chunk (inference/Conan.py:147-155).  Same here: Emformer and decoder step statefully, the vocoder is reset
        and run over the prefix (O(T^2) like the reference; rings sized for the whole utterance)."""
        B, T, _ = src_mel.shape
        if self.st.max_frames < T:
            mr = self.st.max_ref_frames
            self.st.close()
            self.st = self.ctx.streams(self.n, max_frames=T, max_ref_frames=mr, arith=self.arith, flags=self.flags, dev_plan=self.dev_plan)
        self.start(ref_mel, ref_len)
        hop = self.ctx.hop
        wavs, mels, codes = [], [], []
        for pos, emit, chunk in self.chunks(src_mel):
            _, _, c = self.st.emformer_step(self.slots, chunk, want_out=False, want_logits=False)
            m = self.st.decoder_step(self.slots, c[:, :emit].contiguous())
            mels.append(m)
            codes.append(c[:, :emit])
            self.st.reset(self.slots, which=4)
            w = self.st.hifigan_step(self.slots, torch.cat(mels, 1))
            wavs.append(w[:, pos * hop:(pos + emit) * hop])
        return torch.cat(wavs, 1), torch.cat(mels, 1), torch.cat(codes, 1)
