#!/usr/bin/env python3
"""Headline benchmark: chunks/sec + p50 per-chunk latency of the chunkwise streaming VC path
(Emformer -> Conan -> causal HiFi-GAN, 80 ms chunks @16 kHz) on N MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one fused chunk step (conan_step) for all B streams of a rank: B chunks of 80 ms.
Inputs (synthetic mel chunks, reference mels, random-init weights of the egs/conan_emformer.yaml +
egs/hifi_16k320_shuffle.yaml architectures) are resident in HBM before the timed region.
Streams are sharded across ranks (weak scaling: B per GPU fixed); the only collective is the RCCL
gather of the finished audio to rank 0, inside the timed region.  Rank 0 prints ONE JSON line.

Timing: W untimed warm-up steps, barrier + synchronize, exactly K timed steps, barrier + synchronize (Runner.timed).  Before the W
steps the stream-set is primed once as part of its setup (Runner.prime, --prime N, default 100 pipelined steps = 0.13 s; reported as
config.priming_steps): the first steps of a freshly built process are not the stream's throughput (the host enqueues them slower, the
GPU comes out of seconds of idling).  What K and W do to the reading, one box: K = 200 / W = 30 1.308 ms per step, K = 60 / W = 10
1.33 (the defaults), K = 20 / W = 5 1.36-1.38 primed and 1.41-1.42 unprimed - a timed region starts from an empty pipeline, so it pays the
first chunk's Emformer + decoder latency (0.73 ms) before its first vocoder step: 0.037 ms per step at K = 20, 0.004 at K = 200.
Both readings are in the line (round 5): the very same timed(K, W) runs once BEFORE the priming pass -> roofline.ms_per_step_unprimed
(exactly what `--warmup W` alone gives), and once after it -> ms_per_step / value.  roofline.vocoder_alone_ms is the vocoder stage's
launches back to back on one stream (conan_hifigan_step loop, no front-end resident), roofline.frontend_cost_ms = ms_per_step minus it:
what the Emformer + decoder stages cost the pipelined step.
"""
import argparse
import json
import os
import statistics
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue
# are served in submission order.  The pipelined step uses three internal streams beside the caller's; with the gather's
# side stream (N > 1 ranks) that is five, two of them alias, and a stage ends up queued behind another stage's event wait:
# measured 2.59 instead of 1.85 ms per step.  Has to be set before the HIP runtime initialises (conan_amd/__init__.py does
# the same for callers that import the package first).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# multi-process GPU work on this pool needs dmabuf IPC (RCCL otherwise fails with hipIpcGetMemHandle: invalid argument)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA, dense
# bf16 MFMA, dense: 16x the f32-input rate (MI355X_MICROARCH.md, Matrix cores: "F32 ... 1/16 of BF16"; ~2.5 PF spec).  The limb
# kernels (resblock_limb.hip) compute every fp32 product as SIX bf16 products, so their ceiling in ALGORITHMIC fp32 FLOP/s is a
# sixth of it; `achieved` stays algorithmic (2 * M * N * K of the fp32 convolution), never the 6x executed bf16 FLOPs.
PEAK_BF16_MFMA_TFLOPS = 16 * PEAK_F32_MFMA_TFLOPS
PEAK_LIMB_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0


def kernel_peak(name):
    """(peak in algorithmic TFLOP/s, what it is) of a matrix kernel by its name."""
    if "resblock_limb" in name or "conv_limb" in name or "conv_tall" in name:      # (conv_limb_kernel and conv_limb_sk_kernel, its build with the split-K tail)
        return PEAK_LIMB_TFLOPS, "bf16 MFMA dense / 6 limb products per fp32 product"
    return PEAK_F32_MFMA_TFLOPS, "f32 MFMA dense"
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
N_FRAMES, N_REF = 151, 151     # 3 s source + 3 s reference at 50 frames/s
# algorithmic work per chunk per stream (stateful, seg = 4) and bytes per step
GFLOP_PER_FRAME = 2.63 / 4     # vocoder 315.85 + Conan 9.6 MMAC per frame + Emformer 13.07 MMAC per chunk
WEIGHT_BYTES_PER_STEP = 168e6  # vocoder 119.8 MB + Conan per-chunk subset 39 MB + Emformer 8.6 MB, read once per step
STATE_BYTES_PER_STREAM = 1.2e6  # state read+write + I/O per stream per step

# BASELINE.json configs -> bench workloads.  `window` = context frames of the windowed mode
# (Conan decoder + vocoder reset and fed window + chunk frames per step; the Emformer stays stateful), 0 = stateful.
WORKLOADS = {
    "b64": dict(streams=64, chunk_ms=80, window=0, config="BASELINE.json configs[2] (x8 GPUs = configs[3])",
                desc="batch=64 concurrent streams per GPU, 80 ms chunk (seg 4 + rc 2 frames), stateful full Emformer->Conan->HiFi-GAN pipeline"),
    "b1": dict(streams=1, chunk_ms=80, window=0, config="BASELINE.json configs[1], stateful mode",
               desc="batch=1 stream, 80 ms chunk (seg 4 + rc 2 frames), stateful full Emformer->Conan->HiFi-GAN pipeline"),
    "b1win": dict(streams=1, chunk_ms=80, window=8, config="BASELINE.json configs[1], 160 ms context window",
                  desc="batch=1 stream, 80 ms chunk + 160 ms context: Conan/vocoder reset + 12 frames per step, Emformer stateful"),
    "b128s2": dict(streams=128, chunk_ms=40, window=0, config="BASELINE.json configs[4], stateful mode",
                   desc="batch=128 streams, 40 ms chunk (seg 2 + rc 2 frames), stateful full pipeline"),
    "b128s2win": dict(streams=128, chunk_ms=40, window=16, config="BASELINE.json configs[4], 320 ms context window",
                      desc="batch=128 streams, 40 ms chunk + 320 ms context: Conan/vocoder reset + 18 frames per step, Emformer stateful"),
    # extra, non-parity datapoint: the Emformer with torchaudio's memory bank enabled, which the
    # reference's constructor never does (modules/Emformer/emformer.py:14-22)
    "b128s2mem4": dict(streams=128, chunk_ms=40, window=0, memory=4, config="BASELINE.json configs[4] + Emformer max_memory_size=4 (not a reference configuration)",
                       desc="batch=128 streams, 40 ms chunk (seg 2 + rc 2 frames), stateful, Emformer memory bank of 4 segments"),
}


def build_context(device, chunk_ms=80, memory=0):
    from conan_amd import configs, synth
    from conan_amd.runtime import Context
    chp, vhp = dict(configs.conan_hparams(), chunk_size=chunk_ms, emformer_max_memory_size=memory), configs.hifigan_hparams()
    ctx = Context(chp, vhp, device)
    ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
    ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    return ctx, chp, vhp


STREAM_OPTS = {"flags": 0, "dev_plan": None}      # conan_streams_opts.flags / .dev_plan of every stream-set this run creates (--fixed-plan, --dev-plan)


def make_engine(ctx, B, first_stream, window=0, arith="auto"):
    """B streams with their reference set and every full chunk of the 3 s utterance staged in HBM."""
    from conan_amd import synth
    from conan_amd.engine import StreamingVoiceConversionEngine
    eng = StreamingVoiceConversionEngine(ctx, B, max_ref_frames=N_REF + 1, max_frames=window + ctx.cfg.emf_segment if window else None, arith=arith,
                                         flags=STREAM_OPTS["flags"], dev_plan=STREAM_OPTS["dev_plan"])
    src = np.concatenate([synth.mel(N_FRAMES, 1234 + first_stream + s) for s in range(B)])
    ref = np.concatenate([synth.mel(N_REF, 4321 + first_stream + s) for s in range(B)])
    src = torch.from_numpy(src).cuda()
    eng.start(torch.from_numpy(ref).cuda())
    chunks = [c for _, emit, c in eng.chunks(src) if emit == eng.seg]
    return eng, chunks


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline():
    """BASELINE.md §3 / SURVEY.md §8(d) Config 1: the oracle's loop on the host cores, B = 1, same synthetic 3 s
    utterance, two variants - reference semantics (inference/Conan.py:95-156: prefix re-run of Conan incl. the style
    encoders and of the vocoder every chunk, numpy hops) and stateful (the arithmetic the GPU path performs) - at the
    host's best thread count ("all-core": PyTorch's CPU convolutions collapse when over-threaded on a 256-thread
    host, so a short probe picks among 8/16/32/64/all) and at 1 thread.  One warm-up + timed runs, median.  The samples
    are bounded so that the default bench run stays within minutes; each entry says what it covered."""
    from conan_amd import configs, synth
    from oracle import emformer as oemf
    from oracle import hifigan as ohifi
    from oracle import loop as oloop
    from oracle.common import to_torch_sd
    chp, vhp = configs.conan_hparams(), configs.hifigan_hparams()
    esd = to_torch_sd(synth.emformer_state_dict(chp, 0))
    csd = to_torch_sd(synth.conan_state_dict(chp, 0))
    vsd = to_torch_sd(synth.hifigan_state_dict(vhp, 0))
    cfg = oemf.EmformerCfg(chp)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    probe = torch.from_numpy(synth.mel(16, 5)).transpose(1, 2)
    best_t, best_dt = 1, float("inf")
    for nt in sorted({min(avail, c) for c in (8, 16, 32, 64, avail)}):
        torch.set_num_threads(nt)
        ohifi.generator_forward(vsd, vhp, probe)
        t0 = time.perf_counter()
        ohifi.generator_forward(vsd, vhp, probe)
        dtp = time.perf_counter() - t0
        if dtp < best_dt:
            best_t, best_dt = nt, dtp
        if dtp > 5.0:
            break
    src, ref = synth.mel(N_FRAMES, 1234)[0], synth.mel(N_REF, 4321)[0]
    seg = cfg.segment_length

    def run(fn, n_chunks, runs, warm_chunks):
        """-> (median chunks/s over `runs` passes of the first n_chunks chunks, p50 ms, p95 ms of the per-chunk times)"""
        def once(k):
            stamps = [time.perf_counter()]
            fn(esd, cfg, csd, chp, vsd, vhp, src[:k * seg + (2 if k * seg + 2 <= len(src) else 0)], ref, on_chunk=lambda: stamps.append(time.perf_counter()))
            lat = [b - a for a, b in zip(stamps[:-1], stamps[1:])][:k]
            return k / sum(lat), lat
        once(warm_chunks)
        rates, lats = [], []
        for _ in range(runs):
            r, l = once(n_chunks)
            rates.append(r); lats += l
        lats.sort()
        return statistics.median(rates), lats[len(lats) // 2] * 1e3, lats[min(len(lats) - 1, int(len(lats) * 0.95))] * 1e3

    out = {"unit": "chunks/s", "kind": "port", "cpu_model": cpu_model(), "host_threads_available": avail}
    torch.set_num_threads(best_t)
    v, p50, p95 = run(oloop.infer_once_ref, 37, 3, 6)
    out.update({"value": v, "cores": best_t, "p50_ms": p50, "p95_ms": p95})
    sv, sp50, sp95 = run(oloop.infer_once_stateful, 37, 5, 6)
    out.update({"stateful_value": sv, "stateful_p50_ms": sp50, "stateful_p95_ms": sp95})
    torch.set_num_threads(1)
    v1, p1, _ = run(oloop.infer_once_ref, 6, 1, 2)
    s1, sp1, _ = run(oloop.infer_once_stateful, 8, 3, 2)
    out["one_thread"] = {"cores": 1, "value": v1, "p50_ms": p1, "stateful_value": s1, "stateful_p50_ms": sp1,
                         "sample": "ref-semantics: 1 pass over the first 6 chunks (cost grows with the prefix: later chunks are slower); "
                                   "stateful: median of 3 passes over the first 8 chunks"}
    torch.set_num_threads(best_t)
    out["sample"] = (f"oracle loop, B=1, the 3 s utterance (37 full 80 ms chunks), {best_t} threads (probe-chosen of {avail}): reference semantics "
                     "(prefix re-run per chunk, inference/Conan.py:95-156) median of 3 passes after a 6-chunk warm-up = `value`; stateful "
                     "variant median of 5 passes; 1-thread figures in `one_thread`")
    return out


def lib_sha256():
    """sha256 of the libconan_hip.so this process loaded (ties committed profiler summaries to a binary)."""
    import hashlib
    from conan_amd import _lib
    try:
        return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
    except OSError:
        return None


def pmc_summary(tag):
    """The committed summary of the separate rocprofv3 --pmc passes for this workload (PMC counters cannot be collected
    from inside the benchmark): profiles/r<round>_<tag>_pmc.json, made by tools/collect_profiles.sh + tools/summarize_pmc.py
    (newest round first).  The summary records the sha256 of the library it was collected with; `stale` says whether that
    differs from the library loaded now (a summary without a hash counts as stale)."""
    for rnd in ("r6", "r5", "r4"):
        path = os.path.join(REPO, "profiles", f"{rnd}_{tag}_pmc.json")
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        return d, f"profiles/{rnd}_{tag}_pmc.json", d.get("lib_sha256") != lib_sha256()
    return None, None, None


DTYPE = {"f32": "f32", "limb": "f32 (3×bf16-limb products, f32 accumulate)"}


class Runner:
    """One stream-set (engine) of the workload with its output buffers: pipelined / windowed steps, blocking steps, the
    throughput timing, the latency leg and the per-kernel profile.  The headline and the other arithmetic form's datapoint are
    two Runners over the same context, timed by the same code."""

    def __init__(self, ctx, wl, B, rank, world, arith, comm=False):
        from conan_amd.engine import AudioGatherRing
        self.wl, self.B, self.world, self.rank, self.window = wl, B, world, rank, wl["window"]
        self.eng, self.chunks = make_engine(ctx, B, first_stream=rank * B, window=self.window, arith=arith)
        self.arith = self.eng.st.arith                  # 'auto' resolved by the library
        self.hop, self.seg = ctx.hop, self.eng.seg
        self.codes = torch.empty(B, self.seg, dtype=torch.int32, device="cuda")
        self.mel_out = torch.empty(B, self.seg, 80, device="cuda")
        self.wav = torch.empty(B, self.seg * self.hop, device="cuda")
        # Throughput leg, stateful workloads: pipelined steps (conan_step_async) - the stages of consecutive chunks overlap on
        # the library's internal HIP streams.  Audio goes to a small ring of buffers; with more than one rank the RCCL gather
        # runs on its own stream (conan_amd.engine.AudioGatherRing; CONAN_BENCH_COMM=1 exercises that choreography on one rank).
        # Windowed workloads: one blocking windowed step (Emformer step, reset, decoder + vocoder over window + chunk frames).
        self.ring = AudioGatherRing(lambda: torch.empty_like(self.wav), world, rank, nb=4, always=comm,
                                    every=int(os.environ.get("CONAN_BENCH_GATHER_EVERY", "4")))
        self.hist = [torch.randint(0, 100, (B, self.window), dtype=torch.int32, device="cuda")] if self.window else None
        self.j = 0
        self.primed = 0

    def step(self):
        j, eng, seg = self.j, self.eng, self.seg
        self.j += 1
        chunk = self.chunks[j % len(self.chunks)]
        if not self.window:      # only the vocoder stage of the step waits for the gather that last read this buffer
            buf, fence = self.ring.acquire(j, fence=True)
            eng.st.step_async(eng.slots, chunk, buf, emit=seg, codes=self.codes, mel_out=self.mel_out, out_fence=fence)
            self.ring.submit(j, join=eng.st.join)
            return
        buf = self.ring.acquire(j)
        c, w = eng.windowed_step(chunk, self.hist[0])
        self.hist[0] = torch.cat([self.hist[0][:, seg:], c], 1)
        buf.copy_(w)
        self.ring.submit(j, wait_current=True)

    def blocking_step(self):
        eng, seg = self.eng, self.seg
        chunk = self.chunks[self.j % len(self.chunks)]
        self.j += 1
        if self.window:
            cc, _ = eng.windowed_step(chunk, self.hist[0])
            self.hist[0] = torch.cat([self.hist[0][:, seg:], cc], 1)
        else:
            eng.st.step(eng.slots, chunk, emit=seg, codes=self.codes, mel_out=self.mel_out, wav_out=self.wav)

    def barrier(self):
        if not self.window:
            self.eng.st.join()
        self.ring.flush(self.j - 1)
        self.ring.drain()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def prime(self, n):
        """Setup, before the W warm-up steps: n pipelined steps of the workload itself, then a barrier.  A process that has just
        built its context and stream-set is cold in ways W = 5 steps do not cure - Python / ctypes paths run for the first time
        (the host needs 1.2 ms to enqueue a step at first, 0.85 ms later, against 1.3 ms of GPU time per step), the GPU comes out
        of seconds of idling: completion intervals of the first steps after a cold start read 1.66, 1.41, 1.35, 1.32, 1.31 ms
        (means of 5, tools/cold_series.py) - a server's first 30 ms, not its throughput.  Reported as config.priming_steps."""
        if self.window:
            n = min(n, 10)       # (windowed steps are blocking and long: a few are enough)
        for _ in range(n):
            self.step()
        self.barrier()
        self.primed = n

    def timed(self, steps, warmup, marks=False):
        """W untimed + exactly K timed steps between barrier + synchronize on both sides -> seconds (this rank's clock)."""
        for _ in range(warmup):
            self.step()
        self.barrier()
        if marks:
            self.eng.st.profile_mark()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.barrier()
        dt = time.perf_counter() - t0
        if marks:
            self.eng.st.profile_mark()
            torch.cuda.synchronize()
        return dt

    def step_intervals(self, n):
        """Distribution of the step time (outside the timed region, same schedule): the library stamps the completion of every
        pipelined step with a timing event on its own vocoder stream (conan_step_clock) - no extra stream, no extra wait."""
        st = self.eng.st
        st.step_clock(n + 1)
        for _ in range(n + 1):
            self.step()
        self.barrier()
        iv = sorted(st.step_clock_read())
        st.step_clock(0)
        if not iv:
            return None
        return {"n": len(iv), "mean_ms": statistics.fmean(iv), "sigma_ms": statistics.pstdev(iv), "p50_ms": statistics.median(iv),
                "p95_ms": iv[min(len(iv) - 1, int(round(0.95 * (len(iv) - 1))))], "min_ms": iv[0], "max_ms": iv[-1],
                "how": "intervals between consecutive step completions (timing events on the library's vocoder stream), a separate run of pipelined steps after the timed region"}

    def latencies(self, n):
        """per-chunk latency: one blocking step for all B streams, host submit -> audio complete on device"""
        lats = []
        for _ in range(n):
            torch.cuda.synchronize()
            a = time.perf_counter()
            self.blocking_step()
            torch.cuda.synchronize()
            lats.append((time.perf_counter() - a) * 1e3)
        return {"n": len(lats), "p50_ms": statistics.median(lats), "p95_ms": sorted(lats)[min(len(lats) - 1, int(round(0.95 * (len(lats) - 1))))],
                "sigma_ms": statistics.pstdev(lats), "min_ms": min(lats), "max_ms": max(lats)}

    def vocoder_alone(self, n=60, warm=10):
        """ms per vocoder step with nothing else on the chip: conan_hifigan_step for all B streams, enqueued back to back on the
        caller's stream (the launches the pipelined step's vocoder stage makes), one synchronize at the end."""
        if self.window:
            return None
        st, mel = self.eng.st, (torch.rand(self.B, self.seg, 80, device="cuda") * 4 - 5)
        for _ in range(warm):
            st.hifigan_step(self.eng.slots, mel, out=self.wav)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            st.hifigan_step(self.eng.slots, mel, out=self.wav)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def kernel_profile(self, nprof=5):
        """HIP events around every launch of the matrix kernels on their launch stream (blocking steps): per kernel template
        ('family') and per instantiation the summed time, algorithmic FLOPs and launches."""
        st = self.eng.st
        torch.cuda.synchronize()
        st.profile_begin()
        for _ in range(nprof):
            self.blocking_step()
        conv_ms, conv_flops, conv_launches = st.profile_end()
        kernels = sorted(st.profile_kernels(), key=lambda r: -r[1])
        fams = {}
        for kn, ms_, fl_, n_ in kernels:
            f = fams.setdefault(kn.split("<")[0], [0.0, 0.0, 0, []])
            f[0] += ms_; f[1] += fl_; f[2] += n_; f[3].append(kn)
        return {"nprof": nprof, "kernels": kernels, "families": fams, "conv_ms": conv_ms, "conv_flops": conv_flops, "conv_launches": conv_launches}

    def close(self):
        self.eng.st.close()


def dominant(prof):
    """The dominant kernel = the kernel template with the largest summed time: (name, achieved algorithmic TFLOP/s, peak, basis,
    ms per step, launches per step, average launch us, instantiations)."""
    name, (k_ms, k_fl, k_n, insts) = max(prof["families"].items(), key=lambda kv: kv[1][0])
    peak, basis = kernel_peak(name)
    return {"kernel": name, "achieved": k_fl / (k_ms * 1e-3) / 1e12, "peak": peak, "peak_basis": basis, "ms_per_step": k_ms / prof["nprof"],
            "launches_per_step": k_n / prof["nprof"], "avg_launch_us": k_ms * 1e3 / k_n, "gflop_per_launch": k_fl / k_n / 1e9, "instantiations": insts}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launcher_command(gpus, argv, env=None, port=None):
    """`python bench.py --gpus N` without a launcher around it (the shape of the driver's N = 1 command) has to measure N GPUs:
    -> the torch.distributed.run command that starts the N ranks (one process per GPU, RCCL over xGMI, rendezvous on
    127.0.0.1), or None when this process IS a rank already (WORLD_SIZE / RANK set by a launcher: no recursion) or N = 1.
    BASELINE.json configs[3] (512 streams over 8 GPUs) is this path."""
    env = os.environ if env is None else env
    if "WORLD_SIZE" in env or "RANK" in env or "LOCAL_RANK" in env:
        return None
    if gpus <= 1 and env.get("CONAN_BENCH_FORCE_SPAWN") != "1":      # (test hook: the child-process path with one rank, on a 1-GPU box)
        return None
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def spawn_ranks(gpus, argv):
    """Run the N ranks as a CHILD process (never exec: this must also be safe in a process that has touched the GPU), relay
    their output, return their exit code; None when there is nothing to spawn.  Asking for more GPUs than the node has is an
    error here, before anything is launched (torch.cuda.device_count() does not initialise the GPU)."""
    cmd = launcher_command(gpus, argv)
    if cmd is None:
        return None
    have = torch.cuda.device_count()
    if have < gpus:
        print(f"bench.py: --gpus {gpus} but this node has {have} GPU(s)", file=sys.stderr)
        return 2
    import subprocess
    return subprocess.run(cmd, env=dict(os.environ)).returncode


def check_ranks(gpus, world, devices):
    """A rank's view must be the job that was asked for: -> error text or None."""
    if world != gpus:
        return f"--gpus {gpus} but WORLD_SIZE={world}: start bench.py with --gpus equal to the launcher's --nproc-per-node (or without a launcher: it starts its own ranks)"
    if devices < world:
        return f"{world} ranks but {devices} visible GPU(s): one process per GPU"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="b64", choices=sorted(WORKLOADS))
    ap.add_argument("--arith", default="auto", choices=["auto", "f32", "limb"],
                    help="arithmetic of the vocoder's matrix kernels (conan_streams_opts.arith); auto = the library's default, which `value` measures")
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default: the workload's)")
    ap.add_argument("--prime", type=int, default=int(os.environ.get("CONAN_BENCH_PRIME", "100")),
                    help="setup: pipelined steps run once after the stream-set is built, before the W warm-up steps (0: none)")
    ap.add_argument("--fixed-plan", action="store_true", help="CONAN_STREAMS_FIXED_PLAN: the launch plan from max_slots only (a stream's bits do not depend on the other active slots)")
    ap.add_argument("--dev-plan", default=None, help="developer switches of the launch plan, 'NAME=value;...' (conan_streams_opts.dev_plan; A/B runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-steps", type=int, default=40)
    ap.add_argument("--no-b1", action="store_true", help="skip the batch=1 latency leg (keeps profiler summaries to one workload)")
    ap.add_argument("--no-other", action="store_true", help="skip the datapoint of the other arithmetic form")
    ap.add_argument("--marks", action="store_true", help="bracket the timed steps with cnk::profile_mark_kernel dispatches and skip "
                                                         "every other leg (rocprofv3 --pmc passes: tools/summarize_pmc.py keeps the dispatches between the marks)")
    args = ap.parse_args()

    rc = spawn_ranks(args.gpus, sys.argv[1:])      # --gpus N > 1 outside a launcher: this process only starts the N ranks
    if rc is not None:
        raise SystemExit(rc)
    err = check_ranks(args.gpus, int(os.environ.get("WORLD_SIZE", "1")), torch.cuda.device_count())
    if err:
        raise SystemExit("bench.py: " + err)
    from conan_amd.engine import init_distributed
    rank, local, world = init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback in the product path)")
    torch.cuda.set_device(local)
    from conan_amd import _lib
    STREAM_OPTS["flags"] = _lib.STREAMS_FIXED_PLAN if args.fixed_plan else 0
    STREAM_OPTS["dev_plan"] = args.dev_plan
    wl = WORKLOADS[args.workload]
    B = args.streams or wl["streams"]
    window = wl["window"]

    ctx, chp, vhp = build_context(local, wl["chunk_ms"], wl.get("memory", 0))
    run = Runner(ctx, wl, B, rank, world, args.arith, comm=os.environ.get("CONAN_BENCH_COMM", "0") == "1")
    arith = run.arith                     # what the library resolved `auto` to: the form `value` is measured in
    hop, seg = run.hop, run.seg

    # the unprimed reading: exactly W warm-up + K timed steps on the freshly built stream-set (what --warmup alone means) ...
    dt_unprimed = run.timed(args.steps, args.warmup) if (args.prime > 0 and not args.marks) else None
    # ... and the steady-state one behind the priming pass: `value`
    if args.prime > 0:
        run.prime(args.prime)
    dt = run.timed(args.steps, args.warmup, marks=args.marks)
    j = run.j
    # Every rank's own clock, its device, and a check of the exchange: the audio rank 0 gathered for the LAST gathered step must
    # be, bit for bit, what each rank produced for it (one all_gather of integer checksums, outside the timed region).
    ring = run.ring
    ranks = None
    gj = ring.last_gathered if ring.last_gathered is not None else j - 1
    last = ring.last_sent if ring.last_sent is not None else ring.bufs[gj % ring.nb]      # what this rank handed to the last gather
    csum = last.view(torch.int32).to(torch.int64).sum().reshape(1)            # order-independent, exact
    if world > 1:
        mine = torch.tensor([dt, float(torch.cuda.current_device())], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        sums = [torch.zeros_like(csum) for _ in range(world)]
        dist.all_gather(sums, csum)
        dt = max(float(r[0].item()) for r in allr)                              # MAX over ranks = the job's time
        if rank == 0:
            got = [int(g.view(torch.int32).to(torch.int64).sum().item()) for g in ring.gbufs]
            want = [int(x.item()) for x in sums]
            ranks = {"rccl_world": dist.get_world_size(), "backend": dist.get_backend(),
                     "devices": [int(r[1].item()) for r in allr], "device_name": torch.cuda.get_device_name(),
                     "ms_per_step_per_rank": [float(r[0].item()) / args.steps * 1e3 for r in allr],
                     "ms_per_step_min": min(float(r[0].item()) for r in allr) / args.steps * 1e3,
                     "ms_per_step_max": max(float(r[0].item()) for r in allr) / args.steps * 1e3,
                     "gathers": ring.submitted, "gather_every": ring.every,
                     "gather_check": {"step": gj, "ok": got == want, "checksums_rank0_gathered": got, "checksums_ranks_own": want}}
            if ranks["rccl_world"] != args.gpus or len(set(ranks["devices"])) != args.gpus:
                raise SystemExit("bench.py: asked for %d GPUs, ran on RCCL world %d over devices %r" % (args.gpus, ranks["rccl_world"], ranks["devices"]))
            if got != want:
                raise SystemExit("bench.py: the audio gathered on rank 0 differs from what the ranks produced: %r vs %r" % (got, want))
    else:
        ranks = {"rccl_world": 1, "backend": None, "devices": [torch.cuda.current_device()], "device_name": torch.cuda.get_device_name(),
                 "ms_per_step_per_rank": [dt / args.steps * 1e3], "ms_per_step_min": dt / args.steps * 1e3, "ms_per_step_max": dt / args.steps * 1e3,
                 "gathers": ring.submitted, "gather_every": ring.every,
                 "gather_check": ({"step": gj, "ok": True, "note": "single rank: the gather path (CONAN_BENCH_COMM=1) hands rank 0 its own buffer",
                                   "checksums_ranks_own": [int(csum.item())]} if ring.active else None)}
    ms_step = dt / args.steps * 1e3

    step_stats = None
    if not args.marks and not window:
        step_stats = run.step_intervals(min(max(args.steps, 2), 60))
    frames_per_step = (window + seg) if window else seg           # decoder / vocoder frames computed per stream per step

    roof = b1 = cpu = fe = None
    p50 = lat_stats = voc_alone = None
    if not args.marks:
        lat_stats = run.latencies(args.latency_steps)
        p50 = lat_stats["p50_ms"]
        if rank == 0:
            voc_alone = run.vocoder_alone()

    if rank == 0 and not args.marks:
        # roofline of the dominant kernel: HIP events around every launch of the matrix kernels on their stream
        prof = run.kernel_profile()
        nprof, kernels = prof["nprof"], prof["kernels"]
        dom = dominant(prof)
        name, ach = dom["kernel"], dom["achieved"]
        fam = prof["conv_flops"] / (prof["conv_ms"] * 1e-3) / 1e12
        tag = args.workload + ("" if args.arith == "auto" else "_" + args.arith)
        pmc, pmc_src, pmc_stale = pmc_summary(tag)
        traffic = step_bytes = mfma_busy = None
        if pmc:
            hit = [v for k, v in pmc.get("kernels", {}).items() if name in k]
            if hit:       # per launch of the kernel, averaged over its instantiations by their launch counts
                w = sum(v["dispatches_per_step"] for v in hit)
                traffic = sum((v["fetch"] + v["write"]) * v["dispatches_per_step"] for v in hit) / w
                busy = [v for v in hit if "mfma_busy_frac" in v]
                mfma_busy = (sum(v["mfma_busy_frac"] * v["dispatches_per_step"] for v in busy) / sum(v["dispatches_per_step"] for v in busy)) if busy else None
            step_bytes = pmc.get("bytes_per_step")
        flops_step = B * frames_per_step * GFLOP_PER_FRAME * 1e9
        bytes_alg = WEIGHT_BYTES_PER_STEP + B * STATE_BYTES_PER_STREAM * (frames_per_step / seg)
        k_peak, k_peak_basis = dom["peak"], dom["peak_basis"]
        step_tflops = flops_step / (ms_step * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": ach, "peak": k_peak, "unit": "TFLOP/s",
                "frac": ach / k_peak, "peak_basis": k_peak_basis, "frac_of_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
                # limb kernels: the bf16 FLOP/s the MFMAs execute (6 per algorithmic FLOP) against the bf16 dense peak - the same fraction
                "executed_bf16_tflops": (6.0 * ach) if k_peak == PEAK_LIMB_TFLOPS else None, "bf16_dense_peak": PEAK_BF16_MFMA_TFLOPS,
                "traffic": traffic, "traffic_source": pmc_src if traffic is not None else None,
                "traffic_stale": pmc_stale if traffic is not None else None, "lib_sha256": lib_sha256(),
                "mfma_busy_frac_pmc": mfma_busy,
                "kernel": name, "arith": arith, "launches_per_step": dom["launches_per_step"], "avg_launch_us": dom["avg_launch_us"],
                "gflop_per_launch": dom["gflop_per_launch"], "kernel_ms_per_step": dom["ms_per_step"], "share_of_step_time": dom["ms_per_step"] / ms_step,
                # scalar companions of the headline (flat, so that they survive parsers that keep only scalars)
                "ms_per_step": ms_step, "ms_per_step_unprimed": (dt_unprimed / args.steps * 1e3) if dt_unprimed else None,
                "vocoder_alone_ms": voc_alone, "frontend_cost_ms": (ms_step - voc_alone) if voc_alone else None,
                "p50_latency_ms": p50, "p95_latency_ms": lat_stats["p95_ms"] if lat_stats else None,
                "step_interval_p50_ms": step_stats["p50_ms"] if step_stats else None, "step_interval_p95_ms": step_stats["p95_ms"] if step_stats else None,
                "step_gflop_algorithmic": flops_step / 1e9, "step_tflops": step_tflops,
                "step_mfma_frac": step_tflops / PEAK_F32_MFMA_TFLOPS, "step_mfma_frac_basis": "whole step's algorithmic FLOP/s over the f32 MFMA dense peak (157.3): the decoder and Emformer compute on it in both forms",
                "hbm_bytes_per_step_algorithmic": bytes_alg, "hbm_bytes_per_step_counter": step_bytes,
                "hbm_counter_source": pmc_src if step_bytes is not None else None,
                "hbm_GBps_algorithmic": bytes_alg / (ms_step * 1e-3) / 1e9, "hbm_frac_algorithmic": bytes_alg / (ms_step * 1e-3) / 1e9 / PEAK_HBM_GBS,
                "hbm_GBps_counter": (step_bytes / (ms_step * 1e-3) / 1e9) if step_bytes else None, "hbm_peak_GBps": PEAK_HBM_GBS,
                "all_matrix_kernels_ms_per_step": prof["conv_ms"] / nprof, "all_matrix_kernels_tflops": fam,
                "all_matrix_kernels_launches_per_step": prof["conv_launches"] / nprof,
                arith + "_ms_per_step": ms_step, arith + "_frac": ach / k_peak, arith + "_kernel": name, arith + "_peak": k_peak,
                "instantiations": dom["instantiations"],
                "largest_instantiation": {"kernel": kernels[0][0], "ms_per_step": kernels[0][1] / nprof, "tflops": kernels[0][2] / (kernels[0][1] * 1e-3) / 1e12},
                "matrix_kernels": [{"kernel": kn, "launches_per_step": n_ / nprof, "us_per_launch": ms_ * 1e3 / n_, "ms_per_step": ms_ / nprof,
                                    "tflops": fl_ / (ms_ * 1e-3) / 1e12, "peak": kernel_peak(kn)[0], "frac": fl_ / (ms_ * 1e-3) / 1e12 / kernel_peak(kn)[0],
                                    "frac_of_f32_mfma_peak": fl_ / (ms_ * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS} for kn, ms_, fl_, n_ in kernels]}
        # the dominant kernel's fraction per channel width (its instantiations differ by an order of magnitude in tile shape): scalars
        for width in (128, 64, 32):
            sel = [(ms_, fl_) for kn, ms_, fl_, n_ in kernels if kn.startswith(name + "<%d," % width)]
            if sel:
                roof["dominant_frac_c%d" % width] = sum(f for _, f in sel) / (sum(m for m, _ in sel) * 1e-3) / 1e12 / k_peak
        # batch=1 latency configuration (BASELINE.json configs[1]) beside the throughput one
        if args.workload == "b64" and not args.streams and not args.no_b1:
            def small_batch_latency(nb):
                e1, ch1 = make_engine(ctx, nb, first_stream=100000, arith=args.arith)
                c1 = torch.empty(nb, seg, dtype=torch.int32, device="cuda")
                m1 = torch.empty(nb, seg, 80, device="cuda")
                w1 = torch.empty(nb, seg * hop, device="cuda")
                l1 = []
                for k in range(10 + args.latency_steps):
                    torch.cuda.synchronize()
                    a = time.perf_counter()
                    e1.st.step(e1.slots, ch1[k % len(ch1)], emit=seg, codes=c1, mel_out=m1, wav_out=w1)
                    torch.cuda.synchronize()
                    if k >= 10:
                        l1.append((time.perf_counter() - a) * 1e3)
                e1.st.close()
                return statistics.median(l1)
            l1 = small_batch_latency(1)
            b1 = {"workload": WORKLOADS["b1"]["desc"], "p50_latency_ms": l1, "chunks_per_s": 1e3 / l1}
            roof["latency_b1_ms"] = l1
            roof["latency_b4_ms"] = small_batch_latency(4)      # the same blocking step for 4 streams (small-batch path)
        # the step before the path: GPU mel front-end rate for B x 3 s of audio (not part of `value`)
        try:
            w = torch.rand(B, 48000, device="cuda") * 2 - 1
            ctx.wav2mel(w)
            torch.cuda.synchronize()
            a = time.perf_counter()
            for _ in range(5):
                ctx.wav2mel(w)
            torch.cuda.synchronize()
            fe_ms = (time.perf_counter() - a) / 5 * 1e3
            fe = {"workload": f"conan_wav2mel, {B} x 3 s @16 kHz -> [{B},151,80]", "ms": fe_ms, "frames_per_s": B * 151 / (fe_ms * 1e-3)}
        except Exception as e:  # noqa: BLE001  (a front-end failure must not hide the headline measurement)
            fe = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:      # a reported baseline of the N = 1 line only
            cpu = cpu_baseline()
        # The other arithmetic form of the same workload, same process, same timing code (the library default is what `value`
        # measures; this is the comparison point): its step time, its dominant kernel against that kernel's own peak.
        if world == 1 and not args.no_other and not window and not args.streams:
            other = "f32" if arith == "limb" else "limb"
            try:
                run.close()                      # (one stream-set at a time: the other form is timed as the headline was)
                o = Runner(ctx, wl, B, rank, world, other)
                if args.prime > 0:
                    o.prime(args.prime)
                o_dt = o.timed(args.steps, args.warmup)
                o_lat = o.latencies(max(10, args.latency_steps // 2))
                od = dominant(o.kernel_profile())
                o.close()
                roof.update({other + "_ms_per_step": o_dt / args.steps * 1e3, other + "_frac": od["achieved"] / od["peak"], other + "_kernel": od["kernel"],
                             other + "_peak": od["peak"], other + "_achieved": od["achieved"], other + "_peak_basis": od["peak_basis"],
                             other + "_p50_latency_ms": o_lat["p50_ms"], other + "_kernel_ms_per_step": od["ms_per_step"]})
            except Exception as e:  # noqa: BLE001  (the comparison point must not hide the headline measurement)
                roof[other + "_error"] = f"{type(e).__name__}: {e}"

    if world > 1:
        dist.barrier()
    if rank == 0:
        total_chunks = world * B * args.steps
        out = {
            "metric": "chunks/sec (%d ms chunk, 16 kHz), all streams summed; p50 per-chunk latency beside it" % wl["chunk_ms"],
            "value": total_chunks / dt, "unit": "chunks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE[arith], "data": "synthetic",
            "config": {"workload": wl["desc"] if not args.streams else f"batch={B} streams per GPU, {wl['chunk_ms']} ms chunk" + (", windowed" if window else ", stateful"),
                       "name": args.workload, "baseline_config": wl["config"], "arith": arith, "arith_requested": args.arith, "fixed_plan": bool(args.fixed_plan), "dev_plan": args.dev_plan,
                       "streams_per_gpu": B, "global_streams": world * B, "chunk_ms": wl["chunk_ms"], "context_window_frames": window, "sample_rate": 16000,
                       "architecture": "egs/conan_emformer.yaml + egs/hifi_16k320_shuffle.yaml shapes, random-init weights",
                       "parallelism": f"dp{world} (streams sharded by slot range; RCCL gather of audio to rank 0)" if world > 1 else "dp1",
                       "priming_steps": run.primed},
            "p50_latency_ms": p50, "latency_stats": lat_stats, "step_time_stats": step_stats, "ranks": ranks,
            "schedule": ("throughput and latency: one blocking windowed step" if window else
                         "throughput: pipelined steps (the three stages of consecutive chunks overlap on three HIP streams); latency: one blocking fused step"),
            "realtime_streams_supported": (total_chunks / dt) / (1000.0 / wl["chunk_ms"]),
            "roofline": roof, "cpu_baseline": cpu,
        }
        if b1 is not None:
            out["latency_b1"] = b1
        if fe is not None:
            out["frontend"] = fe
        print(json.dumps(out))
    run.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
