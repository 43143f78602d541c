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

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA, dense
N_FRAMES, N_REF = 151, 151     # 3 s source + 3 s reference at 50 frames/s

WORKLOADS = {
    "b64": dict(streams=64, desc="batch=64 concurrent streams per GPU, 80 ms chunk (seg 4 + rc 2 frames), stateful full "
                                 "Emformer->Conan->HiFi-GAN pipeline"),
    "b1": dict(streams=1, desc="batch=1 stream, 80 ms chunk (seg 4 + rc 2 frames), stateful full Emformer->Conan->HiFi-GAN pipeline"),
}


def build_context(device):
    from conan_amd import configs, synth
    from conan_amd.runtime import Context
    chp, vhp = configs.conan_hparams(), configs.hifigan_hparams()
    ctx = Context(chp, vhp, device)
    ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
    ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    return ctx, chp, vhp


def make_engine(ctx, B, first_stream):
    """B streams with their reference set and every full chunk of the 3 s utterance staged in HBM."""
    from conan_amd import synth
    from conan_amd.engine import StreamingVoiceConversionEngine
    eng = StreamingVoiceConversionEngine(ctx, B, max_ref_frames=N_REF + 1)
    src = np.concatenate([synth.mel(N_FRAMES, 1234 + first_stream + s) for s in range(B)])
    ref = np.concatenate([synth.mel(N_REF, 4321 + first_stream + s) for s in range(B)])
    src = torch.from_numpy(src).cuda()
    eng.start(torch.from_numpy(ref).cuda())
    chunks = [c for _, emit, c in eng.chunks(src) if emit == eng.seg]
    return eng, chunks


def cpu_baseline(budget_s=20.0):
    """The oracle's reference-semantics loop (inference/Conan.py:95-156: prefix re-run of Conan and the
    vocoder every chunk, numpy hops) timed on the host cores, B=1, on a time-bounded prefix of the same
    3 s utterance; plus the stateful CPU variant (same arithmetic the GPU path performs)."""
    from conan_amd import configs, synth
    from oracle import emformer as oemf
    from oracle import loop as oloop
    from oracle.common import to_torch_sd
    from oracle import hifigan as ohifi
    chp, vhp = configs.conan_hparams(), configs.hifigan_hparams()
    esd = to_torch_sd(synth.emformer_state_dict(chp, 0))
    csd = to_torch_sd(synth.conan_state_dict(chp, 0))
    vsd = to_torch_sd(synth.hifigan_state_dict(vhp, 0))
    cfg = oemf.EmformerCfg(chp)
    # PyTorch's CPU conv path degrades badly when heavily over-threaded on a many-core host: pick the
    # fastest thread count for this workload (vocoder forward on 16 frames) among a few candidates.
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
    torch.set_num_threads(best_t)
    src, ref = synth.mel(N_FRAMES, 1234)[0], synth.mel(N_REF, 4321)[0]

    class Stop(Exception):
        pass

    def run(fn, budget):
        stamps = [time.perf_counter()]

        def tick():
            stamps.append(time.perf_counter())
            if stamps[-1] - stamps[0] > budget:
                raise Stop()
        try:
            fn(esd, cfg, csd, chp, vsd, vhp, src, ref, on_chunk=tick)
        except Stop:
            pass
        n = len(stamps) - 1
        dt = stamps[-1] - stamps[0]
        lat = sorted(b - a for a, b in zip(stamps[:-1], stamps[1:]))
        return n, dt, (lat[len(lat) // 2] if lat else float("nan"))

    n_ref, t_ref, p50_ref = run(oloop.infer_once_ref, budget_s * 0.7)
    n_st, t_st, p50_st = run(oloop.infer_once_stateful, budget_s * 0.3)
    return {
        "value": n_ref / t_ref, "unit": "chunks/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"oracle ref-semantics loop (prefix re-run per chunk, inference/Conan.py:95-156), B=1, first {n_ref} of 38 "
                  f"chunks of the 3 s utterance in {t_ref:.1f} s; stateful CPU variant over {n_st} chunks",
        "p50_ms": p50_ref * 1e3,
        "stateful_value": n_st / t_st, "stateful_p50_ms": p50_st * 1e3,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="b64", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency-steps", type=int, default=40)
    ap.add_argument("--no-b1", action="store_true", help="skip the batch=1 latency leg (keeps profiler summaries B=64 only)")
    args = ap.parse_args()

    from conan_amd.engine import gather_audio_equal, init_distributed
    rank, local, world = init_distributed()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X (no CPU fallback in the product path)")
    torch.cuda.set_device(local)
    B = args.streams or WORKLOADS[args.workload]["streams"]

    ctx, chp, vhp = build_context(local)
    eng, chunks = make_engine(ctx, B, first_stream=rank * B)
    hop, seg = ctx.hop, eng.seg
    codes = torch.empty(B, seg, dtype=torch.int32, device="cuda")
    mel_out = torch.empty(B, seg, 80, device="cuda")
    wav = torch.empty(B, seg * hop, device="cuda")
    gbufs = [torch.empty_like(wav) for _ in range(world)] if (world > 1 and rank == 0) else None

    # Throughput leg: pipelined steps (conan_step_async) - the front-end of chunk t+1 overlaps the vocoder of chunk t on
    # the library's two internal HIP streams.  Audio goes to a small ring of buffers; with more than one rank the RCCL
    # gather of chunk t runs on its own stream so that it does not serialise the pipeline either.
    NB = 4
    wavs = [torch.empty_like(wav) for _ in range(NB)]
    # CONAN_BENCH_COMM=1 exercises the gather stream / event choreography on a single rank (the gather itself is a
    # no-op there); used to check that path on a one-GPU box
    use_comm = world > 1 or os.environ.get("CONAN_BENCH_COMM", "0") == "1"
    comm = torch.cuda.Stream() if use_comm else None
    gdone = [torch.cuda.Event() for _ in range(NB)] if use_comm else None

    def step(j):
        k = j % NB
        if use_comm and j >= NB:
            torch.cuda.current_stream().wait_event(gdone[k])      # the gather that read this buffer has finished
        eng.st.step_async(eng.slots, chunks[j % len(chunks)], wavs[k], emit=seg, codes=codes, mel_out=mel_out)
        if use_comm:
            with torch.cuda.stream(comm):
                eng.st.join()
                gather_audio_equal(wavs[k], world, rank, gbufs)
                gdone[k].record(comm)

    def barrier():
        eng.st.join()
        if use_comm:
            comm.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    j = 0
    for _ in range(args.warmup):
        step(j); j += 1
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(j); j += 1
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # per-chunk latency: one step for all B streams, host submit -> audio complete on device
    lats = []
    for _ in range(args.latency_steps):
        torch.cuda.synchronize()
        a = time.perf_counter()
        eng.st.step(eng.slots, chunks[j % len(chunks)], emit=seg, codes=codes, mel_out=mel_out, wav_out=wav); j += 1
        torch.cuda.synchronize()
        lats.append((time.perf_counter() - a) * 1e3)
    p50 = statistics.median(lats)

    # roofline of the dominant kernel family (conv_mfma): HIP events around every launch on its stream
    roof = None
    b1 = None
    cpu = None
    fe = None
    if rank == 0:
        nprof = 5
        torch.cuda.synchronize()
        eng.st.profile_begin()
        for _ in range(nprof):
            eng.st.step(eng.slots, chunks[j % len(chunks)], emit=seg, codes=codes, mel_out=mel_out, wav_out=wav); j += 1
        conv_ms, conv_flops, conv_launches = eng.st.profile_end()
        # the dominant kernel = the template instantiation with the largest summed time (the vocoder's streaming tiles)
        name, k_ms, k_fl, k_n = max(eng.st.profile_kernels(), key=lambda r: r[1])
        ach = k_fl / (k_ms * 1e-3) / 1e12
        fam = conv_flops / (conv_ms * 1e-3) / 1e12
        # HBM bytes per launch of that kernel: PMC counters cannot be collected from inside the benchmark, so the figure
        # comes from the committed summary of the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        traffic, traffic_src = None, None
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_b64_pmc_hbm.json")
        if os.path.exists(pmc):
            try:
                kern = json.load(open(pmc))["kernels"]
                hit = [v for k, v in kern.items() if name in k]
                if hit:
                    traffic = hit[0]["fetch"] + hit[0]["write"]
                    traffic_src = "profiles/r1_b64_pmc_hbm.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)"
            except (OSError, ValueError, KeyError):
                pass
        roof = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": name, "launches_per_step": k_n / nprof, "avg_launch_us": k_ms * 1e3 / k_n,
                "gflop_per_launch": k_fl / k_n / 1e9, "share_of_step_time": (k_ms / nprof) / (dt / args.steps * 1e3),
                "all_conv_kernels": {"achieved": fam, "frac": fam / PEAK_F32_MFMA_TFLOPS, "launches_per_step": conv_launches / nprof,
                                     "ms_per_step": conv_ms / nprof, "gflop_per_chunk_per_stream": conv_flops / nprof / B / 1e9}}
        # batch=1 latency configuration (BASELINE.json configs[1]) beside the throughput one
        if B != 1 and not args.no_b1:
            e1, ch1 = make_engine(ctx, 1, first_stream=100000)
            c1 = torch.empty(1, seg, dtype=torch.int32, device="cuda")
            m1 = torch.empty(1, seg, 80, device="cuda")
            w1 = torch.empty(1, seg * hop, device="cuda")
            l1 = []
            for k in range(10 + args.latency_steps):
                torch.cuda.synchronize()
                a = time.perf_counter()
                e1.st.step(e1.slots, ch1[k % len(ch1)], emit=seg, codes=c1, mel_out=m1, wav_out=w1)
                torch.cuda.synchronize()
                if k >= 10:
                    l1.append((time.perf_counter() - a) * 1e3)
            b1 = {"workload": WORKLOADS["b1"]["desc"], "p50_latency_ms": statistics.median(l1),
                  "chunks_per_s": 1e3 / statistics.median(l1)}
            e1.st.close()
        # the step before the path: GPU mel front-end rate for B x 3 s of audio (not part of `value`)
        fe = None
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
        if not args.no_cpu_baseline:
            cpu = cpu_baseline()

    if world > 1:
        dist.barrier()
    if rank == 0:
        total_chunks = world * B * args.steps
        out = {
            "metric": "chunks/sec (80 ms chunk, 16 kHz), all streams summed; p50 per-chunk latency beside it",
            "value": total_chunks / dt, "unit": "chunks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload]["desc"] if not args.streams else f"batch={B} streams per GPU, 80 ms chunk, stateful",
                       "streams_per_gpu": B, "global_streams": world * B, "chunk_ms": 80, "sample_rate": 16000,
                       "architecture": "egs/conan_emformer.yaml + egs/hifi_16k320_shuffle.yaml shapes, random-init weights",
                       "parallelism": f"dp{world} (streams sharded by slot range; RCCL gather of audio to rank 0)" if world > 1 else "dp1"},
            "p50_latency_ms": p50,
            "schedule": "throughput: pipelined steps (front-end of chunk t+1 overlaps the vocoder of chunk t on two HIP streams); latency: one blocking fused step",
            "realtime_streams_supported": (total_chunks / dt) / 12.5,
            "roofline": roof, "cpu_baseline": cpu,
        }
        if b1 is not None:
            out["latency_b1"] = b1
        if fe is not None:
            out["frontend"] = fe
        print(json.dumps(out))
    eng.st.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
