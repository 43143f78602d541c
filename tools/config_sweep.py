"""Timings of the other BASELINE.json configurations (not the bench line): B=1 windowed (160 ms context), 40 ms chunks at
B=128 stateful and windowed (320 ms context).  Developer tool; prints one line per configuration."""
import statistics
import sys
import time

import torch

sys.path.insert(0, ".")
from conan_amd import configs, synth  # noqa: E402
from conan_amd.runtime import Context  # noqa: E402


def ctx_for(chp):
    vhp = configs.hifigan_hparams()
    ctx = Context(chp, vhp, 0, True, True, True)
    ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
    ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    return ctx


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)


def windowed_step(st, slots, chunk, ctx_codes, ctx_frames, seg):
    """configs[1]/[4] windowed mode: Emformer stays stateful (its left context is its own cache); decoder and vocoder
    are reset and fed ctx + chunk frames, only the last `seg` frames are kept."""
    _, _, codes = st.emformer_step(slots, chunk, want_out=False, want_logits=False)
    win = torch.cat([ctx_codes, codes], 1)
    st.reset(slots, which=2 | 4)
    mel = st.decoder_step(slots, win)
    wav = st.hifigan_step(slots, mel)
    return wav[:, -seg * 320:]


def main():
    # configs[1]: B = 1, 80 ms chunk + 160 ms (8-frame) context
    chp = configs.conan_hparams()
    ctx = ctx_for(chp)
    st = ctx.streams(1, max_frames=12, max_ref_frames=256)
    st.reset([0]); st.set_reference([0], torch.from_numpy(synth.mel(151, 4321)).cuda())
    chunk = torch.from_numpy(synth.mel(6, 1)).cuda()
    cc = torch.randint(0, 100, (1, 8), dtype=torch.int32, device="cuda")
    ms = timed(lambda: windowed_step(st, [0], chunk, cc, 8, 4))
    print("configs[1] windowed: B=1, 80 ms chunk + 160 ms context (reset + 12 frames per step): p50 %.2f ms per chunk" % ms)
    st.close(); ctx.close()
    # configs[4]: 40 ms chunks (seg 2), B = 128
    chp = dict(configs.conan_hparams(), chunk_size=40)
    ctx = ctx_for(chp)
    B = 128
    st = ctx.streams(B, max_frames=18, max_ref_frames=256)
    slots = list(range(B))
    st.reset(slots); st.set_reference(slots, torch.from_numpy(synth.mel(151, 4321, B)).cuda())
    chunk = torch.from_numpy(synth.mel(4, 2, B)).cuda()
    wav = torch.empty(B, 2 * 320, device="cuda")
    ms = timed(lambda: st.step(slots, chunk, wav_out=wav))
    print("configs[4] stateful: B=128, 40 ms chunk (seg 2 + rc 2): p50 %.2f ms per step = %.0f chunks/s" % (ms, B / ms * 1e3))

    def pipe():
        for _ in range(8):
            st.step_async(slots, chunk, wav)
        st.join()
    ms = timed(pipe, n=10, warm=2) / 8
    print("configs[4] stateful, pipelined steps: %.2f ms per step = %.0f chunks/s" % (ms, B / ms * 1e3))
    cc = torch.randint(0, 100, (B, 16), dtype=torch.int32, device="cuda")
    ms = timed(lambda: windowed_step(st, slots, chunk, cc, 16, 2), n=10, warm=2)
    print("configs[4] windowed: B=128, 40 ms chunk + 320 ms context (reset + 18 frames per step): p50 %.2f ms per step = %.0f chunks/s" % (ms, B / ms * 1e3))
    st.close(); ctx.close()


if __name__ == "__main__":
    main()
