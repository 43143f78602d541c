"""Experiment: front-end (Emformer + Conan decoder) of step t+1 on one HIP stream while the vocoder of step t runs on
another, optionally on disjoint CU partitions (hipExtStreamCreateWithCUMask).  Developer tool; prints ms/step."""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from conan_amd import configs, synth  # noqa: E402
from conan_amd.runtime import Context  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def make_stream(mask_words=None):
    s = C.c_void_p()
    if mask_words is None:
        rc = hip.hipStreamCreateWithFlags(C.byref(s), 1)  # hipStreamNonBlocking
    else:
        arr = (C.c_uint32 * len(mask_words))(*mask_words)
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(mask_words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def main():
    B, K = 64, 40
    front_cus = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    chp, vhp = configs.conan_hparams(False), configs.hifigan_hparams(False)
    ctx = Context(chp, vhp, 0, True, True, True)
    ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
    ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    st = ctx.streams(B, max_frames=4, max_ref_frames=256)
    slots = list(range(B))
    st.reset(slots)
    st.set_reference(slots, torch.from_numpy(synth.mel(150, 7, B)).cuda())
    chunk = torch.from_numpy(synth.mel(6, 11, B)).cuda()
    hop = ctx.hop
    wav = torch.empty(B, 4 * hop, device="cuda")

    def fused(n):
        for _ in range(n):
            st.step(slots, chunk, wav_out=wav)

    fused(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); fused(K); torch.cuda.synchronize()
    print("single stream, fused conan_step: %.3f ms/step" % ((time.perf_counter() - t0) / K * 1e3))

    def run_pipelined(sf, sv, n):
        ev_f = [torch.cuda.Event(), torch.cuda.Event()]
        ev_v = [torch.cuda.Event(), torch.cuda.Event()]
        mels = [None, None]
        for t in range(n):
            p = t & 1
            with torch.cuda.stream(sf):
                if t >= 2:
                    sf.wait_event(ev_v[p])
                _, _, codes = st.emformer_step(slots, chunk, want_out=False, want_logits=False)
                mels[p] = st.decoder_step(slots, codes)
                ev_f[p].record(sf)
            with torch.cuda.stream(sv):
                sv.wait_event(ev_f[p])
                st.hifigan_step(slots, mels[p], out=wav)
                ev_v[p].record(sv)

    for name, masks in (("two streams, no CU mask", (None, None)),
                        ("two streams, front-end on %d CUs" % front_cus, "mask")):
        if masks == "mask":
            words = 8
            fm = [0] * words; vm = [0] * words
            for i in range(256):
                (fm if i < front_cus else vm)[i // 32] |= 1 << (i % 32)
            sf, sv = make_stream(fm), make_stream(vm)
        else:
            sf, sv = make_stream(), make_stream()
        run_pipelined(sf, sv, 6)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); run_pipelined(sf, sv, K); torch.cuda.synchronize()
        print("%s: %.3f ms/step" % (name, (time.perf_counter() - t0) / K * 1e3))
        # vocoder alone on its stream / front-end alone, for reference
        with torch.cuda.stream(sv):
            mel = torch.zeros(B, 4, 80, device="cuda")
            for _ in range(3): st.hifigan_step(slots, mel, out=wav)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(K): st.hifigan_step(slots, mel, out=wav)
            torch.cuda.synchronize(); tv = (time.perf_counter() - t0) / K * 1e3
        with torch.cuda.stream(sf):
            for _ in range(3):
                codes = st.emformer_step(slots, chunk, want_out=False, want_logits=False)[2]; st.decoder_step(slots, codes)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(K):
                codes = st.emformer_step(slots, chunk, want_out=False, want_logits=False)[2]; st.decoder_step(slots, codes)
            torch.cuda.synchronize(); tf = (time.perf_counter() - t0) / K * 1e3
        print("    alone on their streams: vocoder %.3f ms, front-end %.3f ms" % (tv, tf))


if __name__ == "__main__":
    main()
