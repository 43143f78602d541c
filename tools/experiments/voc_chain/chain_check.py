"""Developer check of voc_chain.hip: the one-launch vocoder step of small stream-sets against the launch plans (CONAN_VOC_CHAIN=0)
on the same inputs: python tools/chain_check.py [slots] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from conan_amd import configs, synth
from conan_amd.runtime import Context


def run(S, steps, chain, frames_list):
    os.environ["CONAN_VOC_CHAIN"] = "1" if chain else "0"      # (the chain is opt-in: the launch plans are faster, see voc_chain_host.hip)
    vhp = configs.hifigan_hparams()
    ctx = Context(None, vhp, 0, False, False, True)
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    st = ctx.streams(S, max_frames=4, max_ref_frames=16, arith="f32")
    ids = list(range(S))
    st.reset(ids)
    mel = torch.from_numpy(synth.mel(4 * steps, 77, S)).cuda()
    outs, p = [], 0
    for k in range(steps):
        f = frames_list[k % len(frames_list)]
        outs.append(st.hifigan_step(ids, mel[:, p:p + f].contiguous()))
        p += f
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(50):
        st.hifigan_step(ids, mel[:, :4].contiguous())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50 * 1e3
    lat = []
    for k in range(30):
        torch.cuda.synchronize()
        a = time.perf_counter()
        st.hifigan_step(ids, mel[:, :4].contiguous())
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - a) * 1e3)
    w = torch.cat(outs, 1).cpu().numpy()
    st.close(); ctx.close()
    return w, dt, sorted(lat)[len(lat) // 2]


if __name__ == "__main__":
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    for fl in ([4], [4, 2, 3, 1]):
        if S * 4 > 16:
            continue
        a, ta, la = run(S, steps, True, fl)
        b, tb, lb = run(S, steps, False, fl)
        err = float(np.abs(a - b).max())
        print(f"slots {S} frames {fl}: chain vs launches max |d wav| {err:.3e} (max |wav| {float(np.abs(b).max()):.3f}); back-to-back {ta:.3f} vs {tb:.3f} ms per step, blocking p50 {la:.3f} vs {lb:.3f} ms", flush=True)
        assert np.isfinite(a).all() and err < 2e-4, err
