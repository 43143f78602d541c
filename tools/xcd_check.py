"""Developer check of the decoder megakernel's xcd mode (single-tile steps: one to four streams) against the separate launches
(CONAN_MEGA_SINGLE=0) on the same inputs, and of the one-launch vocoder step (CONAN_VOC_CHAIN): python tools/xcd_check.py [slots]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from conan_amd import configs, synth
from conan_amd.runtime import Context


def run(S, single, steps=10):
    os.environ["CONAN_MEGA_SINGLE"] = "1" if single else "0"
    chp = configs.conan_hparams()
    ctx = Context(chp, None, 0, False, True, False)
    ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    ctx.finalize()
    st = ctx.streams(S, max_frames=4, max_ref_frames=64)
    ids = list(range(S))
    st.reset(ids)
    st.set_reference(ids, torch.from_numpy(synth.mel(40, 60, S)).cuda())
    codes = torch.from_numpy(synth.codes(4 * steps, S, seed=9)).int().cuda()
    outs = [st.decoder_step(ids, codes[:, 4 * k:4 * k + 4].contiguous()) for k in range(steps)]
    torch.cuda.synchronize()
    c0 = codes[:, :4].contiguous()
    lat = []
    for k in range(40):
        torch.cuda.synchronize()
        a = time.perf_counter()
        st.decoder_step(ids, c0)
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - a) * 1e3)
    m = torch.cat(outs, 1).cpu().numpy()
    st.close(); ctx.close()
    return m, sorted(lat)[len(lat) // 2]


if __name__ == "__main__":
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    a, la = run(S, True)
    b, lb = run(S, False)
    err = float(np.abs(a - b).max())
    print(f"slots {S}: decoder step xcd mode vs separate launches max |d mel| {err:.3e} (max |mel| {float(np.abs(b).max()):.3f}); blocking p50 {la:.3f} vs {lb:.3f} ms", flush=True)
    assert np.isfinite(a).all() and err < 2e-4, err
