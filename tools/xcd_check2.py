"""Developer check: single-tile decoder steps of odd shapes (windowed steps: reset + T frames) xcd mode against the separate launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conan_amd import _lib, configs, synth
from conan_amd.runtime import Context

chp = dict(configs.conan_hparams(), chunk_size=40)
ctx = Context(chp, None, 0, False, True, False)
ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
ctx.finalize()
for S in (1, 2, 3):
    a = ctx.streams(S, max_frames=18, max_ref_frames=64)
    b = ctx.streams(S, max_frames=18, max_ref_frames=64, flags=_lib.STREAMS_SEPARATE_SMALL_STEPS)
    ids = list(range(S))
    ref = torch.from_numpy(synth.mel(40, 60, S)).cuda()
    for st in (a, b):
        st.reset(ids); st.set_reference(ids, ref)
    codes = torch.from_numpy(synth.codes(64, S, seed=9)).int().cuda()
    for T in (2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
        if S * T > 16 and S * T != 16:
            pass
        c = codes[:, :T].contiguous()
        for st in (a, b):
            st.reset(ids, which=2)
        ma, mb = a.decoder_step(ids, c), b.decoder_step(ids, c)
        err = float((ma - mb).abs().max())
        print(f"S {S} T {T} rows {S*T}: max |d mel| {err:.3e}", "" if err < 2e-5 else "  <-- MISMATCH", flush=True)
    a.close(); b.close()
