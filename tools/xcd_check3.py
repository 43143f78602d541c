"""Developer check: windowed engine steps (Emformer step, reset, decoder + vocoder over the window) at 2 streams, xcd mode vs separate launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conan_amd import _lib, configs, synth
from conan_amd.runtime import Context
from conan_amd.engine import StreamingVoiceConversionEngine

chp = dict(configs.conan_hparams(), chunk_size=40); vhp = configs.hifigan_hparams()
ctx = Context(chp, vhp, 0)
ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0)); ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0)); ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
ctx.finalize()
B, CTX, SEG = 2, 16, 2
T = 11 * SEG + 2
src = torch.from_numpy(np.concatenate([synth.mel(T, 1234 + s) for s in range(B)])).cuda()
ref = torch.from_numpy(np.concatenate([synth.mel(40, 4321 + s) for s in range(B)])).cuda()
res = {}
for name, env in (("xcd", "1"), ("launches", "0")):
    os.environ["CONAN_MEGA_SINGLE"] = env
    eng = StreamingVoiceConversionEngine(ctx, B, max_ref_frames=64, max_frames=CTX + SEG)
    eng.start(ref)
    h = torch.zeros(B, 0, dtype=torch.int32, device="cuda")
    out = []
    for k in range(11):
        c, w, m = eng.windowed_step(src[:, k * SEG:k * SEG + SEG + 2].contiguous(), h[:, -CTX:], return_mel=True)
        h = torch.cat([h, c], 1)
        out.append((c.clone(), m.clone()))
    res[name] = out
    eng.st.close()
for k in range(11):
    ca, ma = res["xcd"][k]; cb, mb = res["launches"][k]
    print(f"step {k} window frames {min(2 * k, 16) + 2}: codes equal {bool(torch.equal(ca, cb))} max |d mel| {float((ma - mb).abs().max()):.3e}", flush=True)
