"""Developer check: the limb kernels against the f32 MFMA kernels at odd stream counts (100 / 40 / 130 slots) over ragged steps
(full batch of 4 frames, half the slots with 3 and 1 frames): max |difference| of the audio and the kernels that ran."""
import os, sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from conan_amd import configs, synth
from conan_amd.runtime import Context
vhp = configs.hifigan_hparams()
ctx = Context(None, vhp, 0, False, False, True); ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0)); ctx.finalize()
for S in (100, 40, 130):
    os.environ["CONAN_RB_LIMB"] = "1"
    a = ctx.streams(S, max_frames=4, max_ref_frames=16)
    os.environ.pop("CONAN_RB_LIMB", None)
    b = ctx.streams(S, max_frames=4, max_ref_frames=16)
    ids = list(range(S)); mel = torch.from_numpy(synth.mel(12, 3, S)).cuda()
    for st in (a, b): st.reset(ids)
    worst = 0.0
    for (i, n) in ((0, 4), (4, 3), (7, 1), (8, 4)):
        sub = ids if n == 4 else ids[::2]
        m = mel[sub, i:i + n].contiguous()
        wa = a.hifigan_step(sub, m); wb = b.hifigan_step(sub, m)
        worst = max(worst, float((wa - wb).abs().max()))
    a.profile_begin(); a.hifigan_step(ids, mel[:, :4].contiguous()); a.profile_end()
    print("S", S, "max |limb - f32| over 4 ragged steps", worst, sorted({r[0].split('<')[0] for r in a.profile_kernels()}))
    a.close(); b.close()
