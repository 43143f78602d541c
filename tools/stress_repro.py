"""Repeat the 128-stream f32 case of tests/test_gpu_stress.py (a pipelined and a blocking stream-set overlapping on the device) and
count the runs that end in a bounded-wait give-up:  python tools/stress_repro.py [iterations] [dev_plan] [S] [arith]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from conan_amd import _lib, configs, synth  # noqa: E402
from conan_amd.runtime import Context  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
plan = (sys.argv[2] if len(sys.argv) > 2 else "") or None
S = int(sys.argv[3]) if len(sys.argv) > 3 else 128
arith = sys.argv[4] if len(sys.argv) > 4 else "f32"
chp, vhp = configs.conan_hparams(), configs.hifigan_hparams()
ctx = Context(chp, vhp, 0)
ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
ctx.finalize()
hop = ctx.hop
ids = list(range(S))
ref = torch.from_numpy(synth.mel(40, 8, S)).cuda()
base = torch.from_numpy(synth.mel(4 * 64 + 8, 9, S)).cuda()
fails = 0
for run in range(iters):
    a = ctx.streams(S, 4, 64, arith=arith, dev_plan=plan)
    b = ctx.streams(S, 4, 64, arith=arith, dev_plan=plan)
    try:
        for st in (a, b):
            st.reset(ids); st.set_reference(ids, ref)
        rng = np.random.default_rng(17 + S + run)
        pos = [0] * S
        slots = ids
        bad = 0
        for it in range(400):
            if it % 23 == 22:
                n = int(rng.integers(S // 2, S + 1))
                slots = sorted(rng.choice(S, n, replace=False).tolist())
            elif it % 23 == 11:
                slots = ids
            n = len(slots)
            chunk = torch.stack([base[s, (4 * pos[s]) % 256:(4 * pos[s]) % 256 + 6] for s in slots]).contiguous()
            for s in slots:
                pos[s] += 1
            w = torch.empty(n, 4 * hop, device="cuda")
            a.step_async(slots, chunk, w, emit=4)
            wb = b.step(slots, chunk)[2]
            if it % 50 == 49:
                a.join(); torch.cuda.synchronize()
                bad += int(not torch.equal(w, wb))
        a.join(); torch.cuda.synchronize()
        print(f"run {run}: ok, mismatching checks {bad}", flush=True)
    except _lib.ConanError as e:
        fails += 1
        print(f"run {run}: {e}", flush=True)
    for st in (a, b):
        try:
            st.close()
        except Exception:
            pass
print(f"plan {plan!r} S={S} {arith}: {fails} of {iters} runs gave up in a bounded wait")
