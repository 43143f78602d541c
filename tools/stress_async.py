"""Developer stress test: random slot subsets, partial chunks and resets through blocking steps on one stream-set and pipelined
steps on another; every output of every step must be bit-identical (python tools/stress_async.py [tiny] [steps])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from conan_amd import configs, synth
from conan_amd.runtime import Context
tiny = len(sys.argv) > 1 and sys.argv[1] == "tiny"
chp, vhp = configs.conan_hparams(tiny), configs.hifigan_hparams(tiny)
ctx = Context(chp, vhp, 0, True, True, True)
ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0)); ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0)); ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
ctx.finalize()
S = int(sys.argv[3]) if len(sys.argv) > 3 else 12          # (>= 16: the first vocoder stage runs the pair kernel)
a, b = ctx.streams(S, 4, 64), ctx.streams(S, 4, 64)
ref = torch.from_numpy(synth.mel(40, 8, S)).cuda()
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
src = torch.from_numpy(synth.mel(4 * N + 16, 9, S)).cuda()
for st in (a, b):
    st.reset(list(range(S))); st.set_reference(list(range(S)), ref)
rng = np.random.default_rng(0)
pos = [0] * S
hop = ctx.hop
outs_a, outs_b = [], []
for it in range(N):
    n = int(rng.integers(1, S + 1))
    slots = sorted(rng.choice(S, n, replace=False).tolist()) if rng.random() < 0.5 else rng.permutation(S)[:n].tolist()
    emit = 4 if rng.random() < 0.8 else int(rng.integers(1, 4))
    chunk = torch.stack([src[s, pos[s]:pos[s] + 6] for s in slots]).contiguous()
    if emit < 4:
        chunk = torch.cat([chunk[:, :emit], chunk[:, emit - 1:emit].expand(-1, 6 - emit, -1)], 1).contiguous()
    c, m, w = a.step(slots, chunk, emit=emit)
    outs_a.append((c.clone(), m.clone(), w.clone()))
    cb = torch.empty(n, 4, dtype=torch.int32, device="cuda"); mb = torch.empty(n, emit, 80, device="cuda"); wb = torch.empty(n, emit * hop, device="cuda")
    b.step_async(slots, chunk, wb, emit=emit, codes=cb, mel_out=mb)
    outs_b.append((cb, mb, wb))
    for s in slots: pos[s] += emit
    if it % 97 == 96:          # occasionally interleave a blocking call on the pipelined set (joins first)
        rs = [int(rng.integers(0, S))]
        a.reset(rs); b.reset(rs); pos[rs[0]] = 0
b.join(); torch.cuda.synchronize()
bad = [k for k, ((ca, ma, wa), (cb, mb, wb)) in enumerate(zip(outs_a, outs_b)) if not (torch.equal(ca, cb) and torch.equal(ma, mb) and torch.equal(wa, wb))]
print("steps", N, "mismatching steps:", bad[:10], "OK" if not bad else "FAIL")
