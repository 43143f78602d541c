"""Developer check: the flow of tests/test_gpu_round2.py::test_config4_b128_seg2_windowed_320ms, step by step."""
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
B, CTX, SEG, Tr = 128, 16, 2, 40
nchunks = CTX // SEG + 3
T = nchunks * SEG + 2
src = torch.from_numpy(np.concatenate([synth.mel(T, 1234 + s) for s in range(B)])).cuda()
ref = torch.from_numpy(np.concatenate([synth.mel(Tr, 4321 + s) for s in range(B)])).cuda()
eng = StreamingVoiceConversionEngine(ctx, B, max_ref_frames=64, max_frames=CTX + SEG, arith="f32")
eng.start(ref)
hist = torch.zeros(B, 0, dtype=torch.int32, device="cuda")
outs = []
for k in range(nchunks):
    codes, wav, mel = eng.windowed_step(src[:, k * SEG:k * SEG + SEG + 2].contiguous(), hist[:, -CTX:], return_mel=True)
    hist = torch.cat([hist, codes], 1)
    outs.append((wav, mel))
solo = StreamingVoiceConversionEngine(ctx, 2, max_ref_frames=64, max_frames=CTX + SEG, arith="f32")
for pair in ([5, 100], [63, 64]):
    solo.start(ref[pair])
    h = torch.zeros(2, 0, dtype=torch.int32, device="cuda")
    for kk in range(nchunks):
        c2, w2, m2 = solo.windowed_step(src[pair, kk * SEG:kk * SEG + SEG + 2].contiguous(), h[:, -CTX:], return_mel=True)
        h = torch.cat([h, c2], 1)
        print(f"pair {pair} step {kk}: codes equal {bool(torch.equal(c2, hist[pair][:, kk*SEG:kk*SEG+SEG]))} max |d mel| {float((m2 - outs[kk][1][pair]).abs().max()):.3e} max |d wav| {float((w2 - outs[kk][0][pair]).abs().max()):.3e}", flush=True)
