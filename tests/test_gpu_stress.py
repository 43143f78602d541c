"""Long pipelined runs at serving sizes, and the bounded device-side waits.

Three kernels wait for other workgroups of their own launch (decoder_mega.hip group / grid barriers, emformer_fused.hip cluster
exchange, resblock_pair.hip partner flags and tile mailbox); forward progress rests on dispatch-order arguments.
(1) 400 pipelined steps at 64 and 128 streams with changing slot subsets, all three stages in flight on their streams, in both
arithmetic forms, must reproduce the blocking loop bit for bit.  (2) Each of those waits carries a 50 ms budget: a fault injected
through conan_streams_test_fault must surface as CONAN_ERR_HIP at the next entry point - not as a hang - and leave the context
usable."""
import time

import numpy as np
import pytest
import torch

from conan_amd import configs, synth
from tests.conftest import ARITHS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full_ctx():
    from conan_amd.runtime import Context
    chp, vhp = configs.conan_hparams(), configs.hifigan_hparams()
    ctx = Context(chp, vhp, 0)
    ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
    ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    yield ctx
    ctx.close()


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("S", [64, 128])
def test_400_pipelined_steps_equal_the_blocking_loop(full_ctx, S, arith):
    ctx = full_ctx
    N, hop = 400, ctx.hop
    a = ctx.streams(S, 4, 64, arith=arith)          # pipelined
    b = ctx.streams(S, 4, 64, arith=arith)          # blocking
    ids = list(range(S))
    ref = torch.from_numpy(synth.mel(40, 8, S)).cuda()
    base = torch.from_numpy(synth.mel(4 * 64 + 8, 9, S)).cuda()      # 64 chunks of source per stream, walked cyclically
    for st in (a, b):
        st.reset(ids); st.set_reference(ids, ref)
    rng = np.random.default_rng(17 + S)
    pos = [0] * S
    slots = ids
    res_a, res_b = [], []
    for it in range(N):
        # the slot subset changes now and then (a changed list drains the in-flight stages first: both paths are exercised);
        # sizes are multiples of 4 streams or not, so megakernel steps and separate-launch steps alternate as well
        if it % 23 == 22:
            n = int(rng.integers(S // 2, S + 1))
            slots = sorted(rng.choice(S, n, replace=False).tolist())
        elif it % 23 == 11:
            slots = ids
        n = len(slots)
        chunk = torch.stack([base[s, (4 * pos[s]) % 256:(4 * pos[s]) % 256 + 6] for s in slots]).contiguous()
        for s in slots:
            pos[s] += 1
        c = torch.empty(n, 4, dtype=torch.int32, device="cuda"); m = torch.empty(n, 4, 80, device="cuda"); w = torch.empty(n, 4 * hop, device="cuda")
        a.step_async(slots, chunk, w, emit=4, codes=c, mel_out=m)
        res_a.append((c, m, w))
        res_b.append(tuple(x.clone() for x in b.step(slots, chunk)))
    a.join(); torch.cuda.synchronize()
    bad = [k for k, (x, y) in enumerate(zip(res_a, res_b)) if not all(torch.equal(p, q) for p, q in zip(x, y))]
    assert not bad, f"pipelined steps that differ from the blocking loop: {bad[:10]} of {len(bad)}"
    assert all(torch.isfinite(w).all() for _, _, w in res_a[::37])
    a.close(); b.close()


@pytest.mark.parametrize("kind,name", [(1, "decoder_mega"), (2, "emformer_fused"), (3, "resblock_pair")])
def test_an_unmet_wait_gives_up_and_is_reported(full_ctx, kind, name):
    from conan_amd import _lib
    ctx = full_ctx
    S = 24                                   # megakernel (6 tiles), Emformer clusters (12 groups x 4), pair kernel (>= 16 slots, f32)
    st = ctx.streams(S, 4, 64, arith="f32")
    ids = list(range(S))
    st.reset(ids); st.set_reference(ids, torch.from_numpy(synth.mel(40, 8, S)).cuda())
    chunk = torch.from_numpy(synth.mel(6, 9, S)).cuda()
    c0, m0, w0 = st.step(ids, chunk)         # a healthy step first
    torch.cuda.synchronize()
    assert torch.isfinite(w0).all()
    _lib.check(ctx.lib.conan_streams_test_fault(st.h, kind))
    t0 = time.perf_counter()
    st.step(ids, chunk)                      # enqueues the launch whose wait can never be met; returns (asynchronous)
    torch.cuda.synchronize()                 # ... and the launch ENDS: the waiters give up after their budget
    dt = time.perf_counter() - t0
    assert 0.03 < dt < 5.0, dt               # the 50 ms budget was spent (not a hang, not a no-op)
    with pytest.raises(_lib.ConanError) as ei:
        st.step(ids, chunk)
    assert ei.value.code == _lib.ERR_HIP and name in str(ei.value)
    with pytest.raises(_lib.ConanError):     # sticky: the stream-set stays refused
        st.reset(ids)
    st.close()
    # the context and the GPU are fine: a fresh stream-set reproduces the healthy step
    st2 = ctx.streams(S, 4, 64, arith="f32")
    st2.reset(ids); st2.set_reference(ids, torch.from_numpy(synth.mel(40, 8, S)).cuda())
    c1, m1, w1 = st2.step(ids, chunk)
    torch.cuda.synchronize()
    assert torch.equal(c0, c1) and torch.equal(m0, m1) and torch.equal(w0, w1)
    st2.close()
