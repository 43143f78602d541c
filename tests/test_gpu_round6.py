"""Round-6 GPU tests: the fixed-plan flag (a stream's bits do not depend on which other slots are active), conv_limb's ragged last
tiles with more than two slots per tile (40 ms chunks), two contexts on one device, and the ABI-8 option block."""
import numpy as np
import pytest
import torch

from conan_amd import _lib, configs, synth
from tests.conftest import ARITHS, kernels_of

pytestmark = pytest.mark.gpu


def _ctx(emformer=True, conan=True, hifigan=True, chp=None):
    from conan_amd.runtime import Context
    chp, vhp = chp or configs.conan_hparams(), configs.hifigan_hparams()
    ctx = Context(chp if (emformer or conan) else None, vhp if hifigan else None, 0, emformer, conan, hifigan)
    if emformer:
        ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
    if conan:
        ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    if hifigan:
        ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    return ctx, chp, vhp


@pytest.mark.parametrize("arith", ARITHS)
def test_fixed_plan_stream_bits_do_not_depend_on_the_active_slots(arith):
    """CONAN_STREAMS_FIXED_PLAN (ABI 8): slot 41 of a 64-slot stream-set stepped with 64, 20 and 1 active slots - at different
    positions of the active list, its reference set in a batch of 64 / 20 / alone - gives bit-identical codes, mel and audio over
    12 fused chunk steps (the reference is batch-1 and deterministic per utterance, inference/Conan.py:109-113).  Without the flag
    the three differ (fp32 re-association: split-K factors, kernel forms and the decoder's single-tile form follow the active
    count) - asserted too, so that the test cannot pass vacuously - but stay within the tolerance every form is held to."""
    ctx, chp, _ = _ctx()
    S, K, steps = 64, 41, 12
    src = torch.from_numpy(synth.mel(4 * steps + 2, 1234, S)).cuda()
    ref = torch.from_numpy(synth.mel(40, 4321, S)).cuda()
    lens = [40 - (i % 5) for i in range(S)]
    actives = [list(range(S)), [K] + [i for i in range(0, 57, 3)], [K]]        # K at position 41, 0, 0; 64 / 20 / 1 active
    assert len(actives[1]) == 20 and len(set(actives[1])) == 20

    def run(flags):
        outs = []
        for act in actives:
            st = ctx.streams(S, max_frames=4, max_ref_frames=64, arith=arith, flags=flags)
            st.reset(act)
            idx = torch.tensor(act, device="cuda")
            st.set_reference(act, ref[idx].contiguous(), [lens[i] for i in act])
            k = act.index(K)
            c, m, w = [], [], []
            for t in range(steps):
                chunk = src[idx, 4 * t:4 * t + 6].contiguous()
                cc, mm, ww = st.step(act, chunk)
                c.append(cc[k].clone()); m.append(mm[k].clone()); w.append(ww[k].clone())
            outs.append((torch.cat(c), torch.cat(m), torch.cat(w)))
            st.close()
        return outs

    fixed = run(_lib.STREAMS_FIXED_PLAN)
    for o in fixed[1:]:
        assert torch.equal(o[0], fixed[0][0]), "codes differ"
        assert torch.equal(o[1], fixed[0][1]), float((o[1] - fixed[0][1]).abs().max())
        assert torch.equal(o[2], fixed[0][2]), float((o[2] - fixed[0][2]).abs().max())
    default = run(0)
    assert any(not torch.equal(o[2], default[0][2]) for o in default[1:]), "the default plan did not depend on the active set: the test shows nothing"
    for o in default[1:] + fixed:
        assert torch.equal(o[0], default[0][0])
        assert float((o[1] - default[0][1]).abs().max()) <= 1e-4
        assert float((o[2] - default[0][2]).abs().max()) <= 1e-4
    ctx.close()


def test_fixed_plan_with_40ms_chunks_and_ragged_tiles():
    """The same promise at BASELINE configs[4]'s chunk size (40 ms: 2 frames per step, 16 rows per slot in the C = 256 stage - four
    slots per conv_limb tile): slot 17 of a 32-slot fixed-plan stream-set with 32, 9 and 1 active slots (9 and 1 end in ragged tiles
    where the full set's tiles are whole: the shape is chosen as the full set would, the launch's own last tile may be ragged) - codes,
    mel and audio bit-identical over 16 steps."""
    chp = dict(configs.conan_hparams(), chunk_size=40)
    ctx, chp, _ = _ctx(chp=chp)
    S, K, steps, seg = 32, 17, 16, 2
    src = torch.from_numpy(synth.mel(seg * steps + 2, 77, S)).cuda()
    ref = torch.from_numpy(synth.mel(36, 78, S)).cuda()
    actives = [list(range(S)), [3, 5, 8, K, 20, 21, 22, 30, 31], [K]]
    outs = []
    for act in actives:
        st = ctx.streams(S, max_frames=seg, max_ref_frames=64, flags=_lib.STREAMS_FIXED_PLAN)
        st.reset(act)
        idx = torch.tensor(act, device="cuda")
        st.set_reference(act, ref[idx].contiguous())
        k = act.index(K)
        c, m, w = [], [], []
        for t in range(steps):
            cc, mm, ww = st.step(act, src[idx, seg * t:seg * t + seg + 2].contiguous())
            c.append(cc[k].clone()); m.append(mm[k].clone()); w.append(ww[k].clone())
        outs.append((torch.cat(c), torch.cat(m), torch.cat(w)))
        st.close()
    for o in outs[1:]:
        for x, y in zip(o, outs[0]):
            assert torch.equal(x, y), float((x.float() - y.float()).abs().max())
    assert torch.isfinite(outs[0][2]).all()
    ctx.close()


def test_fixed_plan_at_full_occupancy_runs_the_default_kernels():
    """At 64 of 64 slots active the fixed plan runs the default's kernels (the split-K factors of conv_tall - ups.0, ups.1 - are one
    per launch and sized by max_slots; conv_mfma's split TAILS, whose position follows the active list, are the one thing a fixed-plan
    stream-set drops, and no launch of this size has one any more); audio within fp32 re-association."""
    ctx, _, vhp = _ctx(emformer=False, conan=False)
    S = 64
    ids = list(range(S))
    mel = torch.from_numpy(synth.mel(8, 5, S)).cuda()
    a = ctx.streams(S, max_frames=4, max_ref_frames=16)
    b = ctx.streams(S, max_frames=4, max_ref_frames=16, flags=_lib.STREAMS_FIXED_PLAN)
    for st in (a, b):
        st.reset(ids)
    wa, wb = a.hifigan_step(ids, mel[:, :4].contiguous()), b.hifigan_step(ids, mel[:, :4].contiguous())
    assert float((wa - wb).abs().max()) <= 2e-5
    na = kernels_of(a, lambda: a.hifigan_step(ids, mel[:, 4:].contiguous()))
    nb = kernels_of(b, lambda: b.hifigan_step(ids, mel[:, 4:].contiguous()))
    assert na == nb and na.get("cnk::conv_tall_kernel") == 2, (sorted(na.items()), sorted(nb.items()))
    a.close(); b.close(); ctx.close()


@pytest.mark.parametrize("S", [30, 17, 33, 34, 45])
def test_conv_limb_ragged_tiles_with_several_slots_per_tile(S):
    """40 ms chunks: 2 frames per step = 16 rows per slot in the C = 256 stage, so conv_limb's 64-row tiles hold FOUR slots and a
    ragged last tile stages up to three slots past the active count (the slot table carried one copy of the last
    slot; it now carries kSlotTablePad).  S % 4 in {1, 2} with every slot of the stream-set active (S = max_slots: the table's end),
    limb against the f32 form of the same steps at the cross-form tolerance, the grouped limb launches asserted."""
    ctx, _, vhp = _ctx(emformer=False, conan=False)
    ids = list(range(S))
    mel = torch.from_numpy(synth.mel(12, 77, S)).cuda()
    a = ctx.streams(S, max_frames=2, max_ref_frames=16, arith="limb")
    b = ctx.streams(S, max_frames=2, max_ref_frames=16, arith="f32")
    for st in (a, b):
        st.reset(ids)
    for p in range(0, 12, 2):
        wa, wb = a.hifigan_step(ids, mel[:, p:p + 2].contiguous()), b.hifigan_step(ids, mel[:, p:p + 2].contiguous())
        assert torch.isfinite(wa).all()
        assert float((wa - wb).abs().max()) <= 2e-5, (p, float((wa - wb).abs().max()))
    names = kernels_of(a, lambda: a.hifigan_step(ids, mel[:, :2].contiguous()))
    b.hifigan_step(ids, mel[:, :2].contiguous())            # (the same step for the other set: the two stay in the same state)
    # (five of the stage's six grouped launches: the dilation-5 c1 launch's four-slot window - 4 x (16 + 50) rows - does not fit the
    # kernel's two LDS buffers and takes conv_mfma)
    # (17 / 18 slots of 16 rows are too few tiles for a third of the CUs: those sizes take conv_mfma here and are kept as the f32-plan
    # cross-check of the same slot tables)
    if S >= 30:
        assert sum(n for k, n in names.items() if "conv_limb_kernel<4, 1, 1, 4>" in k) >= 5, sorted(names.items())
    # the same set with only S - 1 and S - 2 slots active (other ragged remainders, entries past n that are live slots of the table)
    for n in (S - 1, S - 2):
        sub = ids[:n]
        x = mel[:n, :2].contiguous()
        wa, wb = a.hifigan_step(sub, x), b.hifigan_step(sub, x)
        assert float((wa - wb).abs().max()) <= 2e-5
    a.close(); b.close(); ctx.close()


def test_two_contexts_on_one_device_share_the_live_count():
    """Two contexts (two models / voices in one server) on one GPU: the count of live stream-sets that decides whether a blocking
    Emformer step may take one workgroup per CU is per DEVICE, not per context - each context's only stream-set is
    not alone.  The Emformer's result does not depend on its cluster size, so the check is behavioural: both contexts step blocking
    and pipelined chunks interleaved, every result equals the single-context run bit for bit, nothing trips a bounded wait."""
    ctx1, chp, _ = _ctx()
    S, steps = 64, 6
    src = torch.from_numpy(synth.mel(4 * steps + 2, 99, S)).cuda()
    ref = torch.from_numpy(synth.mel(40, 98, S)).cuda()
    ids = list(range(S))

    def fresh(ctx, flags=0):
        st = ctx.streams(S, max_frames=4, max_ref_frames=64, flags=flags)
        st.reset(ids); st.set_reference(ids, ref)
        return st

    solo = fresh(ctx1)
    want = [tuple(x.clone() for x in solo.step(ids, src[:, 4 * t:4 * t + 6].contiguous())) for t in range(steps)]
    solo.close()
    ctx2, _, _ = _ctx()
    a, b = fresh(ctx1), fresh(ctx2, flags=_lib.STREAMS_SHARED_DEVICE)
    wav_b = torch.empty(S, 4 * ctx2.hop, device="cuda")
    for t in range(steps):
        chunk = src[:, 4 * t:4 * t + 6].contiguous()
        b.step_async(ids, chunk, wav_b)                    # pipelined step of the other context in flight ...
        got = a.step(ids, chunk)                           # ... beside this context's blocking step
        b.join(); torch.cuda.synchronize()
        for x, y in zip(got, want[t]):
            assert torch.equal(x, y)
        assert torch.equal(wav_b, want[t][2])
    a.close(); b.close(); ctx1.close(); ctx2.close()


def test_option_block_abi8():
    """conan_streams_opts (ABI 8): dev_plan names are validated, the retired bit 4 and unknown bits are refused, FIXED_PLAN and
    SHARED_DEVICE are accepted; a process's environment no longer changes the plan (CONAN_EMF_UNFUSED=1 in the environment is
    ignored by the shipped library: the fused Emformer kernel still runs)."""
    import os
    ctx, chp, _ = _ctx(conan=False, hifigan=False)
    for flags in (_lib.STREAMS_FIXED_PLAN, _lib.STREAMS_SHARED_DEVICE, _lib.STREAMS_FIXED_PLAN | _lib.STREAMS_SHARED_DEVICE | _lib.STREAMS_SEPARATE_SMALL_STEPS):
        ctx.streams(2, flags=flags).close()
    for bad in (4, 32, 1 << 9):
        with pytest.raises(_lib.ConanError) as ei:
            ctx.streams(2, flags=bad)
        assert ei.value.code == _lib.ERR_INVALID
    with pytest.raises(_lib.ConanError) as ei:
        ctx.streams(2, dev_plan="EMF_UNFUSED=1;BOGUS=2")
    assert ei.value.code == _lib.ERR_INVALID and "BOGUS" in str(ei.value)
    chunk = torch.from_numpy(synth.mel(6, 3, 2)).cuda()
    old = os.environ.get("CONAN_EMF_UNFUSED")
    os.environ["CONAN_EMF_UNFUSED"] = "1"
    try:
        st = ctx.streams(2)
    finally:
        if old is None:
            os.environ.pop("CONAN_EMF_UNFUSED", None)
        else:
            os.environ["CONAN_EMF_UNFUSED"] = old
    st.reset([0, 1])
    names = kernels_of(st, lambda: st.emformer_step([0, 1], chunk))
    assert any("emformer_fused_kernel" in k for k in names), sorted(names)
    per_op = ctx.streams(2, dev_plan="EMF_UNFUSED=1")
    per_op.reset([0, 1])
    names = kernels_of(per_op, lambda: per_op.emformer_step([0, 1], chunk))
    assert not any("emformer_fused_kernel" in k for k in names), sorted(names)
    st.close(); per_op.close(); ctx.close()


def test_bench_launcher_child_process_path_with_one_rank():
    """bench.py's own launcher (task 1) on real hardware as far as a 1-GPU box allows: CONAN_BENCH_FORCE_SPAWN=1 makes
    `python bench.py --gpus 1` take the path `--gpus 8` takes on an 8-GPU node - torch.distributed.run started as a CHILD process on
    127.0.0.1, the rank's JSON line relayed, its exit code returned - with one rank (the ranks' RCCL side is covered by the world-1
    test of tests/test_gpu_round5.py and the gloo tests)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["CONAN_BENCH_FORCE_SPAWN"] = "1"
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--prime", "0", "--latency-steps", "2",
                        "--no-cpu-baseline", "--no-other", "--no-b1", "--streams", "8"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["ranks"]["rccl_world"] == 1
