"""Round-5 GPU tests: the decoder megakernel's xcd mode (single-tile steps), the deployment flags of conan_streams_opts, and the reference fixtures at the stream counts where the vocoder's plan
changes."""
import os

import numpy as np
import pytest
import torch

from conan_amd import _lib, configs, synth
from tests.conftest import ARITHS, assert_arith_ran, kernels_of, load_golden

pytestmark = pytest.mark.gpu


def _ctx(emformer=False, conan=True, hifigan=True):
    from conan_amd.runtime import Context
    chp, vhp = configs.conan_hparams(), configs.hifigan_hparams()
    ctx = Context(chp if (emformer or conan) else None, vhp if hifigan else None, 0, emformer, conan, hifigan)
    if emformer:
        ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
    if conan:
        ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    if hifigan:
        ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    return ctx, chp, vhp


@pytest.mark.parametrize("S,K", [(1, 0), (4, 2)])
def test_conan_golden_through_the_xcd_mode_megakernel(S, K):
    """tests/golden/conan_full.npz (Conan.forward of the imported reference, modules/Conan/Conan.py:115-198: 150 frames, hidden 256)
    through slot K of an S-slot stream-set, 4 frames per step: a step is a single row tile (S x 4 <= 16 rows), i.e. ONE
    decoder_mega_kernel<4, 2> launch - the group forms on one XCD at run time, hand-offs through that XCD's L2, flag barriers
    (decoder_mega.hip).  mel_out against the reference's at 1e-4; the ragged last step (2 frames) takes the same launch; the
    separate launches (CONAN_STREAMS_SEPARATE_SMALL_STEPS) agree within fp32 re-association."""
    g = load_golden("conan_full.npz")
    ctx, chp, _ = _ctx(hifigan=False)
    ids = list(range(S))
    refs = torch.from_numpy(synth.mel(150, 50, S)).cuda()
    refs[K] = torch.from_numpy(g["ref"][0]).cuda()
    lens = [150 - 5 * (i % 3) for i in range(S)]
    lens[K] = 150
    codes = torch.from_numpy(synth.codes(150, S, seed=5)).int().cuda()
    codes[K] = torch.from_numpy(g["content"][0]).int().cuda()
    outs = []
    for flags in (0, _lib.STREAMS_SEPARATE_SMALL_STEPS):
        st = ctx.streams(S, max_frames=4, max_ref_frames=160, flags=flags)
        st.reset(ids)
        st.set_reference(ids, refs, lens)
        mel = torch.cat([st.decoder_step(ids, codes[:, p:p + 4].contiguous())[K] for p in range(0, 150, 4)]).cpu().numpy()
        assert mel.shape == (150, 80)
        np.testing.assert_allclose(mel, g["mel_out"][0], atol=1e-4, rtol=1e-4)
        names = kernels_of(st, lambda: st.decoder_step(ids, codes[:, :4].contiguous()))
        if flags == 0:
            assert any("decoder_mega_kernel<4, 2>" in k for k in names) and not any("rowconv_kernel" in k for k in names), sorted(names)
        else:
            assert any("rowconv_kernel" in k for k in names) and not any("decoder_mega_kernel" in k for k in names), sorted(names)
        outs.append(mel)
        st.close()
    np.testing.assert_allclose(outs[0], outs[1], atol=2e-5, rtol=1e-5)
    ctx.close()


def test_xcd_mode_is_bit_reproducible_and_survives_concurrent_stream_sets():
    """Three stream-sets of 1 / 2 / 4 slots step their decoders on three HIP streams at once, 60 steps each - three xcd-mode groups
    compete for XCDs (workgroups are bound to XCDs: with a roll call of the whole grid two groups that each hold an XCD wait for each
    other for ever; the quorum election does not) - and every one reproduces, bit for bit, what it computes alone."""
    ctx, chp, _ = _ctx(hifigan=False)
    sizes = (1, 2, 4)
    sets = [ctx.streams(S, max_frames=4, max_ref_frames=64) for S in sizes]
    refs = [torch.from_numpy(synth.mel(40, 70 + S, S)).cuda() for S in sizes]
    codes = [torch.from_numpy(synth.codes(240, S, seed=11 + S)).int().cuda() for S in sizes]

    def run(concurrent):
        for st, S, r in zip(sets, sizes, refs):
            st.reset(list(range(S))); st.set_reference(list(range(S)), r)
        torch.cuda.synchronize()
        outs = [[] for _ in sizes]
        streams = [torch.cuda.Stream() for _ in sizes] if concurrent else [torch.cuda.current_stream()] * 3
        for k in range(60):
            for q, (st, S) in enumerate(zip(sets, sizes)):
                with torch.cuda.stream(streams[q]):
                    outs[q].append(st.decoder_step(list(range(S)), codes[q][:, 4 * k:4 * k + 4].contiguous()))
        torch.cuda.synchronize()
        return [torch.cat(o, 1) for o in outs]

    alone, together = run(False), run(True)
    for a, b in zip(alone, together):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    for st in sets:
        st.close()
    ctx.close()


def test_single_tile_and_multi_tile_steps_interleave():
    """One stream-set of 2 slots stepping 2, 16, 4, 8, 16, 2 frames after a reset each (windowed serving): 2 x 16 frames are two row
    tiles - the multi-tile launch, groups of 8 - the others one tile - xcd mode.  The two forms keep their own launch sequences (the xcd
    election word must be found in the state the last xcd launch left it in); every step equals the separate launches."""
    ctx, chp, _ = _ctx(hifigan=False)
    ids = [0, 1]
    ref = torch.from_numpy(synth.mel(40, 60, 2)).cuda()
    codes = torch.from_numpy(synth.codes(16, 2, seed=9)).int().cuda()
    a = ctx.streams(2, max_frames=16, max_ref_frames=64)
    b = ctx.streams(2, max_frames=16, max_ref_frames=64, flags=_lib.STREAMS_SEPARATE_SMALL_STEPS)
    for st in (a, b):
        st.reset(ids); st.set_reference(ids, ref)
    for T in (2, 16, 4, 8, 16, 2):
        for st in (a, b):
            st.reset(ids, which=2)
        c = codes[:, :T].contiguous()
        np.testing.assert_allclose(a.decoder_step(ids, c).cpu().numpy(), b.decoder_step(ids, c).cpu().numpy(), atol=2e-5, rtol=1e-5)
    names = kernels_of(a, lambda: a.decoder_step(ids, codes[:, :16].contiguous()))
    assert any("decoder_mega_kernel<4, 3>" in k or "decoder_mega_kernel<6, 3>" in k for k in names), sorted(names)
    a.close(); b.close(); ctx.close()


def test_xcd_mode_election_fault_is_reported():
    """conan_streams_test_fault(1) on a single-tile stream-set: the election word is not in the state the launch expects, nobody can
    claim an XCD - every workgroup gives up after the 50 ms budget, the launch ends, the next entry point returns CONAN_ERR_HIP, and a
    fresh stream-set on the same context reproduces the healthy step bit for bit."""
    ctx, chp, _ = _ctx(hifigan=False)
    ref = torch.from_numpy(synth.mel(40, 3, 2)).cuda()
    codes = torch.from_numpy(synth.codes(8, 2, seed=3)).int().cuda()

    def fresh():
        st = ctx.streams(2, max_frames=4, max_ref_frames=64)
        st.reset([0, 1]); st.set_reference([0, 1], ref)
        return st
    a = fresh()
    want = a.decoder_step([0, 1], codes[:, :4].contiguous()).clone()
    a.close()
    b = fresh()
    _lib.check(ctx.lib.conan_streams_test_fault(b.h, 1))
    b.decoder_step([0, 1], codes[:, :4].contiguous())
    torch.cuda.synchronize()
    with pytest.raises(_lib.ConanError) as ei:
        b.decoder_step([0, 1], codes[:, 4:8].contiguous())
    assert ei.value.code == _lib.ERR_HIP and "xcd election" in str(ei.value)
    b.close()
    c = fresh()
    assert torch.equal(c.decoder_step([0, 1], codes[:, :4].contiguous()), want)
    c.close(); ctx.close()


# the vocoder's plan per stream-set size (csrc/streams.hip build_vocoder, conv_limb_shape): limb stream-sets fuse the C = 128 / 64
# stages from 4 slots on and run the C = 256 stage as grouped conv_limb launches from 16 slots on; f32 stream-sets fuse from 8 slots
# on and take the pair kernel from 16 slots on
@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("S", [3, 4, 15, 16, 17, 40, 96])
def test_vocoder_golden_at_the_plan_switch_boundaries(S, arith):
    """tests/golden/loop_full.npz's mel (the reference loop's mel for a given code sequence) -> wav through slot S - 2 of an S-slot
    stream-set, chunk by chunk, at the stream counts on both sides of every plan switch, both arithmetic forms, the kernels that ran
    asserted per size.  wav against the reference's at 1e-4."""
    g = load_golden("loop_full.npz")
    ctx, _, vhp = _ctx(conan=False)
    K = S - 2
    T = g["mel"].shape[0]
    mels = torch.from_numpy(synth.mel(T, 17, S)).cuda()
    mels[K] = torch.from_numpy(g["mel"]).cuda()
    st = ctx.streams(S, max_frames=4, max_ref_frames=16, arith=arith)
    assert st.arith == arith
    ids = list(range(S))
    st.reset(ids)
    wav = torch.cat([st.hifigan_step(ids, mels[:, p:p + 4].contiguous())[K] for p in range(0, T, 4)]).cpu().numpy()
    np.testing.assert_allclose(wav, g["wav"], atol=1e-4, rtol=0)
    vn = kernels_of(st, lambda: st.hifigan_step(ids, mels[:, :4].contiguous()))
    has = lambda sub: any(sub in k for k in vn)
    if arith == "limb":
        assert has("resblock_limb_kernel<32,")
        assert has("resblock_limb_kernel<128,") == (S >= 4) and has("resblock_limb_kernel<64,") == (S >= 4), sorted(vn)
        # the C = 256 stage: six grouped conv_limb launches from 16 slots on, below that conv_mfma's two-launch plan (then more than
        # conv_pre / ups.0 / ups.1 run on conv_mfma)
        n_mfma = sum(n for k, n in vn.items() if "conv_mfma_kernel" in k)
        assert (n_mfma <= 4) == (S >= 16), (n_mfma, sorted(vn.items()))
        assert (sum(n for k, n in vn.items() if "conv_limb_kernel<4, 1, 1, 4>" in k) >= 6) or S < 16, sorted(vn.items())
        assert not has("resblock_pair_kernel") and not has("resblock_fused_kernel"), sorted(vn)
    else:
        assert_arith_ran(vn, "f32")
        assert has("resblock_fused_kernel<32,")
        assert has("resblock_fused_kernel<128,") == (S >= 8) and has("resblock_pair_kernel") == (S >= 16), sorted(vn)
    st.close(); ctx.close()


def test_deployment_flags_through_the_c_abi():
    """conan_streams_opts.flags: CONAN_STREAMS_FUSED_DECODER_BLOCKS gives the fused conv-block operators (28 instead of 36
    operators: the decoder step still equals the default within fp32 re-association), unknown bits are refused."""
    ctx, chp, _ = _ctx(hifigan=False)
    S = 24
    ids = list(range(S))
    ref = torch.from_numpy(synth.mel(40, 8, S)).cuda()
    codes = torch.from_numpy(synth.codes(16, S)).int().cuda()
    a = ctx.streams(S, 4, 64)
    b = ctx.streams(S, 4, 64, flags=_lib.STREAMS_FUSED_DECODER_BLOCKS)
    for st in (a, b):
        st.reset(ids); st.set_reference(ids, ref)
    for p in range(0, 16, 4):
        ma, mb = a.decoder_step(ids, codes[:, p:p + 4].contiguous()), b.decoder_step(ids, codes[:, p:p + 4].contiguous())
        np.testing.assert_allclose(ma.cpu().numpy(), mb.cpu().numpy(), atol=2e-5, rtol=1e-5)
    a.close(); b.close()
    with pytest.raises(_lib.ConanError) as ei:
        ctx.streams(2, flags=1 << 9)
    assert ei.value.code == _lib.ERR_INVALID
    ctx.close()


_RCCL_WORLD1 = r"""
import socket, sys, torch
import torch.distributed as dist
sys.path.insert(0, ".")
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
try:
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
except Exception as e:
    print(f"RCCL-INIT-FAILED {type(e).__name__}: {e}"); sys.exit(0)
dev = torch.device("cuda", 0)
grp = torch.randn(4, 64, 1280, device=dev)
out = [torch.empty_like(grp)]
side = torch.cuda.Stream(); done = torch.cuda.Event()
with torch.cuda.stream(side):
    side.wait_stream(torch.cuda.current_stream())
    dist.gather(grp, out, dst=0)
    done.record(side)
torch.cuda.current_stream().wait_event(done)
assert torch.equal(out[0], grp)
csum = grp.view(torch.int32).to(torch.int64).sum().reshape(1)
sums = [torch.zeros_like(csum)]
dist.all_gather(sums, csum)
assert int(sums[0].item()) == int(csum.item())
dist.barrier(); torch.cuda.synchronize()
from conan_amd.engine import AudioGatherRing
seen = []
ring = AudioGatherRing(lambda: torch.empty(8, 1280, device=dev), 1, 0, always=True, every=4, on_gathered=lambda j, bufs: seen.append((j, bufs[0].clone())))
for j in range(9):
    buf, fence = ring.acquire(j, fence=True)
    buf.fill_(float(j))
    ring.submit(j, join=None, wait_current=True)
ring.flush(8); ring.drain()
assert [j for j, _ in seen] == list(range(9)) and all(float(b.mean()) == float(j) for j, b in seen)
dist.destroy_process_group()
print("RCCL-WORLD1-OK")
"""


def test_rccl_process_group_runs_the_gather_choreography_world1():
    """SURVEY.md §8e's only exchange, on the real backend: an RCCL ("nccl") process group of ONE rank on this GPU runs the very
    collectives of the N > 1 bench path - the grouped gather of finished audio on a side stream behind an event fence, the all_gather
    of checksums, the barrier - with the tensors the 64-stream workload uses ([4 steps, 64 streams, 1280 samples] fp32), and
    AudioGatherRing with its collective path switched on.  It cannot show scaling; it shows that RCCL initialises on this image and
    that the calls, shapes and stream usage are what RCCL accepts (the multi-rank choreography itself is driven on gloo by
    tests/test_distributed_cpu.py).  In a child process: the process group, and whatever RCCL prints, stay out of this one."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        r = subprocess.run([sys.executable, "-c", _RCCL_WORLD1], cwd=root, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired:
        pytest.skip("the RCCL process group did not come up within 300 s on this box")
    if "RCCL-INIT-FAILED" in r.stdout:
        pytest.skip("RCCL process group could not be initialised here: " + r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and "RCCL-WORLD1-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
