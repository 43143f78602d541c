"""Round-4 GPU tests: the reference-generated fixtures DIRECTLY through the kernels that are the defaults at serving sizes - the
decoder megakernel at the shipped width (hidden 256) and the bf16-limb vocoder kernels at >= 48 slots - and the C-ABI's
arithmetic option."""
import numpy as np
import pytest
import torch

from conan_amd import configs, synth
from conan_amd import _lib
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


def test_conan_golden_through_the_decoder_megakernel():
    """tests/golden/conan_full.npz (Conan.forward of the imported reference, modules/Conan/Conan.py:115-198: 150 frames, hidden
    256) through slot 11 of a 24-slot stream-set WITHOUT taps, 4 frames per step: every full step is one decoder_mega_kernel
    launch (6 row tiles, 6 groups of 8 workgroups); the ragged last step (2 frames) runs the separate launches.  mel_out against
    the reference's at 1e-4."""
    g = load_golden("conan_full.npz")
    ctx, chp, _ = _ctx(hifigan=False)
    S, K = 24, 11
    st = ctx.streams(S, max_frames=4, max_ref_frames=160)
    ids = list(range(S))
    refs = torch.from_numpy(synth.mel(150, 50, S)).cuda()
    refs[K] = torch.from_numpy(g["ref"][0]).cuda()
    lens = [150 - 3 * (i % 7) for i in range(S)]
    lens[K] = 150
    codes = torch.from_numpy(synth.codes(150, S, seed=5)).int().cuda()
    codes[K] = torch.from_numpy(g["content"][0]).int().cuda()
    st.reset(ids)
    st.set_reference(ids, refs, lens)
    mels = [st.decoder_step(ids, codes[:, p:p + 4].contiguous())[K] for p in range(0, 150, 4)]
    mel = torch.cat(mels).cpu().numpy()
    assert mel.shape == (150, 80)
    np.testing.assert_allclose(mel, g["mel_out"][0], atol=1e-4, rtol=1e-4)
    names = kernels_of(st, lambda: st.decoder_step(ids, codes[:, :4].contiguous()))
    assert any("decoder_mega_kernel" in k for k in names) and not any("rowconv_kernel" in k for k in names), sorted(names)
    st.close(); ctx.close()


@pytest.mark.parametrize("arith", ARITHS)
@pytest.mark.parametrize("S,K", [(48, 29), (64, 41), (128, 77)])
def test_loop_golden_at_48_slots(S, K, arith):
    """tests/golden/loop_full.npz (mel / wav of the reference loop inference/Conan.py:95-156 from the imported reference modules
    for a given code sequence) through slot K of an S-slot stream-set, decoder step -> vocoder step per 4-frame chunk as the
    fused chunk step issues them: the decoder as one megakernel launch, the vocoder in both arithmetic forms - from 48 slots on the
    limb form covers every ResBlock stage (conv_limb's grouped launches in the C = 256 stage) and ups.2 / ups.3.  S = 64 is
    BASELINE.json configs[2]'s stream count (the headline), 128 configs[4]'s."""
    g = load_golden("loop_full.npz")
    ctx, chp, vhp = _ctx()
    st = ctx.streams(S, max_frames=4, max_ref_frames=64, arith=arith)
    assert st.arith == arith
    ids = list(range(S))
    T = g["codes"].shape[0]
    refs = torch.from_numpy(synth.mel(40, 60, S)).cuda()
    refs[K] = torch.from_numpy(g["ref"][0]).cuda()
    codes = torch.from_numpy(synth.codes(T, S, seed=9)).int().cuda()
    codes[K] = torch.from_numpy(g["codes"]).int().cuda()
    st.reset(ids)
    st.set_reference(ids, refs)
    mels, wavs = [], []
    for p in range(0, T, 4):
        m = st.decoder_step(ids, codes[:, p:p + 4].contiguous())
        mels.append(m[K]); wavs.append(st.hifigan_step(ids, m)[K])
    mel, wav = torch.cat(mels).cpu().numpy(), torch.cat(wavs).cpu().numpy()
    np.testing.assert_allclose(mel, g["mel"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(wav, g["wav"], atol=1e-4, rtol=0)
    m0 = st.decoder_step(ids, codes[:, :4].contiguous())
    dn = kernels_of(st, lambda: st.decoder_step(ids, codes[:, :4].contiguous()))
    assert any("decoder_mega_kernel" in k for k in dn), sorted(dn)
    vn = kernels_of(st, lambda: st.hifigan_step(ids, m0))
    assert_arith_ran(vn, arith)
    if arith == "limb":
        assert all(any(("resblock_limb_kernel<%d," % c) in k for k in vn) for c in (128, 64, 32)), sorted(vn)
        assert sum(n for k, n in vn.items() if "conv_limb_kernel" in k) >= 6 and not any("resblock_pair_kernel" in k or "resblock_fused_kernel" in k for k in vn), sorted(vn)
    else:
        assert any("resblock_pair_kernel" in k for k in vn) and any("resblock_fused_kernel<128" in k for k in vn), sorted(vn)
    st.close(); ctx.close()


def test_arith_option_through_the_c_abi():
    """conan_streams_create_opts / conan_streams_arith: AUTO resolves to the limb form for a ResBlock1 vocoder and to f32 for a
    context without limb weights (ResBlock2; no vocoder at all), an explicit limb request there is refused with
    CONAN_ERR_UNSUPPORTED, a bad option block with CONAN_ERR_INVALID; conan_streams_create == AUTO."""
    import ctypes as C
    from conan_amd import _lib
    from conan_amd.runtime import Context
    ctx, chp, vhp = _ctx(conan=False)
    a, f, l = ctx.streams(2, arith="auto"), ctx.streams(2, arith="f32"), ctx.streams(2, arith="limb")
    assert (a.arith, f.arith, l.arith) == ("limb", "f32", "limb")
    h = C.c_void_p()
    _lib.check(ctx.lib.conan_streams_create(ctx.h, 2, 4, 16, C.byref(h)))
    assert ctx.lib.conan_streams_arith(h) == _lib.ARITH_LIMB
    ctx.lib.conan_streams_destroy(h)
    for opts, code in ((_lib.StreamsOpts(_lib.ABI_VERSION, 7), _lib.ERR_INVALID), (_lib.StreamsOpts(_lib.ABI_VERSION - 1, 0), _lib.ERR_INVALID),
                       (_lib.StreamsOpts(_lib.ABI_VERSION, 0, 0, 0, None, (C.c_int32 * 2)(0, 1)), _lib.ERR_INVALID),
                       (_lib.StreamsOpts(_lib.ABI_VERSION, 0, 0, 1), _lib.ERR_INVALID),
                       (_lib.StreamsOpts(_lib.ABI_VERSION, 0, 0, 0, b"NOT_A_SWITCH=1"), _lib.ERR_INVALID),      # (an unknown dev_plan name)
                       (_lib.StreamsOpts(_lib.ABI_VERSION, 0, 4), _lib.ERR_INVALID),       # (bit 4: ABI 7's VOCODER_CHAIN, retired)
                       (_lib.StreamsOpts(_lib.ABI_VERSION, 0, 64), _lib.ERR_INVALID)):      # (an unknown flag bit)
        assert ctx.lib.conan_streams_create_opts(ctx.h, 2, 4, 16, C.byref(opts), C.byref(h)) == code
    with pytest.raises(ValueError):
        ctx.streams(2, arith="bf16")
    for s in (a, f, l):
        s.close()
    ctx.close()
    rb2 = Context(None, configs.HIFIGAN_ZERO_RB2_TINY, 0, False, False, True)
    rb2.load_state_dict("hifigan", synth.hifigan_state_dict(configs.HIFIGAN_ZERO_RB2_TINY, 0))
    rb2.finalize()
    s2 = rb2.streams(2, max_frames=8, max_ref_frames=16)
    # (the zero-insertion upsamplers' convs have limb copies, so AUTO may still resolve to limb; ResBlock2 itself has no limb pass)
    assert s2.arith in ("f32", "limb")
    s2.close(); rb2.close()
    dec = Context(chp, None, 0, emformer=False, conan=True, hifigan=False)
    dec.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    dec.finalize()
    d = dec.streams(2, 4, 16)
    assert d.arith == "f32"
    with pytest.raises(_lib.ConanError) as ei:
        dec.streams(2, 4, 16, arith="limb")
    assert ei.value.code == _lib.ERR_UNSUPPORTED
    d.close(); dec.close()


def test_fused_conv_block_operators_equal_the_separate_launches():
    """CONAN_STREAMS_FUSED_DECODER_BLOCKS (off by default because it costs the pipelined step 1.4 % - DESIGN.md): the decoder's
    eight sub-layers [LN -> k5 conv -> GELU] -> [1x1 conv + residual, masks] as ONE megakernel operator each (the 16 x 512 hidden
    tile stays in LDS, the 1x1 conv's K loop is split over the group, the consumer forms x' from 8 partial tensors) against the same
    step as ~38 separate launches: 12 steps of 4 frames at 24 streams, mel within fp32 re-association; the fused program twice gives
    the same bits."""
    from conan_amd.runtime import Context
    chp = configs.conan_hparams()
    ctx = Context(chp, None, 0, emformer=False, conan=True, hifigan=False)
    ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    ctx.finalize()
    S = 24
    a, a2 = ctx.streams(S, 4, 64, flags=_lib.STREAMS_FUSED_DECODER_BLOCKS), ctx.streams(S, 4, 64, flags=_lib.STREAMS_FUSED_DECODER_BLOCKS)
    b = ctx.streams(S, 4, 64, dev_plan="DEC_MEGA=0")
    ids = list(range(S))
    ref = torch.from_numpy(synth.mel(40, 8, S)).cuda()
    codes = torch.from_numpy(synth.codes(48, S)).int().cuda()
    for st in (a, a2, b):
        st.reset(ids); st.set_reference(ids, ref)
    for i in range(0, 48, 4):
        c = codes[:, i:i + 4].contiguous()
        ma, ma2, mb = a.decoder_step(ids, c), a2.decoder_step(ids, c), b.decoder_step(ids, c)
        assert torch.equal(ma, ma2)
        np.testing.assert_allclose(ma.cpu().numpy(), mb.cpu().numpy(), atol=2e-5, rtol=1e-5)
    names = kernels_of(a, lambda: a.decoder_step(ids, codes[:, :4].contiguous()))
    assert any("decoder_mega_kernel" in k for k in names)
    for st in (a, a2, b):
        st.close()
    ctx.close()


def test_grouped_gather_ring_with_event_fences_on_the_gpu():
    """The multi-rank audio hand-off on ONE rank (always=True: side stream, join, event per group of four steps, and the output
    fence as the event recorded behind the gather that last read the group - conan_streams_output_fence_event): 19 pipelined steps
    through engine.AudioGatherRing(every=4) must hand on_gathered every step's audio, in order, bit-identical to the blocking
    loop - buffers are reused every 8 steps, so a fence that let the vocoder overwrite a buffer under its gather would show."""
    from conan_amd.engine import AudioGatherRing
    ctx, chp, vhp = _ctx(emformer=True)
    S, N, hop = 8, 19, ctx.hop
    a, b = ctx.streams(S, 4, 64), ctx.streams(S, 4, 64)
    ids = list(range(S))
    ref = torch.from_numpy(synth.mel(40, 8, S)).cuda()
    src = torch.from_numpy(synth.mel(4 * N + 8, 9, S)).cuda()
    for st in (a, b):
        st.reset(ids); st.set_reference(ids, ref)
    got = {}
    ring = AudioGatherRing(lambda: torch.empty(S, 4 * hop, device="cuda"), 1, 0, always=True, every=4,
                           on_gathered=lambda j, bufs: got.__setitem__(j, bufs[0].clone()))
    assert ring.nb == 8
    fences = 0
    for j in range(N):
        buf, fence = ring.acquire(j, fence=True)
        fences += fence is not None
        assert fence is None or isinstance(fence, torch.cuda.Event)
        a.step_async(ids, src[:, 4 * j:4 * j + 6].contiguous(), buf, emit=4, out_fence=fence)
        ring.submit(j, join=a.join)
    ring.flush(N - 1); ring.drain()
    a.join(); torch.cuda.synchronize()
    assert fences == 3 and ring.submitted == 5 and sorted(got) == list(range(N))     # fences at steps 8, 12 and 16 (a group re-opens); gathers after 3, 7, 11, 15 and the flush
    for j in range(N):
        _, _, w = b.step(ids, src[:, 4 * j:4 * j + 6].contiguous())
        assert torch.equal(w, got[j]), j
    a.close(); b.close(); ctx.close()
