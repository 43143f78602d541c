"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs,
and against the golden vectors produced by the reference.  Tolerances (fp32):
integer intermediates exact (off decision margins), mel max-abs <= 1e-4, pre-tanh rel <= 1e-4,
wav max-abs <= 1e-4."""
import numpy as np
import pytest
import torch

from conan_amd import configs, synth
from tests.conftest import assert_arith_ran, kernels_of, load_golden

pytestmark = pytest.mark.gpu


def _ctx(tiny, emformer=True, conan=True, hifigan=True):
    from conan_amd.runtime import Context
    chp, vhp = configs.conan_hparams(tiny), configs.hifigan_hparams(tiny)
    ctx = Context(chp if (emformer or conan) else None, vhp if hifigan else None, 0, emformer, conan, hifigan)
    if emformer:
        ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
    if conan:
        ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    if hifigan:
        ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    return ctx, chp, vhp


def _streams(ctx, *a, **k):
    """A stream-set in the arithmetic the module fixture is parametrized with (conan_streams_opts.arith)."""
    return ctx.streams(*a, arith=ctx.test_arith, **k)


@pytest.fixture(scope="module", params=["tiny-f32", "full-f32", "full-limb"])
def env(request):
    size, arith = request.param.split("-")
    tiny = size == "tiny"
    ctx, chp, vhp = _ctx(tiny)
    ctx.test_arith = arith
    from oracle.common import to_torch_sd
    sds = {"emformer": to_torch_sd(synth.emformer_state_dict(chp, 0)),
           "conan": to_torch_sd(synth.conan_state_dict(chp, 0)),
           "hifigan": to_torch_sd(synth.hifigan_state_dict(vhp, 0))}
    yield size, ctx, chp, vhp, sds
    ctx.close()


def test_hifigan_stream_vs_oracle_and_golden(env):
    from oracle import hifigan as ohifi
    tag, ctx, chp, vhp, sds = env
    g = load_golden(f"hifigan_{tag}.npz")
    mel0 = torch.from_numpy(g["mel_12"])[0].T                      # [12,80]
    mel1 = torch.from_numpy(synth.mel(12, 77))[0]
    mels = torch.stack([mel0, mel1]).cuda()                        # [2,12,80]
    st = _streams(ctx, 4, max_frames=4, max_ref_frames=16)
    slots = [2, 0]
    st.reset(slots)
    wavs, pres = [], []
    for i in range(0, 12, 4):
        w, p = st.hifigan_step(slots, mels[:, i:i + 4], want_pre_tanh=True)
        wavs.append(w)
        pres.append(p)
    wav = torch.cat(wavs, 1).cpu()
    pre = torch.cat(pres, 1).cpu()
    taps = {}
    ref = ohifi.generator_forward(sds["hifigan"], vhp, mels.cpu().transpose(1, 2), None, taps)[:, 0]
    refpre = taps["pre_tanh"][:, 0]
    scale = float(refpre.abs().max())
    assert float((pre - refpre).abs().max()) <= 1e-4 * max(1.0, scale)
    np.testing.assert_allclose(wav.numpy(), ref.numpy(), atol=1e-4, rtol=0)
    np.testing.assert_allclose(wav[0].numpy(), g["wav_12"], atol=1e-4, rtol=0)     # golden from the reference
    # prefix consistency / causality (hifigan_causal.py:550-680): a different future does not change the past
    st.reset(slots)
    w_a = st.hifigan_step(slots, mels[:, 0:4]).cpu()
    st.reset(slots)
    w_b = st.hifigan_step(slots, mels[:, 0:4]).cpu()
    assert torch.equal(w_a, w_b)                                    # bitwise reproducible (no atomics)
    np.testing.assert_allclose(w_a.numpy(), ref[:, :4 * 320].numpy(), atol=1e-4, rtol=0)
    assert st.arith == ctx.test_arith
    if tag == "full":       # (4 slots: the C = 32 stage is the one whose fused pass has a limb form at this size)
        assert_arith_ran(kernels_of(st, lambda: st.hifigan_step(slots, mels[:, 4:8])), ctx.test_arith)
    st.close()


def test_hifigan_long_steps_and_ragged_frames(env):
    """one 12-frame step == three 4-frame steps == uneven 5+1+6 steps (ring wrap + state carry)."""
    from oracle import hifigan as ohifi
    tag, ctx, chp, vhp, sds = env
    mels = torch.from_numpy(synth.mel(24, 5, 1)).cuda()
    ref = ohifi.generator_forward(sds["hifigan"], vhp, mels.cpu().transpose(1, 2))[:, 0]
    st = _streams(ctx, 1, max_frames=12, max_ref_frames=16)
    for plan in ([12, 12], [4] * 6, [5, 1, 6, 3, 9]):
        st.reset([0])
        out, p = [], 0
        for f in plan:
            out.append(st.hifigan_step([0], mels[:, p:p + f]))
            p += f
        np.testing.assert_allclose(torch.cat(out, 1).cpu().numpy(), ref.numpy(), atol=1e-4, rtol=0)
    st.close()


def _emformer_stream_vs_oracle(env, dev_plan=None):
    from oracle import emformer as oemf
    tag, ctx, chp, vhp, sds = env
    cfg = oemf.EmformerCfg(chp)
    B, T = 3, 72
    mel = torch.from_numpy(synth.mel(T, 1234, B))
    st = _streams(ctx, 4, max_frames=4, max_ref_frames=16, dev_plan=dev_plan)
    slots = [3, 1, 0]
    st.reset(slots)
    state = None
    for pos, emit, chunk in oemf.chunk_iter(mel, cfg.segment_length, cfg.right_context_length):
        o_ref, _, state = oemf.emformer_infer(sds["emformer"], cfg, chunk, torch.full((B,), chunk.shape[1]), state)
        lg_ref, codes_ref = oemf.logits_and_codes(sds["emformer"], o_ref)
        o, lg, codes = st.emformer_step(slots, chunk.cuda())
        np.testing.assert_allclose(o.cpu().numpy(), o_ref.numpy(), atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(lg.cpu().numpy(), lg_ref.numpy(), atol=2e-4, rtol=1e-4)
        top2 = lg_ref.topk(2, -1).values
        safe = (top2[..., 0] - top2[..., 1]) > 1e-3
        assert torch.equal(codes.cpu().long()[safe], codes_ref[safe])
    st.close()


def test_emformer_stream_vs_oracle(env):
    """The fused whole-step kernel (emformer_fused.hip): 3 streams on 2 blocks, 18 chunks (the K/V rings wrap)."""
    _emformer_stream_vs_oracle(env)


def test_emformer_per_op_path_vs_oracle(env):
    """The per-op launch plan that covers shapes the fused kernel does not (conan_streams_opts.dev_plan "EMF_UNFUSED=1" selects it)."""
    _emformer_stream_vs_oracle(env, dev_plan="EMF_UNFUSED=1")


def test_conan_decoder_vs_oracle_and_golden(env):
    from oracle import conan as oconan
    tag, ctx, chp, vhp, sds = env
    g = load_golden(f"conan_{tag}.npz")
    content = torch.from_numpy(g["content"])                      # [1,150]
    ref = torch.from_numpy(g["ref"])                              # [1,150,80]
    # second stream: different codes and a shorter reference (ragged reference lengths)
    content2 = torch.from_numpy(synth.codes(150, 1, seed=3))
    ref2 = torch.from_numpy(synth.mel(150, 999))
    ref2[:, 131:] = 0
    cache1 = oconan.style_pass(sds["conan"], chp, ref)
    cache2 = oconan.style_pass(sds["conan"], chp, ref2[:, :131])
    o1 = oconan.decode_frames(sds["conan"], chp, content, cache1)
    o2 = oconan.decode_frames(sds["conan"], chp, content2, cache2)
    st = _streams(ctx, 4, max_frames=4, max_ref_frames=160)
    slots = [1, 3]
    st.reset(slots)
    st.set_reference(slots, torch.cat([ref, ref2]).cuda(), [150, 131])
    codes = torch.cat([content, content2]).int().cuda()
    mels, bins, uvs, dinps = [], [], [], []
    for i in range(0, 48, 4):
        m, taps = st.decoder_step(slots, codes[:, i:i + 4], taps=True)
        mels.append(m.cpu()); bins.append(taps["pitch_bins"].cpu()); uvs.append(taps["uv_pred"].cpu()); dinps.append(taps["decoder_inp"].cpu())
    mel = torch.cat(mels, 1); bins = torch.cat(bins, 1).long(); uv = torch.cat(uvs, 1); dinp = torch.cat(dinps, 1)
    for k, o in enumerate((o1, o2)):
        np.testing.assert_allclose(uv[k].numpy(), o["uv_pred"][0, :48].numpy(), atol=2e-4, rtol=1e-4)
        # pitch bins: exact away from rounding/decision margins
        ob = o["pitch_bins"][0, :48]
        f0 = o["f0_denorm_pred"][0, :48]
        fm = 1127 * (1 + f0 / 700).log()
        fm = torch.where(fm > 0, (fm - 1127 * np.log(1 + 50 / 700)) * 254 / (1127 * np.log(1 + 900 / 700) - 1127 * np.log(1 + 50 / 700)) + 1, fm)
        safe = ((fm + 0.5) - (fm + 0.5).floor() - 0.5).abs() < 0.49
        safe &= o["uv_pred"][0, :48, 0].abs() > 1e-3
        assert torch.equal(bins[k][safe], ob[safe])
        assert bool((bins[k] == ob).all()), "pitch bin flipped on a decision margin: compare downstream with care"
        np.testing.assert_allclose(dinp[k].numpy(), o["decoder_inp"][0, :48].numpy(), atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(mel[k].numpy(), o["mel_out"][0, :48].numpy(), atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(mel[0].numpy(), g["mel_out"][0, :48], atol=1e-4, rtol=1e-4)      # golden from the reference
    st.close()


def test_fused_step_vs_oracle_loop(env):
    from oracle import emformer as oemf
    from oracle import loop as oloop
    tag, ctx, chp, vhp, sds = env
    cfg = oemf.EmformerCfg(chp)
    T, Tr, B = 26, 40, 2            # ragged tail: 26 = 6*4 + 2
    src = synth.mel(T, 1234, B)
    ref = synth.mel(Tr, 4321, B)
    st = _streams(ctx, B, max_frames=4, max_ref_frames=64)
    slots = list(range(B))
    st.reset(slots)
    st.set_reference(slots, torch.from_numpy(ref).cuda())
    wavs, mels, codes = [], [], []
    for pos, emit, chunk in oemf.chunk_iter(torch.from_numpy(src), cfg.segment_length, cfg.right_context_length):
        c, m, w = st.step(slots, chunk.cuda().contiguous(), emit=emit)
        wavs.append(w.cpu()); mels.append(m.cpu()); codes.append(c.cpu()[:, :emit])
    wav, mel, code = torch.cat(wavs, 1), torch.cat(mels, 1), torch.cat(codes, 1)
    for b in range(B):
        w_ref, m_ref, c_ref = oloop.infer_once_stateful(sds["emformer"], cfg, sds["conan"], chp, sds["hifigan"], vhp, src[b], ref[b])
        assert np.array_equal(code[b].numpy(), c_ref), "code flip (argmax margin); see test_emformer_stream_vs_oracle"
        np.testing.assert_allclose(mel[b].numpy(), m_ref, atol=1e-4, rtol=1e-4)
        np.testing.assert_allclose(wav[b].numpy(), w_ref, atol=1e-4, rtol=0)
    if tag == "full":
        chunk0 = torch.from_numpy(src[:, :cfg.segment_length + cfg.right_context_length]).cuda().contiguous()
        assert_arith_ran(kernels_of(st, lambda: st.step(slots, chunk0)), ctx.test_arith)
    st.close()


@pytest.mark.parametrize("upsample", ["shuffle", "zero", "nn"])
def test_reference_invariants_causality_and_prefix_consistency(upsample):
    """The reference's own known-answer tests for this path (hifigan_causal.py:550-598 verify_causality, :602-672
    verify_prefix_consistency, both at atol 1e-6) run on the HIP generator through the module seam: perturbing mel frames
    after t must leave the first (t+1)*320 samples unchanged, and an 8-frame input must reproduce the first 8*320 samples of
    a 16-frame input with the same prefix.  Rows of a tile are independent dot products, so causality holds BITWISE for the
    causal upsamplers (same launch shapes); the 16-frame forward picks other tile shapes / split-K factors than the 8-frame
    one, so prefix consistency is checked at the reference's 1e-6.  `upsample: nn` fails both - in the reference as well
    (see DESIGN.md f3): two frames of look-ahead."""
    from conan_amd import configs as cfgs
    from conan_amd.modules.vocoder.hifigan.hifigan_causal import HifiGanGenerator
    vhp = dict(cfgs.hifigan_hparams(True), upsample=upsample)
    gen = HifiGanGenerator(vhp)
    gen.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hifigan_state_dict(vhp, 0).items()}, strict=True)
    g = torch.Generator().manual_seed(5)
    T, hop = 8, 320
    x = torch.randn(1, 80, T, generator=g).cuda()
    y0 = gen(x)
    assert y0.shape == (1, 1, T * hop)
    causal = True
    for t in range(T - 1):
        xp = x.clone()
        xp[:, :, t + 1:] += 1e-3 * torch.randn(1, 80, T - t - 1, generator=g).cuda()
        yp = gen(xp)
        n = (t + 1) * hop
        if upsample == "nn":
            causal = causal and bool(torch.allclose(yp[:, :, :n], y0[:, :, :n], atol=1e-6))
        else:
            assert torch.equal(yp[:, :, :n], y0[:, :, :n]), t
            assert not torch.equal(yp[:, :, n:], y0[:, :, n:])
    long = torch.cat([x, torch.randn(1, 80, 8, generator=g).cuda()], 2)
    yl = gen(long)
    assert yl.shape == (1, 1, 16 * hop)
    if upsample == "nn":
        assert not causal and not torch.allclose(yl[:, :, :T * hop], y0, atol=1e-6)
        assert torch.allclose(yl[:, :, :T * hop - 740], y0[:, :, :T * hop - 740], atol=1e-6)      # all but the look-ahead
    else:
        assert torch.allclose(yl[:, :, :T * hop], y0, atol=1e-6)
