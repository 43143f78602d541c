"""Round-3 GPU tests: single-branch vocoders, host plans that take their widths from the checkpoint, the fused first
vocoder stage, fused decoder sub-layers, the memory bank inside the one-launch Emformer step, fenced hand-offs."""
import os

import numpy as np
import pytest
import torch

from conan_amd import configs, synth

pytestmark = pytest.mark.gpu


def _voc_ctx(vhp):
    from conan_amd.runtime import Context
    ctx = Context(None, vhp, 0, False, False, True)
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    return ctx


@pytest.mark.parametrize("tiny,slots", [(True, 1), (False, 1), (False, 9)])
def test_single_branch_vocoder_keeps_the_last_leaky_relu(tiny, slots):
    """resblock_kernel_sizes = [3]: HifiGanGenerator.forward still applies F.leaky_relu before conv_post
    (hifigan_causal.py:324-333) although the branch mean is over one branch.  Streamed in 4-frame steps against the
    oracle's one-shot forward; 9 slots take the fused ResBlock plan at every width, one slot the two-launch plan."""
    from oracle import hifigan as ohifi
    from oracle.common import to_torch_sd
    vhp = dict(configs.hifigan_hparams(tiny), resblock_kernel_sizes=[3], resblock_dilation_sizes=[[1, 3, 5]])
    ctx = _voc_ctx(vhp)
    T = 12
    mel = torch.from_numpy(synth.mel(T, 31, slots)).transpose(1, 2).contiguous()      # [slots, 80, T]
    taps = {}
    ref = ohifi.generator_forward(to_torch_sd(synth.hifigan_state_dict(vhp, 0)), vhp, mel, None, taps)[:, 0].numpy()
    st = ctx.streams(slots, max_frames=4, max_ref_frames=16)
    ids = list(range(slots))
    st.reset(ids)
    x = mel.transpose(1, 2).contiguous().cuda()
    outs, pres = [], []
    for i in range(0, T, 4):
        w, p = st.hifigan_step(ids, x[:, i:i + 4], want_pre_tanh=True)
        outs.append(w); pres.append(p)
    wav, pre = torch.cat(outs, 1).cpu().numpy(), torch.cat(pres, 1).cpu().numpy()
    ref_pre = taps["pre_tanh"][:, 0].numpy()
    np.testing.assert_allclose(pre, ref_pre, atol=1e-4 * max(1.0, np.abs(ref_pre).max()), rtol=0)
    np.testing.assert_allclose(wav, ref, atol=1e-4, rtol=0)
    st.close(); ctx.close()


def test_fence_free_handoffs_match_the_fenced_build_under_concurrency():
    """Split-K partial tiles (conv_mfma) and the Emformer cluster exchange travel as write-through stores + a flag / ticket +
    sc1 loads, without release / acquire fences.  Cross-check: a second stream-set created with dev_plan "FENCED=1" brackets the
    same hand-offs with agent-scope fences; at 1-4 streams all three pipelined stages split K / run clusters concurrently on
    their internal streams.  Every output of 120 pipelined steps must be bit-identical between the two (the split factors
    and cluster sizes are the same, so any difference is a stale read)."""
    from conan_amd.runtime import Context
    chp, vhp = configs.conan_hparams(), configs.hifigan_hparams()
    ctx = Context(chp, vhp, 0)
    ctx.load_state_dict("emformer", synth.emformer_state_dict(chp, 0))
    ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    ctx.load_state_dict("hifigan", synth.hifigan_state_dict(vhp, 0))
    ctx.finalize()
    S, N = 4, 120
    plain = ctx.streams(S, 4, 64)
    fenced = ctx.streams(S, 4, 64, dev_plan="FENCED=1")
    ref = torch.from_numpy(synth.mel(40, 8, S)).cuda()
    src = torch.from_numpy(synth.mel(4 * N + 8, 9, S)).cuda()
    hop = ctx.hop
    rng = np.random.default_rng(3)
    outs = {}
    for name, st in (("plain", plain), ("fenced", fenced)):
        st.reset(list(range(S))); st.set_reference(list(range(S)), ref)
    pos = [0] * S
    res = {"plain": [], "fenced": []}
    for it in range(N):
        n = int(rng.integers(1, S + 1))
        slots = sorted(rng.choice(S, n, replace=False).tolist())
        chunk = torch.stack([src[s, pos[s]:pos[s] + 6] for s in slots]).contiguous()
        for name, st in (("plain", plain), ("fenced", fenced)):
            c = torch.empty(n, 4, dtype=torch.int32, device="cuda"); m = torch.empty(n, 4, 80, device="cuda"); w = torch.empty(n, 4 * hop, device="cuda")
            st.step_async(slots, chunk, w, emit=4, codes=c, mel_out=m)
            res[name].append((c, m, w))
        for s in slots:
            pos[s] += 4
    plain.join(); fenced.join(); torch.cuda.synchronize()
    bad = [k for k, (a, b) in enumerate(zip(res["plain"], res["fenced"])) if not all(torch.equal(x, y) for x, y in zip(a, b))]
    assert not bad, f"steps whose fence-free result differs from the fenced one: {bad[:10]}"
    assert all(torch.isfinite(w).all() for _, _, w in res["plain"])
    plain.close(); fenced.close(); ctx.close()


@pytest.mark.parametrize("arith,S", [("f32", 24), ("limb", 24), ("limb", 48)])
def test_wide_stage_pairs_match_reference_goldens(arith, S):
    """The first vocoder stage (C = 256) on exact-f32 stream-sets of >= 16 slots runs resblock_pair.hip: a pair of workgroups per
    (branch, stream) tile, xt history ring instead of a halo; bf16-limb stream-sets of >= 16 slots run its convs as conv_limb's
    grouped launches.  Reference goldens (tools/make_goldens.py: wav_150, wav_12,
    pre_tanh_12) streamed through slot 11 of a 24-slot stream-set while the other slots carry other streams, in steps
    of 4 frames with a ragged tail (150 = 37 x 4 + 2; then 12 frames as 3 + 1 + 4 + 2 + 2: 24, 8, 32, 16 and 16 rows in
    the wide stage)."""
    from tests.conftest import load_golden
    vhp = configs.hifigan_hparams()
    ctx = _voc_ctx(vhp)
    g = load_golden("hifigan_full.npz")
    st = ctx.streams(S, max_frames=4, max_ref_frames=16, arith=arith)
    ids = list(range(S))
    st.reset(ids)
    mel = torch.from_numpy(synth.mel(150, 77, S)).cuda()                               # other streams
    mel[11] = torch.from_numpy(g["mel_150"][0].T).cuda()
    outs = []
    for i in range(0, 150, 4):
        outs.append(st.hifigan_step(ids, mel[:, i:i + 4].contiguous())[11])
    wav = torch.cat(outs).cpu().numpy()
    np.testing.assert_allclose(wav, g["wav_150"], atol=1e-4, rtol=0)
    # ragged steps on a subset of the slots, pre-tanh against the golden too
    sub = [3, 11, 20]
    st.reset(sub)
    m12 = torch.from_numpy(g["mel_12"]).transpose(1, 2).contiguous().cuda().expand(3, -1, -1).contiguous()
    pres, wavs, p = [], [], 0
    for n in (3, 1, 4, 2, 2):
        w, pr = st.hifigan_step(sub, m12[:, p:p + n].contiguous(), want_pre_tanh=True)
        wavs.append(w); pres.append(pr); p += n
    wav12, pre12 = torch.cat(wavs, 1).cpu().numpy(), torch.cat(pres, 1).cpu().numpy()
    for r in range(3):
        np.testing.assert_allclose(wav12[r], g["wav_12"], atol=1e-4, rtol=0)
        np.testing.assert_allclose(pre12[r], g["pre_tanh_12"][0], atol=1e-4 * max(1.0, np.abs(g["pre_tanh_12"]).max()), rtol=0)
    # which kernels carried the wide first stage: the f32 pair kernel (every f32 stream-set of >= 16 slots), conv_limb's grouped
    # launches (limb stream-sets of >= 16 slots: 3 dilations x (c1, c2); from 16 slots on ups.2 + ups.3 as well: a tile for every second CU)
    from tests.conftest import assert_arith_ran, kernels_of
    names = kernels_of(st, lambda: st.hifigan_step(ids, mel[:, :4].contiguous()))
    assert_arith_ran(names, arith)
    if arith == "limb":
        assert sum(n for k, n in names.items() if "conv_limb_kernel" in k) >= 6 and not any("resblock_pair_kernel" in k for k in names), sorted(names)
    else:
        assert any("resblock_pair_kernel" in k for k in names), sorted(names)
    st.close(); ctx.close()


def test_wide_stage_pairs_equal_the_two_launch_plan_per_stage():
    """Same streams through a 24-slot stream-set (pair kernel in the first stage) and through a 24-slot stream-set created
    with dev_plan "RB_NOPAIR=1" (conv_mfma two-launch plan there): per-stage tensors and audio agree to fp32 re-association."""
    vhp = configs.hifigan_hparams()
    ctx = _voc_ctx(vhp)
    S = 24
    a = ctx.streams(S, max_frames=4, max_ref_frames=16, arith="f32")       # (the pair kernel is the exact-f32 form of this stage)
    b = ctx.streams(S, max_frames=4, max_ref_frames=16, arith="f32", dev_plan="RB_NOPAIR=1")
    ids = list(range(S))
    mel = torch.from_numpy(synth.mel(24, 5, S)).cuda()
    for st in (a, b):
        st.reset(ids)
    for i in range(0, 24, 4):
        wa, pa, ca, ua = a.hifigan_step_taps(ids, mel[:, i:i + 4].contiguous())
        wb, pb, cb, ub = b.hifigan_step_taps(ids, mel[:, i:i + 4].contiguous())
        for x, y in zip(ua, ub):
            s = max(1.0, float(y.abs().max()))
            assert float((x - y).abs().max()) <= 2e-5 * s
        assert float((pa - pb).abs().max()) <= 2e-5 * max(1.0, float(pb.abs().max()))
        assert float((wa - wb).abs().max()) <= 2e-5
    a.close(); b.close(); ctx.close()


def test_decoder_megakernel_equals_the_separate_launches():
    """The decoder step as one persistent launch (decoder_mega.hip: row-tile groups, group barriers, agent-scope activation
    accesses, fused feed-forward) against the same step as ~38 separate launches (dev_plan "DEC_MEGA=0") on the same streams:
    20 steps of 4 frames at 24 streams (6 row tiles), then the ragged tail (3 frames: separate launches in both).  mel within
    fp32 re-association (the fused feed-forward sums its hidden units in a different order), and the megakernel twice gives
    the same bits (no stale reads between its operators)."""
    from conan_amd.runtime import Context
    chp, vhp = configs.conan_hparams(), configs.hifigan_hparams()
    ctx = Context(chp, None, 0, emformer=False, conan=True, hifigan=False)
    ctx.load_state_dict("conan", synth.conan_state_dict(chp, 0))
    ctx.finalize()
    S = 24
    a = ctx.streams(S, 4, 64); a2 = ctx.streams(S, 4, 64)
    b = ctx.streams(S, 4, 64, dev_plan="DEC_MEGA=0")
    ids = list(range(S))
    ref = torch.from_numpy(synth.mel(40, 8, S)).cuda()
    codes = torch.from_numpy(synth.codes(83, S)).int().cuda()
    for st in (a, a2, b):
        st.reset(ids); st.set_reference(ids, ref)
    for i in list(range(0, 80, 4)) + [80]:
        c = codes[:, i:i + 4 if i < 80 else 83].contiguous()
        ma, ma2, mb = a.decoder_step(ids, c), a2.decoder_step(ids, c), b.decoder_step(ids, c)
        assert torch.equal(ma, ma2)
        np.testing.assert_allclose(ma.cpu().numpy(), mb.cpu().numpy(), atol=2e-5, rtol=1e-5)
    a.close(); a2.close(); b.close(); ctx.close()


@pytest.mark.parametrize("streams", [1, 4, 12])
def test_decoder_widths_come_from_the_checkpoint(streams):
    """A checkpoint with a 64-channel uv predictor and a 512-wide aligner feed-forward (the reference hard-codes 128 / 2048 in
    its constructors; hidden_size 128 here): the host plans size their buffers and kernels from the loaded tensors.  Style
    pass + stateful decoder steps against the oracle; integer intermediates exact; 12 streams take the megakernel."""
    from conan_amd.runtime import Context
    from oracle import conan as oconan
    from oracle.common import to_torch_sd
    chp = dict(configs.conan_hparams(), hidden_size=128, uv_predictor_hidden=64, align_ffn_dim=512)
    sd_np = synth.conan_state_dict(chp, 0)
    assert sd_np["uv_predictor.conv.4.0.conv.weight"].shape[0] == 64 and sd_np["align.layers.0.linear1.weight"].shape == (512, 128)
    ctx = Context(chp, None, 0, emformer=False, conan=True, hifigan=False)
    ctx.load_state_dict("conan", sd_np)
    ctx.finalize()
    sd = to_torch_sd(sd_np)
    T, Tr = 16, 40
    ref = torch.from_numpy(synth.mel(Tr, 21, streams))
    codes = torch.from_numpy(synth.codes(T, streams))
    st = ctx.streams(streams, 4, 64)
    ids = list(range(streams))
    st.reset(ids); st.set_reference(ids, ref.cuda())
    mels, bins, uvs = [], [], []
    for p in range(0, T, 4):
        m, tp = st.decoder_step(ids, codes[:, p:p + 4].int().cuda(), taps=True)
        bins.append(tp["pitch_bins"].cpu()); uvs.append(tp["uv_pred"].cpu())
    st.reset(ids, which=2)
    for p in range(0, T, 4):                       # and without taps (12 streams: the megakernel)
        mels.append(st.decoder_step(ids, codes[:, p:p + 4].int().cuda()).cpu())
    mel, bins, uvs = torch.cat(mels, 1).numpy(), torch.cat(bins, 1).numpy(), torch.cat(uvs, 1).numpy()
    for b in range(streams):
        cache = oconan.style_pass(sd, chp, ref[b:b + 1])
        want = oconan.decode_frames(sd, chp, codes[b:b + 1].long(), cache, {})
        np.testing.assert_allclose(uvs[b], want["uv_pred"][0].numpy(), atol=2e-4, rtol=1e-4)
        safe = np.abs(want["uv_pred"][0, :, 0].numpy()) > 1e-3
        assert np.array_equal(bins[b][safe], want["pitch_bins"][0].numpy()[safe]) if "pitch_bins" in want else True
        np.testing.assert_allclose(mel[b], want["mel_out"][0].numpy(), atol=1e-4, rtol=1e-4)
    st.close(); ctx.close()
    # a third aligner layer is refused when the checkpoint is packed, not in a step
    from conan_amd import _lib
    bad = dict(sd_np)
    for k, v in sd_np.items():
        if k.startswith("align.layers.1."):
            bad[k.replace("layers.1.", "layers.2.")] = v
    ctx2 = Context(chp, None, 0, emformer=False, conan=True, hifigan=False)
    ctx2.load_state_dict("conan", bad)
    with pytest.raises(_lib.ConanError) as ei:
        ctx2.finalize()
    assert ei.value.code == _lib.ERR_SHAPE
    ctx2.close()


def _two_stream_sets(ctx, S, *envs, arith="limb"):
    """Stream-sets created with the developer switches of each env (conan_streams_opts.dev_plan; {"CONAN_X": "1"} -> "X=1")."""
    return [ctx.streams(S, max_frames=4, max_ref_frames=16, arith=arith,
                        dev_plan=";".join(f"{k[len('CONAN_'):]}={v}" for k, v in env.items()) or None) for env in envs]


def _kernel_names(st, ids, mel):
    st.profile_begin()
    st.hifigan_step(ids, mel)
    st.profile_end()
    return {r[0]: r[3] for r in st.profile_kernels()}       # kernel name -> launches


@pytest.mark.parametrize("env,other", [({}, "f32"), ({"CONAN_RB_PAIR": "1"}, "pair")])
def test_bf16_limb_kernels_equal_the_f32_mfma_kernels(env, other):
    """resblock_limb.hip / conv_limb.hip form every fp32 product from three bf16 limbs per operand (six bf16 MFMA products,
    accumulated in fp32); conan_streams_opts.arith selects the form per stream-set.
    64 streams through a limb stream-set - limb kernels in the C = 128 / 64 / 32 ResBlock stages, in ups.2 / ups.3 and, for
    stream-sets of >= 16 slots, conv_limb's grouped launches (three problems of 3 / 7 / 11 taps per launch, list-scheduled tiles)
    for the ResBlock convs of the C = 256 stage - and through a default one (exact-f32 MFMA, pair kernel in the first stage) or
    a limb one created with the developer switch CONAN_RB_PAIR=1 (limb kernels, but the f32 pair kernel in the first stage): per-stage tensors,
    pre-tanh and audio agree to fp32 re-association - the same bound the pair / two-launch cross-check uses."""
    vhp = configs.hifigan_hparams()
    ctx = _voc_ctx(vhp)
    S = 64
    a = ctx.streams(S, max_frames=4, max_ref_frames=16, arith="limb")
    if other == "pair":
        (b,) = _two_stream_sets(ctx, S, env)[:1]
    else:
        b = ctx.streams(S, max_frames=4, max_ref_frames=16, arith="f32")
    ids = list(range(S))
    mel = torch.from_numpy(synth.mel(16, 9, S)).cuda()
    for st in (a, b):
        st.reset(ids)
    for i in range(0, 16, 4):
        wa, pa, ca, ua = a.hifigan_step_taps(ids, mel[:, i:i + 4].contiguous())
        wb, pb, cb, ub = b.hifigan_step_taps(ids, mel[:, i:i + 4].contiguous())
        for x, y in zip(ua, ub):
            s = max(1.0, float(y.abs().max()))
            assert float((x - y).abs().max()) <= 2e-5 * s
        assert float((pa - pb).abs().max()) <= 2e-5 * max(1.0, float(pb.abs().max()))
        assert float((wa - wb).abs().max()) <= 2e-5
    # the kernels that ran (a silent fall-back to the f32 kernels would pass the comparison above)
    na, nb = _kernel_names(a, ids, mel[:, :4].contiguous()), _kernel_names(b, ids, mel[:, :4].contiguous())
    assert any("resblock_limb_kernel<128" in k for k in na) and any("resblock_limb_kernel<64" in k for k in na) and any("resblock_limb_kernel<32" in k for k in na)
    # ups.2 / ups.3 + 3 dilations x (c1, c2) grouped launches on conv_limb, no pair kernel
    assert sum(n for k, n in na.items() if "conv_limb_kernel" in k) == 8 and not any("resblock_pair_kernel" in k for k in na)
    assert na.get("cnk::conv_tall_kernel") == 2       # ups.0 and ups.1 (round 6: the split-K limb GEMM of conv_tall.hip)
    if other == "pair":
        assert sum(n for k, n in nb.items() if "conv_limb_kernel" in k) == 2 and nb.get("cnk::conv_tall_kernel") == 2 and any("resblock_pair_kernel" in k for k in nb)
        assert any("resblock_limb_kernel<128" in k for k in nb)
    else:
        assert not any("limb" in k for k in nb) and any("resblock_pair_kernel" in k for k in nb)
    a.close(); b.close(); ctx.close()
