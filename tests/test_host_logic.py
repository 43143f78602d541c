"""CPU tests of the host side: C-ABI library loads and exports every declared symbol (no compute without a GPU),
config/ckpt compatibility helpers, state_dict contracts, error conventions, stream sharding."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import yaml

from conan_amd import _lib, configs, specs, synth


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = _lib.declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib._PROTOS, f"{n} declared in include/conan_hip.h but not bound in _lib.py"
    assert lib.conan_abi_version() == _lib.ABI_VERSION


def test_cfg_struct_matches_header_and_hparams():
    cfg = _lib.make_cfg(configs.conan_hparams(), configs.hifigan_hparams())
    assert C.sizeof(cfg) == 4 * 81          # 81 int32 fields, include/conan_hip.h
    assert C.sizeof(_lib.MelCfg) == 4 * 13  # conan_mel_cfg
    assert cfg.voc_upsample == 0 and cfg.voc_resblock == 1
    z = _lib.make_cfg(None, configs.HIFIGAN_ZERO_RB2, emformer=False, conan=False)
    assert z.voc_upsample == 1 and z.voc_resblock == 2 and z.voc_rb_num_dil == 2
    assert cfg.models == 7 and cfg.hidden_size == 256 and cfg.emf_segment == 4 and cfg.emf_right_context == 2
    assert list(cfg.voc_up_rates)[:4] == [8, 5, 4, 2] and list(cfg.voc_up_kernels)[:4] == [16, 10, 8, 4]
    assert [list(r)[:3] for r in cfg.voc_rb_dilations][:3] == [[1, 3, 5]] * 3
    assert _lib.make_cfg(None, configs.HIFIGAN_NN, emformer=False, conan=False).voc_upsample == 2
    with pytest.raises(_lib.ConanError):
        _lib.make_cfg(None, dict(configs.hifigan_hparams(), upsample="linear"))
    with pytest.raises(_lib.ConanError):
        _lib.make_cfg(dict(configs.conan_hparams(), f0_gen="flow"), None)


def test_ctx_create_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.lib()
    cfg = _lib.make_cfg(configs.conan_hparams(True), configs.hifigan_hparams(True))
    h = C.c_void_p()
    rc = lib.conan_ctx_create(0, C.byref(cfg), C.byref(h))
    assert rc == _lib.ERR_HIP and b"device" in lib.conan_last_error().lower()
    from conan_amd.runtime import Context
    with pytest.raises(RuntimeError):
        Context(configs.conan_hparams(True), configs.hifigan_hparams(True))


def test_bad_cfg_rejected_before_touching_the_gpu():
    lib = _lib.lib()
    cfg = _lib.make_cfg(configs.conan_hparams(True), configs.hifigan_hparams(True))
    cfg.abi_version = 99
    h = C.c_void_p()
    assert lib.conan_ctx_create(0, C.byref(cfg), C.byref(h)) == _lib.ERR_INVALID


def test_state_dict_contracts():
    from conan_amd.modules.Conan.Conan import Conan
    from conan_amd.modules.Emformer.emformer import EmformerDistillModel
    from conan_amd.modules.vocoder.hifigan.hifigan_causal import HifiGanGenerator
    chp, vhp = configs.conan_hparams(), configs.hifigan_hparams()
    g = HifiGanGenerator(vhp)
    assert len(g.state_dict()) == 234 and sum(p.numel() for p in g.parameters()) == 29972450
    c = Conan(0, chp)
    assert len(c.state_dict()) == 271 and sum(p.numel() for p in c.parameters()) == 54922868
    assert "prosody_extractor.vqvae.embedding" in dict(c.named_buffers())
    e = EmformerDistillModel(chp, output_dim=100)
    keys = list(e.state_dict().keys())
    assert keys[0].startswith("emformer.emformer_layers.0.attention.") and "proj.weight" in keys
    assert abs(sum(p.numel() for p in e.parameters()) - 2.145e6) < 2e4
    # a reference-style checkpoint round-trips through load_state_dict unchanged
    sd = {k: torch.from_numpy(v) for k, v in synth.hifigan_state_dict(vhp, 3).items()}
    g.load_state_dict(sd, strict=True)
    assert torch.equal(g.state_dict()["ups.0.conv.conv.weight_v"], sd["ups.0.conv.conv.weight_v"])
    with pytest.raises(ValueError):
        c.forward(torch.zeros(1, 4, dtype=torch.long), ref=None, infer=True)      # Conan.py:152-155
    with pytest.raises(NotImplementedError):
        HifiGanGenerator(dict(vhp, upsample="linear"))
    # `upsample: nn` carries the reference's ConvTranspose1d tree: weight [Cin, Cout, k], weight_g per INPUT channel, `_cache` buffer
    gn = HifiGanGenerator(dict(vhp, upsample="nn"))
    nsd = gn.state_dict()
    C0 = vhp["upsample_initial_channel"]
    assert tuple(nsd["ups.0.deconv.weight_v"].shape) == (C0, C0 // 2, 16) and tuple(nsd["ups.0.deconv.weight_g"].shape) == (C0, 1, 1)
    assert tuple(nsd["ups.0.deconv.bias"].shape) == (C0 // 2,) and tuple(nsd["ups.0._cache"].shape) == (1, C0, 7)
    assert "ups.0._cache" in dict(gn.named_buffers()) and gn.one_shot


def test_hparams_yaml_chain(tmp_path, monkeypatch):
    from conan_amd.utils.commons import hparams as H
    (tmp_path / "egs" / "bases").mkdir(parents=True)
    (tmp_path / "egs" / "bases" / "root.yaml").write_text(yaml.safe_dump({"a": 1, "nested": {"x": 1, "y": 2}, "lst": [1, 2]}))
    (tmp_path / "egs" / "bases" / "mid.yaml").write_text(yaml.safe_dump({"base_config": "./root.yaml", "a": 2, "b": "s"}))
    (tmp_path / "egs" / "top.yaml").write_text(yaml.safe_dump({"base_config": ["egs/bases/mid.yaml"], "nested": {"y": 3}, "c": True}))
    monkeypatch.chdir(tmp_path)
    hp = H.set_hparams("egs/top.yaml", hparams_str="a=5,nested.x=7,lst=[3 4],c=False", print_hparams=False)
    assert hp["a"] == 5 and hp["b"] == "s" and hp["nested"] == {"x": 7, "y": 3} and hp["lst"] == [3, 4] and hp["c"] is False
    assert H.hparams["a"] == 5 and hp["work_dir"] == ""
    v = H.set_hparams("egs/bases/mid.yaml", global_hparams=False, print_hparams=False)
    assert v["a"] == 2 and H.hparams["a"] == 5          # vocoder config does not clobber the global dict


def test_ckpt_utils(tmp_path):
    from conan_amd.utils.commons import ckpt_utils as K
    from conan_amd.modules.vocoder.hifigan.hifigan_causal import HifiGanGenerator
    vhp = configs.hifigan_hparams(True)
    sd = {k: torch.from_numpy(v) for k, v in synth.hifigan_state_dict(vhp, 5).items()}
    torch.save({"state_dict": {"model_gen": sd}, "global_step": 10}, tmp_path / "model_ckpt_steps_10.ckpt")
    torch.save({"state_dict": {"model_gen": {k: v * 0 for k, v in sd.items()}}}, tmp_path / "model_ckpt_steps_2.ckpt")
    g = HifiGanGenerator(vhp)
    K.load_ckpt(g, str(tmp_path), "model_gen")                       # highest step wins
    assert torch.equal(g.state_dict()["conv_pre.conv.bias"], sd["conv_pre.conv.bias"])
    flat = {"state_dict": {f"model.{k}": v for k, v in sd.items()}}
    torch.save(flat, tmp_path / "flat.ckpt")
    g2 = HifiGanGenerator(vhp)
    K.load_ckpt(g2, str(tmp_path / "flat.ckpt"), "model")
    assert torch.equal(g2.state_dict()["conv_post.conv.weight_v"], sd["conv_post.conv.weight_v"])
    with pytest.raises(AssertionError):
        K.load_ckpt(g, str(tmp_path / "nothing_here"))


def test_synth_is_deterministic_and_weightnorm_positive():
    a = synth.hifigan_state_dict(configs.hifigan_hparams(True), 0)
    b = synth.hifigan_state_dict(configs.hifigan_hparams(True), 0)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert all((v > 0).all() for k, v in a.items() if k.endswith("weight_g"))
    assert synth.mel(151, 1234).shape == (1, 151, 80) and synth.mel(4, 1).min() >= -6 and synth.mel(4, 1).max() <= 1.5


def test_shard_range_partitions_streams():
    from conan_amd.engine import shard_range
    for total, world in ((512, 8), (10, 4), (3, 8)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_wav_io_roundtrip_and_pad(tmp_path):
    """utils/audio: save_wav (utils/audio/io.py:7-13) writes 16-bit PCM that load_wav reads back; librosa_pad_lr
    (utils/audio/__init__.py:13-22) pads to a whole number of hops plus one frame."""
    from conan_amd.utils.audio import librosa_pad_lr, load_wav
    from conan_amd.utils.audio.io import save_wav
    x = (0.5 * np.sin(np.arange(4000) * 0.05)).astype(np.float32)
    path = str(tmp_path / "a.wav")
    save_wav(x, path, 16000)
    y = load_wav(path, 16000)
    assert y.shape == x.shape and np.abs(y - x).max() <= 2.0 / 32768    # truncation to int16 (as the reference writes) + the 32767/32768 scale
    with pytest.raises(ValueError):
        load_wav(path, 22050)
    with pytest.raises(NotImplementedError):
        save_wav(x, str(tmp_path / "a.mp3"), 16000)
    assert librosa_pad_lr(np.zeros(1000), 1024, 320, 1) == (0, 280)
    assert librosa_pad_lr(np.zeros(960), 1024, 320, 2) == (160, 160)


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree exists in the build container only")
@pytest.mark.parametrize("cfg,table", [("egs/conan_emformer.yaml", "CONAN_EMFORMER"), ("egs/hifi_16k320_shuffle.yaml", "HIFIGAN_16K320_SHUFFLE")])
def test_set_hparams_equals_the_reference_parser_and_configs_py(cfg, table):
    """conan_amd's yaml-chain parser against the reference's own utils/commons/hparams.py run in a subprocess on the
    reference's egs/ files (key for key), and conan_amd/configs.py against those resolved chains."""
    import json
    import subprocess
    import sys
    code = ("import json,sys; sys.dont_write_bytecode=True\n"
            "from utils.commons.hparams import set_hparams\n"
            f"hp = set_hparams(config={cfg!r}, exp_name='', print_hparams=False, global_hparams=False)\n"
            "print('JSON' + json.dumps(hp, sort_keys=True, default=str))\n")
    env = dict(os.environ, PYTHONPATH=REFERENCE, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], cwd=REFERENCE, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    ref = json.loads([l for l in out.stdout.splitlines() if l.startswith("JSON")][-1][4:])
    from conan_amd.utils.commons import hparams as hpm
    cwd = os.getcwd()
    try:
        os.chdir(REFERENCE)
        ours = hpm.set_hparams(config=cfg, exp_name="", print_hparams=False, global_hparams=False)
    finally:
        os.chdir(cwd)
    ours = json.loads(json.dumps(ours, sort_keys=True, default=str))
    assert set(ours) == set(ref), (sorted(set(ours) ^ set(ref)))
    for k in ref:
        assert ours[k] == ref[k], k
    # the hot-path tables in configs.py carry the resolved values of those chains
    tab = getattr(configs, table)
    skip = {"tiny", "emformer_input_dim", "emformer_output_dim", "emformer_mode", "content_embedding_dim", "num_mels", "profile_infer", "work_dir"}
    for k, v in tab.items():
        if k in skip or k not in ref:
            continue
        assert ref[k] == v or (isinstance(v, float) and abs(float(ref[k]) - v) < 1e-12), (k, ref[k], v)


def test_state_dicts_from_checkpoints_like_the_reference_constructor(tmp_path):
    """inference/Conan.py:34-52: Conan from hp['work_dir'] (newest model_ckpt_steps_*.ckpt, 'model'), the vocoder from
    hp['vocoder_ckpt'] (config.yaml + 'model_gen'), the Emformer from hp['emformer_ckpt'] - host side, no GPU."""
    from conan_amd.inference.Conan import state_dicts_from_checkpoints
    chp, vhp = configs.conan_hparams(True), configs.hifigan_hparams(True)
    sds = {"emformer": synth.emformer_state_dict(chp, 5), "conan": synth.conan_state_dict(chp, 5), "hifigan": synth.hifigan_state_dict(vhp, 5)}
    t = lambda d: {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in d.items()}  # noqa: E731
    wd, vd, ed = tmp_path / "conan", tmp_path / "hifigan_vc", tmp_path / "emformer"
    for d in (wd, vd, ed):
        d.mkdir()
    torch.save({"state_dict": {"model": t(sds["conan"])}}, str(wd / "model_ckpt_steps_200.ckpt"))
    torch.save({"state_dict": {"model": {k: v * 0 for k, v in t(sds["conan"]).items()}}}, str(wd / "model_ckpt_steps_30.ckpt"))
    torch.save({"state_dict": {"model_gen": t(sds["hifigan"]), "model_disc": {}}}, str(vd / "model_ckpt_steps_7.ckpt"))
    (vd / "config.yaml").write_text(yaml.safe_dump(vhp))
    torch.save({"state_dict": {"model": t(sds["emformer"])}}, str(ed / "model_ckpt_steps_9.ckpt"))
    hp = dict(chp, work_dir=str(wd), vocoder_ckpt=str(vd), emformer_ckpt=str(ed))
    got, got_vhp = state_dicts_from_checkpoints(hp)
    assert got_vhp["upsample_rates"] == vhp["upsample_rates"] and got_vhp["upsample_initial_channel"] == vhp["upsample_initial_channel"]
    for name in sds:
        want = {k: v for k, v in sds[name].items() if np.asarray(v).dtype.kind == "f"}
        assert set(want) <= set(got[name]), (name, sorted(set(want) - set(got[name]))[:5])
        for k, v in want.items():
            assert np.array_equal(np.asarray(v, dtype=np.float32), got[name][k]), (name, k)
    with pytest.raises(AssertionError):
        state_dicts_from_checkpoints(dict(hp, emformer_ckpt=str(tmp_path / "missing")))
    with pytest.raises(ValueError):
        state_dicts_from_checkpoints(dict(hp, vocoder="NoSuchVocoder"))


def _bf16_rne(x):
    """float32 array -> the float32 value of its bf16 rounding (round-to-nearest-even on the bit pattern, as ctx.hip packs the
    weights and v_cvt_pk_bf16_f32 rounds on the device)."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def test_three_bf16_limbs_carry_an_fp32_value_and_six_products_its_product():
    """The arithmetic resblock_limb.hip / conv_limb.hip rest on: x = h + m + l with h = bf16(x), m = bf16(x - h),
    l = bf16(x - h - m) reproduces an fp32 x to <= 2^-24 |x| (exactly, for most values), both subtractions are exact in fp32,
    and the six limb products hh + hm + mh + hl + mm + lh - each exact in fp32 - miss x * w by <= 2^-22 |x w| before any
    accumulation rounding, i.e. the sum is an fp32-grade product."""
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-12, 8, 200000))).astype(np.float32)
    w = (rng.standard_normal(200000) * np.exp(rng.uniform(-12, 2, 200000))).astype(np.float32)

    def limbs(v):
        h = _bf16_rne(v)
        r1 = (v - h).astype(np.float32)
        assert np.array_equal(r1.astype(np.float64), v.astype(np.float64) - h.astype(np.float64))       # exact subtraction
        m = _bf16_rne(r1)
        r2 = (r1 - m).astype(np.float32)
        assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - m.astype(np.float64))
        return h, m, _bf16_rne(r2)

    xh, xm, xl = limbs(x)
    wh, wm, wl = limbs(w)
    rec = xh.astype(np.float64) + xm.astype(np.float64) + xl.astype(np.float64)
    assert np.max(np.abs(rec - x.astype(np.float64)) / np.abs(x.astype(np.float64))) <= 2.0 ** -24
    assert np.mean(rec == x.astype(np.float64)) > 0.9
    # each limb product has <= 16 significant bits: exact in fp32
    for a, b in ((xh, wh), (xh, wm), (xm, wh), (xh, wl), (xm, wm), (xl, wh)):
        p32 = (a * b).astype(np.float32)
        assert np.array_equal(p32.astype(np.float64), a.astype(np.float64) * b.astype(np.float64))
    six = sum(a.astype(np.float64) * b.astype(np.float64) for a, b in ((xl, wh), (xm, wm), (xh, wl), (xm, wh), (xh, wm), (xh, wh)))
    exact = x.astype(np.float64) * w.astype(np.float64)
    assert np.max(np.abs(six - exact) / np.abs(exact)) <= 2.0 ** -22
    # ... against which the fp32 product itself is off by up to 2^-24
    assert np.max(np.abs((x * w).astype(np.float64) - exact) / np.abs(exact)) <= 2.0 ** -24


def test_ctypes_mirror_matches_the_c_header(tmp_path):
    """include/conan_hip.h compiled as plain C (gcc): the size of every struct the ctypes binding mirrors, the enum values of
    conan_status / conan_arith and the ABI version must be what conan_amd/_lib.py declares - a field added on one side only
    would otherwise shift every argument behind it silently."""
    import shutil
    import subprocess
    from conan_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("gcc not present")
    src = tmp_path / "probe.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "conan_hip.h"\nint main(void) {\n'
                   '  printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(conan_cfg), sizeof(conan_mel_cfg), sizeof(conan_decoder_taps), sizeof(conan_hifigan_taps),\n'
                   '         sizeof(conan_streams_opts), sizeof(void*));\n'
                   '  printf("%d %d %d %d\\n", CONAN_HIP_ABI_VERSION, CONAN_ARITH_AUTO, CONAN_ARITH_F32, CONAN_ARITH_LIMB);\n'
                   '  printf("%d %d %d %d %d %d %d\\n", CONAN_OK, CONAN_ERR_INVALID, CONAN_ERR_MISSING, CONAN_ERR_SHAPE, CONAN_ERR_HIP, CONAN_ERR_STATE, CONAN_ERR_UNSUPPORTED);\n'
                   '  printf("%d %d %d %d\\n", CONAN_MAX_UPS, CONAN_MAX_RESBLOCKS, CONAN_MAX_DILATIONS, CONAN_MAX_DEC_BLOCKS);\n'
                   '  printf("%d %d %d %d %zu\\n", CONAN_STREAMS_FUSED_DECODER_BLOCKS, CONAN_STREAMS_SEPARATE_SMALL_STEPS, CONAN_STREAMS_FIXED_PLAN, CONAN_STREAMS_SHARED_DEVICE, offsetof(conan_streams_opts, dev_plan));\n  return 0;\n}\n')
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.dirname(_lib.HEADER_PATH), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    sizes = [int(x) for x in out[0].split()]
    assert sizes == [C.sizeof(_lib.ConanCfg), C.sizeof(_lib.MelCfg), C.sizeof(_lib.DecoderTaps), C.sizeof(_lib.HifiganTaps), C.sizeof(_lib.StreamsOpts), C.sizeof(C.c_void_p)]
    assert [int(x) for x in out[1].split()] == [_lib.ABI_VERSION, _lib.ARITH_AUTO, _lib.ARITH_F32, _lib.ARITH_LIMB]
    assert [int(x) for x in out[2].split()] == [_lib.OK, _lib.ERR_INVALID, _lib.ERR_MISSING, _lib.ERR_SHAPE, _lib.ERR_HIP, _lib.ERR_STATE, _lib.ERR_UNSUPPORTED]
    assert [int(x) for x in out[3].split()] == [_lib.MAX_UPS, _lib.MAX_RESBLOCKS, _lib.MAX_DILATIONS, _lib.MAX_DEC_BLOCKS]
    assert [int(x) for x in out[4].split()] == [_lib.STREAMS_FUSED_DECODER_BLOCKS, _lib.STREAMS_SEPARATE_SMALL_STEPS, _lib.STREAMS_FIXED_PLAN, _lib.STREAMS_SHARED_DEVICE, _lib.StreamsOpts.dev_plan.offset]
    # every entry point the header declares has a prototype in the binding, and the other way round
    assert sorted(_lib._PROTOS) == _lib.declared_symbols()


def test_shipped_library_reads_no_environment_variable():
    """Since ABI 8 the launch plan is a function of the arguments (conan_streams_opts.flags / .dev_plan), never of the process
    environment: the only getenv call in the library's sources is the one inside dev_getenv's `make DEV=1` branch, and the built
    library carries none of the former CONAN_* variable names that were plan switches."""
    import glob
    import re
    csrc = os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc")
    hits = []
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.inc"))):
        text = open(path).read()
        for m in re.finditer(r"(?<![A-Za-z_])getenv\(", text):
            line = text.count("\n", 0, m.start()) + 1
            hits.append((os.path.basename(path), line))
    # kernels.h: dev_getenv's body; streams.h: conan_streams::dev()'s CONAN_DEV_SWITCHES branch
    assert sorted(set(f for f, _ in hits)) == ["kernels.h", "streams.h"], hits
    for f, line in hits:
        src = open(os.path.join(csrc, f)).read().split("\n")
        window = "\n".join(src[max(0, line - 4):line])
        assert "#ifdef CONAN_DEV_SWITCHES" in window, (f, line)
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"CONAN_EMF_UNFUSED", b"CONAN_FENCED", b"CONAN_MEGA_BLK", b"CONAN_RB_NOLIMB", b"CONAN_DEC_MEGA", b"CONAN_SKIP_STAGE"):
        assert name not in blob, name
