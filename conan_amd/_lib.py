"""ctypes binding of libconan_hip.so (include/conan_hip.h).

There is no CPU fallback: importing symbols works without a GPU (so the CPU test-suite can check
that the library loads and exports every declared symbol), but every compute entry point needs a
MI355X and the library must have been built (`python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libconan_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "conan_hip.h")

ABI_VERSION = 8
MAX_UPS, MAX_RESBLOCKS, MAX_DILATIONS, MAX_DEC_BLOCKS = 8, 4, 4, 16
MODEL_EMFORMER, MODEL_CONAN, MODEL_HIFIGAN = 1, 2, 4

ARITH_AUTO, ARITH_F32, ARITH_LIMB = 0, 1, 2
ARITH_NAMES = {"auto": ARITH_AUTO, "f32": ARITH_F32, "limb": ARITH_LIMB}

OK, ERR_INVALID, ERR_MISSING, ERR_SHAPE, ERR_HIP, ERR_STATE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6


class ConanCfg(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("hidden_size", C.c_int32), ("num_mels", C.c_int32), ("content_vocab", C.c_int32),
        ("content_kernel", C.c_int32), ("dec_kernel", C.c_int32), ("dec_num_blocks", C.c_int32),
        ("dec_dilations", C.c_int32 * MAX_DEC_BLOCKS), ("dec_layers_in_block", C.c_int32),
        ("dec_post_kernel", C.c_int32), ("predictor_kernel", C.c_int32), ("nvq", C.c_int32),
        ("silent_token", C.c_int32),
        ("emf_input_dim", C.c_int32), ("emf_heads", C.c_int32), ("emf_ffn_dim", C.c_int32),
        ("emf_layers", C.c_int32), ("emf_segment", C.c_int32), ("emf_left_context", C.c_int32),
        ("emf_right_context", C.c_int32), ("emf_output_dim", C.c_int32),
        ("voc_initial_channel", C.c_int32), ("voc_num_ups", C.c_int32),
        ("voc_up_rates", C.c_int32 * MAX_UPS), ("voc_up_kernels", C.c_int32 * MAX_UPS),
        ("voc_num_resblocks", C.c_int32), ("voc_rb_kernels", C.c_int32 * MAX_RESBLOCKS),
        ("voc_rb_num_dil", C.c_int32), ("voc_rb_dilations", (C.c_int32 * MAX_DILATIONS) * MAX_RESBLOCKS),
        ("models", C.c_int32), ("voc_upsample", C.c_int32), ("voc_resblock", C.c_int32),
        ("emf_max_memory_size", C.c_int32), ("emf_tanh_on_mem", C.c_int32)]


class StreamsOpts(C.Structure):
    """conan_streams_opts (include/conan_hip.h)."""
    _fields_ = [("abi_version", C.c_int32), ("arith", C.c_int32), ("flags", C.c_int32), ("reserved0", C.c_int32),
                ("dev_plan", C.c_char_p), ("reserved", C.c_int32 * 2)]


# conan_streams_opts.flags (bit 4 - ABI 7's VOCODER_CHAIN - is retired and rejected)
STREAMS_FUSED_DECODER_BLOCKS, STREAMS_SEPARATE_SMALL_STEPS, STREAMS_FIXED_PLAN, STREAMS_SHARED_DEVICE = 1, 2, 8, 16


class ConanError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libconan_hip error {code}: {msg}")
        self.code = code


_PROTOS = {
    "conan_last_error": (C.c_char_p, []),
    "conan_abi_version": (C.c_int, []),
    "conan_ctx_create": (C.c_int, [C.c_int, C.POINTER(ConanCfg), C.POINTER(C.c_void_p)]),
    "conan_ctx_destroy": (C.c_int, [C.c_void_p]),
    "conan_ctx_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "conan_ctx_finalize": (C.c_int, [C.c_void_p]),
    "conan_streams_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "conan_streams_create_opts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(StreamsOpts), C.POINTER(C.c_void_p)]),
    "conan_streams_arith": (C.c_int, [C.c_void_p]),
    "conan_streams_destroy": (C.c_int, [C.c_void_p]),
    "conan_streams_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "conan_set_reference": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "conan_emformer_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "conan_emformer_head_dim": (C.c_int, [C.c_void_p, C.c_char_p]),
    "conan_emformer_project": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "conan_decoder_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "conan_hifigan_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "conan_hifigan_step_taps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "conan_set_style": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "conan_get_prosody_ids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "conan_decoder_step_taps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "conan_get_style": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "conan_wav2mel": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "conan_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "conan_step_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "conan_streams_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "conan_streams_output_fence": (C.c_int, [C.c_void_p, C.c_void_p]),
    "conan_streams_output_fence_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "conan_streams_test_fault": (C.c_int, [C.c_void_p, C.c_int]),
    "conan_profile_mark": (C.c_int, [C.c_void_p, C.c_void_p]),
    "conan_step_clock": (C.c_int, [C.c_void_p, C.c_int]),
    "conan_step_clock_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
    "conan_step_timeline": (C.c_int, [C.c_void_p, C.c_int]),
    "conan_step_timeline_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int]),
    "conan_profile_begin": (C.c_int, [C.c_void_p]),
    "conan_profile_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "conan_profile_kernel": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "conan_hop_size": (C.c_int, [C.c_void_p]),
    "conan_ctx_weight_bytes": (C.c_int64, [C.c_void_p]),
    "conan_streams_state_bytes": (C.c_int64, [C.c_void_p]),
}

_lib = None


def declared_symbols():
    """Function names declared in include/conan_hip.h."""
    with open(HEADER_PATH) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(conan_[a-z_0-9]+)\s*\(", txt)))


def lib():
    """The loaded library.  Fails loudly when it has not been built: no CPU fallback exists."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()').  conan_amd has no CPU fallback.")
        # PyTorch-ROCm bundles its own HIP runtime; load it first so that libconan_hip.so binds to the same
        # libamdhip64 (two runtimes in one process do not see each other's devices / streams).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.conan_abi_version() != ABI_VERSION:
            raise ImportError("libconan_hip.so ABI version mismatch; rebuild it")
        _lib = l
    return _lib


def check(rc):
    if rc < 0:
        raise ConanError(rc, lib().conan_last_error().decode("utf-8", "replace"))
    return rc


class MelCfg(C.Structure):
    """conan_mel_cfg (include/conan_hip.h)."""
    _fields_ = [("fft_size", C.c_int32), ("hop_size", C.c_int32), ("win_length", C.c_int32), ("num_mels", C.c_int32),
                ("sample_rate", C.c_int32), ("fmin", C.c_float), ("fmax", C.c_float), ("eps", C.c_float),
                ("vmin", C.c_float), ("vmax", C.c_float), ("framing", C.c_int32), ("natural_log", C.c_int32),
                ("mag_eps", C.c_float)]


class DecoderTaps(C.Structure):
    """conan_decoder_taps (include/conan_hip.h)."""
    _fields_ = [("uv_pred", C.c_void_p), ("f0_denorm_pred", C.c_void_p), ("pitch_bins", C.c_void_p), ("decoder_inp", C.c_void_p),
                ("content_embed_proj", C.c_void_p), ("attn", C.c_void_p * 2)]


class HifiganTaps(C.Structure):
    """conan_hifigan_taps (include/conan_hip.h)."""
    _fields_ = [("conv_pre_act", C.c_void_p), ("ups", C.c_void_p * MAX_UPS), ("stage_out", C.c_void_p * MAX_UPS)]


def make_cfg(conan_hp=None, hifigan_hp=None, emformer=True, conan=True, hifigan=True):
    """conan_cfg from the reference's hparams dicts (utils/commons/hparams.py)."""
    c = ConanCfg()
    c.abi_version = ABI_VERSION
    models = 0
    hp = conan_hp or {}
    if conan_hp is not None and conan:
        models |= MODEL_CONAN
        if hp.get("decoder_type", "conv") != "conv" or hp.get("f0_gen", "orig") != "orig" or not hp.get("style", True):
            raise ConanError(ERR_UNSUPPORTED, "only decoder_type='conv', f0_gen='orig', style=true (egs/conan_emformer.yaml) is on the hot path")
        c.hidden_size = hp["hidden_size"]
        c.content_vocab = 102
        c.content_kernel = hp["kernel_size"]
        c.dec_kernel = hp["dec_kernel_size"]
        dd = list(hp["dec_dilations"])
        c.dec_num_blocks = len(dd)
        for i, d in enumerate(dd):
            c.dec_dilations[i] = d
        c.dec_layers_in_block = hp["layers_in_block"]
        c.dec_post_kernel = hp.get("dec_post_net_kernel", 3)
        c.predictor_kernel = hp["predictor_kernel"]
        c.nvq = hp["nVQ"]
        c.silent_token = hp["silent_token"]
    c.num_mels = (conan_hp or hifigan_hp or {}).get("audio_num_mel_bins", 80)
    if conan_hp is not None and emformer:
        models |= MODEL_EMFORMER
        c.emf_input_dim = 80
        c.emf_heads = 8
        c.emf_ffn_dim = 2048
        c.emf_layers = hp["emformer_layers"]
        c.emf_segment = hp["chunk_size"] // 20
        c.emf_left_context = 50
        c.emf_right_context = hp["right_context"]
        # mode == 'both': the streaming loop projects with proj1 (80 -> 100), inference/Conan.py:117-118
        c.emf_output_dim = 100 if hp.get("mode", None) == "both" else hp.get("emformer_output_dim", 100)
        # not a key of the reference's yaml files (modules/Emformer/emformer.py:14-22 leaves torchaudio's defaults, 0 /
        # False); read when present so that the memory bank can be exercised
        c.emf_max_memory_size = int(hp.get("emformer_max_memory_size", 0))
        c.emf_tanh_on_mem = int(bool(hp.get("emformer_tanh_on_mem", False)))
    if hifigan_hp is not None and hifigan:
        v = hifigan_hp
        up = v.get("upsample", "shuffle")
        if up not in ("shuffle", "zero", "nn"):
            raise ConanError(ERR_UNSUPPORTED, "upsample='%s': 'shuffle', 'zero' or 'nn' (hifigan_causal.py:287-293)" % up)
        c.voc_upsample = {"shuffle": 0, "zero": 1, "nn": 2}[up]
        c.voc_resblock = 1 if str(v.get("resblock", "1")) == "1" else 2
        if len({len(ds) for ds in v["resblock_dilation_sizes"]}) != 1:
            raise ConanError(ERR_UNSUPPORTED, "resblock branches with different numbers of dilations")
        models |= MODEL_HIFIGAN
        c.voc_initial_channel = v.get("upsample_initial_channel", 512)
        c.voc_num_ups = len(v["upsample_rates"])
        for i, (r, k) in enumerate(zip(v["upsample_rates"], v["upsample_kernel_sizes"])):
            c.voc_up_rates[i] = r
            c.voc_up_kernels[i] = k
        c.voc_num_resblocks = len(v["resblock_kernel_sizes"])
        c.voc_rb_num_dil = len(v["resblock_dilation_sizes"][0])
        for b, (k, ds) in enumerate(zip(v["resblock_kernel_sizes"], v["resblock_dilation_sizes"])):
            c.voc_rb_kernels[b] = k
            for j, d in enumerate(ds):
                c.voc_rb_dilations[b][j] = d
    c.models = models
    return c
