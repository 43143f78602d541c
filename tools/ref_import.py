"""Import the read-only reference (/root/reference) in THIS container with its
non-arithmetic third-party imports stubbed.  Used only by
tools/make_goldens.py to generate fixtures; never travels to the GPU box."""
import os, sys, types

REF = "/root/reference"


def _stub(name, attrs=()):
    m = types.ModuleType(name)
    m.__path__ = []
    for a in attrs:
        setattr(m, a, None)
    sys.modules[name] = m
    return m


def install():
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present")
    for n in ["librosa", "librosa.filters", "librosa.util", "librosa.core", "librosa.feature",
              "pyloudnorm", "torchdyn", "torchdyn.core", "webrtcvad", "skimage", "skimage.transform",
              "soundfile", "resemblyzer", "parselmouth"]:
        _stub(n)
    sys.modules["librosa.filters"].mel = None
    sys.modules["librosa.util"].normalize = None
    sys.modules["librosa.util"].pad_center = None
    sys.modules["librosa.util"].tiny = None
    sys.modules["torchdyn.core"].NeuralODE = object
    sys.modules["skimage.transform"].resize = None
    sys.modules["resemblyzer"].VoiceEncoder = object
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(REF)


def build_conan():
    from utils.commons.hparams import set_hparams, hparams
    hp = set_hparams(config="egs/conan_emformer.yaml", exp_name="", print_hparams=False)
    from modules.Conan.Conan import Conan
    m = Conan(0, hp).eval()
    return m, hp


def build_vocoder():
    from utils.commons.hparams import set_hparams
    hp = set_hparams("egs/hifi_16k320_shuffle.yaml", exp_name="", print_hparams=False, global_hparams=False)
    from modules.vocoder.hifigan.hifigan_causal import HifiGanGenerator
    g = HifiGanGenerator(hp).eval()
    return g, hp
