"""StreamingVoiceConversion with the reference's interface (inference/Conan.py:20-166), running the chunk loop
through the fused HIP step.  Inputs are mel spectrograms ('ref_mel' / 'src_mel') or waveforms / wav paths ('ref_wav' /
'src_wav', inference/Conan.py:72-80): the mel front-end (inference/Conan.py:57-70) runs on the GPU (conan_wav2mel)."""
from typing import Dict

import numpy as np
import torch

from .. import configs
from ..engine import StreamingVoiceConversionEngine
from ..runtime import Context


def state_dicts_from_checkpoints(hp: Dict):
    """The three models exactly as inference/Conan.py:34-52 builds them: Conan(0, hp) + load_ckpt(work_dir, strict=False);
    get_vocoder_cls(hp['vocoder'])() -> `{vocoder_ckpt}/config.yaml` + newest checkpoint's 'model_gen'
    (tasks/tts/vocoder_infer/hifigan.py:13-20); EmformerDistillModel(hp, output_dim=100) + load_ckpt(emformer_ckpt,
    strict=False).  Returns ({'emformer','conan','hifigan'} -> host state_dict, vocoder config).  Host-side only."""
    from ..modules import _tree
    from ..modules.Conan.Conan import Conan
    from ..modules.Emformer.emformer import EmformerDistillModel
    from ..modules.vocoder.hifigan.hifigan_causal import HifiGanGenerator
    from ..tasks.tts.vocoder_infer import hifigan as _registers_hifigan  # noqa: F401  (@register_vocoder('HifiGAN'))
    from ..tasks.tts.vocoder_infer.base_vocoder import get_vocoder_cls
    from ..utils.commons.ckpt_utils import load_ckpt
    from ..utils.commons.hparams import set_hparams
    model = Conan(0, hp)
    load_ckpt(model, hp["work_dir"], strict=False)
    vocoder_cls = get_vocoder_cls(hp["vocoder"])
    if vocoder_cls is None:
        raise ValueError(f"Vocoder '{hp['vocoder']}' is not registered. Check vocoder name and registration.")
    base_dir = hp["vocoder_ckpt"]
    vocoder_hp = set_hparams(f"{base_dir}/config.yaml", global_hparams=False, print_hparams=False)
    gen = HifiGanGenerator(vocoder_hp)
    load_ckpt(gen, base_dir, "model_gen")
    emformer = EmformerDistillModel(hp, output_dim=100)
    load_ckpt(emformer, hp["emformer_ckpt"], strict=False)
    sds = {"conan": _tree.host_state_dict(model), "hifigan": _tree.host_state_dict(gen), "emformer": _tree.host_state_dict(emformer)}
    return sds, vocoder_hp


class StreamingVoiceConversion:
    tokens_per_chunk: int = 4

    def __init__(self, hp: Dict, vocoder_hp: Dict = None, state_dicts: Dict = None, device: int = 0):
        """StreamingVoiceConversion(hp) as the reference constructs it (inference/Conan.py:26-55): the models come from
        hp['work_dir'] / hp['vocoder_ckpt'] / hp['emformer_ckpt'] through load_ckpt, then the vocoder is warmed with 4
        zero frames (:54-55).  `vocoder_hp` / `state_dicts` ({'emformer','conan','hifigan'} -> state_dict) build the
        same object from in-memory weights instead (no checkpoint files)."""
        if not torch.cuda.is_available():
            raise RuntimeError("StreamingVoiceConversion needs a HIP device (no CPU fallback)")
        self.hparams = hp
        self.device = f"cuda:{device}"
        if hp.get("vocoder", "HifiGAN") != "HifiGAN":
            raise ValueError(f"Vocoder '{hp['vocoder']}' is not registered. Check vocoder name and registration.")
        if state_dicts is None:
            state_dicts, ckpt_vocoder_hp = state_dicts_from_checkpoints(hp)
            vocoder_hp = vocoder_hp or ckpt_vocoder_hp
        vocoder_hp = vocoder_hp or configs.hifigan_hparams()
        self.ctx = Context(hp, vocoder_hp, device)
        for name in ("emformer", "conan", "hifigan"):
            self.ctx.load_state_dict(name, state_dicts[name])
        self.ctx.finalize()
        self.engine = None
        self._vocoder_warm_zero()

    def _vocoder_warm_zero(self):
        """inference/Conan.py:54-55: one vocoder pass over 4 zero frames (first-launch costs); state is reset per utterance."""
        st = self.ctx.streams(1, max_frames=4, max_ref_frames=4)
        st.reset([0], which=4)
        st.hifigan_step([0], torch.zeros(1, 4, self.ctx.cfg.num_mels, device=self.device))
        torch.cuda.synchronize()
        st.close()

    def _wav_to_mel(self, wav) -> torch.Tensor:
        """inference/Conan.py:57-70 on the GPU: path or float array -> clipped log-mel [T, 80] (cuda)."""
        from ..utils.audio import load_wav
        hp = self.hparams
        if isinstance(wav, str):
            wav = load_wav(wav, hp["audio_sample_rate"])
        if hp.get("loud_norm", False):
            raise NotImplementedError("loud_norm is off on the inference path (egs_bases/tts/dataset_params.yaml:15)")
        return self.ctx.wav2mel(torch.as_tensor(np.asarray(wav), dtype=torch.float32), fft_size=hp["fft_size"], hop_size=hp["hop_size"],
                                win_length=hp["win_size"], num_mels=hp["audio_num_mel_bins"], fmin=hp["fmin"], fmax=hp["fmax"],
                                sample_rate=hp["audio_sample_rate"], mel_vmin=hp["mel_vmin"], mel_vmax=hp["mel_vmax"])[0]

    def infer_once(self, inp: Dict):
        """inp: {'ref_wav', 'src_wav'} (paths or float arrays, like inference/Conan.py:72-80) or {'ref_mel': [Tr,80],
        'src_mel': [T,80]} (numpy / torch).  Returns (wav np[N], mel np[T,80]) like inference/Conan.py:166."""
        if "ref_mel" in inp and "src_mel" in inp:
            ref = torch.as_tensor(np.asarray(inp["ref_mel"]), dtype=torch.float32, device=self.device)[None]
            src = torch.as_tensor(np.asarray(inp["src_mel"]), dtype=torch.float32, device=self.device)[None]
        elif "ref_wav" in inp and "src_wav" in inp:
            ref = self._wav_to_mel(inp["ref_wav"])[None]
            src = self._wav_to_mel(inp["src_wav"])[None]
        else:
            raise ValueError("pass 'ref_wav' / 'src_wav' or 'ref_mel' / 'src_mel'")
        if self.engine is None or self.engine.st.max_ref_frames < ref.shape[1]:
            self.engine = StreamingVoiceConversionEngine(self.ctx, 1, max_ref_frames=max(256, ref.shape[1]))
        wav, mel, _ = self.engine.infer(src, ref)
        return wav[0].cpu().numpy(), mel[0].cpu().numpy()
