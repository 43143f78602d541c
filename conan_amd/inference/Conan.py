"""StreamingVoiceConversion with the reference's interface (inference/Conan.py:20-166), running the chunk loop
through the fused HIP step.  Inputs are mel spectrograms: the librosa wav front-end (inference/Conan.py:57-70) is
the step before the hot path and is not re-implemented here."""
from typing import Dict

import numpy as np
import torch

from .. import configs
from ..engine import StreamingVoiceConversionEngine
from ..runtime import Context


class StreamingVoiceConversion:
    tokens_per_chunk: int = 4

    def __init__(self, hp: Dict, vocoder_hp: Dict = None, state_dicts: Dict = None, device: int = 0):
        """hp: Conan/Emformer hparams (utils.commons.hparams.set_hparams('egs/conan_emformer.yaml'));
        vocoder_hp: the vocoder's own config; state_dicts: {'emformer','conan','hifigan'} -> state_dict
        (as extracted by utils.commons.ckpt_utils.extract_state_dict from the reference's checkpoints)."""
        if not torch.cuda.is_available():
            raise RuntimeError("StreamingVoiceConversion needs a HIP device (no CPU fallback)")
        self.hparams = hp
        self.device = f"cuda:{device}"
        vocoder_hp = vocoder_hp or configs.hifigan_hparams()
        if hp.get("vocoder", "HifiGAN") != "HifiGAN":
            raise ValueError(f"Vocoder '{hp['vocoder']}' is not registered. Check vocoder name and registration.")
        self.ctx = Context(hp, vocoder_hp, device)
        if state_dicts is None:
            raise ValueError("state_dicts with 'emformer', 'conan' and 'hifigan' entries are required")
        for name in ("emformer", "conan", "hifigan"):
            self.ctx.load_state_dict(name, state_dicts[name])
        self.ctx.finalize()
        self.engine = None

    def infer_once(self, inp: Dict):
        """inp: {'ref_mel': [Tr,80], 'src_mel': [T,80]} (numpy / torch).  Returns (wav np[N], mel np[T,80])
        like inference/Conan.py:166."""
        if "ref_mel" not in inp or "src_mel" not in inp:
            raise NotImplementedError("pass 'ref_mel' / 'src_mel' (clipped log-mel, inference/Conan.py:57-70); "
                                      "the wav front-end is outside the hot path")
        ref = torch.as_tensor(np.asarray(inp["ref_mel"]), dtype=torch.float32, device=self.device)[None]
        src = torch.as_tensor(np.asarray(inp["src_mel"]), dtype=torch.float32, device=self.device)[None]
        if self.engine is None or self.engine.st.max_ref_frames < ref.shape[1]:
            self.engine = StreamingVoiceConversionEngine(self.ctx, 1, max_ref_frames=max(256, ref.shape[1]))
        wav, mel, _ = self.engine.infer(src, ref)
        return wav[0].cpu().numpy(), mel[0].cpu().numpy()
