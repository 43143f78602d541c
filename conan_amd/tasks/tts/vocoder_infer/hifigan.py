"""`HifiGAN` vocoder wrapper with the reference's numpy seam (tasks/tts/vocoder_infer/hifigan.py:11-31):
spec2wav(mel: np.float32[T, 80]) -> np.float32[T*hop], computed by the HIP generator."""
import os

import numpy as np
import torch

from ....modules.vocoder.hifigan.hifigan_causal import HifiGanGenerator
from ....utils.commons.ckpt_utils import load_ckpt
from ....utils.commons.hparams import hparams, set_hparams
from .base_vocoder import BaseVocoder, register_vocoder


@register_vocoder("HifiGAN")
class HifiGAN(BaseVocoder):
    def __init__(self, config=None, state_dict=None):
        """Reference behaviour (no arguments): read `{hparams['vocoder_ckpt']}/config.yaml` and the newest
        checkpoint's 'model_gen'.  `config` / `state_dict` allow construction from in-memory objects."""
        if config is None:
            base_dir = hparams["vocoder_ckpt"]
            config = set_hparams(f"{base_dir}/config.yaml", global_hparams=False, print_hparams=False)
            self.model = HifiGanGenerator(config)
            load_ckpt(self.model, base_dir, "model_gen")
        else:
            self.model = HifiGanGenerator(config)
            if state_dict is not None:
                self.model.load_state_dict(state_dict)
        self.config = config
        if not torch.cuda.is_available():
            raise RuntimeError("conan_amd vocoder needs a HIP device (no CPU fallback)")
        self.device = torch.device("cuda")
        self.model.to(self.device)
        self.model.eval()

    def spec2wav(self, mel, **kwargs):
        with torch.no_grad():
            c = torch.as_tensor(np.asarray(mel), dtype=torch.float32).unsqueeze(0).to(self.device).transpose(2, 1)
            y = self.model(c).view(-1)
        return y.cpu().numpy()
