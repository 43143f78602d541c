"""Batch file runner with the reference's interface (inference/run_voice_conversion.py:14-140): a JSON list of
{ref_wav, src_wav, output_name} pairs -> converted wavs.  Where the reference converts one pair at a time, this runner
feeds up to `streams` pairs at once to the multi-stream engine (each pair is an independent stream; shorter sources are
padded with their last frame, which is exactly what the chunk loop does for the look-ahead of a final chunk,
inference/Conan.py:100-110, so every file's output equals its single-stream conversion)."""
import json
import os
import time

import numpy as np
import torch

from ..engine import StreamingVoiceConversionEngine
from ..utils.audio.io import save_wav
from .Conan import StreamingVoiceConversion


class VoiceConversionRunner:
    def __init__(self, config_file="voice_conversion_config.json", hparams=None, vocoder_hp=None, state_dicts=None,
                 output_dir="test_output_hifigan", streams=16, device=0):
        self.config_file = config_file
        self.config = self.load_config()
        self.output_dir = output_dir
        os.makedirs(self.output_dir, exist_ok=True)
        self.hparams = hparams
        self.vc = StreamingVoiceConversion(hparams, vocoder_hp, state_dicts, device)
        self.streams = int(streams)
        self.engine = None

    def load_config(self):
        if not os.path.exists(self.config_file):
            raise FileNotFoundError(f"Configuration file {self.config_file} not found")
        with open(self.config_file, "r") as f:
            config = json.load(f)
        config.setdefault("total_pairs", len(config["conversion_pairs"]))
        return config

    def run_single_conversion(self, pair, pair_idx):
        try:
            wav_pred, _ = self.vc.infer_once({"ref_wav": pair["ref_wav"], "src_wav": pair["src_wav"]})
            out = os.path.join(self.output_dir, pair["output_name"])
            save_wav(wav_pred, out, self.hparams["audio_sample_rate"])
            return True, out
        except Exception as e:  # noqa: BLE001  (per-file errors are reported, like the reference runner)
            return False, str(e)

    @torch.no_grad()
    def run_batch(self, pairs):
        """Convert len(pairs) <= streams pairs concurrently; returns the list of output paths."""
        vc = self.vc
        src = [vc._wav_to_mel(p["src_wav"]) for p in pairs]
        ref = [vc._wav_to_mel(p["ref_wav"]) for p in pairs]
        B = len(pairs)
        T, Tr = max(m.shape[0] for m in src), max(m.shape[0] for m in ref)
        srcb = torch.stack([torch.cat([m, m[-1:].expand(T - m.shape[0], -1)]) for m in src])
        refb = torch.stack([torch.cat([m, m.new_zeros(Tr - m.shape[0], m.shape[1])]) for m in ref])
        ref_len = [m.shape[0] for m in ref]
        if self.engine is None or self.engine.n != B or self.engine.st.max_ref_frames < Tr:
            if self.engine is not None:
                self.engine.st.close()
            self.engine = StreamingVoiceConversionEngine(vc.ctx, B, max_ref_frames=max(256, Tr))
        wav, _, _ = self.engine.infer(srcb, refb, ref_len)
        hop = vc.ctx.hop
        outs = []
        for k, p in enumerate(pairs):
            out = os.path.join(self.output_dir, p["output_name"])
            save_wav(wav[k, :src[k].shape[0] * hop].cpu().numpy(), out, self.hparams["audio_sample_rate"])
            outs.append(out)
        return outs

    def run_all_conversions(self, start_idx=0, end_idx=None, batch_size=50):
        pairs = self.config["conversion_pairs"]
        end_idx = len(pairs) if end_idx is None else end_idx
        ok, failed, errors, t0 = 0, 0, [], time.time()
        progress_file = os.path.join(self.output_dir, "conversion_progress.json")
        i = start_idx
        while i < end_idx:
            group = pairs[i:min(i + self.streams, end_idx)]
            try:
                self.run_batch(group)
                ok += len(group)
            except Exception:  # noqa: BLE001  (fall back to per-file conversion so that one bad file does not sink the group)
                for j, p in enumerate(group):
                    success, res = self.run_single_conversion(p, i + j)
                    ok += success
                    failed += (not success)
                    if not success:
                        errors.append(f"Pair {i + j}: {res}")
            i += len(group)
            if (i - start_idx) % batch_size < len(group) or i >= end_idx:
                with open(progress_file, "w") as f:
                    json.dump({"processed": i - start_idx, "total": end_idx - start_idx, "successful": ok, "failed": failed,
                               "elapsed_time": time.time() - t0, "errors": errors}, f, indent=2)
        return {"successful": ok, "failed": failed, "errors": errors, "elapsed_time": time.time() - t0}
