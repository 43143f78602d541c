"""Hot-path hyper-parameters.

These dicts hold only the keys the streaming path reads; the values are the
resolved ones of the reference's yaml chains (egs/conan_emformer.yaml ->
egs/egs_bases/tts/fs.yaml -> base.yaml -> config_base.yaml, and
egs/hifi_16k320_shuffle.yaml -> egs_bases/tts/vocoder/hifigan.yaml), cited
per key.  A user with the reference's `egs/` tree can instead call
`conan_amd.hparams.set_hparams('egs/conan_emformer.yaml')`, which parses the
yaml chain exactly like utils/commons/hparams.py:25-131.
"""
import copy

# egs/conan_emformer.yaml + egs/egs_bases/tts/fs.yaml
CONAN_EMFORMER = {
    "hidden_size": 256,            # conan_emformer.yaml:76
    "kernel_size": 3,              # conan_emformer.yaml:77 (content_proj)
    "audio_num_mel_bins": 80,      # dataset_params
    "audio_sample_rate": 16000,    # conan_emformer.yaml:29
    "hop_size": 320,               # conan_emformer.yaml:30
    "win_size": 1024, "fft_size": 1024,   # conan_emformer.yaml:33-34
    "fmin": 80, "fmax": 7600,             # conan_emformer.yaml:37-38
    "loud_norm": False,                   # egs_bases/tts/dataset_params.yaml:15
    "decoder_type": "conv",        # conan_emformer.yaml:56
    "dec_dilations": [1, 1, 1, 1],  # fs.yaml
    "dec_kernel_size": 5,          # fs.yaml
    "dec_post_net_kernel": 3,      # fs.yaml
    "layers_in_block": 2,          # fs.yaml
    "enc_dec_norm": "ln",          # fs.yaml
    "dropout": 0.0,
    "enc_layers": 4, "dec_layers": 4,
    "use_spk_id": False, "use_spk_embed": False,
    "use_pitch_embed": True,       # conan_emformer.yaml:46
    "predictor_hidden": -1,        # fs.yaml
    "predictor_kernel": 5,         # fs.yaml
    "predictor_grad": 1.0,
    "dec_inp_add_noise": False,
    "f0_gen": "orig",              # conan_emformer.yaml:47
    "style": True,                 # conan_emformer.yaml
    "nVQ": 512, "lambda_commit": 0.25, "vae_dropout": 0.0,
    "vq_start": 20500, "forcing": 20000,
    "silent_token": 57, "content_embedding_dim": 102,
    "mel_vmin": -6.0, "mel_vmax": 1.5,
    # Emformer (modules/Emformer/emformer.py:14-22)
    "emformer_layers": 6, "chunk_size": 80, "right_context": 2,
    "emformer_input_dim": 80, "emformer_output_dim": 100, "emformer_mode": None,
    "vocoder": "HifiGAN", "vocoder_ckpt": "checkpoints/hifigan_vc",
    "work_dir": "", "emformer_ckpt": "checkpoints/emformer_test2",
    "profile_infer": False,
}

# egs/hifi_16k320_shuffle.yaml + egs/egs_bases/tts/vocoder/hifigan.yaml
HIFIGAN_16K320_SHUFFLE = {
    "audio_num_mel_bins": 80, "num_mels": 80,
    "audio_sample_rate": 16000, "hop_size": 320,
    "upsample_rates": [8, 5, 4, 2],            # hifi_16k320_shuffle.yaml:4
    "upsample_kernel_sizes": [16, 10, 8, 4],   # hifi_16k320_shuffle.yaml:5
    "upsample_initial_channel": 512,           # hifigan.yaml
    "resblock": "1",
    "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    "upsample": "shuffle",                     # hifi_16k320_shuffle.yaml:20
    "use_pitch_embed": False,
}

# Tiny variants for fast unit tests; same topology, small widths.
CONAN_TINY = dict(copy.deepcopy(CONAN_EMFORMER), hidden_size=32, nVQ=16, emformer_layers=2,
                  tiny=True)
HIFIGAN_TINY = dict(copy.deepcopy(HIFIGAN_16K320_SHUFFLE), upsample_initial_channel=64)


# vocoder config.yaml variants the generator accepts (hifigan_causal.py:287-303): `upsample: zero` + `resblock: "2"`
HIFIGAN_ZERO_RB2 = dict(copy.deepcopy(HIFIGAN_16K320_SHUFFLE), upsample="zero", resblock="2",
                        resblock_dilation_sizes=[[1, 3], [1, 3], [1, 3]])
HIFIGAN_ZERO_RB2_TINY = dict(copy.deepcopy(HIFIGAN_ZERO_RB2), upsample_initial_channel=64)
# `upsample: nn` (CausalUpsampleBlock1, transposed convolution): whole-utterance / whole-window forward only
HIFIGAN_NN = dict(copy.deepcopy(HIFIGAN_16K320_SHUFFLE), upsample="nn")
HIFIGAN_NN_TINY = dict(copy.deepcopy(HIFIGAN_NN), upsample_initial_channel=64)


def conan_hparams(tiny=False):
    return copy.deepcopy(CONAN_TINY if tiny else CONAN_EMFORMER)


def hifigan_hparams(tiny=False):
    return copy.deepcopy(HIFIGAN_TINY if tiny else HIFIGAN_16K320_SHUFFLE)
