/*
 * conan_hip.h -- C-ABI of libconan_hip.so, the MI355X (gfx950) implementation of the
 * chunkwise streaming voice-conversion hot path of User-tian/Conan.
 *
 * The reference has no FFI (it is pure Python/PyTorch); its seams are Python-level.  Each entry
 * point below names the reference interface it replaces.  All tensor pointers marked _dev are
 * DEVICE pointers owned by the caller (e.g. PyTorch-ROCm `tensor.data_ptr()`), fp32 contiguous
 * row-major as documented; the library owns only its handles.  Work is enqueued asynchronously on
 * the hipStream_t passed in (`stream`, as void*; NULL = the default stream).  Functions return 0
 * on success or a negative conan_status and never throw; conan_last_error() is thread-local.
 * One conan_streams handle may be driven by one host thread at a time; distinct handles are
 * independent.
 */
#ifndef CONAN_HIP_H
#define CONAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CONAN_HIP_ABI_VERSION 8

typedef enum conan_status {
  CONAN_OK = 0,
  CONAN_ERR_INVALID = -1,     /* bad argument (the reference raises ValueError / assert)          */
  CONAN_ERR_MISSING = -2,     /* a required state_dict tensor was not loaded (strict load failure) */
  CONAN_ERR_SHAPE = -3,       /* tensor shape does not match the configured architecture          */
  CONAN_ERR_HIP = -4,         /* HIP runtime error                                                */
  CONAN_ERR_STATE = -5,       /* call order violation (e.g. step before finalize/set_reference)    */
  CONAN_ERR_UNSUPPORTED = -6  /* configuration outside the implemented hot path                   */
} conan_status;

#define CONAN_MAX_UPS 8
#define CONAN_MAX_RESBLOCKS 4
#define CONAN_MAX_DILATIONS 4
#define CONAN_MAX_DEC_BLOCKS 16

/* Architecture hyper-parameters: the subset of the reference's `hparams` the hot path reads
 * (utils/commons/hparams.py:25-131; egs/conan_emformer.yaml, egs/hifi_16k320_shuffle.yaml). */
typedef struct conan_cfg {
  int32_t abi_version;            /* must be CONAN_HIP_ABI_VERSION */
  /* Conan (modules/Conan/Conan.py:46-113, modules/tts/fs.py:49-79) */
  int32_t hidden_size;            /* hparams['hidden_size'] (256) */
  int32_t num_mels;               /* audio_num_mel_bins (80) */
  int32_t content_vocab;          /* nn.Embedding(102, H) */
  int32_t content_kernel;         /* hparams['kernel_size'] (3) */
  int32_t dec_kernel;             /* dec_kernel_size (5) */
  int32_t dec_num_blocks;         /* len(dec_dilations) (4) */
  int32_t dec_dilations[CONAN_MAX_DEC_BLOCKS];
  int32_t dec_layers_in_block;    /* layers_in_block (2) */
  int32_t dec_post_kernel;        /* dec_post_net_kernel (3) */
  int32_t predictor_kernel;       /* predictor_kernel (5) */
  int32_t nvq;                    /* nVQ (512) */
  int32_t silent_token;           /* silent_token (57) */
  /* Emformer (modules/Emformer/emformer.py:14-25) */
  int32_t emf_input_dim;          /* 80 */
  int32_t emf_heads;              /* 8 */
  int32_t emf_ffn_dim;            /* 2048 */
  int32_t emf_layers;             /* emformer_layers (6) */
  int32_t emf_segment;            /* chunk_size // 20 (4) */
  int32_t emf_left_context;       /* 50 */
  int32_t emf_right_context;      /* right_context (2) */
  int32_t emf_output_dim;         /* proj out (100) */
  /* HiFi-GAN (modules/vocoder/hifigan/hifigan_causal.py:273-312) */
  int32_t voc_initial_channel;    /* upsample_initial_channel (512) */
  int32_t voc_num_ups;
  int32_t voc_up_rates[CONAN_MAX_UPS];
  int32_t voc_up_kernels[CONAN_MAX_UPS];
  int32_t voc_num_resblocks;      /* len(resblock_kernel_sizes) (3) */
  int32_t voc_rb_kernels[CONAN_MAX_RESBLOCKS];
  int32_t voc_rb_num_dil;
  int32_t voc_rb_dilations[CONAN_MAX_RESBLOCKS][CONAN_MAX_DILATIONS];
  /* which sub-models this context holds (bit 0 Emformer, bit 1 Conan, bit 2 HiFi-GAN) */
  int32_t models;
  /* vocoder config.yaml choices (hifigan_causal.py:287-303); 0 = the shipped egs/hifi_16k320_shuffle.yaml values */
  int32_t voc_upsample;           /* 0: 'shuffle' (CausalUpsampleBlock3), 1: 'zero' (CausalUpsampleBlock2), 2: 'nn' (CausalUpsampleBlock1,
                                     hifigan_causal.py:60-145: looks ahead, so every vocoder step must follow a reset of its slots
                                     and carry the whole utterance or window; CONAN_ERR_STATE otherwise) */
  int32_t voc_resblock;           /* 0 or 1: ResBlock1, 2: ResBlock2 */
  /* torchaudio.models.Emformer(max_memory_size=, tanh_on_mem=): the memory bank.  modules/Emformer/emformer.py:14-22
   * never passes them (0 / false), so shipped checkpoints run without a bank; > 0 enables the summary vector,
   * the per-layer memory ring and the memory tokens in the attention (BASELINE.json north_star "memory-bank update") */
  int32_t emf_max_memory_size;
  int32_t emf_tanh_on_mem;
} conan_cfg;

#define CONAN_MODEL_EMFORMER 1
#define CONAN_MODEL_CONAN 2
#define CONAN_MODEL_HIFIGAN 4

typedef struct conan_ctx conan_ctx;          /* per device: packed weights, workspaces */
typedef struct conan_streams conan_streams;  /* per-slot streaming state               */

const char* conan_last_error(void);
int conan_abi_version(void);

/* Replaces model construction: Conan(0, hp) inference/Conan.py:34-38, EmformerDistillModel(hp,
 * output_dim=100) :47-52, HifiGanGenerator(config) tasks/tts/vocoder_infer/hifigan.py:17. */
int conan_ctx_create(int device, const conan_cfg* cfg, conan_ctx** out);
int conan_ctx_destroy(conan_ctx* ctx);

/* Replaces load_ckpt / nn.Module.load_state_dict (utils/commons/ckpt_utils.py:26-66).
 * `key` is "<model>.<state_dict key>" with model in {"emformer","conan","hifigan"}; `host`
 * is a HOST pointer to fp32 data of `shape[0..ndim)`; the data is copied.  Unknown keys are
 * ignored (strict=False behaviour) and reported through the return value 1. */
int conan_ctx_load_tensor(conan_ctx* ctx, const char* key, const float* host, const int64_t* shape, int ndim);

/* Fold weight-norm (w = g*v/||v||, hifigan_causal.py:45), permute pixel-shuffle output channels,
 * repack every weight to the kernels' [tap][Cin/4][Cout][4] layout and upload.  Fails with
 * CONAN_ERR_MISSING naming the first required tensor that was not loaded. */
int conan_ctx_finalize(conan_ctx* ctx);

/* Per-slot state: Emformer K/V rings + past length, Conan conv rings + cached style/prosody,
 * HiFi-GAN conv rings.  max_frames = largest number of mel frames one step may carry;
 * max_ref_frames = longest reference mel conan_set_reference may receive. */
int conan_streams_create(conan_ctx* ctx, int max_slots, int max_frames, int max_ref_frames, conan_streams** out);
int conan_streams_destroy(conan_streams* s);

/* Arithmetic of the vocoder's matrix kernels (HifiGanGenerator's Conv1d products, hifigan_causal.py:191-244, 314-333).  Both
 * forms compute fp32 convolutions with fp32 accumulation and fp32 results; they differ in how an fp32 x fp32 product is formed:
 *   CONAN_ARITH_F32   every product on the f32-input MFMA (v_mfma_f32_*_f32);
 *   CONAN_ARITH_LIMB  every fp32 operand split exactly into three bf16 limbs (x = h + m + l), a product as the six largest of
 *                     the nine limb products on the bf16 MFMA (error of the dropped terms <= 2^-23 |x w|, below the rounding of
 *                     the fp32 accumulation; tests/test_gpu_arith.py holds both forms against float64: measured 0.8-1.1 x the
 *                     f32 kernels' rms error).  The split is exact for |x| >= 2^-109 (the third limb may be a bf16 denormal: the
 *                     bf16 MFMA honours them, measured at 2^-110); smaller operands lose low bits of the third limb, an absolute
 *                     error below 2^-130.
 *   CONAN_ARITH_AUTO  the library's default: LIMB wherever the context holds limb weights and a limb kernel exists for the launch
 *                     (ResBlock1 stages; upsamplers whose tiles fill the chip), F32 elsewhere (ResBlock2, decoder, Emformer).
 * The choice is a property of the stream-set, fixed at creation, reported by conan_streams_arith().  What is and is not invariant:
 * a step is bit-reproducible for a given stream-set, list of active slots and frame count (also pipelined against blocking steps, and
 * after a reset).  WHICH launches of an AUTO stream-set run a limb kernel depends on max_slots (the fused limb passes exist from 4
 * slots on for the C = 128 / 64 stages, always for C = 32; the C = 256 stage's grouped limb convs from 16) and, for the plain convs
 * (ups.2 / ups.3, the C = 256 stage), on whether the step's ACTIVE slots give the launch enough tiles to fill the chip - few active
 * slots in a large stream-set take the f32 kernels there, exactly like the split-K factor of the f32 conv kernel follows the active
 * count.  A stream's audio therefore differs between steps with different active sets by fp32 re-association / the form of a
 * product, within the tolerance every form is held to (tests/test_gpu_configs.py: a stream inside a batch of 64 against the same
 * stream alone, 2e-5; tests/test_gpu_round5.py: both forms against the reference goldens at the plan-switch sizes 3 .. 40).
 * CONAN_STREAMS_FIXED_PLAN (below) removes that dependence: the plan then follows max_slots only. */
typedef enum conan_arith { CONAN_ARITH_AUTO = 0, CONAN_ARITH_F32 = 1, CONAN_ARITH_LIMB = 2 } conan_arith;
/* Deployment choices of a stream-set (conan_streams_opts.flags):
 *   CONAN_STREAMS_FUSED_DECODER_BLOCKS  the decoder's conv blocks [LN -> k5 conv -> GELU] -> [1x1 conv + residual] as ONE operator each
 *                                       of the persistent decoder launch: 2 % less blocking latency at 64 streams, 1.4 % MORE time per
 *                                       pipelined step (every group member reads eight partial tensors) - for latency-bound serving;
 *   CONAN_STREAMS_SEPARATE_SMALL_STEPS  decoder steps of a single row tile (slots x frames <= 16: one to four streams) as ~38
 *                                       separate launches instead of the persistent launch in xcd mode (0.41 against
 *                                       0.28 ms per one-stream step) - an A/B switch, and a way out on parts whose workgroup -> XCD
 *                                       placement gives an XCD fewer than 8 workgroups of a 256-workgroup launch;
 *   (bit 4, CONAN_STREAMS_VOCODER_CHAIN of ABI 7, is retired: the one-launch vocoder step of small stream-sets measured slower
 *   than the launches it replaced at every size and left the library - tools/experiments/voc_chain; the bit is rejected)
 *   CONAN_STREAMS_FIXED_PLAN            (ABI 8) every launch-plan choice - kernel form (limb / f32), tile shape, split-K factor, merged
 *                                       stage tails, the decoder step's single-tile or multi-tile form - is made from max_slots
 *                                       and the step's frame count ONLY, never from the number of slots active in the step: slot k's
 *                                       audio is bit-identical whichever other slots step with it (the reference is batch-1 and
 *                                       deterministic per utterance, inference/Conan.py:109-113; a serving API can promise the
 *                                       same).  Costs throughput only when few slots of a large stream-set are active (the plan is
 *                                       the full set's); at full occupancy the plan IS the default's;
 *   CONAN_STREAMS_SHARED_DEVICE         (ABI 8) other PROCESSES drive this GPU too: never give a launch that waits inside itself
 *                                       (Emformer clusters) the whole chip.  Inside one process the library counts the live stream-sets
 *                                       per device itself (over all contexts) and takes the whole-chip shape only for blocking steps
 *                                       of the only one. */
enum { CONAN_STREAMS_FUSED_DECODER_BLOCKS = 1, CONAN_STREAMS_SEPARATE_SMALL_STEPS = 2, CONAN_STREAMS_FIXED_PLAN = 8, CONAN_STREAMS_SHARED_DEVICE = 16 };
typedef struct conan_streams_opts {
  int32_t abi_version;   /* must be CONAN_HIP_ABI_VERSION */
  int32_t arith;         /* conan_arith */
  int32_t flags;         /* bitwise or of CONAN_STREAMS_* (0: the defaults) */
  int32_t reserved0;     /* must be 0 */
  /* NULL in deployments.  Developer / test switches of the launch plan as "NAME=value;NAME=value" (A/B runs, the cross-checks of
   * tests/test_gpu_round3.py: e.g. "FENCED=1", "EMF_UNFUSED=1", "DEC_MEGA=0"); unknown names are CONAN_ERR_INVALID.  Since ABI 8 the
   * shipped library reads NO environment variable: a process's plan is a function of the arguments it passes, not of its
   * environment (until ABI 7 these were CONAN_* environment variables; `make DEV=1` builds still accept those).  The one runtime
   * knob that stays in the environment is HIP's own GPU_MAX_HW_QUEUES (Multi-GPU). */
  const char* dev_plan;
  int32_t reserved[2];   /* must be 0 */
} conan_streams_opts;
/* conan_streams_create with options (opts == NULL: all defaults, i.e. conan_streams_create). */
int conan_streams_create_opts(conan_ctx* ctx, int max_slots, int max_frames, int max_ref_frames, const conan_streams_opts* opts,
                              conan_streams** out);
/* The arithmetic this stream-set's vocoder launches use where both forms exist: CONAN_ARITH_F32 or CONAN_ARITH_LIMB
 * (AUTO resolved); negative conan_status on a null handle. */
int conan_streams_arith(const conan_streams* s);

/* Start of utterance for the given slots (replaces `state = None` inference/Conan.py:92 and the
 * zero left-padding of every causal conv).  which = bitmask of CONAN_MODEL_*. */
int conan_streams_reset(conan_streams* s, const int32_t* slots, int n, int which, void* stream);

/* Per-utterance style pass = the reference-mel-only part of Conan.forward
 * (modules/Conan/Conan.py:157-159 encode_spk_embed, :221-245 prosody tokens, K/V of the aligner).
 * ref_mel_dev[n][max_len][num_mels] (rows >= ref_len[i] ignored), ref_len host array. */
int conan_set_reference(conan_streams* s, const int32_t* slots, int n, const float* ref_mel_dev,
                        const int32_t* ref_len, int max_len, void* stream);

/* One streaming step of torchaudio Emformer.infer + proj + argmax
 * (inference/Conan.py:115-124).  chunk_dev[n][seg+rc][D]; out_dev[n][seg][D] (may be NULL);
 * logits_dev[n][seg][K] (may be NULL); codes_dev[n][seg] int32 (may be NULL). */
int conan_emformer_step(conan_streams* s, const int32_t* slots, int n, const float* chunk_dev,
                        float* out_dev, float* logits_dev, int32_t* codes_dev, void* stream);
/* The output heads of EmformerDistillModel applied to features that did NOT just come out of a step
 * (modules/Emformer/emformer.py:25 `proj`, :29-30 `proj1` / `proj2`; used by `inference()` :95-97 on the concatenated
 * per-chunk features): y_dev[rows][K_head] = x_dev[rows][D] @ W_head^T + b_head.  head = "proj", "proj1" or "proj2" as named
 * in the checkpoint.  CONAN_ERR_MISSING when the checkpoint holds no such head. */
int conan_emformer_project(conan_streams* s, const char* head, const float* x_dev, int rows, float* y_dev, void* stream);
/* Output width K_head of that head (nn.Linear(input_dim, K_head).out_features), 0 when the checkpoint has none. */
int conan_emformer_head_dim(conan_streams* s, const char* head);


/* `frames` new content codes per slot -> `frames` mel rows: Conan.forward(infer=True) restricted to
 * the new frames (modules/Conan/Conan.py:140-198 given the cached style pass).
 * codes_dev[n][frames] int32; mel_out_dev[n][frames][num_mels].  Optional taps (may be NULL):
 * uv_pred_dev[n][frames][2], f0_dev[n][frames], bins_dev[n][frames] int32, decoder_inp_dev[n][frames][H]. */
int conan_decoder_step(conan_streams* s, const int32_t* slots, int n, int frames, const int32_t* codes_dev,
                       float* mel_out_dev, float* uv_pred_dev, float* f0_dev, int32_t* bins_dev,
                       float* decoder_inp_dev, void* stream);

/* The remaining entries of the dict Conan.forward returns (modules/Conan/Conan.py:170-198; SURVEY.md §8b seam 3) as
 * optional device outputs of a decoder step; any pointer may be NULL.
 *   content_embed_proj[n][frames][H]   content_proj(content_embedding(content))              (Conan.py:140-142)
 *   attn[l][n][frames][max_tokens]     head-averaged cross-attention weights of ProsodyAligner layer l = 0, 1 over
 *                                      the slot's prosody tokens (prosody_util.py:119-161; the list in ret['attn']);
 *                                      max_tokens = ceil(max_ref_frames / 4) as given to conan_streams_create
 *                                      (conan_get_style reports it), entries past the slot's token count are 0 */
typedef struct conan_decoder_taps {
  float* uv_pred;               /* [n][frames][2] */
  float* f0_denorm_pred;        /* [n][frames] */
  int32_t* pitch_bins;          /* [n][frames] */
  float* decoder_inp;           /* [n][frames][H] */
  float* content_embed_proj;    /* [n][frames][H] */
  float* attn[2];               /* per aligner layer: [n][frames][max_tokens] */
} conan_decoder_taps;
int conan_decoder_step_taps(conan_streams* s, const int32_t* slots, int n, int frames, const int32_t* codes_dev,
                            float* mel_out_dev, const conan_decoder_taps* taps, void* stream);
/* Conan.forward(spk_embed=...) (modules/Conan/Conan.py:146-149): replace the cached global style vector of the slots
 * by a caller-provided one, style_dev[n][H].  The prosody tokens still come from the reference mel
 * (get_prosody(pitch_inp, ref, ...), Conan.py:166), so conan_set_reference must have run for the slots. */
int conan_set_style(conan_streams* s, const int32_t* slots, int n, const float* style_dev, void* stream);
/* VQ code indices of the slots' prosody tokens (VQEmbeddingEMA.encode argmin, prosody_util.py:34-46), cached by
 * conan_set_reference: ids_dev[n][max_tokens] int32 (entries past the slot's token count are -1), count_dev[n] int32
 * (may be NULL). */
int conan_get_prosody_ids(conan_streams* s, const int32_t* slots, int n, int32_t* ids_dev, int32_t* count_dev, void* stream);

/* style_embed of the slots' current reference (encode_spk_embed, Conan.py:200-219, cached by conan_set_reference):
 * style_dev[n][H].  max_tokens_out (may be NULL) receives the attn row width of conan_decoder_taps. */
int conan_get_style(conan_streams* s, const int32_t* slots, int n, float* style_dev, int32_t* max_tokens_out, void* stream);

/* `frames` new mel rows per slot -> frames*hop samples: HifiGanGenerator.forward restricted to the
 * new frames (hifigan_causal.py:314-333).  mel_dev[n][frames][num_mels]; wav_out_dev[n][frames*hop];
 * pre_tanh_dev optional. */
int conan_hifigan_step(conan_streams* s, const int32_t* slots, int n, int frames, const float* mel_dev,
                       float* wav_out_dev, float* pre_tanh_dev, void* stream);

/* conan_hifigan_step with optional taps of the generator's intermediate tensors (the forward hooks a reference
 * maintainer would register on conv_pre / ups[i], hifigan_causal.py:319-322), channel-last; any pointer may be NULL.
 *   conv_pre_act[n][frames][C0]             leaky_relu(conv_pre(mel), 0.1): the tensor ups[0] consumes
 *   ups[i][n][frames*rate_i][C_i]           output of ups[i] after the pixel shuffle (rate_i = prod(up_rates[0..i]))
 *   stage_out[i][n][frames*rate_i][C_i]     the MRF stage's output (see the field) */
typedef struct conan_hifigan_taps {
  float* conv_pre_act;
  float* ups[CONAN_MAX_UPS];
  float* stage_out[CONAN_MAX_UPS];   /* [n][frames*rate_i][C_i]: leaky_relu(mean_j resblocks[i*num_kernels + j](ups[i] output)), the tensor
                                        ups[i+1] / conv_post consumes (hifigan_causal.py:324-331) */
} conan_hifigan_taps;
int conan_hifigan_step_taps(conan_streams* s, const int32_t* slots, int n, int frames, const float* mel_dev,
                            float* wav_out_dev, float* pre_tanh_dev, const conan_hifigan_taps* taps, void* stream);

/* Fused chunk step = one iteration of the loop inference/Conan.py:95-156 for n slots:
 * mel_chunk_dev[n][seg+rc][D] -> codes_dev[n][seg] (int32), mel_out_dev[n][seg][num_mels],
 * wav_out_dev[n][seg*hop].  emit = number of leading frames that are real (<= seg). */
int conan_step(conan_streams* s, const int32_t* slots, int n, int emit, const float* mel_chunk_dev,
               int32_t* codes_dev, float* mel_out_dev, float* wav_out_dev, void* stream);

/* Pipelined conan_step: same arguments, same results bit for bit.  The Emformer + decoder of this call run on an
 * internal HIP stream and the vocoder on a second one, so the front-end of chunk t+1 overlaps the vocoder of chunk t
 * (in the reference the three stages of consecutive chunks are strictly serial, inference/Conan.py:95-156).
 * `stream` is the caller's stream: the inputs are taken as ready in its order at call time.  Outputs are complete in
 * the order of a stream that has passed conan_streams_join(); every buffer handed to a pipelined step must stay valid
 * until then.  All other stream-ordered entry points join pending pipelined work first, so the two styles can be mixed. */
int conan_step_async(conan_streams* s, const int32_t* slots, int n, int emit, const float* mel_chunk_dev,
                     int32_t* codes_dev, float* mel_out_dev, float* wav_out_dev, void* stream);
/* Make `stream` wait (device side, non-blocking for the host) for every step enqueued by conan_step_async. */
int conan_streams_join(conan_streams* s, void* stream);
/* Output fence for the NEXT conan_step_async call: its vocoder stage (the only stage that writes wav_out_dev) first waits,
 * device side, for everything enqueued on `fence_stream` up to now - e.g. the side stream on which a collective still
 * reads the buffer that the next step reuses.  Making the caller's `stream` wait for that instead would hold back the
 * step's Emformer and decoder stages too, which do not touch the buffer, and drain the pipeline (measured: 1.81 -> 2.6 ms
 * per step at 64 streams).  One-shot: cleared by the step that consumes it. */
int conan_streams_output_fence(conan_streams* s, void* fence_stream);
/* The same with an event the caller has ALREADY recorded (hipEvent_t as void*) - e.g. right behind the one collective that read the
 * buffer: the vocoder stage then waits for that collective only, not for whatever else has been enqueued on its stream since (a
 * later gather's join waits for the newest step's audio: fencing on the stream's tail would make step t's vocoder wait for the
 * side stream to have seen step t-1's completion - a cross-stream round trip per fence).  One-shot like the stream form. */
int conan_streams_output_fence_event(conan_streams* s, void* event);

/* Test hook for the bounded device-side waits.  Three kernels wait for other workgroups of their own launch (decoder step:
 * group / grid barriers; Emformer step: cluster exchange; first vocoder stage: partner flags); each such wait carries a 50 ms
 * budget, after which the waiter records a code and leaves, the launch finishes with meaningless results, and every later
 * stream-ordered entry point on the stream-set returns CONAN_ERR_HIP (the stream-set must then be destroyed; the context and
 * the process stay usable).  conan_streams_test_fault(s, kind) makes the NEXT launch of kind 1 (decoder megakernel), 2 (Emformer
 * clusters) or 3 (pair kernel) wait for an arrival that never comes, so that tests can walk that path on a healthy GPU. */
int conan_streams_test_fault(conan_streams* s, int kind);

/* Mel front-end (the step before the hot path): librosa_wav2spec as used by
 * StreamingVoiceConversion._wav_to_mel (utils/audio/__init__.py:37-84, inference/Conan.py:57-70), loud_norm off:
 * centred zero-padded STFT (periodic Hann) -> magnitude -> Slaney mel filterbank -> log10(max(eps, .)) -> clip.
 * wav_dev[n][samples] fp32 in [-1, 1]; mel_out_dev[n][frames][num_mels] with frames = 1 + samples / hop_size
 * (*frames_out, may be NULL).  fmin / fmax < 0 mean 0 / Nyquist.  Not re-entrant per context (shared workspace).
 * framing 1 / natural_log 1 / mag_eps 1e-9 select the earlier loop's front-end (inference/Conan_previous.py:100-121:
 * reflect padding of (fft_size - hop_size) / 2 samples per side, torch.stft(center=False), sqrt(re^2 + im^2 + 1e-9),
 * ln(max(eps, .)), frames = samples / hop_size; pass vmin / vmax = -/+ 1e30 for "no clip"). */
typedef struct conan_mel_cfg {
  int32_t fft_size, hop_size, win_length, num_mels, sample_rate;
  float fmin, fmax, eps, vmin, vmax;
  int32_t framing;      /* 0: centred frames, zero padding of fft_size / 2 (librosa.stft, pad_mode='constant');
                           1: reflect padding of (fft_size - hop_size) / 2, frames start at the padded signal's first sample */
  int32_t natural_log;  /* 0: log10, 1: ln */
  float mag_eps;        /* added under the magnitude's square root (0 for librosa's |X|) */
} conan_mel_cfg;
int conan_wav2mel(conan_ctx* ctx, const conan_mel_cfg* cfg, const float* wav_dev, int n, int samples,
                  float* mel_out_dev, int32_t* frames_out, void* stream);

/* Measurement hook (replaces the reference's Timer('hifigan') around the vocoder forward,
 * utils/commons/meters.py:21-42, tasks/tts/vocoder_infer/hifigan.py:28): between begin and end every
 * launch of the conv_mfma kernel family is bracketed by HIP events on its launch stream.  end() waits
 * for them and returns the summed kernel time, the algorithmic FLOPs (2*M*N*K of the convolutions,
 * unpadded) and the launch count. */
int conan_profile_begin(conan_streams* s);
int conan_profile_end(conan_streams* s, double* conv_ms, double* conv_flops, int64_t* conv_launches);
/* After conan_profile_end: the same totals per kernel template instantiation (index 0,1,...; returns 1 and fills the
 * outputs, 0 past the last one).  `name` is the kernel name as rocprofv3 prints it. */
int conan_profile_kernel(conan_streams* s, int index, char* name, int name_cap, double* ms, double* flops, int64_t* launches);

/* Enqueue one dispatch of the empty kernel cnk::profile_mark_kernel on `stream`: a marker that tools/summarize_pmc.py
 * uses to keep only the timed steps of a rocprofv3 counter-collection run (bench.py --marks). */
int conan_profile_mark(conan_streams* s, void* stream);
/* Step-time distribution of pipelined steps without touching their schedule: conan_step_clock(s, capacity > 0) makes every
 * following conan_step_async record one timing event on the internal vocoder stream when the step's audio is complete
 * (up to `capacity` steps; 0 switches it off); conan_step_clock_read waits for the last recorded step and writes the
 * intervals between consecutive completions in milliseconds to ms_out[cap], returning their count. */
int conan_step_clock(conan_streams* s, int capacity);
int conan_step_clock_read(conan_streams* s, double* ms_out, int cap);
/* Stage timeline of pipelined steps (developer diagnostics): with capacity > 0 every following conan_step_async records timing
 * events at the start and at the end of its Emformer, decoder and vocoder stage on their internal streams;
 * conan_step_timeline_read writes, per recorded step, the six times in milliseconds since the first step's first event
 * {emf start, emf end, dec start, dec end, voc start, voc end} to ms_out[cap_steps][6] and returns the number of steps. */
int conan_step_timeline(conan_streams* s, int capacity);
int conan_step_timeline_read(conan_streams* s, double* ms_out, int cap_steps);

/* Introspection for tests / INTEGRATION.md. */
int conan_hop_size(const conan_ctx* ctx);             /* prod(upsample_rates) */
int64_t conan_ctx_weight_bytes(const conan_ctx* ctx); /* packed device weight bytes */
int64_t conan_streams_state_bytes(const conan_streams* s);

#ifdef __cplusplus
}
#endif
#endif /* CONAN_HIP_H */
