// Mel front-end on the GPU: librosa_wav2spec as called by StreamingVoiceConversion._wav_to_mel
// (utils/audio/__init__.py:37-84, inference/Conan.py:57-70) for waveforms already in device memory:
//   centred, zero-padded frames x periodic Hann -> |rfft| -> Slaney mel filterbank -> log10(max(eps, .)) -> clip.
// The DFT is a GEMM against a precomputed [n_fft x 2*(n_fft/2+1)] cos|sin matrix and the filterbank a second GEMM, both
// through conv_mfma (k = 1) in exact fp32; 1.1 MMAC per frame, < 0.4 % of the vocoder's work per frame.
#include <cmath>

#include "host_common.h"

namespace cnk {

struct FrameArgs { const float* wav; const float* win; float* out; int n, samples, frames, hop, n_fft; };

__global__ __launch_bounds__(256) void stft_frames_kernel(const FrameArgs a) {
  const long long row = blockIdx.x;                       // i * frames + f
  const int i = (int)(row / a.frames), f = (int)(row - (long long)i * a.frames);
  const float* x = a.wav + (long long)i * a.samples;
  const int s0 = f * a.hop - a.n_fft / 2;                 // center=True, pad_mode='constant'
  for (int k = threadIdx.x; k < a.n_fft; k += blockDim.x) {
    const int s = s0 + k;
    a.out[row * a.n_fft + k] = (s >= 0 && s < a.samples) ? x[s] * a.win[k] : 0.f;
  }
}

struct MagArgs { const float* y; float* mag; long long rows; int nb, nbp, cmag; };   // y[row][2*nbp], mag[row][cmag]

__global__ __launch_bounds__(256) void stft_mag_kernel(const MagArgs a) {
  const long long row = blockIdx.x;
  const float* y = a.y + row * 2 * a.nbp;
  for (int b = threadIdx.x; b < a.cmag; b += blockDim.x) {
    float v = 0.f;
    if (b < a.nb) { const float re = y[b], im = y[a.nbp + b]; v = sqrtf(re * re + im * im); }
    a.mag[row * a.cmag + b] = v;
  }
}

struct LogMelArgs { const float* x; float* y; long long total; float eps, vmin, vmax; };

__global__ __launch_bounds__(256) void logmel_kernel(const LogMelArgs a) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.total) return;
  const float v = log10f(fmaxf(a.eps, a.x[e]));
  a.y[e] = fminf(fmaxf(v, a.vmin), a.vmax);
}

}  // namespace cnk

namespace {

// librosa.core.convert (htk=False): Slaney's Auditory Toolbox mel scale
double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

cnk::ConvArgs linear_args(const ch::PackedConv& pc, float* x, int xC, float* y, int rows) {
  cnk::ConvArgs a; memset(&a, 0, sizeof(a));
  a.x = ch::lin_ref(x, rows, xC);
  a.y = ch::lin_ref(y, rows, pc.Cout);
  a.res = ch::null_ref(); a.m1 = ch::null_ref(); a.m2 = ch::null_ref();
  a.w = pc.w; a.bias = pc.bias;
  a.Cin = pc.Cin; a.Cin_pad = pc.Cin_pad; a.Cin_alloc = pc.Cin_alloc; a.Cout = pc.Cout; a.Cout_pad = pc.Cout_pad;
  a.ktaps = 1; a.dil = 1; a.pad_left = 0; a.T = rows; a.n = 1;
  a.in_act = cnk::ACT_NONE; a.out_act = cnk::ACT_NONE; a.out_scale = 1.f; a.shuffle_r = 1;
  return a;
}

void run_linear(conan_ctx* ctx, const cnk::ConvArgs& a, hipStream_t st) {
  cnk::ConvGroup g; memset(&g, 0, sizeof(g));
  g.p[0] = a; g.ksplit = 1;
  const long long tiles64 = (long long)((a.T + 63) / 64) * ((a.Cout + 63) / 64);
  cnk::launch_conv(g, 1, tiles64 >= ctx->num_cu ? cnk::CFG_64x64 : cnk::CFG_32x32_K4, st, ctx->num_cu);
}

}  // namespace

void conan_ctx::wav2mel(const conan_mel_cfg& m, const float* wav, int n, int samples, float* mel_out, hipStream_t st) {
  using ch::Error;
  if (m.fft_size < 64 || (m.fft_size & 3) || m.fft_size > 4096 || m.hop_size < 1 || m.win_length < 1 || m.win_length > m.fft_size ||
      m.num_mels < 1 || m.num_mels > 512 || m.sample_rate < 1 || !(m.eps > 0.f))
    throw Error(CONAN_ERR_INVALID, "mel front-end configuration");
  if (n < 1 || samples < 1) throw Error(CONAN_ERR_INVALID, "wav2mel batch / samples");
  const int N = m.fft_size, NB = N / 2 + 1, NBP = (NB + 3) & ~3, CM = NBP;
  const double fmin = m.fmin < 0 ? 0.0 : m.fmin, fmax = m.fmax < 0 ? m.sample_rate / 2.0 : m.fmax;
  char key[160];
  snprintf(key, sizeof(key), "fe.%d.%d.%d.%d.%g.%g", N, m.win_length, m.num_mels, m.sample_rate, fmin, fmax);
  const std::string k(key);
  if (!convs.count(k + ".dft")) {
    // periodic Hann (scipy.signal.get_window('hann', win_length, fftbins=True)), centred in the FFT frame (pad_center)
    std::vector<float> win(N, 0.f);
    const int lp = (N - m.win_length) / 2;
    const double PI = 3.14159265358979323846;
    for (int i = 0; i < m.win_length; ++i) win[lp + i] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * i / m.win_length));
    vecs[k + ".win"] = upload(win);
    // DFT as a Linear weight [out = 2*NBP][in = N]: rows 0..NB-1 cos, rows NBP..NBP+NB-1 sin (sign is irrelevant for |.|)
    std::vector<float> W((size_t)2 * NBP * N, 0.f);
    for (int b = 0; b < NB; ++b)
      for (int t = 0; t < N; ++t) {
        const double ph = 2.0 * PI * (double)(((long long)b * t) % N) / N;
        W[(size_t)b * N + t] = (float)std::cos(ph);
        W[(size_t)(NBP + b) * N + t] = (float)std::sin(ph);
      }
    pack_conv(k + ".dft", W, nullptr, 2 * NBP, N, 1);
    // librosa.filters.mel(htk=False, norm='slaney', dtype=float32)
    std::vector<double> mel_f(m.num_mels + 2);
    const double m0 = hz_to_mel(fmin), m1 = hz_to_mel(fmax);
    for (int i = 0; i < m.num_mels + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (m.num_mels + 1));
    std::vector<float> B((size_t)m.num_mels * CM, 0.f);
    for (int i = 0; i < m.num_mels; ++i) {
      const float enorm = (float)(2.0 / (mel_f[i + 2] - mel_f[i]));
      for (int b = 0; b < NB; ++b) {
        const double f = (m.sample_rate / 2.0) * b / (NB - 1);
        const double lower = (f - mel_f[i]) / (mel_f[i + 1] - mel_f[i]), upper = (mel_f[i + 2] - f) / (mel_f[i + 2] - mel_f[i + 1]);
        float w = (float)std::max(0.0, std::min(lower, upper));
        w *= enorm;
        B[(size_t)i * CM + b] = w;
      }
    }
    pack_conv(k + ".mel", B, nullptr, m.num_mels, CM, 1);
  }
  const int frames = 1 + samples / m.hop_size;
  const long long rows = (long long)n * frames;
  if (rows > (1ll << 19)) throw Error(CONAN_ERR_INVALID, "wav2mel: more than 2^19 frames in one call (32-bit tile offsets)");
  // workspace: frames [rows][N] | spectrum [rows][2*NBP] | magnitude [rows][CM] | mel [rows][num_mels]
  const size_t need = (size_t)rows * ((size_t)N + 2 * NBP + CM + m.num_mels);
  if (need > fe_ws_floats) {      // grow: the previous block is released once the stream has drained
    if (fe_ws) {
      HIP_CHECK(hipStreamSynchronize(st));
      for (size_t i = 0; i < allocs.size(); ++i) if (allocs[i] == fe_ws) { allocs.erase(allocs.begin() + i); break; }
      HIP_CHECK(hipFree(fe_ws));
    }
    fe_ws = dev_alloc(need, false); fe_ws_floats = need;
  }
  float* fr = fe_ws; float* spec = fr + (size_t)rows * N; float* mag = spec + (size_t)rows * 2 * NBP; float* melraw = mag + (size_t)rows * CM;
  { cnk::FrameArgs a{wav, vec(k + ".win"), fr, n, samples, frames, m.hop_size, N}; hipLaunchKernelGGL(cnk::stft_frames_kernel, dim3((unsigned)rows), dim3(256), 0, st, a); }
  run_linear(this, linear_args(conv(k + ".dft"), fr, N, spec, (int)rows), st);
  { cnk::MagArgs a{spec, mag, rows, NB, NBP, CM}; hipLaunchKernelGGL(cnk::stft_mag_kernel, dim3((unsigned)rows), dim3(256), 0, st, a); }
  run_linear(this, linear_args(conv(k + ".mel"), mag, CM, melraw, (int)rows), st);
  { const long long total = rows * m.num_mels; cnk::LogMelArgs a{melraw, mel_out, total, m.eps, m.vmin, m.vmax};
    hipLaunchKernelGGL(cnk::logmel_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a); }
}
