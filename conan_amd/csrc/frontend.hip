// Mel front-end on the GPU: librosa_wav2spec as called by StreamingVoiceConversion._wav_to_mel
// (utils/audio/__init__.py:37-84, inference/Conan.py:57-70) for waveforms already in device memory:
//   centred, zero-padded frames x periodic Hann -> |rfft| -> Slaney mel filterbank -> log10(max(eps, .)) -> clip.
// The DFT sums and the filterbank products are accumulated in f64 (VALU; 1.1 MMAC per frame, < 0.4 % of the vocoder's
// work per frame): what is left against librosa is librosa's own float32 FFT rounding.
#include <cmath>

#include "host_common.h"

namespace cnk {

struct FrameArgs { const float* wav; const float* win; float* out; int n, samples, frames, hop, n_fft, framing; };

__global__ __launch_bounds__(256) void stft_frames_kernel(const FrameArgs a) {
  const long long row = blockIdx.x;                       // i * frames + f
  const int i = (int)(row / a.frames), f = (int)(row - (long long)i * a.frames);
  const float* x = a.wav + (long long)i * a.samples;
  if (a.framing == 0) {
    const int s0 = f * a.hop - a.n_fft / 2;               // center=True, pad_mode='constant'
    for (int k = threadIdx.x; k < a.n_fft; k += blockDim.x) {
      const int s = s0 + k;
      a.out[row * a.n_fft + k] = (s >= 0 && s < a.samples) ? x[s] * a.win[k] : 0.f;
    }
  } else {
    // F.pad(y, ((n_fft - hop) / 2,) * 2, mode='reflect') then torch.stft(center=False) (inference/Conan_previous.py:112-116)
    const int s0 = f * a.hop - (a.n_fft - a.hop) / 2;
    for (int k = threadIdx.x; k < a.n_fft; k += blockDim.x) {
      int s = s0 + k;
      if (s < 0) s = -s;
      if (s >= a.samples) s = 2 * (a.samples - 1) - s;
      a.out[row * a.n_fft + k] = x[s] * a.win[k];
    }
  }
}

// |rfft| of the windowed frames: the DFT sums run in f64 (a 1024-term f32 sum loses the bins 5 decades below the frame
// peak, which the log then magnifies; MI355X's f64 vector rate makes the exact sum free at this size: 1 MMAC per frame).
// Block = 256 bins of one frame; the frame and the twiddle table tw[t] = (cos, sin)(2 pi t / N) sit in LDS as f64.
struct DftArgs { const float* fr; const double2* tw; float* mag; int n_fft, nb, cmag; float mag_eps; };

__global__ __launch_bounds__(256) void dft_mag_kernel(const DftArgs a) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  double* x = dsm;                                         // [n_fft]
  double2* tw = reinterpret_cast<double2*>(dsm + a.n_fft); // [n_fft]
  const long long row = blockIdx.x;
  for (int t = threadIdx.x; t < a.n_fft; t += blockDim.x) { x[t] = (double)a.fr[row * a.n_fft + t]; tw[t] = a.tw[t]; }
  __syncthreads();
  const int b = blockIdx.y * blockDim.x + threadIdx.x;
  if (b >= a.cmag) return;
  float v = 0.f;
  if (b < a.nb) {
    double re = 0.0, im = 0.0;
    const int mask = a.n_fft - 1;                          // n_fft is a power of two (checked on the host)
    int idx = 0;
    for (int t = 0; t < a.n_fft; ++t) {
      const double2 w = tw[idx];
      re = fma(x[t], w.x, re); im = fma(x[t], w.y, im);
      idx = (idx + b) & mask;
    }
    v = (float)sqrt(re * re + im * im + (double)a.mag_eps);
  }
  a.mag[row * a.cmag + b] = v;
}

// mel = log10(max(eps, filterbank . |X|)) clipped; one thread per (frame, mel bin), f64 sum over the bins of its triangle
struct MelArgs { const float* mag; const float* fb; const int* lo; const int* hi; float* y; long long rows; int nmel, cmag; float eps, vmin, vmax; int natural_log; };

__global__ __launch_bounds__(256) void mel_log_kernel(const MelArgs a) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.rows * a.nmel) return;
  const long long row = e / a.nmel;
  const int m = (int)(e - row * a.nmel);
  const float* mg = a.mag + row * a.cmag;
  const float* w = a.fb + (long long)m * a.cmag;
  double s = 0.0;
  for (int b = a.lo[m]; b < a.hi[m]; ++b) s = fma((double)w[b], (double)mg[b], s);
  const float c = fmaxf(a.eps, (float)s);
  const float v = a.natural_log ? logf(c) : log10f(c);
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

}  // namespace

int conan_mel_frames(const conan_mel_cfg& m, int samples) {
  if (m.framing == 0) return 1 + samples / m.hop_size;
  const int padded = samples + 2 * ((m.fft_size - m.hop_size) / 2);
  return padded < m.fft_size ? 0 : (padded - m.fft_size) / m.hop_size + 1;
}

void conan_ctx::wav2mel(const conan_mel_cfg& m, const float* wav, int n, int samples, float* mel_out, hipStream_t st) {
  using ch::Error;
  if (m.fft_size < 64 || (m.fft_size & (m.fft_size - 1)) || m.fft_size > 2048 || m.hop_size < 1 || m.win_length < 1 || m.win_length > m.fft_size ||
      m.num_mels < 1 || m.num_mels > 512 || m.sample_rate < 1 || !(m.eps > 0.f))
    throw Error(CONAN_ERR_INVALID, "mel front-end configuration");
  if (n < 1 || samples < 1) throw Error(CONAN_ERR_INVALID, "wav2mel batch / samples");
  if (m.framing < 0 || m.framing > 1 || !(m.mag_eps >= 0.f)) throw Error(CONAN_ERR_INVALID, "mel front-end framing / mag_eps");
  // reflect padding needs pad < samples (torch raises otherwise); frames = (samples + 2 pad - n_fft) / hop + 1
  if (m.framing == 1 && ((m.fft_size - m.hop_size) / 2 >= samples || m.hop_size > m.fft_size)) throw Error(CONAN_ERR_INVALID, "wav2mel: reflect padding longer than the signal");
  const int N = m.fft_size, NB = N / 2 + 1, CM = (NB + 3) & ~3;
  const double fmin = m.fmin < 0 ? 0.0 : m.fmin, fmax = m.fmax < 0 ? m.sample_rate / 2.0 : m.fmax;
  char key[160];
  snprintf(key, sizeof(key), "fe.%d.%d.%d.%d.%g.%g", N, m.win_length, m.num_mels, m.sample_rate, fmin, fmax);
  const std::string k(key);
  if (!vecs.count(k + ".tw")) {
    // periodic Hann (scipy.signal.get_window('hann', win_length, fftbins=True)), centred in the FFT frame (pad_center)
    std::vector<float> win(N, 0.f);
    const int lp = (N - m.win_length) / 2;
    const double PI = 3.14159265358979323846;
    for (int i = 0; i < m.win_length; ++i) win[lp + i] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * i / m.win_length));
    vecs[k + ".win"] = upload(win);
    // twiddles (cos, sin)(2 pi t / N) as f64 pairs (the sign of the sine is irrelevant for |.|)
    std::vector<float> tw((size_t)4 * N);
    for (int t = 0; t < N; ++t) {
      const double c = std::cos(2.0 * PI * t / N), sn = std::sin(2.0 * PI * t / N);
      memcpy(&tw[(size_t)4 * t], &c, 8); memcpy(&tw[(size_t)4 * t + 2], &sn, 8);
    }
    vecs[k + ".tw"] = upload(tw);
    // librosa.filters.mel(htk=False, norm='slaney', dtype=float32), plus each triangle's bin range
    std::vector<double> mel_f(m.num_mels + 2);
    const double m0 = hz_to_mel(fmin), m1 = hz_to_mel(fmax);
    for (int i = 0; i < m.num_mels + 2; ++i) mel_f[i] = mel_to_hz(m0 + (m1 - m0) * i / (m.num_mels + 1));
    std::vector<float> B((size_t)m.num_mels * CM, 0.f), range((size_t)2 * m.num_mels);
    for (int i = 0; i < m.num_mels; ++i) {
      const float enorm = (float)(2.0 / (mel_f[i + 2] - mel_f[i]));
      int lo = NB, hi = 0;
      for (int b = 0; b < NB; ++b) {
        const double f = (m.sample_rate / 2.0) * b / (NB - 1);
        const double lower = (f - mel_f[i]) / (mel_f[i + 1] - mel_f[i]), upper = (mel_f[i + 2] - f) / (mel_f[i + 2] - mel_f[i + 1]);
        float w = (float)std::max(0.0, std::min(lower, upper));
        w *= enorm;
        B[(size_t)i * CM + b] = w;
        if (w != 0.f) { lo = std::min(lo, b); hi = std::max(hi, b + 1); }
      }
      if (hi <= lo) { lo = 0; hi = 0; }
      const int32_t l32 = lo, h32 = hi;
      memcpy(&range[i], &l32, 4); memcpy(&range[(size_t)m.num_mels + i], &h32, 4);
    }
    vecs[k + ".fb"] = upload(B);
    vecs[k + ".range"] = upload(range);
  }
  const int frames = conan_mel_frames(m, samples);
  if (frames < 1) throw Error(CONAN_ERR_INVALID, "wav2mel: signal shorter than one frame");
  const long long rows = (long long)n * frames;
  if (rows > (1ll << 19)) throw Error(CONAN_ERR_INVALID, "wav2mel: more than 2^19 frames in one call");
  // workspace: frames [rows][N] | magnitude [rows][CM]
  const size_t need = (size_t)rows * ((size_t)N + CM);
  if (need > fe_ws_floats) {      // grow: the previous block is released once the stream has drained
    if (fe_ws) {
      HIP_CHECK(hipStreamSynchronize(st));
      for (size_t i = 0; i < allocs.size(); ++i) if (allocs[i] == fe_ws) { allocs.erase(allocs.begin() + i); break; }
      HIP_CHECK(hipFree(fe_ws));
    }
    fe_ws = dev_alloc(need, false); fe_ws_floats = need;
  }
  float* fr = fe_ws; float* mag = fr + (size_t)rows * N;
  { cnk::FrameArgs a{wav, vec(k + ".win"), fr, n, samples, frames, m.hop_size, N, m.framing}; hipLaunchKernelGGL(cnk::stft_frames_kernel, dim3((unsigned)rows), dim3(256), 0, st, a); }
  { cnk::DftArgs a{fr, reinterpret_cast<const double2*>(vec(k + ".tw")), mag, N, NB, CM, m.mag_eps};
    hipLaunchKernelGGL(cnk::dft_mag_kernel, dim3((unsigned)rows, (unsigned)((CM + 255) / 256)), dim3(256), (size_t)N * 24, st, a); }
  { const float* rg = vec(k + ".range");
    cnk::MelArgs a{mag, vec(k + ".fb"), reinterpret_cast<const int*>(rg), reinterpret_cast<const int*>(rg) + m.num_mels, mel_out, rows, m.num_mels, CM, m.eps, m.vmin, m.vmax, m.natural_log};
    const long long total = rows * m.num_mels;
    hipLaunchKernelGGL(cnk::mel_log_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a); }
}
