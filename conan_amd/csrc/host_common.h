// Host-side runtime of libconan_hip.so: context (weights), streams (per-slot state), launch plans.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <atomic>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/conan_hip.h"
#include "kernels.h"

namespace ch {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

void hip_check(hipError_t e, const char* what);

using cnk::dev_getenv;

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  int64_t numel() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};

// A conv / linear weight repacked for conv_mfma: [Cout_pad/64][taps][Cin_alloc/4][64][4], zero padded: per group of
// 64 output columns the K rows are 1 KiB apart whatever Cout is (a [K][Cout_pad] layout makes a W tile touch one
// page per row when Cout is large).
struct PackedConv {
  float* w = nullptr;
  float* bias = nullptr;   // [Cout_pad] (zeros when the layer has no bias)
  int Cin = 0, Cin_pad = 0, Cin_alloc = 0, Cout = 0, Cout_pad = 0, k = 1;
  int shuffle_r = 1;
  float* wl = nullptr;     // bf16 limbs for conv_limb.hip (vocoder upsamplers / wide ResBlock convs; bits behind a float pointer)
  float* wf = nullptr;     // fragment-major copy for rowconv.hip (decoder-step layers only), Cout padded to wf_cout_pad
  int wf_cout_pad = 0;
};

struct Ring {
  float* base = nullptr;
  int L = 0, C = 0, rate = 1;
  long long slot_stride = 0;
  cnk::TRef ref(int off = 0) const {
    cnk::TRef r; r.base = base; r.slot_stride = slot_stride; r.C = C; r.lmask = L - 1; r.rate = rate; r.off = off; r.mode = 0; r.pad_ = 0;
    return r;
  }
  long long floats_per_slot() const { return slot_stride; }
};

struct Lin {   // batch-indexed linear buffer [n][rows][C]
  float* base = nullptr;
  int rows = 0, C = 0;
  cnk::TRef ref(int off = 0) const {
    cnk::TRef r; r.base = base; r.slot_stride = (long long)rows * C; r.C = C; r.lmask = 0; r.rate = 0; r.off = off; r.mode = 1; r.pad_ = 0;
    return r;
  }
};

inline cnk::TRef lin_ref(float* p, int rows, int C, int off = 0) {
  Lin l; l.base = p; l.rows = rows; l.C = C; return l.ref(off);
}
inline cnk::TRef null_ref() { cnk::TRef r; memset(&r, 0, sizeof(r)); r.mode = 1; return r; }

inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace ch

int conan_mel_frames(const conan_mel_cfg& m, int samples);   // frontend.hip: frames conan_wav2mel writes per waveform

// Stream-sets that exist right now on a device (conan_streams_create .. _destroy), over ALL contexts of this process: launches that
// wait inside themselves (Emformer clusters on whole CUs, the decoder's resident groups, the f32 pair kernel) can only be given
// the whole chip when nothing else of the process can overlap them on that device - two contexts on one GPU each counted
// themselves "alone" while the count was per context.  Other PROCESSES on the device are not visible here:
// CONAN_STREAMS_SHARED_DEVICE says so.
inline std::atomic<int>& device_live_streams(int device) {
  static std::atomic<int> counts[64];
  return counts[device & 63];
}

struct conan_ctx {
  int device = 0;
  conan_cfg cfg;
  bool finalized = false;
  std::map<std::string, ch::HostTensor> raw;
  std::map<std::string, ch::PackedConv> convs;
  std::map<std::string, float*> vecs;
  std::map<std::string, float> scalars;
  std::vector<void*> allocs;
  int64_t weight_bytes = 0;
  int hop = 1;
  int num_cu = 256;
  bool has_limb_weights = false;   // finalize packed bf16-limb copies of the vocoder's conv weights (resblock_limb.hip / conv_limb.hip)

  float* dev_alloc(size_t floats, bool zero = true);
  float* upload(const std::vector<float>& v);
  const ch::HostTensor& get(const std::string& key) const;
  bool has(const std::string& key) const { return raw.count(key) != 0; }
  const ch::PackedConv& conv(const std::string& name) const;
  float* vec(const std::string& name) const;
  float* vec_or_null(const std::string& name) const;
  void pack_conv(const std::string& name, const std::vector<float>& W, const float* bias, int Cout, int Cin, int k,
                 int shuffle_r = 1);
  void add_rowconv_weights(const std::string& name, const std::vector<float>& W);   // second, fragment-major copy
  void pack_from_keys(const std::string& name, const std::string& wkey, const std::string& bkey, int shuffle_r = 1);
  void pack_weightnorm(const std::string& name, const std::string& prefix, int shuffle_r = 1, bool rowconv = false);
  void pack_fragments(const std::string& name, const std::string& prefix);   // resblock_fused.hip operand layout
  void fold_weightnorm(const std::string& prefix, std::vector<float>& W, std::vector<float>& bias, int& Cout, int& Cin, int& k) const;
  void upload_vec(const std::string& name, const std::string& key);
  void finalize_hifigan();
  void finalize_conan();
  void finalize_emformer();
  // mel front-end (frontend.hip): tables are built on first use per configuration; one workspace, regrown when a call needs more
  float* fe_ws = nullptr; size_t fe_ws_floats = 0;
  void wav2mel(const conan_mel_cfg& m, const float* wav, int n, int samples, float* mel_out, hipStream_t st);
  ~conan_ctx();
};
