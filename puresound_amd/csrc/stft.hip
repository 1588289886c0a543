// Conv-STFT encoder / iSTFT decoder helpers (ConvSTFT, lobe/encoder.py:358-456 and lobe/stft.py:103-125 of
// mcw519/PureSound).  The analysis kernels wsin/wcos are TRAINABLE parameters in every recipe, so the
// transform is a dense product against learned filters, not an FFT: the two dense products (analysis,
// synthesis with the Hermitian extension folded into the weight) run on ps_conv1x1_f32; this file holds the
// byte-moving steps around them.
#include "ps_common.h"

namespace ps {

// frames[n][k][t] = wav[n][t*hop + k]  (k < win, t < T): the strided framing the reference gets from conv1d's
// stride.  One thread per (k, 4 frames): 16-byte coalesced stores, gathered dword loads (L1/L2 resident).
__global__ __launch_bounds__(256) void frame_kernel(const float* __restrict__ wav, float* __restrict__ frames,
                                                    int L, int win, int hop, int T, int ldt) {
  const int n = blockIdx.z, k = blockIdx.y;
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t >= ldt) return;
  const float* x = wav + (size_t)n * L + k;
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (t + e < T) ? x[(size_t)(t + e) * hop] : 0.f;
  *reinterpret_cast<f32x4*>(frames + ((size_t)n * win + k) * ldt + t) = v;
}

__device__ __forceinline__ float act(float m, int a) {
  if (a == PS_ACT_RELU) return relu_keep_nan(m);
  if (a == PS_ACT_SIGMOID) return 1.f / (1.f + expf(-m));
  return m;
}

// Complex mask on a complex representation stored as channel halves [re ; im] (apply_tf_masks complex/complex,
// base_nn.py:56-61, _mul_c :97-112):  y_re = x_re*m_re - x_im*m_im,  y_im = x_re*m_im + x_im*m_re.
__global__ __launch_bounds__(256) void complex_mask_kernel(const float* __restrict__ feats,
                                                           const float* __restrict__ mask, float* __restrict__ out,
                                                           int half, int ldt, int mask_act) {
  const int n = blockIdx.z, c = blockIdx.y;
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t >= ldt) return;
  const size_t re = ((size_t)n * 2 * half + c) * ldt + t, im = re + (size_t)half * ldt;
  const f32x4 xr = *reinterpret_cast<const f32x4*>(feats + re), xi = *reinterpret_cast<const f32x4*>(feats + im);
  f32x4 mr = *reinterpret_cast<const f32x4*>(mask + re), mi = *reinterpret_cast<const f32x4*>(mask + im);
  f32x4 yr, yi;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a = act(mr[e], mask_act), b = act(mi[e], mask_act);
    yr[e] = xr[e] * a - xi[e] * b;
    yi[e] = xr[e] * b + xi[e] * a;
  }
  *reinterpret_cast<f32x4*>(out + re) = yr;
  *reinterpret_cast<f32x4*>(out + im) = yi;
}

// Complex mask in polar form (_apply_complex_mask_on_polar, base_nn.py:161-190), both operands as channel halves
// [re ; im]: magnitudes multiply (the mask's through tanh), phases add.
__global__ __launch_bounds__(256) void polar_mask_kernel(const float* __restrict__ feats, const float* __restrict__ mask,
                                                         float* __restrict__ out, int half, int ldt) {
  const int n = blockIdx.z, c = blockIdx.y;
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t >= ldt) return;
  const size_t re = ((size_t)n * 2 * half + c) * ldt + t, im = re + (size_t)half * ldt;
  const f32x4 xr = *reinterpret_cast<const f32x4*>(feats + re), xi = *reinterpret_cast<const f32x4*>(feats + im);
  const f32x4 mr = *reinterpret_cast<const f32x4*>(mask + re), mi = *reinterpret_cast<const f32x4*>(mask + im);
  f32x4 yr, yi;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float tf_mag = sqrtf(xr[e] * xr[e] + xi[e] * xi[e] + 1e-8f);
    const float tf_phase = atan2f(xi[e], xr[e]);
    float mask_mag = sqrtf(mr[e] * mr[e] + mi[e] * mi[e] + 1e-8f);
    const float mask_phase = atan2f(mi[e] / (mask_mag + 1e-8f), mr[e] / (mask_mag + 1e-8f));
    mask_mag = tanhf(mask_mag);
    const float est_mag = tf_mag * mask_mag, est_phase = tf_phase + mask_phase;
    yr[e] = est_mag * cosf(est_phase);
    yi[e] = est_mag * sinf(est_phase);
  }
  *reinterpret_cast<f32x4*>(out + re) = yr;
  *reinterpret_cast<f32x4*>(out + im) = yi;
}

// "MagPhase" output of the conv-STFT (lobe/encoder.py:384-389) from the analysis product's channel halves
// [re ; im] (im = -conv(x, wsin)): mags = re^2 + im^2, sqrt(mags + 1e-8) when the kernels are trainable;
// phase = atan2(im + 0.0, re).  Output rows [mags ; phase].
__global__ __launch_bounds__(256) void magphase_kernel(const float* __restrict__ spec, float* __restrict__ out, int half,
                                                       int ldt, int take_sqrt) {
  const int n = blockIdx.z, c = blockIdx.y;
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t >= ldt) return;
  const size_t re = ((size_t)n * 2 * half + c) * ldt + t, im = re + (size_t)half * ldt;
  const f32x4 xr = *reinterpret_cast<const f32x4*>(spec + re), xi = *reinterpret_cast<const f32x4*>(spec + im);
  f32x4 mg, ph;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float p = xr[e] * xr[e] + xi[e] * xi[e];
    mg[e] = take_sqrt ? sqrtf(p + 1e-8f) : p;
    ph[e] = atan2f(xi[e] + 0.0f, xr[e]);
  }
  *reinterpret_cast<f32x4*>(out + re) = mg;
  *reinterpret_cast<f32x4*>(out + im) = ph;
}

// Real mask on real features (apply_tf_masks real/real, base_nn.py:52-54) for encoders whose decoder is not fused with
// the mask product (the STFT front end of the tse_unet_tcn presets): y = x * act(m), rows of ldt frames.
__global__ __launch_bounds__(256) void real_mask_kernel(const float* __restrict__ feats, const float* __restrict__ mask,
                                                        float* __restrict__ out, int ldt, int mask_act) {
  const size_t row = blockIdx.y;
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t >= ldt) return;
  const f32x4 x = *reinterpret_cast<const f32x4*>(feats + row * ldt + t);
  const f32x4 m = *reinterpret_cast<const f32x4*>(mask + row * ldt + t);
  f32x4 y;
#pragma unroll
  for (int e = 0; e < 4; ++e) y[e] = x[e] * act(m[e], mask_act);
  *reinterpret_cast<f32x4*>(out + row * ldt + t) = y;
}

// Overlap-add of synthesis frames (ConvSTFT.inverse, lobe/encoder.py:432-454): every frame is multiplied by the
// window and divided by n_fft, overlapping samples are summed, then divided by the overlap-added squared window
// wherever that exceeds 1e-10 (the reference does this with a boolean-mask index, i.e. a host sync; here the
// window sum is recomputed per sample in the same loop).  One thread per output sample.
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames,
                                                        const float* __restrict__ window, float* __restrict__ out,
                                                        int n_fft, int hop, int T, int ldt, int out_mode) {
  const int n = blockIdx.y;
  const int Lout = (T - 1) * hop + n_fft;
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= Lout) return;
  int t_hi = g / hop;
  if (t_hi > T - 1) t_hi = T - 1;
  const int t_lo = (g - n_fft + 1 <= 0) ? 0 : (g - n_fft + 1 + hop - 1) / hop;
  const float inv = (float)n_fft;
  float acc = 0.f, wsum = 0.f;
  for (int t = t_lo; t <= t_hi; ++t) {
    const int s = g - t * hop;
    const float w = window[s];
    acc += frames[((size_t)n * n_fft + s) * ldt + t] * w / inv;
    wsum += w * w;
  }
  if (wsum > 1e-10f) acc = acc / wsum;
  if (out_mode == PS_OUT_CLAMP) acc = clamp1_keep_nan(acc);
  if (out_mode == PS_OUT_SIGMOID) acc = 1.f / (1.f + expf(-acc));
  out[(size_t)n * Lout + g] = acc;
}

static int launched(const char* who) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", who, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // namespace ps

using namespace ps;

extern "C" int ps_frame_f32(const float* wav, float* frames, int N, int L, int win, int hop, int T, int ldt,
                            void* stream) {
  if (!wav || !frames || N <= 0 || win <= 0 || hop <= 0 || L < win || T != (L - win) / hop + 1 || ldt < T ||
      ldt % kTileT != 0 || win > 65535) {
    set_error("ps_frame_f32: bad argument (N=%d L=%d win=%d hop=%d T=%d ldt=%d)", N, L, win, hop, T, ldt);
    return PS_E_INVALID;
  }
  LaunchTimer timer("frame", (hipStream_t)stream);
  hipLaunchKernelGGL(frame_kernel, dim3((ldt / 4 + 255) / 256, win, N), dim3(256), 0, (hipStream_t)stream, wav,
                     frames, L, win, hop, T, ldt);
  return launched("ps_frame_f32");
}

extern "C" int ps_complex_mask_f32(const float* feats, const float* mask, float* out, int N, int half, int ldt,
                                   int mask_act, void* stream) {
  if (!feats || !mask || !out || N <= 0 || half <= 0 || half > 65535 || ldt <= 0 || ldt % kTileT != 0 ||
      mask_act < PS_ACT_LINEAR || mask_act > PS_ACT_SIGMOID) {
    set_error("ps_complex_mask_f32: bad argument (N=%d half=%d ldt=%d act=%d)", N, half, ldt, mask_act);
    return PS_E_INVALID;
  }
  LaunchTimer timer("complex_mask", (hipStream_t)stream);
  hipLaunchKernelGGL(complex_mask_kernel, dim3((ldt / 4 + 255) / 256, half, N), dim3(256), 0, (hipStream_t)stream,
                     feats, mask, out, half, ldt, mask_act);
  return launched("ps_complex_mask_f32");
}

extern "C" int ps_polar_mask_f32(const float* feats, const float* mask, float* out, int N, int half, int ldt,
                                 void* stream) {
  if (!feats || !mask || !out || N <= 0 || half <= 0 || half > 65535 || ldt <= 0 || ldt % kTileT != 0) {
    set_error("ps_polar_mask_f32: bad argument (N=%d half=%d ldt=%d)", N, half, ldt);
    return PS_E_INVALID;
  }
  LaunchTimer timer("polar_mask", (hipStream_t)stream);
  hipLaunchKernelGGL(polar_mask_kernel, dim3((ldt / 4 + 255) / 256, half, N), dim3(256), 0, (hipStream_t)stream, feats,
                     mask, out, half, ldt);
  return launched("ps_polar_mask_f32");
}

extern "C" int ps_magphase_f32(const float* spec, float* out, int N, int half, int ldt, int take_sqrt, void* stream) {
  if (!spec || !out || N <= 0 || half <= 0 || half > 65535 || ldt <= 0 || ldt % kTileT != 0) {
    set_error("ps_magphase_f32: bad argument (N=%d half=%d ldt=%d)", N, half, ldt);
    return PS_E_INVALID;
  }
  LaunchTimer timer("magphase", (hipStream_t)stream);
  hipLaunchKernelGGL(magphase_kernel, dim3((ldt / 4 + 255) / 256, half, N), dim3(256), 0, (hipStream_t)stream, spec, out,
                     half, ldt, take_sqrt);
  return launched("ps_magphase_f32");
}

extern "C" int ps_real_mask_f32(const float* feats, const float* mask, float* out, int64_t rows, int ldt, int mask_act,
                                void* stream) {
  if (!feats || !mask || !out || rows <= 0 || rows > 65535 * 32768LL || ldt <= 0 || ldt % 4 || mask_act < PS_ACT_LINEAR ||
      mask_act > PS_ACT_SIGMOID || ((uintptr_t)feats & 15) || ((uintptr_t)mask & 15) || ((uintptr_t)out & 15)) {
    set_error("ps_real_mask_f32: bad argument (rows=%lld ldt=%d mask_act=%d)", (long long)rows, ldt, mask_act);
    return PS_E_INVALID;
  }
  LaunchTimer timer("real_mask", (hipStream_t)stream);
  const int64_t chunk = 65535;
  for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
    const int64_t nr = rows - r0 < chunk ? rows - r0 : chunk;
    hipLaunchKernelGGL(real_mask_kernel, dim3((ldt / 4 + 255) / 256, (unsigned)nr), dim3(256), 0, (hipStream_t)stream,
                       feats + r0 * ldt, mask + r0 * ldt, out + r0 * ldt, ldt, mask_act);
  }
  return launched("ps_real_mask_f32");
}

extern "C" int ps_istft_ola_f32(const float* frames, const float* window, float* out, int N, int n_fft, int hop,
                                int T, int ldt, int out_mode, void* stream) {
  if (!frames || !window || !out || N <= 0 || n_fft <= 0 || hop <= 0 || T <= 0 || ldt < T ||
      out_mode < PS_OUT_CLAMP || out_mode > PS_OUT_NONE) {
    set_error("ps_istft_ola_f32: bad argument (N=%d n_fft=%d hop=%d T=%d)", N, n_fft, hop, T);
    return PS_E_INVALID;
  }
  const int Lout = (T - 1) * hop + n_fft;
  LaunchTimer timer("istft_ola", (hipStream_t)stream);
  hipLaunchKernelGGL(istft_ola_kernel, dim3((Lout + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, frames,
                     window, out, n_fft, hop, T, ldt, out_mode);
  return launched("ps_istft_ola_f32");
}

// Magnitude lobe (lobe/trivial.py:21-59) on the [re rows; im rows] channel layout: y[h] = sqrt(re[h+d]^2 + im[h+d]^2 + 1e-8)
// (d = 1 drops the first bin), optionally log1p.
namespace ps {
__global__ __launch_bounds__(256) void magnitude_kernel(const float* __restrict__ x, float* __restrict__ y, int half,
                                                        int drop, int T, int ldt, int log1p_) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int h = blockIdx.y, n = blockIdx.z;
  if (t >= T) return;
  const int hout = half - drop;
  const float re = x[((size_t)n * 2 * half + h + drop) * ldt + t];
  const float im = x[((size_t)n * 2 * half + half + h + drop) * ldt + t];
  // kind 0: sqrt(|.|^2 + 1e-8); 1: log1p of it; 2: |.|^2 (power); 3: |.|^2 + 1e-8 (ConvMelSpectrogram, encoder.py:532)
  float m = re * re + im * im;
  if (log1p_ <= 1) m = sqrtf(m + 1e-8f);
  if (log1p_ == 1) m = log1pf(m);
  if (log1p_ == 3) m += 1e-8f;
  y[((size_t)n * hout + h) * ldt + t] = m;
}
}  // namespace ps

extern "C" int ps_magnitude_f32(const float* x, float* y, int N, int half, int drop_first, int log1p, int T, int ldt,
                                void* stream) {
  using namespace ps;
  if (!x || !y || N <= 0 || half <= 0 || T <= 0 || ldt < T || (drop_first != 0 && drop_first != 1) ||
      half - drop_first <= 0 || half > 65535 || N > 65535 || log1p < 0 || log1p > 3) {
    set_error("ps_magnitude_f32: bad argument (N=%d half=%d T=%d)", N, half, T);
    return PS_E_INVALID;
  }
  {
    LaunchTimer timer("magnitude", (hipStream_t)stream);
    hipLaunchKernelGGL(magnitude_kernel, dim3((T + 255) / 256, half - drop_first, N), dim3(256), 0, (hipStream_t)stream, x,
                       y, half, drop_first, T, ldt, log1p);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_magnitude_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// SpecAugment's masked fill (lobe/trivial.py:306-335 of the reference: torchaudio.functional.mask_along_axis picks the
// span on the host, this writes it): rows [lo, hi) of every utterance (axis 1) or frames [lo, hi) of every row (axis 2) of
// y [N][rows][ld] := x with that span replaced by `value`.
namespace ps {
__global__ __launch_bounds__(256) void fill_span_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int ld,
                                                        int axis, int lo, int hi, float value, long long total) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= total) return;
  const int t = (int)(i % ld), r = (int)((i / ld) % rows);
  f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
  if (axis == 1) {
    if (r >= lo && r < hi) v = f32x4{value, value, value, value};
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (t + e >= lo && t + e < hi) v[e] = value;
  }
  *reinterpret_cast<f32x4*>(y + i) = v;
}
}  // namespace ps

extern "C" int ps_fill_span_f32(const float* x, float* y, int N, int rows, int ld, int axis, int lo, int hi, float value,
                                void* stream) {
  if (!x || !y || N <= 0 || rows <= 0 || ld <= 0 || ld % 4 || (axis != 1 && axis != 2) || lo < 0 || hi < lo) {
    set_error("ps_fill_span_f32: bad argument (N=%d rows=%d ld=%d axis=%d span=[%d,%d))", N, rows, ld, axis, lo, hi);
    return PS_E_INVALID;
  }
  const long long total = (long long)N * rows * ld;
  LaunchTimer timer("fill_span", (hipStream_t)stream);
  hipLaunchKernelGGL(ps::fill_span_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                     rows, ld, axis, lo, hi, value, total);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_fill_span_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
