// Attentive statistics pooling tail (AttentiveStatisticsPooling, lobe/pooling.py:109-126 of mcw519/PureSound):
// softmax over frames of the attention logits, attention-weighted mean and standard deviation of x.
// One workgroup per (utterance, channel) row; the row (<= 16 KiB at T = 3999) is streamed three times
// (max, normaliser + mean, variance) and stays in L1/L2 after the first pass.  Two-pass variance as in the
// reference (sum a*(x-mean)^2), not E[x^2]-mean^2.
#include "ps_common.h"

namespace ps {

__device__ __forceinline__ float block_max(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  v = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return v;
}

// `lens` (optional, [N] relative lengths): frame t of utterance n takes part iff (float)t < lens[n] * (float)T, the
// fp32 comparison of length_to_mask(lengths * L) (lobe/pooling.py:9-50, 100-107 of the reference); masked frames get
// weight exp(-inf) = 0, i.e. they are left out of every sum.  No valid frame at all gives NaN as the reference does.
__global__ __launch_bounds__(256) void attn_stats_pool_kernel(const float* __restrict__ logits,
                                                              const float* __restrict__ x,
                                                              const float* __restrict__ lens, float* __restrict__ out,
                                                              int C, int T_all, int ldt, float eps) {
  __shared__ float red[4];
  const int c = blockIdx.x, n = blockIdx.y;
  int T = T_all;
  if (lens) {  // valid frames are a prefix: count them with the reference's comparison
    const float lim = lens[n] * (float)T_all;
    int lo = 0, hi = T_all;  // first t with !((float)t < lim)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((float)mid < lim) lo = mid + 1; else hi = mid;
    }
    T = lo;
  }
  const float* lr = logits + ((size_t)n * C + c) * ldt;
  const float* xr = x + ((size_t)n * C + c) * ldt;
  float m = -INFINITY;
  for (int t = threadIdx.x; t < T; t += 256) m = fmaxf(m, lr[t]);
  m = block_max(m, red);
  float s = 0.f, s1 = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    const float e = expf(lr[t] - m);
    s += e;
    s1 += e * xr[t];
  }
  s = block_sum(s, red);
  s1 = block_sum(s1, red);
  const float mean = s1 / s;
  float s2 = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) {
    const float e = expf(lr[t] - m);
    const float d = xr[t] - mean;
    s2 += e * d * d;
  }
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    out[(size_t)n * 2 * C + c] = mean;
    out[(size_t)n * 2 * C + C + c] = sqrtf((s2 / s) < eps ? eps : (s2 / s))  /* clamp(eps), NaN kept */;
  }
}

// Rows of up to 4096 frames (the 4 s utterances of the benchmarks: 3999): the row of logits and the row of x are read
// ONCE, as four 16-byte loads each per thread, and the three passes above run on registers -- the same sums in the same
// per-thread order, so the results are those of the kernel above bit for bit.
__global__ __launch_bounds__(256) void attn_stats_pool_reg_kernel(const float* __restrict__ logits,
                                                                  const float* __restrict__ x,
                                                                  const float* __restrict__ lens, float* __restrict__ out,
                                                                  int C, int T_all, int ldt, float eps) {
  __shared__ float red[4];
  const int c = blockIdx.x, n = blockIdx.y;
  int T = T_all;
  if (lens) {
    const float lim = lens[n] * (float)T_all;
    int lo = 0, hi = T_all;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((float)mid < lim) lo = mid + 1; else hi = mid;
    }
    T = lo;
  }
  const float* lr = logits + ((size_t)n * C + c) * ldt;
  const float* xr = x + ((size_t)n * C + c) * ldt;
  // element (i, e) of this thread: frame t = 1024 i + 4 threadIdx.x + e  (the looping kernel walks t = threadIdx.x + 256 k:
  // other frames per thread, but every block-wide sum is the same set of terms; fp32 sums differ in the last bits only)
  f32x4 lv[4], xv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t0 = 1024 * i + 4 * threadIdx.x;
    const bool in = t0 < ldt;  // (rows are padded to a multiple of 4 frames: a 16-byte piece is inside or outside)
    lv[i] = in ? *reinterpret_cast<const f32x4*>(lr + t0) : f32x4{0.f, 0.f, 0.f, 0.f};
    xv[i] = in ? *reinterpret_cast<const f32x4*>(xr + t0) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (1024 * i + 4 * (int)threadIdx.x + e < T) m = fmaxf(m, lv[i][e]);
  m = block_max(m, red);
  float s = 0.f, s1 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool on = 1024 * i + 4 * (int)threadIdx.x + e < T;
      const float w = on ? expf(lv[i][e] - m) : 0.f;
      lv[i][e] = w;  // (the weight replaces the logit: the third pass needs only it)
      s += w;
      s1 += on ? w * xv[i][e] : 0.f;
    }
  s = block_sum(s, red);
  s1 = block_sum(s1, red);
  const float mean = s1 / s;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool on = 1024 * i + 4 * (int)threadIdx.x + e < T;
      const float d = xv[i][e] - mean;
      s2 += on ? lv[i][e] * d * d : 0.f;
    }
  s2 = block_sum(s2, red);
  if (threadIdx.x == 0) {
    out[(size_t)n * 2 * C + c] = mean;
    out[(size_t)n * 2 * C + C + c] = sqrtf((s2 / s) < eps ? eps : (s2 / s));
  }
}

// The attention map itself (forward(..., return_weight=True), lobe/pooling.py:109-113): softmax over the valid frames of
// each (utterance, channel) row, 0 on masked frames; the frames beyond T_all of the padded output row are cleared.
__global__ __launch_bounds__(256) void attn_weights_kernel(const float* __restrict__ logits,
                                                           const float* __restrict__ lens, float* __restrict__ out, int C,
                                                           int T_all, int ldt) {
  __shared__ float red[4];
  const int c = blockIdx.x, n = blockIdx.y;
  int T = T_all;
  if (lens) {
    const float lim = lens[n] * (float)T_all;
    int lo = 0, hi = T_all;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((float)mid < lim) lo = mid + 1; else hi = mid;
    }
    T = lo;
  }
  const float* lr = logits + ((size_t)n * C + c) * ldt;
  float* o = out + ((size_t)n * C + c) * ldt;
  float m = -INFINITY;
  for (int t = threadIdx.x; t < T; t += 256) m = fmaxf(m, lr[t]);
  m = block_max(m, red);
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += 256) s += expf(lr[t] - m);
  s = block_sum(s, red);
  // (no valid frame: softmax of all -inf is NaN in the reference; 0 / 0 here)
  const float inv = 1.f / s;
  for (int t = threadIdx.x; t < ldt; t += 256) o[t] = t < T ? expf(lr[t] - m) * inv : (t < T_all && T == 0 ? NAN : 0.f);
}

}  // namespace ps

extern "C" int ps_attn_weights_f32(const float* logits, const float* lengths, float* out, int N, int C, int T, int ldt,
                                   void* stream) {
  using namespace ps;
  if (!logits || !out || N <= 0 || C <= 0 || T <= 0 || ldt < T || N > 65535) {
    set_error("ps_attn_weights_f32: bad argument (N=%d C=%d T=%d ldt=%d)", N, C, T, ldt);
    return PS_E_INVALID;
  }
  {
    LaunchTimer timer("attn_weights", (hipStream_t)stream);
    hipLaunchKernelGGL(attn_weights_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, logits, lengths, out, C, T, ldt);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_attn_weights_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int ps_attn_stats_pool_len_f32(const float* logits, const float* x, const float* lengths, float* out, int N,
                                          int C, int T, int ldt, float eps, void* stream) {
  using namespace ps;
  if (!logits || !x || !out || N <= 0 || C <= 0 || T <= 0 || ldt < T || N > 65535) {
    set_error("ps_attn_stats_pool_f32: bad argument (N=%d C=%d T=%d ldt=%d)", N, C, T, ldt);
    return PS_E_INVALID;
  }
  {
    LaunchTimer timer("attn_stats_pool", (hipStream_t)stream);
    // rows that fit 16 values per thread: one pass over memory (ps_debug_flags bit 0 keeps the three-pass kernel: tests)
    if (T <= 4096 && ldt % 4 == 0 && !(((uintptr_t)logits | (uintptr_t)x) & 15) && !(g_debug_flags & 1))
      hipLaunchKernelGGL(attn_stats_pool_reg_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, logits, x, lengths,
                         out, C, T, ldt, eps);
    else
      hipLaunchKernelGGL(attn_stats_pool_kernel, dim3(C, N), dim3(256), 0, (hipStream_t)stream, logits, x, lengths, out,
                         C, T, ldt, eps);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_attn_stats_pool_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int ps_attn_stats_pool_f32(const float* logits, const float* x, float* out, int N, int C, int T, int ldt,
                                      float eps, void* stream) {
  return ps_attn_stats_pool_len_f32(logits, x, nullptr, out, N, C, T, ldt, eps, stream);
}
