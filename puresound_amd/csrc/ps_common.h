// Shared device helpers for libpuresound_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/puresound_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace ps {

constexpr int kWave = 64;
constexpr int kTileT = 128;  // padded-row granule: ldt % kTileT == 0

void set_error(const char* fmt, ...);
extern void* g_debug_buffer;  // ps_debug_buffer(): where diagnostic stamps go (NULL in production)
extern int g_debug_flags;  // ps_debug_flags(): kernel ablation switches for profiling builds, 0 in production

// Compute units of the CURRENT device (hipGetDevice), cached per device id: a process that drives several GPUs -- one
// per thread, or hipSetDevice between calls -- gets each device's own count.
int device_cus();

// Brackets one kernel launch with hipEvents when ps_profile_enable(1) is active (no-op otherwise).
struct LaunchTimer {
  LaunchTimer(const char* kernel, hipStream_t stream);
  ~LaunchTimer();
  int slot_;
  hipStream_t stream_;
};

// Per-utterance scalars of a global norm, derived from the producer's partial sums.
struct NormScalars {
  float mean;
  float rstd;
};

// 64-lane butterfly sum of a double (two 32-bit shuffles per step).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}

// 64-lane butterfly maximum (fmaxf: a NaN lane is ignored)
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  return v;
}

// Sum (a, b) over a 256-thread workgroup; every thread receives the totals.  `red` holds 8 doubles.
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[wave * 2] = a;
    red[wave * 2 + 1] = b;
  }
  __syncthreads();
  a = red[0] + red[2] + red[4] + red[6];
  b = red[1] + red[3] + red[5] + red[7];
  __syncthreads();
}

// Combine the producer's partials for utterance n into mean / rstd (fp64 combine, deterministic order
// per thread, butterfly across threads).  biased variance, var = E[x^2] - mean^2.
__device__ __forceinline__ NormScalars load_norm_scalars(const ps_prologue& p, int n, double* red) {
  NormScalars s{0.f, 1.f};
  if (p.norm != PS_NORM_GLOBAL) return s;
  double a = 0.0, b = 0.0;
  const double* src = p.stats + (size_t)n * p.parts * 2;
  for (int i = threadIdx.x; i < p.parts; i += blockDim.x) {
    a += src[2 * i];
    b += src[2 * i + 1];
  }
  block_sum2(a, b, red);
  const double mean = a / p.count;
  double var = b / p.count - mean * mean;
  var = var > 0.0 ? var : 0.0;
  s.mean = (float)mean;
  s.rstd = (float)(1.0 / sqrt(var + (double)p.eps));
  return s;
}

// conv1x1_small.hip: rows of at most 64 frames (no global-norm prologue, no statistics); enqueue only
int conv1x1_small_launch(const float* x, const float* wt, float* y, int N, int K, int M, int T, int ldt,
                         const ps_prologue* pro, const float* bias, const float* bias_n, const float* res,
                         hipStream_t stream);

__device__ __forceinline__ float prelu(float v, float slope) { return v >= 0.f ? v : slope * v; }
// torch.relu / clamp keep a NaN a NaN (fmaxf / fminf would return the other operand): the reference's own look-ahead
// probe (base_nn.py:740-777) reads NaNs off the output
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }
__device__ __forceinline__ float clamp1_keep_nan(float v) { return v < -1.f ? -1.f : (v > 1.f ? 1.f : v); }

}  // namespace ps
