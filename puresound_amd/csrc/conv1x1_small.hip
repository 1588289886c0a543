// 1x1 convolution for rows of at most 64 frames (the streaming step: frames = concurrent streams; Mem-LSTM state
// rows: frames = segments).  The big kernel (conv1x1.hip) owns 256 x 128 output tiles, which at T <= 64 leaves one
// to four workgroups busy with mostly-padding tiles; here a workgroup owns 16 output channels x all frames, so
// M / 16 workgroups share the weight read, and its four waves split K (partials meet in LDS, summed in a fixed order,
// so the result is deterministic).  v_mfma_f32_16x16x4_f32: A = W[16 rows][4 k] from the packed transposed weights,
// B = x[4 k][16 frames] straight from the activation rows, up to four 16-frame column blocks per wave.
// Same epilogue terms as the big kernel (bias, per-utterance bias, residual) and the per-channel part of its
// prologue (affine norm, PReLU, ReLU-before / tanh-after); no global-norm prologue, no output statistics.
#include "ps_common.h"

namespace ps {

struct SmallArgs {
  const float* x;
  const float* wt;
  float* y;
  const float* bias;
  const float* bias_n;
  const float* res;
  ps_prologue pro;
  int K, Kp, M, T, ldt;
};

__device__ __forceinline__ float small_transform(const SmallArgs& a, float v, int k, float slope) {
  if (a.pro.pre_relu) v = fmaxf(v, 0.f);
  if (a.pro.norm == PS_NORM_AFFINE) v = v * a.pro.gamma[k] + a.pro.beta[k];
  if (a.pro.prelu) v = prelu(v, slope);
  if (a.pro.post_tanh) v = tanhf(v);
  return v;
}

template <int NCB, bool TR>
__global__ __launch_bounds__(256) void conv1x1_small_kernel(SmallArgs a) {
  __shared__ f32x4 part[4][NCB][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * 16, n = blockIdx.y;
  const int m = m0 + r;
  const float* wp = a.wt + ((size_t)(m >> 8) * a.Kp) * 256 + (m & 255);
  const float* xp = a.x + (size_t)n * a.K * a.ldt + r;
  const float slope = (TR && a.pro.prelu) ? a.pro.slope[0] : 1.f;
  f32x4 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk4 = a.Kp / 4;
#pragma unroll 4
  for (int i = w; i < nk4; i += 4) {
    const int k = 4 * i + kq;
    const float av = wp[(size_t)k * 256];
    const bool kin = k < a.K;
    float bv[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      float v = kin ? xp[(size_t)k * a.ldt + cb * 16] : 0.f;
      if constexpr (TR) v = kin ? small_transform(a, v, k, slope) : 0.f;
      bv[cb] = v;
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[cb], acc[cb], 0, 0, 0);
  }
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) part[w][cb][lane] = acc[cb];
  __syncthreads();
  // thread (cb, lane) finishes the 4 outputs the accumulator lane holds: rows 4*kq + reg, frame cb*16 + r
  for (int idx = threadIdx.x; idx < NCB * 64; idx += 256) {
    const int cb = idx >> 6, ln = idx & 63;
    const f32x4 s = (part[0][cb][ln] + part[1][cb][ln]) + (part[2][cb][ln] + part[3][cb][ln]);
    const int t = cb * 16 + (ln & 15);
    if (t >= a.T) continue;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int mo = m0 + 4 * (ln >> 4) + reg;
      if (mo >= a.M) continue;
      float v = s[reg];
      if (a.bias) v += a.bias[mo];
      if (a.bias_n) v += a.bias_n[(size_t)n * a.M + mo];
      const size_t off = ((size_t)n * a.M + mo) * a.ldt + t;
      if (a.res) v += a.res[off];
      a.y[off] = v;
    }
  }
}

int conv1x1_small_launch(const float* x, const float* wt, float* y, int N, int K, int M, int T, int ldt,
                         const ps_prologue* pro, const float* bias, const float* bias_n, const float* res,
                         hipStream_t stream) {
  SmallArgs a{};
  a.x = x;
  a.wt = wt;
  a.y = y;
  a.bias = bias;
  a.bias_n = bias_n;
  a.res = res;
  bool tr = false;
  if (pro) {
    a.pro = *pro;
    tr = pro->norm != PS_NORM_NONE || pro->prelu || pro->pre_relu || pro->post_tanh;
  }
  a.K = K;
  a.Kp = (K + 15) / 16 * 16;
  a.M = M;
  a.T = T;
  a.ldt = ldt;
  const int ncb = (T + 15) / 16;
  dim3 grid((M + 15) / 16, N);
  LaunchTimer timer("conv1x1", stream);
#define PS_SMALL(NCB)                                                                          \
  if (tr)                                                                                      \
    hipLaunchKernelGGL((conv1x1_small_kernel<NCB, true>), grid, dim3(256), 0, stream, a);      \
  else                                                                                         \
    hipLaunchKernelGGL((conv1x1_small_kernel<NCB, false>), grid, dim3(256), 0, stream, a);
  if (ncb == 1) {
    PS_SMALL(1)
  } else if (ncb == 2) {
    PS_SMALL(2)
  } else {
    PS_SMALL(4)
  }
#undef PS_SMALL
  return 0;
}

}  // namespace ps
