// Conv2d / ConvTranspose2d of the U-Net family as an implicit GEMM (unet.py:100-165 of mcw519/PureSound): the tap
// matrix that ps_unfold2d_f32 would materialise (K = Cin*kf*kt rows per output frequency row, 36 GB per forward of
// ns_dpcrn_v0 at 32 x 4 s) is never written; the B operand of the MFMA is gathered straight from the input rows.
//
//   out[n][co][fo][t] = act( bias[co] + sum_k W[co][k] * B_fo[k][t] ),   k = (ci, jf, jt)
//   conv:        B_fo[k][t] = x[n][ci][fo*sf + jf*df - pf][t + jt*dt - pt]
//   transposed:  B_fo[k][t] = x[n][ci][(fo + pf - jf*df)/sf][t + pt - jt*dt]   (when divisible)       0 outside the input
//
// A workgroup owns one output frequency row fo of one utterance, 128 frames and MB*32 output channels; wave w owns
// frames [32w, 32w+32) of the tile and all its channels (MB accumulators of v_mfma_f32_32x32x2_f32), so every B element is
// needed by exactly one wave and goes from global memory to its lane without LDS.  The k -> (source row, time shift)
// table of this fo lives in LDS.  Weights come from the packed transposed layout of ps_conv1x1_f32 (L2 resident).
// Eval BatchNorm2d is folded into W / bias by the caller; the activation is the epilogue.
#include <type_traits>

#include "ps_common.h"

namespace ps {

struct Conv2dArgs {
  const float* x1;
  const float* x2;
  const float* wt;
  const float* bias;
  const float* slope;
  float* y;
  int C1, C2, Fin, T, Tin, ld, kf, kt, sf, df, dt, pf, pt, Fout, M, K, Kp, transposed, act;
};

constexpr int C2D_MAXK = 4096;  // table entries (Cin*kf*kt rounded up to 16)
constexpr int C2D_UN = 8;       // k-pairs whose loads are in flight together

__device__ __forceinline__ float act_apply(float u, int kind, float s) {
  switch (kind) {
    case 1: return relu_keep_nan(u);
    case 2: return u >= 0.f ? u : s * u;
    case 3: return u * tanhf(u > 20.f ? u : log1pf(expf(u)));
    case 4: return 1.f / (1.f + expf(-u));
    case 5: return tanhf(u);
    default: return u;
  }
}

template <int MB>
__global__ __launch_bounds__(256) void conv2d_kernel(Conv2dArgs a) {
  // dynamic LDS, 2 * Kp ints (round 4: the static 32 KiB tables held a CU to five workgroups whatever K was)
  extern __shared__ int c2d_tab[];
  int* const tab_off = c2d_tab;            // element offset of the source row inside the utterance (bit 30: second source), -1 = zero row
  int* const tab_shift = c2d_tab + a.Kp;  // frame shift
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 31, lk = lane >> 5;
  const int t0 = blockIdx.x * 128, fo = blockIdx.y;
  const int mtiles = (a.M + 32 * MB - 1) / (32 * MB);
  const int n = blockIdx.z / mtiles, m0 = (blockIdx.z % mtiles) * 32 * MB;

  for (int k = tid; k < a.Kp; k += 256) {
    int off = -1, sh = 0;
    if (k < a.K) {
      const int jt = k % a.kt, jf = (k / a.kt) % a.kf, ci = k / (a.kt * a.kf);
      int fi;
      bool ok = true;
      if (!a.transposed) {
        fi = fo * a.sf + jf * a.df - a.pf;
        sh = jt * a.dt - a.pt;
      } else {
        const int num = fo + a.pf - jf * a.df;
        ok = num >= 0 && num % a.sf == 0;
        fi = num / a.sf;
        sh = a.pt - jt * a.dt;
      }
      if (ok && fi >= 0 && fi < a.Fin)
        off = ci < a.C1 ? (ci * a.Fin + fi) * a.ld : (((ci - a.C1) * a.Fin + fi) * a.ld) | (1 << 30);
    }
    tab_off[k] = off;
    tab_shift[k] = sh;
  }
  __syncthreads();

  const float* x1n = a.x1 + (size_t)n * a.C1 * a.Fin * a.ld;
  const float* x2n = a.x2 ? a.x2 + (size_t)n * a.C2 * a.Fin * a.ld : a.x1;
  const int tcol = t0 + 32 * w + lr;
  f32x16 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

  // Operands of C2D_UN k-pairs per batch; the loads of batch i + 1 are issued before the MFMAs of batch i (round 4: with
  // load -> wait -> MFMA in sequence the kernel ran at a third of the fp32 MFMA rate, every wave waiting out an L2 round
  // trip per batch).
  const int npairs = a.Kp / 2;
  float av[2][C2D_UN][MB], bv[2][C2D_UN];
  auto fetch = [&](int p0, auto buf_c) {
    constexpr int buf = decltype(buf_c)::value;
#pragma unroll
    for (int u = 0; u < C2D_UN; ++u) {
      const int k = 2 * (p0 + u) + lk;  // Kp is a multiple of 16 = 2 * C2D_UN: always < Kp
      const int off = tab_off[k];
      const int ti = tcol + tab_shift[k];
      const bool ok = off >= 0 && ti >= 0 && ti < a.Tin;
      const float* src = (off & (1 << 30)) ? x2n : x1n;
      const int idx = ok ? (off & ((1 << 30) - 1)) + ti : 0;
      const float v = src[idx];   // unconditional load of a valid address, masked afterwards
      bv[buf][u] = ok ? v : 0.f;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int m = m0 + 32 * mb + lr;  // the packed weight is zero padded to 256 rows per tile
        av[buf][u][mb] = a.wt[((size_t)(m >> 8) * a.Kp + k) * 256 + (m & 255)];
      }
    }
  };
  auto multiply = [&](auto buf_c) {
    constexpr int buf = decltype(buf_c)::value;
#pragma unroll
    for (int u = 0; u < C2D_UN; ++u)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u][mb], bv[buf][u], acc[mb], 0, 0, 0);
  };
  using b0 = std::integral_constant<int, 0>;
  using b1 = std::integral_constant<int, 1>;
  fetch(0, b0{});
  for (int p0 = 0; p0 < npairs; p0 += 2 * C2D_UN) {
    if (p0 + C2D_UN < npairs) fetch(p0 + C2D_UN, b1{});
    multiply(b0{});
    if (p0 + C2D_UN < npairs) {
      if (p0 + 2 * C2D_UN < npairs) fetch(p0 + 2 * C2D_UN, b0{});
      multiply(b1{});
    }
  }

  const float s = a.slope ? a.slope[0] : 0.f;
  if (tcol < a.ld) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < a.M) {
          float v = acc[mb][r] + (a.bias ? a.bias[m] : 0.f);
          v = act_apply(v, a.act, s);
          a.y[(((size_t)n * a.M + m) * a.Fout + fo) * a.ld + tcol] = tcol < a.T ? v : 0.f;
        }
      }
  }
}

}  // namespace ps

using namespace ps;

extern "C" int ps_conv2d_f32(const float* x1, int C1, const float* x2, int C2, const float* wt, const float* bias,
                             float* y, int N, int M, int Fin, int T_in, int T, int ld, int kf, int kt, int stride_f,
                             int dil_f, int dil_t, int pad_f, int pad_t, int Fout, int transposed, int act,
                             const float* slope, void* stream) {
  if (!x1 || !wt || !y || N <= 0 || M <= 0 || C1 <= 0 || C2 < 0 || (C2 > 0 && !x2) || Fin <= 0 || Fout <= 0 || T <= 0 ||
      T_in <= 0 || ld < T || ld < T_in || ld % 128 || kf <= 0 || kt <= 0 || stride_f <= 0 || dil_f <= 0 || dil_t <= 0 ||
      Fout > 65535 || act < 0 || act > 5 || (act == 2 && !slope)) {
    set_error("ps_conv2d_f32: bad argument (N=%d M=%d C=%d+%d F=%d->%d T=%d k=%dx%d act=%d)", N, M, C1, C2, Fin, Fout, T, kf,
              kt, act);
    return PS_E_INVALID;
  }
  const long long K = (long long)(C1 + C2) * kf * kt;
  const int Kp = (int)((K + 15) / 16 * 16);
  if (Kp > C2D_MAXK) {
    set_error("ps_conv2d_f32: Cin*kf*kt = %lld exceeds %d", K, C2D_MAXK);
    return PS_E_UNSUPPORTED;
  }
  if ((long long)(C1 > C2 ? C1 : C2) * Fin * ld >= (1LL << 30)) {
    set_error("ps_conv2d_f32: one utterance of the input exceeds 2^30 elements");
    return PS_E_UNSUPPORTED;
  }
  const int mb = M <= 32 ? 1 : M <= 64 ? 2 : 4;
  const int mtiles = (M + 32 * mb - 1) / (32 * mb);
  if ((long long)N * mtiles > 65535) {
    set_error("ps_conv2d_f32: N * channel tiles exceeds the grid limit");
    return PS_E_UNSUPPORTED;
  }
  Conv2dArgs a{x1, x2, wt, bias, slope, y, C1, C2, Fin, T, T_in, ld, kf, kt, stride_f, dil_f, dil_t, pad_f, pad_t,
               Fout, M, (int)K, Kp, transposed, act};
  dim3 grid(ld / 128, Fout, N * mtiles);
  {
    LaunchTimer timer("conv2d", (hipStream_t)stream);
    const size_t lds = (size_t)2 * Kp * sizeof(int);
    if (mb == 1)
      hipLaunchKernelGGL((conv2d_kernel<1>), grid, dim3(256), lds, (hipStream_t)stream, a);
    else if (mb == 2)
      hipLaunchKernelGGL((conv2d_kernel<2>), grid, dim3(256), lds, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL((conv2d_kernel<4>), grid, dim3(256), lds, (hipStream_t)stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_conv2d_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
