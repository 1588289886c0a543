// 2-D convolutional maskers (Unet / UnetTcn / DPCRN, puresound/nnet/unet.py, dpcrn.py of mcw519/PureSound) on the
// library's rows: a 4-D activation [N][CH][F][T] is stored as [N][CH*F] rows of ld frames.  A Conv2d / ConvTranspose2d
// over (frequency, time) becomes a GEMM on the matrix pipe (ps_conv1x1_f32 with M = Cout over "frames" = (f, t)
// flattened) after this file's gather kernel has laid the taps side by side:
//
//   ps_unfold2d_f32   y[n][(ci*kf + jf)*kt + jt][fo*ld + t] = x[n][ci][fi][ti]   (0 outside the input)
//        conv:        fi = fo*sf + jf*df - pf,          ti = t + jt*dt - pt      (nn.ZeroPad2d + nn.Conv2d, unet.py:112-128)
//        transposed:  fi = (fo + pf - jf*df) / sf if divisible,  ti = t + t_shift - jt*dt
//                     (nn.ConvTranspose2d + the time trim that follows it, unet.py:139-165, 252-256)
//        the input channels may come from two tensors (the decoder's torch.cat([x, skip], 1), unet.py:250).
//   ps_activation_f32 relu / prelu / mish / sigmoid / tanh in place (lobe/activation.py)
#include "ps_common.h"

namespace ps {

struct Unfold2dArgs {
  const float* x1;
  const float* x2;
  float* y;
  int C1, C2, Fin, T, Tin, ld, kf, kt, sf, df, dt, pf, pt, Fout, transposed;
  int n0;  // first utterance of this launch (the grid's z axis holds at most 65535 (utterance, tap row) pairs)
};

__global__ __launch_bounds__(256) void unfold2d_kernel(Unfold2dArgs a) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int fo = blockIdx.y;
  const int K = (a.C1 + a.C2) * a.kf * a.kt;
  const int n = a.n0 + blockIdx.z / K, row = blockIdx.z % K;
  if (t >= a.ld) return;
  const int jt = row % a.kt, jf = (row / a.kt) % a.kf, ci = row / (a.kt * a.kf);
  float v = 0.f;
  if (t < a.T) {
    int fi, ti;
    bool ok;
    if (!a.transposed) {
      fi = fo * a.sf + jf * a.df - a.pf;
      ti = t + jt * a.dt - a.pt;
      ok = true;
    } else {
      const int num = fo + a.pf - jf * a.df;
      ok = num >= 0 && num % a.sf == 0;
      fi = num / a.sf;
      ti = t + a.pt - jt * a.dt;  // pt carries the trim shift here
    }
    if (ok && fi >= 0 && fi < a.Fin && ti >= 0 && ti < a.Tin) {
      const float* src = ci < a.C1 ? a.x1 + ((size_t)n * a.C1 + ci) * a.Fin * a.ld
                                   : a.x2 + ((size_t)n * a.C2 + (ci - a.C1)) * a.Fin * a.ld;
      v = src[(size_t)fi * a.ld + ti];
    }
  }
  a.y[(((size_t)n * K + row) * a.Fout + fo) * a.ld + t] = v;
}

__global__ __launch_bounds__(256) void activation_kernel(float* __restrict__ x, const float* __restrict__ slope, int kind,
                                                         int T, int ld) {
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  const size_t row = blockIdx.y;
  if (t >= ld) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ld + t);
  const float s = slope ? slope[0] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float u = (t + e < T) ? v[e] : 0.f;  // pad columns are cleared: nothing non-finite survives a layer
    switch (kind) {
      case 1: u = relu_keep_nan(u); break;
      case 2: u = u >= 0.f ? u : s * u; break;
      case 3: u = u * tanhf(u > 20.f ? u : log1pf(expf(u))); break;  // mish = x tanh(softplus(x))
      case 4: u = 1.f / (1.f + expf(-u)); break;
      case 5: u = tanhf(u); break;
      default: break;
    }
    v[e] = (t + e < T) ? u : 0.f;
  }
  *reinterpret_cast<f32x4*>(x + row * ld + t) = v;
}

// gLN over [CH, F, T] (GlobLN on a 4-D map, lobe/norm.py:20-34) + activation, in place.  The statistics come from the
// producing GEMM's partial slabs, which also cover the pad columns of every frequency row; there the taps are zero and
// the GEMM output is exactly its bias, so the caller passes that contribution (corr_sum, corr_sq) to be taken out.
struct NormActArgs {
  float* x;
  ps_prologue pro;
  double corr_sum, corr_sq;
  const float* slope;
  int rows_per_channel, kind, rows_per_utt, T, ld;
};

__global__ __launch_bounds__(256) void norm_activation_kernel(NormActArgs a) {
  __shared__ double red[8];
  const int n = blockIdx.z, row = blockIdx.y;
  double mean = 0.0;
  float rstd = 1.f;
  if (a.pro.norm == PS_NORM_GLOBAL) {  // (PS_NORM_AFFINE: a folded BatchNorm, gamma / beta are its scale / shift)
    double sa = 0.0, sq = 0.0;
    const double* src = a.pro.stats + (size_t)n * a.pro.parts * 2;
    for (int i = threadIdx.x; i < a.pro.parts; i += 256) {
      sa += src[2 * i];
      sq += src[2 * i + 1];
    }
    block_sum2(sa, sq, red);
    sa -= a.corr_sum;
    sq -= a.corr_sq;
    mean = sa / a.pro.count;
    double var = sq / a.pro.count - mean * mean;
    var = var > 0.0 ? var : 0.0;
    rstd = (float)(1.0 / sqrt(var + (double)a.pro.eps));
  }
  const int ch = row / a.rows_per_channel;
  const float sc = a.pro.gamma[ch] * rstd, sh = a.pro.beta[ch] - (float)mean * sc;
  const float s = a.slope ? a.slope[0] : 0.f;
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t >= a.ld) return;
  float* p = a.x + ((size_t)n * a.rows_per_utt + row) * a.ld + t;
  f32x4 v = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float u = v[e] * sc + sh;
    switch (a.kind) {
      case 1: u = relu_keep_nan(u); break;
      case 2: u = u >= 0.f ? u : s * u; break;
      case 3: u = u * tanhf(u > 20.f ? u : log1pf(expf(u))); break;
      case 4: u = 1.f / (1.f + expf(-u)); break;
      case 5: u = tanhf(u); break;
      default: break;
    }
    v[e] = (t + e < a.T) ? u : 0.f;
  }
  *reinterpret_cast<f32x4*>(p) = v;
}

// partial (sum, sum of squares) of the valid frames of an utterance's rows: stats [N][kRowStatsParts][2] fp64, the
// slab layout the global-norm prologues and ps_norm_activation_f32 consume (GlobLN called on its own)
constexpr int kRowStatsParts = 64;
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, int rows,
                                                        int T, int ld) {
  __shared__ double red[8];
  const int p = blockIdx.x, n = blockIdx.y;
  double sa = 0.0, sq = 0.0;
  for (int r = p; r < rows; r += kRowStatsParts) {
    const float* row = x + ((size_t)n * rows + r) * ld;
    float fa = 0.f, fq = 0.f;  // (a row's share per thread is short: fp32 inside, fp64 across rows)
    for (int t = threadIdx.x; t < T; t += 256) {
      const float v = row[t];
      fa += v;
      fq += v * v;
    }
    sa += fa;
    sq += fq;
  }
  block_sum2(sa, sq, red);
  if (threadIdx.x == 0) {
    stats[((size_t)n * kRowStatsParts + p) * 2] = sa;
    stats[((size_t)n * kRowStatsParts + p) * 2 + 1] = sq;
  }
}

__global__ __launch_bounds__(256) void add_rows_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ y, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  reinterpret_cast<f32x4*>(y)[i] = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
}

static int unet_status(const char* who) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", who, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // namespace ps

using namespace ps;

extern "C" int ps_unfold2d_f32(const float* x1, int C1, const float* x2, int C2, float* y, int N, int Fin, int T_in,
                               int T, int ld, int kf, int kt, int stride_f, int dil_f, int dil_t, int pad_f, int pad_t, int Fout,
                               int transposed, void* stream) {
  if (!x1 || !y || N <= 0 || C1 <= 0 || C2 < 0 || (C2 > 0 && !x2) || Fin <= 0 || Fout <= 0 || T <= 0 || T_in <= 0 || ld < T || ld < T_in ||
      kf <= 0 || kt <= 0 || stride_f <= 0 || dil_f <= 0 || dil_t <= 0 || Fout > 65535) {
    set_error("ps_unfold2d_f32: bad argument (N=%d C=%d+%d F=%d->%d T=%d k=%dx%d)", N, C1, C2, Fin, Fout, T, kf, kt);
    return PS_E_INVALID;
  }
  const long long K = (long long)(C1 + C2) * kf * kt;
  if (K > 65535) {
    set_error("ps_unfold2d_f32: Cin * kf * kt = %lld exceeds the grid limit 65535", K);
    return PS_E_UNSUPPORTED;
  }
  Unfold2dArgs a{x1, x2, y, C1, C2, Fin, T, T_in, ld, kf, kt, stride_f, dil_f, dil_t, pad_f, pad_t, Fout, transposed, 0};
  {
    LaunchTimer timer("unfold2d", (hipStream_t)stream);
    const int per = (int)(65535 / K);  // utterances per launch (tse_unet_tcn at 32 utterances: 81,920 tap rows)
    for (int n0 = 0; n0 < N; n0 += per) {
      a.n0 = n0;
      const int nb = N - n0 < per ? N - n0 : per;
      hipLaunchKernelGGL(unfold2d_kernel, dim3((ld + 255) / 256, Fout, (unsigned)(nb * K)), dim3(256), 0, (hipStream_t)stream, a);
    }
  }
  return unet_status("ps_unfold2d_f32");
}

extern "C" int ps_activation_f32(float* x, int kind, const float* slope, int64_t rows, int T, int ld, void* stream) {
  if (!x || rows <= 0 || T <= 0 || ld < T || ld % 4 || ((uintptr_t)x & 15) || kind < 0 || kind > 5 ||
      (kind == 2 && !slope)) {
    set_error("ps_activation_f32: bad argument (rows=%lld T=%d ld=%d kind=%d)", (long long)rows, T, ld, kind);
    return PS_E_INVALID;
  }
  LaunchTimer timer("activation", (hipStream_t)stream);
  const int64_t chunk = 65535;
  for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
    const int64_t nr = rows - r0 < chunk ? rows - r0 : chunk;
    hipLaunchKernelGGL(activation_kernel, dim3((ld / 4 + 255) / 256, (unsigned)nr), dim3(256), 0, (hipStream_t)stream,
                       x + r0 * ld, slope, kind, T, ld);
  }
  return unet_status("ps_activation_f32");
}

extern "C" int ps_add_f32(const float* a, const float* b, float* y, int64_t count, void* stream) {
  if (!a || !b || !y || count <= 0 || count % 4 || ((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)y & 15) ||
      count / 4 > 0x7fffffffLL * 256) {
    set_error("ps_add_f32: bad argument (count=%lld must be a positive multiple of 4, pointers 16-byte aligned)",
              (long long)count);
    return PS_E_INVALID;
  }
  const size_t n4 = (size_t)count / 4;
  LaunchTimer timer("add", (hipStream_t)stream);
  hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, y, n4);
  return unet_status("ps_add_f32");
}

extern "C" int ps_row_stats_parts(void) { return ps::kRowStatsParts; }

extern "C" int ps_row_stats_f64(const float* x, double* stats, int N, int rows, int T, int ld, void* stream) {
  using namespace ps;
  if (!x || !stats || N <= 0 || N > 65535 || rows <= 0 || T <= 0 || ld < T) {
    set_error("ps_row_stats_f64: bad argument (N=%d rows=%d T=%d ld=%d)", N, rows, T, ld);
    return PS_E_INVALID;
  }
  {
    LaunchTimer timer("row_stats", (hipStream_t)stream);
    hipLaunchKernelGGL(row_stats_kernel, dim3(kRowStatsParts, N), dim3(256), 0, (hipStream_t)stream, x, stats, rows, T, ld);
  }
  return unet_status("ps_row_stats_f64");
}

extern "C" int ps_norm_activation_f32(float* x, const ps_prologue* pro, double corr_sum, double corr_sq,
                                      int rows_per_channel, int kind, const float* slope, int N, int rows_per_utt, int T,
                                      int ld, void* stream) {
  const bool global = pro && pro->norm == PS_NORM_GLOBAL;
  if (!x || !pro || (!global && pro->norm != PS_NORM_AFFINE) ||
      (global && (!pro->stats || pro->parts <= 0 || pro->count <= 0)) || !pro->gamma ||
      !pro->beta || N <= 0 || N > 65535 || rows_per_utt <= 0 || rows_per_utt > 65535 || rows_per_channel <= 0 ||
      T <= 0 || ld < T || ld % 4 || ((uintptr_t)x & 15) || kind < 0 || kind > 5 || (kind == 2 && !slope)) {
    set_error("ps_norm_activation_f32: bad argument (N=%d rows=%d T=%d ld=%d kind=%d)", N, rows_per_utt, T, ld, kind);
    return PS_E_INVALID;
  }
  NormActArgs a{x, *pro, corr_sum, corr_sq, slope, rows_per_channel, kind, rows_per_utt, T, ld};
  {
    LaunchTimer timer("norm_activation", (hipStream_t)stream);
    hipLaunchKernelGGL(norm_activation_kernel, dim3((ld / 4 + 255) / 256, rows_per_utt, N), dim3(256), 0,
                       (hipStream_t)stream, a);
  }
  return unet_status("ps_norm_activation_f32");
}
