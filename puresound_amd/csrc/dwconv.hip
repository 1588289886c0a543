// Fused depthwise dilated convolution of the TCN block (lobe/cnn.py:62-74 of mcw519/PureSound):
//   a = PReLU(norm(x))   (the in_conv's gLN/bN1d + PReLU, applied on load)
//   y[n][h][t] = b[h] + sum_j w[h][j] * a[n][h][t + j*dilation - left]      (a == 0 outside [0,T))
// plus per-workgroup partial (sum, sumsq) of y for the next global norm.
//
// HBM-bound streaming kernel: one thread owns 4 consecutive frames (16-byte loads/stores), a workgroup
// owns DW_ROWS channels x 1024 frames.  Taps whose offset is a multiple of 4 frames are aligned 16-byte
// loads of the neighbouring vectors (they hit L1/L2: the same rows were just streamed); other offsets
// (dilation 1, 2 or odd bases) fall back to dword loads.
#include "ps_common.h"

namespace ps {

constexpr int DW_ROWS = 16;
constexpr int DW_FRAMES = 1024;  // 256 threads x 4
constexpr int DW_MAXP = 8;

struct DwArgs {
  const float* x;
  const float* w;
  const float* b;
  float* y;
  double* ostats;
  ps_prologue pro;
  int H, T, ldt, P, dilation, left;
};

__global__ __launch_bounds__(256) void dwconv_kernel(DwArgs a) {
  __shared__ double red[8];
  const int tid = threadIdx.x;
  const int n = blockIdx.z;
  const int h0 = blockIdx.y * DW_ROWS;
  const int t = blockIdx.x * DW_FRAMES + tid * 4;

  const NormScalars ns = load_norm_scalars(a.pro, n, red);
  const bool has_norm = a.pro.norm != PS_NORM_NONE;
  const float slope = a.pro.prelu ? a.pro.slope[0] : 1.f;
  const bool aligned = (a.dilation % 4 == 0) && (a.left % 4 == 0);

  float fsum = 0.f, fsq = 0.f;
  if (t < a.T) {
    for (int r = 0; r < DW_ROWS; ++r) {
      const int h = h0 + r;
      if (h >= a.H) break;
      const float* xr = a.x + ((size_t)n * a.H + h) * a.ldt;
      const float sc = has_norm ? a.pro.gamma[h] * ns.rstd : 1.f;
      const float sh = has_norm ? a.pro.beta[h] : 0.f;
      const float bias = a.b ? a.b[h] : 0.f;
      f32x4 out{bias, bias, bias, bias};
      for (int j = 0; j < a.P; ++j) {
        const float wj = a.w[h * a.P + j];
        const int tt = t + j * a.dilation - a.left;
        f32x4 v{0.f, 0.f, 0.f, 0.f};
        if (aligned) {
          if (tt >= 0 && tt < a.T) v = *reinterpret_cast<const f32x4*>(xr + tt);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (tt + e >= 0 && tt + e < a.T) v[e] = xr[tt + e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u = (v[e] - ns.mean) * sc + sh;
          if (a.pro.prelu) u = prelu(u, slope);
          u = (tt + e >= 0 && tt + e < a.T) ? u : 0.f;
          out[e] += wj * u;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (t + e < a.T) {
          fsum += out[e];
          fsq += out[e] * out[e];
        }
      *reinterpret_cast<f32x4*>(a.y + ((size_t)n * a.H + h) * a.ldt + t) = out;
    }
  }
  if (a.ostats) {
    double s = fsum, q = fsq;
    block_sum2(s, q, red);
    if (tid == 0) {
      const int parts = gridDim.x * gridDim.y;
      double* dst = a.ostats + ((size_t)n * parts + blockIdx.y * gridDim.x + blockIdx.x) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

}  // namespace ps

extern "C" int ps_dwconv_stats_parts(int H, int T) {
  if (H <= 0 || T <= 0) return 0;
  return ((T + ps::DW_FRAMES - 1) / ps::DW_FRAMES) * ((H + ps::DW_ROWS - 1) / ps::DW_ROWS);
}

extern "C" int ps_dwconv_f32(const float* x, const float* w, const float* b, float* y, int N, int H, int T,
                             int ldt, int P, int dilation, int left, const ps_prologue* pro, double* ostats,
                             void* stream) {
  using namespace ps;
  if (!x || !w || !y || N <= 0 || H <= 0 || T <= 0 || P <= 0 || P > DW_MAXP || dilation <= 0 || left < 0) {
    set_error("ps_dwconv_f32: bad argument (N=%d H=%d T=%d P=%d dilation=%d left=%d)", N, H, T, P, dilation, left);
    return PS_E_INVALID;
  }
  if (ldt < T || ldt % kTileT != 0 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) {
    set_error("ps_dwconv_f32: ldt=%d must be a multiple of %d >= T=%d and pointers 16-byte aligned", ldt, kTileT, T);
    return PS_E_ALIGN;
  }
  DwArgs a{};
  a.x = x;
  a.w = w;
  a.b = b;
  a.y = y;
  a.ostats = ostats;
  if (pro) {
    a.pro = *pro;
    if (a.pro.norm != PS_NORM_NONE && (!a.pro.gamma || !a.pro.beta)) {
      set_error("ps_dwconv_f32: norm prologue needs gamma/beta");
      return PS_E_INVALID;
    }
    if (a.pro.norm == PS_NORM_GLOBAL && (!a.pro.stats || a.pro.parts <= 0 || a.pro.count <= 0)) {
      set_error("ps_dwconv_f32: PS_NORM_GLOBAL prologue needs stats/parts/count");
      return PS_E_INVALID;
    }
    if (a.pro.prelu && !a.pro.slope) {
      set_error("ps_dwconv_f32: prelu prologue needs slope");
      return PS_E_INVALID;
    }
  } else {
    a.pro.norm = PS_NORM_NONE;
  }
  a.H = H;
  a.T = T;
  a.ldt = ldt;
  a.P = P;
  a.dilation = dilation;
  a.left = left;
  dim3 grid((T + DW_FRAMES - 1) / DW_FRAMES, (H + DW_ROWS - 1) / DW_ROWS, N);
  {
    LaunchTimer timer("dwconv", (hipStream_t)stream);
    hipLaunchKernelGGL(dwconv_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_dwconv_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
