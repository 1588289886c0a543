// Fused 1x1 convolution for the Conv-TasNet TCN blocks: exact-fp32 MFMA GEMM with the producer's
// normalisation + PReLU applied while staging the activation tile, and bias / residual / partial
// statistics in the epilogue.
//
//   y[n][m][t] = sum_k W[m][k] * pro(x[n][k][t]) + bias[m] (+ bias_n[n][m]) (+ res[n][m][t])
//
// Reference arithmetic replaced: conv_tasnet.py:43-49,65,85-88 and lobe/cnn.py:75-79 of mcw519/PureSound.
//
// Mapping to CDNA4: D = A*B with A = W (rows = output channel m) and B = activations (cols = frame t),
// v_mfma_f32_32x32x2_f32.  Both operands are staged k-major in LDS ([k][m] and [k][t]) so that the 32
// lanes of a half-wave read 32 consecutive dwords (conflict-free ds_read_b32); the weight is stored
// pre-transposed in HBM for that reason.  A 256-thread workgroup owns a 256(m) x 128(t) output tile;
// each of its 4 waves owns 64(m) x 128(t) = 2x4 MFMA tiles (128 accumulator VGPRs), so one k-pair costs
// 2+4 LDS reads for 8 MFMAs.  K is consumed in steps of 16 through a 2-deep LDS ring; the global loads
// of step s+1 are issued before the MFMAs of step s and written to LDS after them (one barrier per step).
#include "ps_common.h"

namespace ps {

constexpr int BM = 256;
constexpr int BT = 128;
constexpr int BK = 16;

struct Conv1x1Args {
  const float* x;
  const float* wt;
  float* y;
  const float* bias;
  const float* bias_n;
  const float* res;
  double* ostats;
  ps_prologue pro;
  int K, M, T, ldt, Mp;
  unsigned long long* stamps;  // ps_debug_buffer(): per-workgroup s_memtime stamps (diagnostic builds/runs only)
  int dbg;  // ablation switches (ps_debug_flags): 1 no stores, 2 no MFMA, 4 no in-loop global loads, 8 no stats
};

__global__ __launch_bounds__(256, 2) void conv1x1_kernel(Conv1x1Args a) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BT];
  __shared__ double red[8];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int n = blockIdx.z;
  const int m0 = blockIdx.y * BM;
  const int t0 = blockIdx.x * BT;

  unsigned long long t_begin = 0, t_pro = 0, t_loop = 0;
  if (a.stamps) t_begin = __builtin_amdgcn_s_memtime();
  const NormScalars ns = load_norm_scalars(a.pro, n, red);
  const bool transform = a.pro.norm != PS_NORM_NONE || a.pro.prelu;
  const float slope = a.pro.prelu ? a.pro.slope[0] : 1.f;

  const float* xg = a.x + (size_t)n * a.K * a.ldt + t0;
  const float* wg = a.wt + m0;

  // staging coordinates
  const int a_row = tid >> 6;         // +4j, j = 0..3
  const int a_col = (tid & 63) * 4;   // 0..252
  const int b_row = tid >> 5;         // +8j, j = 0..1
  const int b_col = (tid & 31) * 4;   // 0..124

  f32x4 ra[4], rb[2];
  float gm[2], bt[2];
  bool valid[2];
  const bool has_norm = a.pro.norm != PS_NORM_NONE;  // kernel-uniform

  // Issue the global loads of one K-step.  Nothing here consumes a loaded value, so the compiler places
  // no s_waitcnt between these loads and the MFMAs that follow: the tile lands while the matrix pipe works.
  // Rows k >= K are clamped to a valid address and zeroed in store_step (branch-free issue).
  auto load_step = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      ra[j] = *reinterpret_cast<const f32x4*>(wg + (size_t)(k0 + a_row + 4 * j) * a.Mp + a_col);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + b_row + 8 * j;
      valid[j] = k < a.K;
      const int kc = valid[j] ? k : a.K - 1;
      rb[j] = *reinterpret_cast<const f32x4*>(xg + (size_t)kc * a.ldt + b_col);
      if (has_norm) {
        gm[j] = a.pro.gamma[kc];
        bt[j] = a.pro.beta[kc];
      }
    }
  };

  auto store_step = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&As[buf][a_row + 4 * j][a_col]) = ra[j];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 v = rb[j];
      if (transform) {
        const float sc = has_norm ? gm[j] * ns.rstd : 1.f;
        const float sh = has_norm ? bt[j] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = prelu((v[e] - ns.mean) * sc + sh, slope);
      }
      if (!valid[j]) v = f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(&Bs[buf][b_row + 8 * j][b_col]) = v;
    }
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ti][r] = 0.f;

  const int nsteps = (a.K + BK - 1) / BK;
  const bool wave_active = (m0 + wave * 64) < a.M;  // wave-uniform: skip MFMAs on all-padding rows
  const int lr = lane & 31;
  const int lk = lane >> 5;

  load_step(0);
  store_step(0);
  __syncthreads();
  if (a.stamps) t_pro = __builtin_amdgcn_s_memtime();

  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps && !(a.dbg & 4)) load_step((s + 1) * BK);
    if (wave_active && !(a.dbg & 2)) {
      // fragment registers are double-buffered: the LDS reads of k-pair kk+1 are in flight while the
      // eight MFMAs of k-pair kk issue, so no MFMA waits on an LDS round trip inside a K-step.
      float av[2][2], bv[2][4];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) av[0][mi] = As[buf][lk][wave * 64 + mi * 32 + lr];
#pragma unroll
      for (int ti = 0; ti < 4; ++ti) bv[0][ti] = Bs[buf][lk][ti * 32 + lr];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < BK / 2) {
          const int k = 2 * (kk + 1) + lk;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) av[nxt][mi] = As[buf][k][wave * 64 + mi * 32 + lr];
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) bv[nxt][ti] = Bs[buf][k][ti * 32 + lr];
        }
        // keep the prefetch reads ahead of this k-pair's MFMAs (hipcc otherwise sinks them to their use)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ti = 0; ti < 4; ++ti)
            acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][mi], bv[cur][ti], acc[mi][ti], 0, 0, 0);
      }
    }
    if (s + 1 < nsteps) store_step(buf ^ 1);
    __syncthreads();
  }

  if (a.stamps) t_loop = __builtin_amdgcn_s_memtime();
  // ---- epilogue: bias, residual, store, partial statistics -------------------------------------
  // Pad columns (t >= T, always inside the padded row) are stored too: nothing reads them as data, and
  // keeping the stores unconditional lets the compiler batch the residual loads instead of serialising
  // 128 load->add->store round trips behind per-element branches.  Statistics mask them out.
  double ssum = 0.0, ssq = 0.0;
  if (wave_active) {
    float* yg = a.y + (size_t)n * a.M * a.ldt + t0 + lr;
    const float* rg = a.res ? a.res + (size_t)n * a.M * a.ldt + t0 + lr : nullptr;
    const float* bn = a.bias_n ? a.bias_n + (size_t)n * a.M : nullptr;
    const int row_base = m0 + wave * 64 + 4 * lk;
    const bool full_rows = (m0 + wave * 64 + 64) <= a.M;  // wave-uniform
    float cmask[4];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti) cmask[ti] = (t0 + ti * 32 + lr) < a.T ? 1.f : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      float fsum = 0.f, fsq = 0.f;
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        float rv[4][4];
        float bs[4];
        int rows[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = row_base + mi * 32 + rr + 8 * rq;
          rows[rr] = full_rows ? row : (row < a.M ? row : a.M - 1);  // clamp: loads stay in bounds
          bs[rr] = (a.bias ? a.bias[rows[rr]] : 0.f) + (bn ? bn[rows[rr]] : 0.f);
          if (rg) {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) rv[rr][ti] = rg[(size_t)rows[rr] * a.ldt + ti * 32];
          }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = row_base + mi * 32 + rr + 8 * rq;
          const bool row_ok = full_rows || row < a.M;
#pragma unroll
          for (int ti = 0; ti < 4; ++ti) {
            float v = acc[mi][ti][rq * 4 + rr] + bs[rr];
            const float vm = row_ok ? v * cmask[ti] : 0.f;
            fsum += vm;
            fsq += vm * vm;
            if (rg) v += rv[rr][ti];
            if (a.dbg & 1) {
              asm volatile("" ::"v"(v));
            } else if (row_ok) {
              yg[(size_t)row * a.ldt + ti * 32] = v;
            }
          }
        }
      }
      ssum += (double)fsum;
      ssq += (double)fsq;
    }
  }
  if (a.ostats && !(a.dbg & 8)) {
    block_sum2(ssum, ssq, red);
    if (tid == 0) {
      const int parts = gridDim.x * gridDim.y;
      double* dst = a.ostats + ((size_t)n * parts + blockIdx.y * gridDim.x + blockIdx.x) * 2;
      dst[0] = ssum;
      dst[1] = ssq;
    }
  }
  if (a.stamps && tid == 0) {
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const size_t wgid = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    unsigned long long* d = a.stamps + wgid * 6;
    d[0] = t_begin;
    d[1] = t_pro;
    d[2] = t_loop;
    d[3] = t_end;
    d[4] = hwid;
    d[5] = xcc;
  }
}

}  // namespace ps

extern "C" int ps_conv1x1_f32(const float* x, const float* wt, float* y, int N, int K, int M, int T, int ldt,
                              const ps_prologue* pro, const float* bias, const float* bias_n,
                              const float* res, double* ostats, void* stream) {
  using namespace ps;
  if (!x || !wt || !y || N <= 0 || K <= 0 || M <= 0 || T <= 0) {
    set_error("ps_conv1x1_f32: null pointer or non-positive size (N=%d K=%d M=%d T=%d)", N, K, M, T);
    return PS_E_INVALID;
  }
  if (ldt < T || ldt % kTileT != 0 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)wt & 15)) {
    set_error("ps_conv1x1_f32: ldt=%d must be a multiple of %d >= T=%d and pointers 16-byte aligned", ldt,
              kTileT, T);
    return PS_E_ALIGN;
  }
  Conv1x1Args a{};
  a.x = x;
  a.wt = wt;
  a.y = y;
  a.bias = bias;
  a.bias_n = bias_n;
  a.res = res;
  a.ostats = ostats;
  if (pro) {
    a.pro = *pro;
    if (a.pro.norm == PS_NORM_GLOBAL && (!a.pro.stats || a.pro.parts <= 0 || a.pro.count <= 0 || !a.pro.gamma ||
                                         !a.pro.beta)) {
      set_error("ps_conv1x1_f32: PS_NORM_GLOBAL prologue needs stats/parts/count/gamma/beta");
      return PS_E_INVALID;
    }
    if (a.pro.norm == PS_NORM_AFFINE && (!a.pro.gamma || !a.pro.beta)) {
      set_error("ps_conv1x1_f32: PS_NORM_AFFINE prologue needs gamma/beta");
      return PS_E_INVALID;
    }
    if (a.pro.prelu && !a.pro.slope) {
      set_error("ps_conv1x1_f32: prelu prologue needs slope");
      return PS_E_INVALID;
    }
  } else {
    a.pro.norm = PS_NORM_NONE;
  }
  a.K = K;
  a.M = M;
  a.T = T;
  a.ldt = ldt;
  a.Mp = (M + BM - 1) / BM * BM;
  a.dbg = g_debug_flags;
  a.stamps = (unsigned long long*)g_debug_buffer;
  dim3 grid((T + BT - 1) / BT, (M + BM - 1) / BM, N);
  {
    LaunchTimer timer("conv1x1", (hipStream_t)stream);
    hipLaunchKernelGGL(conv1x1_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_conv1x1_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
