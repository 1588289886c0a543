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
  if (a.pro.pre_relu) v = relu_keep_nan(v);
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
  // explicit batches: all loads of UN k-steps are issued before their MFMAs (the loop is latency bound)
  constexpr int UN = 8;
  for (int i0 = w; i0 < nk4; i0 += 4 * UN) {
    float av[UN], bv[UN][NCB];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = 4 * (i0 + 4 * u) + kq;
      const bool in = (i0 + 4 * u) < nk4;
      const bool kin = in && k < a.K;
      av[u] = in ? wp[(size_t)k * 256] : 0.f;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) bv[u][cb] = kin ? xp[(size_t)k * a.ldt + cb * 16] : 0.f;
    }
    if constexpr (TR) {
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int k = 4 * (i0 + 4 * u) + kq;
        const bool kin = (i0 + 4 * u) < nk4 && k < a.K;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) bv[u][cb] = kin ? small_transform(a, bv[u][cb], k, slope) : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][cb], acc[cb], 0, 0, 0);
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

// ---- fused streaming-step forms --------------------------------------------------------------------------------
// The reduction above leaves every finishing thread with four CONSECUTIVE output rows of one frame.  Packing the weight
// rows so that those four rows belong together turns two elementwise kernels of the streaming step into epilogues:
//   FiLM   rows (2c, 2c+1) = (cond_scale row c, cond_bias row c)  ->  y[c] = scale * x[c] + bias      (x = the GEMM input)
//   cell   rows 4u .. 4u+3 = gates i, f, g, o of hidden unit u    ->  c' = sig(f) c + sig(i) tanh(g), h' = sig(o) tanh(c')
struct FusedArgs {
  SmallArgs g;
  float* c_state;  // cell: [N][H][ld_state], in place
  float* h_out;    // cell: [N][H][ld_state]
  int ld_state;
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

template <int NCB, int EPI>  // EPI 1 = FiLM, 2 = LSTM cell
__device__ __forceinline__ void small_fused_body(const FusedArgs& fa, const int n) {
  const SmallArgs& a = fa.g;
  __shared__ f32x4 part[4][NCB][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * 16;
  const int t0 = blockIdx.z * (NCB * 16);
  const int m = m0 + r;
  const float* wp = a.wt + ((size_t)(m >> 8) * a.Kp) * 256 + (m & 255);
  const float* xp = a.x + (size_t)n * a.K * a.ldt + t0 + r;
  f32x4 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The kernel is one chain of memory round trips on a nearly idle chip (the streaming step: 64 frames): what the
  // epilogue needs -- bias, per-frame term, the FiLM multiplicands / the cell state of this thread's (frame, four rows) --
  // is requested HERE, next to the GEMM operands, instead of behind the reduction (round 3: one round trip less per launch)
  constexpr int NE = (NCB * 64 + 255) / 256;  // epilogue elements per thread
  float e_bias[NE][4], e_res[NE][4], e_x[NE][2];
#pragma unroll
  for (int q = 0; q < NE; ++q) {
    const int idx = threadIdx.x + 256 * q, cb = idx >> 6, ln = idx & 63;
    const int t = t0 + cb * 16 + (ln & 15), mq = m0 + 4 * (ln >> 4);
    const bool on = idx < NCB * 64 && t < a.T && mq < a.M;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      e_bias[q][reg] = (on && a.bias) ? a.bias[mq + reg] : 0.f;
      e_res[q][reg] = (on && a.res) ? a.res[((size_t)n * a.M + mq + reg) * a.ldt + t] : 0.f;
    }
    if constexpr (EPI == 1) {
      const int c0 = mq >> 1;
      e_x[q][0] = on ? a.x[((size_t)n * a.K + c0) * a.ldt + t] : 0.f;
      e_x[q][1] = on ? a.x[((size_t)n * a.K + c0 + 1) * a.ldt + t] : 0.f;
    } else {
      const int u = mq >> 2, H = a.M >> 2;
      e_x[q][0] = on ? fa.c_state[((size_t)n * H + u) * fa.ld_state + t] : 0.f;
      e_x[q][1] = 0.f;
    }
  }
  const int nk4 = a.Kp / 4;
  // explicit batches: all loads of UN k-steps are issued before their MFMAs (the loop is latency bound)
  constexpr int UN = 8;
  for (int i0 = w; i0 < nk4; i0 += 4 * UN) {
    float av[UN], bv[UN][NCB];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = 4 * (i0 + 4 * u) + kq;
      const bool in = (i0 + 4 * u) < nk4;
      av[u] = in ? wp[(size_t)k * 256] : 0.f;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) bv[u][cb] = (in && k < a.K) ? xp[(size_t)k * a.ldt + cb * 16] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
        acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][cb], acc[cb], 0, 0, 0);
  }
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) part[w][cb][lane] = acc[cb];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NE; ++q) {
    const int idx = threadIdx.x + 256 * q;
    if (idx >= NCB * 64) continue;
    const int cb = idx >> 6, ln = idx & 63;
    f32x4 s = (part[0][cb][ln] + part[1][cb][ln]) + (part[2][cb][ln] + part[3][cb][ln]);
    const int t = t0 + cb * 16 + (ln & 15);
    const int mq = m0 + 4 * (ln >> 4);  // first of the four rows
    if (t >= a.T || mq >= a.M) continue;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      if (a.bias) s[reg] += e_bias[q][reg];
      if (a.res) s[reg] += e_res[q][reg];
    }
    if constexpr (EPI == 1) {
      const int c0 = mq >> 1, C = a.M >> 1;
      a.y[((size_t)n * C + c0) * a.ldt + t] = s[0] * e_x[q][0] + s[1];
      a.y[((size_t)n * C + c0 + 1) * a.ldt + t] = s[2] * e_x[q][1] + s[3];
    } else {
      const int u = mq >> 2, H = a.M >> 2;
      const size_t so = ((size_t)n * H + u) * fa.ld_state + t;
      const float cn = sigm(s[1]) * e_x[q][0] + sigm(s[0]) * tanhf(s[2]);
      fa.c_state[so] = cn;
      fa.h_out[so] = sigm(s[3]) * tanhf(cn);
    }
  }
}

template <int NCB, int EPI>
__global__ __launch_bounds__(256) void conv1x1_small_fused_kernel(FusedArgs fa) {
  small_fused_body<NCB, EPI>(fa, blockIdx.y);
}

// Several independent one-utterance launches of the kernel above as ONE launch (the streaming wavefront: the (hop, block)
// cells of an anti-diagonal share nothing but their shapes): blockIdx.y picks the cell's arguments.
constexpr int MAX_CELLS = PS_MAX_CELLS;
struct FusedCells {
  FusedArgs c[MAX_CELLS];
};

template <int NCB, int EPI>
__global__ __launch_bounds__(256) void conv1x1_small_fused_cells_kernel(FusedCells m) {
  small_fused_body<NCB, EPI>(m.c[blockIdx.y], 0);
}

// Projection + LayerNorm + residual (+ the NEXT block's input LayerNorm) of the streaming step: a workgroup owns ALL M
// output channels of 16 frames (waves split the channel blocks, each walks the whole K), the [M][16] tile meets in
// LDS and every frame is normalised over its M channels there:
//   y  = res + LN(W x + b; gamma, beta, eps)         y2 = LN(y; gamma2, beta2, eps2)   (optional)
struct ProjLnArgs {
  const float* x;
  const float* wt;
  const float* bias;
  const float* gamma;
  const float* beta;
  const float* res;
  float* y;
  const float* gamma2;
  const float* beta2;
  float* y2;
  float* x_copy;  // optional: the input rows are also written here (same layout), frames [0, T)
  int res_inside;  // 0: y = res + LN(W x + b)   1: y = LN(W x + b + res)  (post-norm transformer blocks)
  float eps, eps2;
  int K, Kp, M, T, ldt;
  int N;  // (the persistent row kernel walks utterances itself)
  float* y_amax;  // row kernel only: [N][4 * ceil(T / 128)] partial maxima of |y| (the next fp16x2 GEMM's input range)
};

constexpr int PLN_MAXM = 256;

// NRB: row blocks of 16 output channels: 8 (M <= 128) or 16 (M <= 256).  NW: waves that split K (4, or 8 for the short
// rows of the streaming step, where the kernel is one latency chain on a handful of CUs: twice the loads in flight,
// half the trips and half the MFMAs per wave).
template <int NRB, int NW>
__device__ __forceinline__ void proj_layernorm_body(const ProjLnArgs& a, const int n) {
  constexpr int NT = 64 * NW, CP = NT / 16;  // threads; channel parts of the LayerNorm phase
  __shared__ f32x4 part[NW][NRB][64];
  __shared__ float tile[NRB * 16][17];
  __shared__ float red[CP][17];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 15, kq = lane >> 4;
  const int t0 = blockIdx.x * 16;
  const float* xp = a.x + (size_t)n * a.K * a.ldt + t0 + r;
  // the four waves split K; every wave accumulates all row blocks of its K share
  f32x4 acc[NRB];
#pragma unroll
  for (int j = 0; j < NRB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk4 = a.Kp / 4;
  constexpr int UN = NRB == 8 ? 4 : 2;
  for (int i0 = w; i0 < nk4; i0 += NW * UN) {
    float bv[UN], av[UN][NRB];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = 4 * (i0 + NW * u) + kq;
      const bool in = (i0 + NW * u) < nk4;
      bv[u] = (in && k < a.K) ? xp[(size_t)k * a.ldt] : 0.f;
#pragma unroll
      for (int j = 0; j < NRB; ++j) {
        const int m = j * 16 + r;  // weights are zero padded to 256 rows per tile
        av[u][j] = in ? a.wt[((size_t)(m >> 8) * a.Kp + k) * 256 + (m & 255)] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = 4 * (i0 + NW * u) + kq;
      if (a.x_copy && (i0 + NW * u) < nk4 && k < a.K && t0 + r < a.T)
        a.x_copy[((size_t)n * a.K + k) * a.ldt + t0 + r] = bv[u];
#pragma unroll
      for (int j = 0; j < NRB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][j], bv[u], acc[j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < NRB; ++j) part[w][j][lane] = acc[j];
  __syncthreads();
  for (int idx = threadIdx.x; idx < NRB * 64; idx += NT) {
    const int j = idx >> 6, ln = idx & 63;
    f32x4 s = (part[0][j][ln] + part[1][j][ln]) + (part[2][j][ln] + part[3][j][ln]);
    if constexpr (NW == 8) s += (part[4][j][ln] + part[5][j][ln]) + (part[6][j][ln] + part[7][j][ln]);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int m = j * 16 + 4 * (ln >> 4) + reg;
      if (m < a.M) {
        float v = s[reg] + (a.bias ? a.bias[m] : 0.f);
        if (a.res_inside && a.res && t0 + (ln & 15) < a.T) v += a.res[((size_t)n * a.M + m) * a.ldt + t0 + (ln & 15)];
        tile[m][ln & 15] = v;
      }
    }
  }
  __syncthreads();
  // 16 frames x CP channel parts
  const int f = threadIdx.x & 15, cp = threadIdx.x >> 4;
  const int t = t0 + f;
  auto frame_stats = [&](float& mean, float& rstd, float eps) {
    float s = 0.f;
    for (int m = cp; m < a.M; m += CP) s += tile[m][f];
    red[cp][f] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int p = 0; p < CP; ++p) tot += red[p][f];
    mean = tot / (float)a.M;
    __syncthreads();
    float q = 0.f;
    for (int m = cp; m < a.M; m += CP) {
      const float dv = tile[m][f] - mean;
      q += dv * dv;
    }
    red[cp][f] = q;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int p = 0; p < CP; ++p) tot += red[p][f];
    rstd = 1.f / sqrtf(tot / (float)a.M + eps);
    __syncthreads();
  };
  float mean, rstd;
  frame_stats(mean, rstd, a.eps);
  for (int m = cp; m < a.M; m += CP) {
    float v = (tile[m][f] - mean) * rstd * a.gamma[m] + a.beta[m];
    const size_t off = ((size_t)n * a.M + m) * a.ldt + t;
    if (!a.res_inside && a.res && t < a.T) v += a.res[off];
    if (t < a.T) a.y[off] = v;
    tile[m][f] = v;
  }
  if (a.y2) {
    __syncthreads();
    frame_stats(mean, rstd, a.eps2);
    if (t < a.T)
      for (int m = cp; m < a.M; m += CP)
        a.y2[((size_t)n * a.M + m) * a.ldt + t] = (tile[m][f] - mean) * rstd * a.gamma2[m] + a.beta2[m];
  }
}

template <int NRB, int NW = 4>
__global__ __launch_bounds__(64 * NW) void proj_layernorm_kernel(ProjLnArgs a) {
  proj_layernorm_body<NRB, NW>(a, blockIdx.y);
}

struct ProjLnCells {  // (see FusedCells)
  ProjLnArgs c[MAX_CELLS];
};

template <int NRB, int NW = 4>
__global__ __launch_bounds__(64 * NW) void proj_layernorm_cells_kernel(ProjLnCells m) {
  proj_layernorm_body<NRB, NW>(m.c[blockIdx.y], 0);
}

// The same operator on LONG rows (the offline DPRNN / SkiM / DPCRN paths: every frame of a 32 x 4 s batch goes through
// it twice per block): the kernel above owns 16 frames per workgroup and moves 64-byte pieces -- 141 us per launch at
// 32 x 4000 frames, C = 128, H = 64 (163 MB: 1.2 TB/s).  Here a wave owns 32 frames x ALL output channels as NB
// 32 x 32 accumulator blocks of v_mfma_f32_32x32x2_f32 (exact fp32 products): lanes 0-31 / 32-63 hold two rows of 32
// consecutive frames, so every global access is two 128-byte pieces, the projection matrix sits in LDS ([k][m], rows
// padded by 32 floats: the two half-waves read different banks), and a frame's LayerNorm needs the lane's own registers
// and ONE exchange with lane ^ 32.  Two-pass variance as the reference.
// Two workgroups per CU for M <= 128: one wave per SIMD (the flat-address version's 418 registers) left the tile's two load
// latencies, its 128 MFMAs and its 64 stores strictly in sequence (67 us per launch on config 4's rows, 53 now).
template <int NB>
__global__ __launch_bounds__(256, NB == 4 ? 2 : 1) void proj_layernorm_rows_kernel(ProjLnArgs a) {
  constexpr int MB = NB * 32, LDW = MB + 32;
  extern __shared__ __attribute__((aligned(16))) float pl_smem[];
  float* wl = pl_smem;                 // [Kp][LDW]
  float* gl = wl + (size_t)a.Kp * LDW;  // gamma[MB] | beta[MB] | bias[MB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the buffer resources below depend on it)
  const int lr = lane & 31, lh = lane >> 5;
  // the projection matrix goes to LDS ONCE per (persistent) workgroup: 16-byte loads, eight in flight per thread (a
  // dword-at-a-time loop of dependent load / store pairs cost 20 us per workgroup -- more than its tiles)
  {
    constexpr int V4 = MB / 4;  // float4 pieces per k-row (the packer's rows are 256 floats: zero beyond M and K)
    const int nv = a.Kp * V4;
    for (int i0 = tid; i0 < nv; i0 += 256 * 8) {
      f32x4 wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u;
        const int k = i / V4, m4 = i % V4;
        wv[u] = i < nv ? *reinterpret_cast<const f32x4*>(a.wt + (size_t)k * 256 + 4 * m4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u;
        if (i < nv) *reinterpret_cast<f32x4*>(wl + (i / V4) * LDW + 4 * (i % V4)) = wv[u];
      }
    }
  }
  for (int m = tid; m < MB; m += 256) {
    gl[m] = m < a.M ? a.gamma[m] : 0.f;
    gl[MB + m] = m < a.M ? a.beta[m] : 0.f;
    gl[2 * MB + m] = (m < a.M && a.bias) ? a.bias[m] : 0.f;
  }
  __syncthreads();
  const int tb = (a.T + 127) / 128;  // 128-frame blocks per utterance
  for (int tile = blockIdx.x; tile < tb * a.N; tile += gridDim.x) {
  const int n = tile / tb, t0 = (tile % tb) * 128 + wave * 32;
  if (t0 >= a.T) {
    if (a.y_amax && lane == 0) a.y_amax[(size_t)n * tb * 4 + (tile % tb) * 4 + wave] = 0.f;
    continue;
  }
  const int t = t0 + lr;
  const bool live = t < a.T;
  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // addressing: buffer instructions -- one resource per utterance, the row as a scalar offset (compile-time channel x ldt),
  // ONE per-lane offset, masked lanes switched off by an out-of-range offset.  (Flat addresses cost this kernel 64-bit
  // vector arithmetic per access, an exec-mask branch around every guarded one and ~250 scalar registers spilled into
  // vector lanes: 2300 vector instructions per tile against 128 MFMAs.)
  constexpr unsigned OOB = 0x7ffffff0u;
  const int ldt = a.ldt;
  const unsigned vo_x = t < ldt ? (unsigned)(lh * ldt + t) * 4u : OOB;   // activation row k0 + 2u + lh
  const unsigned vo_y = live ? (unsigned)(4 * lh * ldt + t) * 4u : OOB;  // channel 32 j + (r & 3) + 8 (r >> 2) + 4 lh
  // M % 4 == 0 (the launcher's condition): the four channels 32 j + 8 rq + {0..3} + 4 lh of a lane lie on one side of M,
  // so 4 NB lane offsets (valid or out of range) serve all 16 NB residual loads and stores
  const int mlim = a.M - 4 * lh, klim = a.K - lh;
  unsigned vq[NB][4];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) vq[j][rq] = 32 * j + 8 * rq < mlim ? vo_y : OOB;
  const __amdgpu_buffer_rsrc_t xr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)n * a.K * ldt, 0, a.K * ldt * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.res ? a.res : a.y) + (size_t)n * a.M * ldt, 0, a.res ? a.M * ldt * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr =
      __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)n * a.M * ldt, 0, a.M * ldt * 4, 0x00020000);
  // the residual values of the tile are requested up front, next to the activations: behind the MFMAs they would be 16 NB
  // dependent loads per lane with nothing left to hide them (the first version: 40 us per tile)
  float rv[NB][16];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cu = 32 * j + (r & 3) + 8 * (r >> 2);
      rv[j][r] = __builtin_bit_cast(
          float, __builtin_amdgcn_raw_buffer_load_b32(rr, vq[j][r >> 2], cu * ldt * 4, 0));
    }
  constexpr int KU = 16;                                // k-pairs whose activation loads are in flight together
  for (int k0 = 0; k0 < a.Kp; k0 += 2 * KU) {
    float bv[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int ku = k0 + 2 * u;
      bv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, ku < klim ? vo_x : OOB, ku * ldt * 4, 0));
    }
#pragma unroll
    for (int u = 0; u < KU; ++u) {
      const int k = k0 + 2 * u + lh;
      if (k0 + 2 * u < a.Kp) {
        const float* wr = wl + k * LDW + lr;
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[j * 32], bv[u], acc[j], 0, 0, 0);
      }
    }
  }
  // element (j, r) of this lane: channel 32 j + (r & 3) + 8 (r >> 2) + 4 lh, frame t
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int c0 = 32 * j + 8 * rq + 4 * lh;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(gl + 2 * MB + c0);
#pragma unroll
      for (int r3 = 0; r3 < 4; ++r3) {
        // (channels >= M: zero weight rows, zero bias, residual read out of range -- v is 0 without a mask)
        float v = acc[j][rq * 4 + r3] + b4[r3];
        if (a.res_inside) v += rv[j][rq * 4 + r3];
        acc[j][rq * 4 + r3] = v;
        s += v;
      }
    }
  s += __shfl_xor(s, 32, 64);
  const float mean = s / (float)a.M;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float dv = vq[j][r >> 2] != OOB ? acc[j][r] - mean : 0.f;
      q += dv * dv;
    }
  q += __shfl_xor(q, 32, 64);
  const float rstd = 1.f / sqrtf(q / (float)a.M + a.eps);
  float amx = 0.f;
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int c0 = 32 * j + 8 * rq + 4 * lh;
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gl + c0), be4 = *reinterpret_cast<const f32x4*>(gl + MB + c0);
#pragma unroll
      for (int r3 = 0; r3 < 4; ++r3) {
        float v = (acc[j][rq * 4 + r3] - mean) * rstd * g4[r3] + be4[r3];
        if (!a.res_inside) v += rv[j][rq * 4 + r3];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, vq[j][rq],
                                              (32 * j + 8 * rq + r3) * ldt * 4, 0);
        amx = fmaxf(amx, vq[j][rq] != OOB ? fabsf(v) : 0.f);
      }
    }
  if (a.y_amax) {
    amx = wave_max(amx);
    if (lane == 0) a.y_amax[(size_t)n * tb * 4 + (tile % tb) * 4 + wave] = amx;
  }
  }  // tiles
}

// The same for K = 64 input channels and M = 128 (the projection behind a unidirectional 64-unit LSTM): a tile's 32
// k-pairs of activations fit in registers, so the loads of the NEXT tile are issued right behind the tile's last MFMA
// and land under its LayerNorm and stores, and the residual loads are issued in front of the MFMAs and land under them.
// All row accesses are buffer instructions: one resource per utterance, the row as a scalar offset (compile-time
// channel x ldt), ONE per-lane offset register, lanes past T switched off by an out-of-range offset -- the flat-address
// version of this kernel spent 2300 vector instructions per tile on 64-bit address arithmetic, exec-mask branches and
// scalar registers spilled into vector lanes, three times its MFMA issue time.  K = 64 and M = 128 exactly: no masks.
__global__ __launch_bounds__(256, 2) void proj_layernorm_rows64_kernel(ProjLnArgs a) {
  constexpr int NB = 4, MB = NB * 32, LDW = MB + 32, KP2 = 32;
  constexpr unsigned OOB = 0x7ffffff0u;  // past every resource's num_records: loads return 0, stores are dropped
  extern __shared__ __attribute__((aligned(16))) float pl_smem[];
  float* wl = pl_smem;                // [64][LDW]
  float* gl = wl + (size_t)64 * LDW;  // gamma[MB] | beta[MB] | bias[MB]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the buffer resources below depend on it)
  const int lr = lane & 31, lh = lane >> 5;
  const int ldt = a.ldt;
  const int tb = (a.T + 127) / 128;
  const int ntiles = tb * a.N;
  const int xslab = a.K * ldt * 4, yslab = a.M * ldt * 4;  // bytes per utterance (< 2^31: checked by the launcher)

  float bv[KP2];
  auto load_bv = [&](int tile) {  // activations of this wave's 32 frames of `tile`: row k = 2u + lh
    const bool ok = tile < ntiles;
    const int n = ok ? tile / tb : 0, t0 = (ok ? (tile % tb) * 128 : 0) + wave * 32, t = t0 + lr;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x) + (size_t)n * a.K * ldt, 0, (ok && t0 < a.T) ? xslab : 0, 0x00020000);
    // (frames beyond T inside the row are padding: computed, not stored; rows k >= K lie past num_records)
    const unsigned vo = t < ldt ? (unsigned)(lh * ldt + t) * 4u : OOB;
#pragma unroll
    for (int u = 0; u < KP2; ++u)
      bv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, vo, 2 * u * ldt * 4, 0));
  };
  load_bv(blockIdx.x);  // (in flight while the weights are staged)

  {
    constexpr int V4 = MB / 4;
    const int nv = 64 * V4;
    for (int i0 = tid; i0 < nv; i0 += 256 * 8) {
      f32x4 wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u;
        const int k = i / V4, m4 = i % V4;
        wv[u] = (i < nv && k < a.Kp) ? *reinterpret_cast<const f32x4*>(a.wt + (size_t)k * 256 + 4 * m4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u;
        if (i < nv) *reinterpret_cast<f32x4*>(wl + (i / V4) * LDW + 4 * (i % V4)) = wv[u];
      }
    }
  }
  for (int m = tid; m < MB; m += 256) {
    gl[m] = m < a.M ? a.gamma[m] : 0.f;
    gl[MB + m] = m < a.M ? a.beta[m] : 0.f;
    gl[2 * MB + m] = (m < a.M && a.bias) ? a.bias[m] : 0.f;
  }
  __syncthreads();

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n = tile / tb, t0 = (tile % tb) * 128 + wave * 32;
    if (t0 >= a.T) {  // (wave-uniform; its activations were requested from an empty resource)
      if (a.y_amax && lane == 0) a.y_amax[(size_t)n * tb * 4 + (tile % tb) * 4 + wave] = 0.f;
      load_bv(tile + gridDim.x);
      continue;
    }
    const int t = t0 + lr;
    const bool live = t < a.T;
    const unsigned vo = live ? (unsigned)(4 * lh * ldt + t) * 4u : OOB;  // channel 32 j + (r & 3) + 8 (r >> 2) + 4 lh
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.res ? a.res : a.y) + (size_t)n * a.M * ldt, 0, a.res ? yslab : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr =
        __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)n * a.M * ldt, 0, yslab, 0x00020000);
    float rv[NB][16];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = 32 * j + (r & 3) + 8 * (r >> 2);
        rv[j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, vo, cu * ldt * 4, 0));
      }
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int u = 0; u < KP2; ++u) {
      const float* wr = wl + (2 * u + lh) * LDW + lr;
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[j * 32], bv[u], acc[j], 0, 0, 0);
    }
    load_bv(tile + gridDim.x);

    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int c0 = 32 * j + 8 * rq + 4 * lh;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(gl + 2 * MB + c0);
#pragma unroll
        for (int r3 = 0; r3 < 4; ++r3) {
          float v = acc[j][rq * 4 + r3] + b4[r3];
          if (a.res_inside) v += rv[j][rq * 4 + r3];
          acc[j][rq * 4 + r3] = v;
          s += v;
        }
      }
    s += __shfl_xor(s, 32, 64);
    const float mean = s / (float)a.M;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float dv = acc[j][r] - mean;
        q += dv * dv;
      }
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.f / sqrtf(q / (float)a.M + a.eps);
    float amx = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int c0 = 32 * j + 8 * rq + 4 * lh;
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(gl + c0), be4 = *reinterpret_cast<const f32x4*>(gl + MB + c0);
#pragma unroll
        for (int r3 = 0; r3 < 4; ++r3) {
          float v = (acc[j][rq * 4 + r3] - mean) * rstd * g4[r3] + be4[r3];
          if (!a.res_inside) v += rv[j][rq * 4 + r3];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, vo, (32 * j + 8 * rq + r3) * ldt * 4, 0);
          amx = fmaxf(amx, live ? fabsf(v) : 0.f);
        }
      }
    if (a.y_amax) {
      amx = wave_max(amx);
      if (lane == 0) a.y_amax[(size_t)n * tb * 4 + (tile % tb) * 4 + wave] = amx;
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

using namespace ps;

static int small_status(const char* who) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", who, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

static int fused_launch(const char* who, int epi, FusedArgs& fa, int N, hipStream_t stream) {
  const SmallArgs& a = fa.g;
  const int ncb = a.T <= 16 ? 1 : a.T <= 32 ? 2 : 4;
  dim3 grid((a.M + 15) / 16, N, (a.T + ncb * 16 - 1) / (ncb * 16));
  LaunchTimer timer(who, stream);
#define PS_FUSED(NCB)                                                                              \
  if (epi == 1)                                                                                    \
    hipLaunchKernelGGL((conv1x1_small_fused_kernel<NCB, 1>), grid, dim3(256), 0, stream, fa);      \
  else                                                                                             \
    hipLaunchKernelGGL((conv1x1_small_fused_kernel<NCB, 2>), grid, dim3(256), 0, stream, fa);
  if (ncb == 1) {
    PS_FUSED(1)
  } else if (ncb == 2) {
    PS_FUSED(2)
  } else {
    PS_FUSED(4)
  }
#undef PS_FUSED
  return 0;
}

extern "C" int ps_film_conv_f32(const float* x, const float* wt_pairs, const float* res_pairs, float* y, int N, int C,
                                int T, int ldt, void* stream) {
  if (!x || !wt_pairs || !y || N <= 0 || C <= 0 || C % 2 || T <= 0 || ldt < T || N > 65535) {
    set_error("ps_film_conv_f32: bad argument (N=%d C=%d T=%d ldt=%d; C must be even)", N, C, T, ldt);
    return PS_E_INVALID;
  }
  FusedArgs fa{};
  fa.g.x = x;
  fa.g.wt = wt_pairs;
  fa.g.y = y;
  fa.g.res = res_pairs;
  fa.g.K = C;
  fa.g.Kp = (C + 15) / 16 * 16;
  fa.g.M = 2 * C;
  fa.g.T = T;
  fa.g.ldt = ldt;
  fused_launch("film_conv", 1, fa, N, (hipStream_t)stream);
  return small_status("ps_film_conv_f32");
}

extern "C" int ps_lstm_gates_cell_f32(const float* xh, const float* wt_units, const float* bias_units, float* c,
                                      float* h, int N, int K, int H, int T, int ldt, int ld_state, void* stream) {
  if (!xh || !wt_units || !c || !h || N <= 0 || K <= 0 || H <= 0 || T <= 0 || ldt < T || ld_state < T || N > 65535) {
    set_error("ps_lstm_gates_cell_f32: bad argument (N=%d K=%d H=%d T=%d)", N, K, H, T);
    return PS_E_INVALID;
  }
  FusedArgs fa{};
  fa.g.x = xh;
  fa.g.wt = wt_units;
  fa.g.bias = bias_units;
  fa.g.K = K;
  fa.g.Kp = (K + 15) / 16 * 16;
  fa.g.M = 4 * H;
  fa.g.T = T;
  fa.g.ldt = ldt;
  fa.c_state = c;
  fa.h_out = h;
  fa.ld_state = ld_state;
  fused_launch("lstm_gates_cell", 2, fa, N, (hipStream_t)stream);
  return small_status("ps_lstm_gates_cell_f32");
}

// ---- the cells of a streaming anti-diagonal as one launch each (N = 1 per cell, shapes shared) -------------------------
static int fused_cells_launch(const char* who, int epi, const FusedCells& fc, int ncells, hipStream_t stream) {
  const SmallArgs& a = fc.c[0].g;
  const int ncb = a.T <= 16 ? 1 : a.T <= 32 ? 2 : 4;
  dim3 grid((a.M + 15) / 16, ncells, (a.T + ncb * 16 - 1) / (ncb * 16));
  LaunchTimer timer(who, stream);
#define PS_FUSED(NCB)                                                                                    \
  if (epi == 1)                                                                                          \
    hipLaunchKernelGGL((conv1x1_small_fused_cells_kernel<NCB, 1>), grid, dim3(256), 0, stream, fc);      \
  else                                                                                                   \
    hipLaunchKernelGGL((conv1x1_small_fused_cells_kernel<NCB, 2>), grid, dim3(256), 0, stream, fc);
  if (ncb == 1) {
    PS_FUSED(1)
  } else if (ncb == 2) {
    PS_FUSED(2)
  } else {
    PS_FUSED(4)
  }
#undef PS_FUSED
  return small_status(who);
}

extern "C" int ps_film_conv_cells_f32(const ps_film_cell* cells, int ncells, int C, int T, int ldt, void* stream) {
  if (!cells || ncells <= 0 || ncells > PS_MAX_CELLS || C <= 0 || C % 2 || T <= 0 || ldt < T) {
    set_error("ps_film_conv_cells_f32: bad argument (ncells=%d of at most %d, C=%d T=%d ldt=%d; C must be even)", ncells,
              PS_MAX_CELLS, C, T, ldt);
    return PS_E_INVALID;
  }
  FusedCells fc{};
  for (int i = 0; i < ncells; ++i) {
    if (!cells[i].x || !cells[i].wt_pairs || !cells[i].y) {
      set_error("ps_film_conv_cells_f32: cell %d has a null pointer", i);
      return PS_E_INVALID;
    }
    SmallArgs& g = fc.c[i].g;
    g.x = cells[i].x, g.wt = cells[i].wt_pairs, g.y = cells[i].y, g.res = cells[i].res_pairs;
    g.K = C, g.Kp = (C + 15) / 16 * 16, g.M = 2 * C, g.T = T, g.ldt = ldt;
  }
  return fused_cells_launch("ps_film_conv_cells_f32", 1, fc, ncells, (hipStream_t)stream);
}

extern "C" int ps_lstm_gates_cell_cells_f32(const ps_gates_cell* cells, int ncells, int K, int H, int T, int ldt,
                                            int ld_state, void* stream) {
  if (!cells || ncells <= 0 || ncells > PS_MAX_CELLS || K <= 0 || H <= 0 || T <= 0 || ldt < T || ld_state < T) {
    set_error("ps_lstm_gates_cell_cells_f32: bad argument (ncells=%d of at most %d, K=%d H=%d T=%d)", ncells, PS_MAX_CELLS,
              K, H, T);
    return PS_E_INVALID;
  }
  FusedCells fc{};
  for (int i = 0; i < ncells; ++i) {
    if (!cells[i].xh || !cells[i].wt_units || !cells[i].c || !cells[i].h) {
      set_error("ps_lstm_gates_cell_cells_f32: cell %d has a null pointer", i);
      return PS_E_INVALID;
    }
    FusedArgs& fa = fc.c[i];
    fa.g.x = cells[i].xh, fa.g.wt = cells[i].wt_units, fa.g.bias = cells[i].bias_units;
    fa.g.K = K, fa.g.Kp = (K + 15) / 16 * 16, fa.g.M = 4 * H, fa.g.T = T, fa.g.ldt = ldt;
    fa.c_state = cells[i].c, fa.h_out = cells[i].h, fa.ld_state = ld_state;
  }
  return fused_cells_launch("ps_lstm_gates_cell_cells_f32", 2, fc, ncells, (hipStream_t)stream);
}

extern "C" int ps_proj_layernorm_cells_f32(const ps_projln_cell* cells, int ncells, int res_inside, int K, int M, int T,
                                           int ldt, void* stream) {
  if (!cells || ncells <= 0 || ncells > PS_MAX_CELLS || K <= 0 || M <= 0 || T <= 0 || ldt < T) {
    set_error("ps_proj_layernorm_cells_f32: bad argument (ncells=%d of at most %d, K=%d M=%d T=%d)", ncells, PS_MAX_CELLS,
              K, M, T);
    return PS_E_INVALID;
  }
  if (M > PLN_MAXM) {
    set_error("ps_proj_layernorm_cells_f32: M=%d > %d output channels", M, PLN_MAXM);
    return PS_E_UNSUPPORTED;
  }
  ProjLnCells pc{};
  for (int i = 0; i < ncells; ++i) {
    const ps_projln_cell& c = cells[i];
    if (!c.x || !c.wt || !c.gamma || !c.beta || !c.y || (c.y2 && (!c.gamma2 || !c.beta2))) {
      set_error("ps_proj_layernorm_cells_f32: cell %d has a null pointer", i);
      return PS_E_INVALID;
    }
    pc.c[i] = ProjLnArgs{c.x, c.wt, c.bias, c.gamma, c.beta, c.res, c.y, c.gamma2, c.beta2, c.y2, c.x_copy, res_inside,
                         c.eps, c.eps2, K, (K + 15) / 16 * 16, M, T, ldt, 1, nullptr};
  }
  hipStream_t s = (hipStream_t)stream;
  LaunchTimer timer("proj_layernorm", s);
  const dim3 grid((T + 15) / 16, ncells);
  // (the same choice as ps_proj_layernorm_f32 makes for one cell: results stay bit-identical to the per-cell launches)
  if (M <= 128 && (long long)((T + 15) / 16) <= 64)
    hipLaunchKernelGGL((proj_layernorm_cells_kernel<8, 8>), grid, dim3(512), 0, s, pc);
  else if (M <= 128)
    hipLaunchKernelGGL((proj_layernorm_cells_kernel<8>), grid, dim3(256), 0, s, pc);
  else
    hipLaunchKernelGGL((proj_layernorm_cells_kernel<16>), grid, dim3(256), 0, s, pc);
  return small_status("ps_proj_layernorm_cells_f32");
}

extern "C" int ps_proj_layernorm_amax_parts(int T) { return T > 0 ? 4 * ((T + 127) / 128) : 0; }

extern "C" int ps_proj_layernorm_f32(const float* x, const float* wt, const float* bias, const float* gamma,
                                     const float* beta, float eps, const float* res, float* y, const float* gamma2,
                                     const float* beta2, float eps2, float* y2, float* x_copy, int res_inside, int N,
                                     int K, int M, int T, int ldt, void* stream) {
  return ps_proj_layernorm_amax_f32(x, wt, bias, gamma, beta, eps, res, y, gamma2, beta2, eps2, y2, x_copy, res_inside, N, K,
                                    M, T, ldt, nullptr, stream);
}

extern "C" int ps_proj_layernorm_amax_f32(const float* x, const float* wt, const float* bias, const float* gamma,
                                          const float* beta, float eps, const float* res, float* y, const float* gamma2,
                                          const float* beta2, float eps2, float* y2, float* x_copy, int res_inside, int N,
                                          int K, int M, int T, int ldt, float* y_amax, void* stream) {
  if (!x || !wt || !gamma || !beta || !y || N <= 0 || K <= 0 || M <= 0 || T <= 0 || ldt < T || N > 65535 ||
      (y2 && (!gamma2 || !beta2))) {
    set_error("ps_proj_layernorm_f32: bad argument (N=%d K=%d M=%d T=%d)", N, K, M, T);
    return PS_E_INVALID;
  }
  if (M > PLN_MAXM) {
    set_error("ps_proj_layernorm_f32: M=%d > %d output channels", M, PLN_MAXM);
    return PS_E_UNSUPPORTED;
  }
  ProjLnArgs a{x, wt, bias, gamma, beta, res, y, gamma2, beta2, y2, x_copy, res_inside, eps, eps2, K, (K + 15) / 16 * 16, M, T, ldt, N, y_amax};
  // long rows without the streaming step's extras: the row kernel (ps_debug_flags bit 4 keeps the 16-frame kernel: tests
  // compare the two on the same data)
  const int kp = (K + 15) / 16 * 16;
  const int nb = M <= 128 ? 4 : 8;
  const size_t lds = ((size_t)kp * (nb * 32 + 32) + 3 * nb * 32) * sizeof(float);
  if (!y2 && !x_copy && T >= 128 && M % 4 == 0 && lds <= 128 * 1024 && (long long)(M > K ? M : K) * ldt * 4 < (1ll << 31) &&
      !(g_debug_flags & 16)) {
    LaunchTimer timer("proj_layernorm", (hipStream_t)stream);
    const long long tiles = (long long)((T + 127) / 128) * N;
    // one persistent workgroup per CU (the kernel holds a tile's accumulators AND its residual values: 256 registers, one
    // wave per SIMD): the projection matrix is staged once, not once per tile
    // persistent: as many workgroups as are resident (two per CU by registers for M <= 128, one by LDS for wide K)
    const long long per_cu = nb == 4 ? (lds <= 76 * 1024 ? 2 : 1) : 1;
    const long long slots = (long long)device_cus() * per_cu;
    dim3 grid((unsigned)(tiles < slots ? tiles : slots));
    if (K == 64 && M == 128 && !(g_debug_flags & (1 << 21))) {  // (debug bit 21: the unpipelined kernel, for the tests)
      const long long slots2 = (long long)device_cus() * 2;  // (three per CU, 168 registers and 11 spills: 44.9 us against 42.5)
      const size_t lds64 = ((size_t)64 * (4 * 32 + 32) + 3 * 4 * 32) * sizeof(float);
      hipLaunchKernelGGL(proj_layernorm_rows64_kernel, dim3((unsigned)(tiles < slots2 ? tiles : slots2)), dim3(256), lds64,
                         (hipStream_t)stream, a);
    } else {
      static const bool big_lds = [] {  // dynamic LDS beyond 64 KiB has to be asked for, once per kernel
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&proj_layernorm_rows_kernel<4>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(&proj_layernorm_rows_kernel<8>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
      }();
      if (!big_lds && lds > 64 * 1024) {
        set_error("ps_proj_layernorm_f32: %zu bytes of LDS refused by the runtime", lds);
        return PS_E_UNSUPPORTED;
      }
      if (nb == 4)
        hipLaunchKernelGGL((proj_layernorm_rows_kernel<4>), grid, dim3(256), lds, (hipStream_t)stream, a);
      else
        hipLaunchKernelGGL((proj_layernorm_rows_kernel<8>), grid, dim3(256), lds, (hipStream_t)stream, a);
    }
    return small_status("ps_proj_layernorm_f32");
  }
  if (y_amax) {
    set_error("ps_proj_layernorm_amax_f32: the maxima are an output of the row kernel only (T >= 128, no second norm / copy, "
              "K * M within its LDS budget); got N=%d K=%d M=%d T=%d", N, K, M, T);
    return PS_E_UNSUPPORTED;
  }
  {
    LaunchTimer timer("proj_layernorm", (hipStream_t)stream);
    // few workgroups (the streaming step: one per 16 streams): the kernel is a latency chain, eight waves split K
    if (M <= 128 && (long long)N * ((T + 15) / 16) <= 64)
      hipLaunchKernelGGL((proj_layernorm_kernel<8, 8>), dim3((T + 15) / 16, N), dim3(512), 0, (hipStream_t)stream, a);
    else if (M <= 128)
      hipLaunchKernelGGL((proj_layernorm_kernel<8>), dim3((T + 15) / 16, N), dim3(256), 0, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL((proj_layernorm_kernel<16>), dim3((T + 15) / 16, N), dim3(256), 0, (hipStream_t)stream, a);
  }
  return small_status("ps_proj_layernorm_f32");
}
