// Recurrent-masker kernels (DPRNN, SkiM, StreamingSkiM of mcw519/PureSound) on the padded channel-major layout
// [N][C][ldt] the rest of the library uses, so every Linear / 1x1 conv of these models is a ps_conv1x1_f32 call:
//
//   ps_lstm_f32            the recurrence of a 1-layer nn.LSTM (both directions) over gate pre-activations that a
//                          1x1 conv already produced (W_ih x + b_ih + b_hh); dprnn.py:155-171, skim.py:215-222
//   ps_chan_layernorm_f32  nn.LayerNorm(C) / ChanLN over the channels of each frame with the residual add, PReLU,
//                          sigmoid and gating product that follow it in the reference; dprnn.py:157-172,
//                          skim.py:85-98,226, lobe/trivial.py:61-126,160
//   ps_film_apply_f32      FiLM modulation scale * x + bias; lobe/trivial.py:162-167
//
// LSTM recurrence.  One workgroup owns LS = 4 sequences and all 4H gate rows (thread g = gate row).  W_hh^T stays in
// registers (H <= 64) or is streamed from L2 (coalesced [k][g] rows), h_{t-1} is broadcast from LDS, the four gate
// pre-activations of a unit meet in LDS for the cell update.  A sequence is a strided walk through the frame axis
// (frame = q*q_stride + step*step_stride), which covers the intra-segment pass (q = segment, steps = K
// contiguous frames), the inter-segment pass (q = position in segment, steps = S frames K apart) and the
// streaming step (q = stream, steps = 1).  Gate loads are issued LSTM_PF steps ahead.
#include "ps_common.h"

namespace ps {

constexpr int LS = 4;        // sequences per workgroup
constexpr int LSTM_PF = 8;   // gate pre-activation prefetch distance (steps)
constexpr int LSTM_WREG = 64;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

struct LstmK {
  ps_lstm_args a;
};

template <bool WREG>
__global__ __launch_bounds__(WREG ? 256 : 1024) void lstm_kernel(LstmK k) {
  extern __shared__ float smem[];
  const ps_lstm_args& a = k.a;
  const int H = a.H, G = 4 * H;
  f32x4* hbuf = reinterpret_cast<f32x4*>(smem);          // [H]  current h of the LS sequences
  f32x4* abuf = hbuf + H;                                  // [4H] gate pre-activations
  const int g = threadIdx.x;
  const int q0 = blockIdx.x * LS, n = blockIdx.y, d = blockIdx.z;
  const bool row = g < G;
  const int DH = a.D * H;

  // recurrent weights of gate row g: column g of W_hh^T [H][4H]
  const float* wt = a.whh_t + (size_t)d * H * G + g;
  float w[LSTM_WREG];
  if constexpr (WREG) {
#pragma unroll
    for (int kk = 0; kk < LSTM_WREG; ++kk) w[kk] = (row && kk < H) ? wt[(size_t)kk * G] : 0.f;
  }

  // cell-update role of this thread: unit j of sequence i (threads [0, H*LS))
  const int j = g % H, i = g / H;
  const bool cell = g < H * LS && q0 + i < a.Q;
  float c = 0.f, h = 0.f;
  if (cell && (a.h0 || a.c0)) {
    const int b = n * a.Q + q0 + i - a.state_shift;
    if (b >= 0) {
      const int nn = b / a.Q, qq = b % a.Q;
      const size_t off = ((size_t)nn * DH + d * H + j) * a.ldq + qq;
      if (a.h0) h = a.h0[off];
      if (a.c0) c = a.c0[off];
    }
  }
  if (g < H * LS) reinterpret_cast<float*>(hbuf)[j * LS + i] = (q0 + i < a.Q) ? h : 0.f;

  // gate pre-activation addressing of row g
  const float* grow = a.gx + ((size_t)n * a.D * G + (size_t)d * G + g) * a.ldt;
  float* hrow = a.hout + ((size_t)n * DH + d * H + j) * a.ldt;
  const int steps = a.steps;
  const bool rev = d == 1;
  float pre[LSTM_PF][LS];
#pragma unroll
  for (int u = 0; u < LSTM_PF; ++u) {
    const int s = u;
    const int ts = rev ? steps - 1 - s : s;
#pragma unroll
    for (int e = 0; e < LS; ++e)
      pre[u][e] = (row && s < steps && q0 + e < a.Q) ? grow[(size_t)(q0 + e) * a.q_stride + (size_t)ts * a.step_stride]
                                                     : 0.f;
  }
  __syncthreads();

  for (int s0 = 0; s0 < steps; s0 += LSTM_PF) {
#pragma unroll
    for (int u = 0; u < LSTM_PF; ++u) {
      const int s = s0 + u;
      if (s < steps) {  // uniform
        f32x4 acc{pre[u][0], pre[u][1], pre[u][2], pre[u][3]};
        {  // refill this ring slot with step s + LSTM_PF
          const int sn = s + LSTM_PF;
          const int ts = rev ? steps - 1 - sn : sn;
#pragma unroll
          for (int e = 0; e < LS; ++e)
            pre[u][e] = (row && sn < steps && q0 + e < a.Q)
                            ? grow[(size_t)(q0 + e) * a.q_stride + (size_t)ts * a.step_stride]
                            : 0.f;
        }
        if constexpr (WREG) {
#pragma unroll
          for (int kk = 0; kk < LSTM_WREG; ++kk) {
            if (kk < H) {
              const f32x4 hv = hbuf[kk];
              acc += w[kk] * hv;
            }
          }
        } else {
          if (row) {
#pragma unroll 8
            for (int kk = 0; kk < H; ++kk) {
              const f32x4 hv = hbuf[kk];
              acc += wt[(size_t)kk * G] * hv;
            }
          }
        }
        if (row) abuf[g] = acc;
        __syncthreads();
        if (g < H * LS) {
          const float* ab = reinterpret_cast<const float*>(abuf);
          const float gi = sigmoidf_(ab[(j)*LS + i]);
          const float gf = sigmoidf_(ab[(H + j) * LS + i]);
          const float gg = tanhf(ab[(2 * H + j) * LS + i]);
          const float go = sigmoidf_(ab[(3 * H + j) * LS + i]);
          c = gf * c + gi * gg;
          h = go * tanhf(c);
          reinterpret_cast<float*>(hbuf)[j * LS + i] = h;
          if (cell) {
            const int ts = rev ? steps - 1 - s : s;
            hrow[(size_t)(q0 + i) * a.q_stride + (size_t)ts * a.step_stride] = h;
          }
        }
        __syncthreads();
      }
    }
  }
  if (cell) {
    const size_t off = ((size_t)n * DH + d * H + j) * a.ldq + q0 + i;
    if (a.h_last) a.h_last[off] = h;
    if (a.c_last) a.c_last[off] = c;
  }
}

// ---- LayerNorm over channels ---------------------------------------------------------------------------------
struct ClnArgs {
  const float* x;
  const float* gamma;
  const float* beta;
  const float* res;
  const float* slope;
  const float* mul;
  float* y;
  float eps;
  int sigmoid;
  int C, T, ldt;
};

// 64 frames x 4 channel quarters per workgroup; three passes over the (L1/L2 resident) 64 x C tile: mean,
// centred second moment (the reference's two-pass variance), normalise + epilogue.
__global__ __launch_bounds__(256) void chan_layernorm_kernel(ClnArgs a) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane, n = blockIdx.y;
  const bool live = t < a.T;
  const size_t base = (size_t)n * a.C * a.ldt + (live ? t : 0);
  float s = 0.f;
  for (int ch = part; ch < a.C; ch += 4) s += live ? a.x[base + (size_t)ch * a.ldt] : 0.f;
  red[part][lane] = s;
  __syncthreads();
  const float mean = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) / (float)a.C;
  __syncthreads();
  float q = 0.f;
  for (int ch = part; ch < a.C; ch += 4) {
    const float dv = live ? a.x[base + (size_t)ch * a.ldt] - mean : 0.f;
    q += dv * dv;
  }
  red[part][lane] = q;
  __syncthreads();
  const float var = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) / (float)a.C;
  const float rstd = 1.f / sqrtf(var + a.eps);
  if (!live) return;
  const float slope = a.slope ? a.slope[0] : 1.f;
  for (int ch = part; ch < a.C; ch += 4) {
    const size_t off = base + (size_t)ch * a.ldt;
    float v = (a.x[off] - mean) * rstd * a.gamma[ch] + a.beta[ch];
    if (a.slope) v = prelu(v, slope);
    if (a.sigmoid) v = sigmoidf_(v);
    if (a.mul) v *= a.mul[off];
    if (a.res) v += a.res[off];
    a.y[off] = v;
  }
}

__global__ __launch_bounds__(256) void film_apply_kernel(const float* __restrict__ x, const float* __restrict__ sb,
                                                         float* __restrict__ y, int C, int T, int ldt) {
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int ch = blockIdx.y, n = blockIdx.z;
  if (t >= T) return;
  const size_t xo = ((size_t)n * C + ch) * ldt + t;
  const size_t so = ((size_t)n * 2 * C + ch) * ldt + t;
  const size_t bo = so + (size_t)C * ldt;
  const f32x4 xv = *reinterpret_cast<const f32x4*>(x + xo);
  const f32x4 sv = *reinterpret_cast<const f32x4*>(sb + so);
  const f32x4 bv = *reinterpret_cast<const f32x4*>(sb + bo);
  *reinterpret_cast<f32x4*>(y + xo) = sv * xv + bv;
}

static int launch_status(const char* who) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", who, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // namespace ps

using namespace ps;

extern "C" int ps_lstm_f32(const ps_lstm_args* args, void* stream) {
  if (!args) {
    set_error("ps_lstm_f32: null args");
    return PS_E_INVALID;
  }
  const ps_lstm_args& a = *args;
  if (!a.gx || !a.whh_t || !a.hout || a.N <= 0 || a.H <= 0 || a.D < 1 || a.D > 2 || a.Q <= 0 || a.steps <= 0 ||
      a.q_stride < 0 || a.step_stride < 0 || a.ldt <= 0 || a.N > 65535) {
    set_error("ps_lstm_f32: bad argument (N=%d H=%d D=%d Q=%d steps=%d)", a.N, a.H, a.D, a.Q, a.steps);
    return PS_E_INVALID;
  }
  if (4 * a.H > 1024) {
    set_error("ps_lstm_f32: hidden size %d > 256 is not supported", a.H);
    return PS_E_UNSUPPORTED;
  }
  const long long last = (long long)(a.Q - 1) * a.q_stride + (long long)(a.steps - 1) * a.step_stride;
  if (last >= a.ldt) {
    set_error("ps_lstm_f32: the last frame %lld lies outside the row (ldt=%d)", last, a.ldt);
    return PS_E_INVALID;
  }
  if ((a.h0 || a.c0 || a.h_last || a.c_last) && a.ldq < a.Q) {
    set_error("ps_lstm_f32: ldq=%d < Q=%d", a.ldq, a.Q);
    return PS_E_INVALID;
  }
  if (a.state_shift != 0 && a.state_shift != 1) {
    set_error("ps_lstm_f32: state_shift must be 0 or 1");
    return PS_E_INVALID;
  }
  LstmK k{a};
  const int threads = (4 * a.H + 63) / 64 * 64;
  const size_t lds = (size_t)5 * a.H * sizeof(f32x4);
  dim3 grid((a.Q + LS - 1) / LS, a.N, a.D);
  {
    LaunchTimer timer("lstm", (hipStream_t)stream);
    if (a.H <= LSTM_WREG)
      hipLaunchKernelGGL((lstm_kernel<true>), grid, dim3(threads), lds, (hipStream_t)stream, k);
    else
      hipLaunchKernelGGL((lstm_kernel<false>), grid, dim3(threads), lds, (hipStream_t)stream, k);
  }
  return launch_status("ps_lstm_f32");
}

extern "C" int ps_chan_layernorm_f32(const float* x, const float* gamma, const float* beta, float eps,
                                     const float* prelu_slope, int sigmoid, const float* mul, const float* res,
                                     float* y, int N, int C, int T, int ldt, void* stream) {
  if (!x || !gamma || !beta || !y || N <= 0 || C <= 0 || T <= 0 || ldt < T || N > 65535) {
    set_error("ps_chan_layernorm_f32: bad argument (N=%d C=%d T=%d ldt=%d)", N, C, T, ldt);
    return PS_E_INVALID;
  }
  ClnArgs a{x, gamma, beta, res, prelu_slope, mul, y, eps, sigmoid, C, T, ldt};
  {
    LaunchTimer timer("chan_layernorm", (hipStream_t)stream);
    hipLaunchKernelGGL(chan_layernorm_kernel, dim3((T + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, a);
  }
  return launch_status("ps_chan_layernorm_f32");
}

extern "C" int ps_film_apply_f32(const float* x, const float* scale_bias, float* y, int N, int C, int T, int ldt,
                                 void* stream) {
  if (!x || !scale_bias || !y || N <= 0 || C <= 0 || T <= 0 || ldt < T || C > 65535 || N > 65535) {
    set_error("ps_film_apply_f32: bad argument (N=%d C=%d T=%d ldt=%d)", N, C, T, ldt);
    return PS_E_INVALID;
  }
  if (ldt % 4 || ((uintptr_t)x & 15) || ((uintptr_t)scale_bias & 15) || ((uintptr_t)y & 15)) {
    set_error("ps_film_apply_f32: rows must be 16-byte aligned");
    return PS_E_ALIGN;
  }
  {
    LaunchTimer timer("film_apply", (hipStream_t)stream);
    hipLaunchKernelGGL(film_apply_kernel, dim3((T + 1023) / 1024, C, N), dim3(256), 0, (hipStream_t)stream, x,
                       scale_bias, y, C, T, ldt);
  }
  return launch_status("ps_film_apply_f32");
}
