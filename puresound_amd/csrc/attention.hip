// Self-attention of the DPARN bottleneck (MhaSelfAttenLayer / nn.MultiheadAttention, lobe/attention.py:38-232 of
// mcw519/PureSound) on the library's rows.  Q, K, V come from ONE ps_conv1x1_f32 over the in_proj weight as
// [N][3E][ld'] rows (q rows, then k rows, then v rows; head h = rows h*dh .. h*dh+dh-1 of each third).  A sequence is the
// same strided walk over the frame axis the LSTM kernel uses (frame = q*q_stride + pos*pos_stride): DPARN attends along
// frequency for every frame, i.e. q = t, q_stride = 1, L = F positions, pos_stride = ld.
//
// One workgroup = one head of ATT_SQ consecutive sequences; thread (sequence, query position).  K and V of the head are
// staged in LDS ([d][key][sequence]); a thread walks the keys in blocks with a running maximum / normaliser (online
// softmax), so L is not bounded by registers.  Sequences are short here (L = 64, dh = 16) and the attention is < 1 % of
// the block's FLOPs: the kernel is written for clarity, the GEMMs around it do the work.
#include "ps_common.h"

namespace ps {

constexpr int ATT_SQ = 4;     // sequences per workgroup
constexpr int ATT_MAXD = 64;  // head dimension

struct AttArgs {
  const float* qkv;
  float* out;
  int E, heads, Q, q_stride, L, pos_stride, ld, causal;
  float scale;
};

__global__ __launch_bounds__(ATT_SQ * 64) void self_attention_kernel(AttArgs a) {
  extern __shared__ float sm[];  // k[dh][L][SQ] | v[dh][L][SQ]
  const int dh = a.E / a.heads;
  const int h = blockIdx.y, n = blockIdx.z;
  const int q0 = blockIdx.x * ATT_SQ;
  const int lanes = blockDim.x / ATT_SQ;  // query positions handled per pass
  float* ks = sm;
  float* vs = sm + (size_t)dh * a.L * ATT_SQ;
  const float* base = a.qkv + (size_t)n * 3 * a.E * a.ld;
  // stage K and V of this head for the ATT_SQ sequences
  for (int idx = threadIdx.x; idx < dh * a.L * ATT_SQ; idx += blockDim.x) {
    const int s = idx % ATT_SQ, pos = (idx / ATT_SQ) % a.L, d = idx / (ATT_SQ * a.L);
    const int q = q0 + s;
    float kv = 0.f, vv = 0.f;
    if (q < a.Q) {
      const size_t fr = (size_t)q * a.q_stride + (size_t)pos * a.pos_stride;
      kv = base[(size_t)(a.E + h * dh + d) * a.ld + fr];
      vv = base[(size_t)(2 * a.E + h * dh + d) * a.ld + fr];
    }
    ks[idx] = kv;
    vs[idx] = vv;
  }
  __syncthreads();
  const int s = threadIdx.x % ATT_SQ;
  const int q = q0 + s;
  for (int i = threadIdx.x / ATT_SQ; i < a.L; i += lanes) {
    if (q >= a.Q) continue;
    const size_t fr = (size_t)q * a.q_stride + (size_t)i * a.pos_stride;
    float qv[ATT_MAXD];
#pragma unroll
    for (int d = 0; d < ATT_MAXD; ++d) qv[d] = d < dh ? base[(size_t)(h * dh + d) * a.ld + fr] * a.scale : 0.f;
    float m = -INFINITY, z = 0.f;
    float o[ATT_MAXD];
#pragma unroll
    for (int d = 0; d < ATT_MAXD; ++d) o[d] = 0.f;
    const int jend = a.causal ? i + 1 : a.L;
    for (int j = 0; j < jend; ++j) {
      float sc = 0.f;
#pragma unroll
      for (int d = 0; d < ATT_MAXD; ++d)
        if (d < dh) sc = fmaf(qv[d], ks[((size_t)d * a.L + j) * ATT_SQ + s], sc);
      const float mn = fmaxf(m, sc);
      const float corr = expf(m - mn), p = expf(sc - mn);
      z = z * corr + p;
#pragma unroll
      for (int d = 0; d < ATT_MAXD; ++d)
        if (d < dh) o[d] = o[d] * corr + p * vs[((size_t)d * a.L + j) * ATT_SQ + s];
      m = mn;
    }
    const float rz = 1.f / z;
    float* dst = a.out + (size_t)n * a.E * a.ld + fr;
#pragma unroll
    for (int d = 0; d < ATT_MAXD; ++d)
      if (d < dh) dst[(size_t)(h * dh + d) * a.ld] = o[d] * rz;
  }
}

// ---- round 4: the same attention for L <= 64 positions, head dimension known at compile time -----------------------------
// ns_dparn_v0_causal at 32 x 4 s runs four of these launches over 16,032 sequences x 8 heads of 64 positions (dh = 16):
// 33.6 GFLOP and 2.1 GB each -- and the kernel above took 11.8 ms per launch, 58 % of the forward
// (profiles/r04_dparn_kernels_before.txt): its loops run to ATT_MAXD = 64 with the head dimension as a run-time predicate
// (4 x the instructions at dh = 16, 128 registers of q / o), every key costs 2 dh one-float LDS reads and two exps (online
// softmax), and neighbouring workgroups -- which share every 128-byte line of the q / k / v rows, 16 bytes each -- sit on
// eight different XCDs, so every line comes from HBM eight times.
// Here: DH is a template parameter; K and V sit in LDS as [key][sequence][DH + 4] and are read 16 bytes at a time (a
// broadcast per sequence); a thread (sequence, query) computes its L <= 64 scores into registers, takes the maximum,
// exponentiates once per key and accumulates P V -- softmax(q k / sqrt(dh)) v in the reference's order of operations;
// workgroup ids are renumbered so that the eight workgroups sharing a line run on ONE XCD, back to back.
template <int DH, int SQ>
__global__ __launch_bounds__(SQ * 64) void self_attention_l64_kernel(AttArgs a) {
  constexpr int RS = DH + 4;  // floats per (key, sequence) row: 16-byte aligned, rows of the 4 sequences in different banks
  extern __shared__ __attribute__((aligned(16))) float sm[];  // k[L][SQ][RS] | v[L][SQ][RS]
  const int h = blockIdx.y, n = blockIdx.z;
  const int per = gridDim.x >> 3;  // (the grid is a multiple of 8 workgroups along x)
  const int q0 = ((blockIdx.x & 7) * per + (blockIdx.x >> 3)) * SQ;
  float* ks = sm;
  float* vs = sm + a.L * SQ * RS;
  const float* base = a.qkv + (size_t)n * 3 * a.E * a.ld;
  for (int idx = threadIdx.x; idx < DH * a.L * SQ; idx += SQ * 64) {
    const int s = idx % SQ, pos = (idx / SQ) % a.L, d = idx / (SQ * a.L);
    const int q = q0 + s;
    float kv = 0.f, vv = 0.f;
    if (q < a.Q) {
      const size_t fr = (size_t)q * a.q_stride + (size_t)pos * a.pos_stride;
      kv = base[(size_t)(a.E + h * DH + d) * a.ld + fr];
      vv = base[(size_t)(2 * a.E + h * DH + d) * a.ld + fr];
    }
    ks[(pos * SQ + s) * RS + d] = kv;
    vs[(pos * SQ + s) * RS + d] = vv;
  }
  const int s = threadIdx.x % SQ, i = threadIdx.x / SQ;
  const int q = q0 + s;
  const bool live = q < a.Q && i < a.L;
  const size_t fr = live ? (size_t)q * a.q_stride + (size_t)i * a.pos_stride : 0;
  float qv[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) qv[d] = live ? base[(size_t)(h * DH + d) * a.ld + fr] * a.scale : 0.f;
  __syncthreads();
  if (!live) return;
  const int jend = a.causal ? i + 1 : a.L;
  float sc[64];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    float acc = -INFINITY;
    if (j < jend) {  // (j < jend <= L <= 64)
      const f32x4* kr = reinterpret_cast<const f32x4*>(ks + (j * SQ + s) * RS);
      acc = 0.f;
#pragma unroll
      for (int d4 = 0; d4 < DH / 4; ++d4) {
        const f32x4 k4 = kr[d4];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = fmaf(qv[4 * d4 + e], k4[e], acc);
      }
      m = fmaxf(m, acc);
    }
    sc[j] = acc;
  }
  float z = 0.f;
  float o[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    if (j < jend) {
      const float p = __expf(sc[j] - m);  // (v_exp_f32, ~2 ulp: 1.28 -> 1.20 ms per launch against expf)
      z += p;
      const f32x4* vr = reinterpret_cast<const f32x4*>(vs + (j * SQ + s) * RS);
#pragma unroll
      for (int d4 = 0; d4 < DH / 4; ++d4) {
        const f32x4 v4 = vr[d4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[4 * d4 + e] = fmaf(p, v4[e], o[4 * d4 + e]);
      }
    }
  }
  const float rz = 1.f / z;
  float* dst = a.out + (size_t)n * a.E * a.ld + fr;
#pragma unroll
  for (int d = 0; d < DH; ++d) dst[(size_t)(h * DH + d) * a.ld] = o[d] * rz;
}

// y[n][c][pos*pos_stride + q*q_stride] = x[...] + pe[pos][c]   (PositionalEncoding.forward, lobe/attention.py:27-35)
__global__ __launch_bounds__(256) void add_position_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                           float* __restrict__ y, int E, int Q, int q_stride,
                                                           int pos_stride, int ld) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  const int pos = blockIdx.y;
  const int n = blockIdx.z / E, c = blockIdx.z % E;
  if (q >= Q) return;
  const size_t off = ((size_t)n * E + c) * ld + (size_t)pos * pos_stride + (size_t)q * q_stride;
  y[off] = x[off] + pe[(size_t)pos * E + c];
}

}  // namespace ps

using namespace ps;

static int att_status(const char* who) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", who, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int ps_self_attention_f32(const float* qkv, float* out, int N, int E, int heads, int Q, int q_stride, int L,
                                     int pos_stride, int ld, int causal, void* stream) {
  if (!qkv || !out || N <= 0 || E <= 0 || heads <= 0 || E % heads || Q <= 0 || L <= 0 || q_stride < 0 ||
      pos_stride < 0 || ld <= 0 || N > 65535 || heads > 65535) {
    set_error("ps_self_attention_f32: bad argument (N=%d E=%d heads=%d Q=%d L=%d)", N, E, heads, Q, L);
    return PS_E_INVALID;
  }
  const int dh = E / heads;
  const size_t lds = (size_t)2 * dh * L * ATT_SQ * sizeof(float);
  if (dh > ATT_MAXD || lds > 64 * 1024) {
    set_error("ps_self_attention_f32: head_dim %d (max %d) or K/V stage of %zu bytes (max 64 KiB) not supported", dh,
              ATT_MAXD, lds);
    return PS_E_UNSUPPORTED;
  }
  if ((long long)(Q - 1) * q_stride + (long long)(L - 1) * pos_stride >= ld) {
    set_error("ps_self_attention_f32: the last frame lies outside the row (ld=%d)", ld);
    return PS_E_INVALID;
  }
  AttArgs a{qkv, out, E, heads, Q, q_stride, L, pos_stride, ld, causal, 1.f / sqrtf((float)dh)};
  // L <= 64 positions with a head dimension of 16 / 32 / 64: the register-score kernel (ps_debug_flags bit 23 keeps the
  // general one; tests run both)
  // (8 sequences per workgroup when their K / V fit: a workgroup then uses 32 bytes of every 128-byte line it touches
  //  instead of 16, and the L2 -> L1 fill traffic halves; ps_debug_flags bit 21 keeps 4)
  const size_t lds64 = (size_t)2 * L * ATT_SQ * (dh + 4) * sizeof(float);
  if (L <= 64 && (dh == 16 || dh == 32 || dh == 64) && lds64 <= 64 * 1024 && !(g_debug_flags & (1 << 23))) {
    LaunchTimer timer("self_attention", (hipStream_t)stream);
    const bool wide = dh == 16 && 2 * lds64 <= 96 * 1024 && !(g_debug_flags & (1 << 21));
    const int sq = wide ? 8 : ATT_SQ;
    dim3 g((((Q + sq - 1) / sq) + 7) / 8 * 8, heads, N);
    if (wide) {
      static bool raised = false;
      if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&self_attention_l64_kernel<16, 8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        raised = true;
      }
      hipLaunchKernelGGL((self_attention_l64_kernel<16, 8>), g, dim3(512), 2 * lds64, (hipStream_t)stream, a);
    } else if (dh == 16)
      hipLaunchKernelGGL((self_attention_l64_kernel<16, ATT_SQ>), g, dim3(ATT_SQ * 64), lds64, (hipStream_t)stream, a);
    else if (dh == 32)
      hipLaunchKernelGGL((self_attention_l64_kernel<32, ATT_SQ>), g, dim3(ATT_SQ * 64), lds64, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL((self_attention_l64_kernel<64, ATT_SQ>), g, dim3(ATT_SQ * 64), lds64, (hipStream_t)stream, a);
    return att_status("ps_self_attention_f32");
  }
  {
    LaunchTimer timer("self_attention", (hipStream_t)stream);
    hipLaunchKernelGGL(self_attention_kernel, dim3((Q + ATT_SQ - 1) / ATT_SQ, heads, N), dim3(ATT_SQ * 64), lds,
                       (hipStream_t)stream, a);
  }
  return att_status("ps_self_attention_f32");
}

extern "C" int ps_add_position_f32(const float* x, const float* pe, float* y, int N, int E, int Q, int q_stride, int L,
                                   int pos_stride, int ld, void* stream) {
  if (!x || !pe || !y || N <= 0 || E <= 0 || Q <= 0 || L <= 0 || L > 65535 || (long long)N * E > 65535 ||
      (long long)(Q - 1) * q_stride + (long long)(L - 1) * pos_stride >= ld) {
    set_error("ps_add_position_f32: bad argument (N=%d E=%d Q=%d L=%d ld=%d)", N, E, Q, L, ld);
    return PS_E_INVALID;
  }
  {
    LaunchTimer timer("add_position", (hipStream_t)stream);
    hipLaunchKernelGGL(add_position_kernel, dim3((Q + 255) / 256, L, N * E), dim3(256), 0, (hipStream_t)stream, x, pe, y,
                       E, Q, q_stride, pos_stride, ld);
  }
  return att_status("ps_add_position_f32");
}
