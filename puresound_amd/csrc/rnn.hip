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

// ---- GRU / Elman RNN cells (SingleRNN(rnn_type="GRU" | "RNN"), lobe/rnn.py:19-35: no recipe builds them) -------------------
// The structure of lstm_kernel<false>: LS sequences per workgroup, a thread per gate row that walks W_hh^T's column with h in
// LDS, then a thread per (unit, sequence) for the cell.  gx holds W_ih x + b_ih (+ b_hh except for the GRU's n gate, whose
// hidden bias sits inside the reset product: bhn [D][H]).  KIND 0: h' = tanh(gx + W_hh h); 2: GRU, gates r, z, n.
template <int KIND>
__global__ __launch_bounds__(1024) void rnn_kernel(LstmK k, const float* __restrict__ bhn) {
  constexpr int NG = KIND == 2 ? 3 : 1;
  extern __shared__ float smem[];
  const ps_lstm_args& a = k.a;
  const int H = a.H, G = NG * H;
  f32x4* hbuf = reinterpret_cast<f32x4*>(smem);  // [H]  current h of the LS sequences
  f32x4* abuf = hbuf + H;                          // [G]  W_hh h (+ gx, + b_hn)
  f32x4* xbuf = abuf + G;                          // [H]  GRU: gx of the n gate
  const int g = threadIdx.x;
  const int q0 = blockIdx.x * LS, n = blockIdx.y, d = blockIdx.z;
  const bool row = g < G;
  const int DH = a.D * H;
  const float* wt = a.whh_t + (size_t)d * H * G + g;
  const int j = g % H, i = g / H;  // cell role: unit j of sequence i
  const bool cell = g < H * LS && q0 + i < a.Q;
  float h = 0.f;
  if (cell && a.h0) h = a.h0[((size_t)n * DH + d * H + j) * a.ldq + q0 + i];
  if (g < H * LS) reinterpret_cast<float*>(hbuf)[j * LS + i] = h;
  const float* grow = a.gx + ((size_t)n * a.D * G + (size_t)d * G + g) * a.ldt;
  float* hrow = a.hout + ((size_t)n * DH + d * H + j) * a.ldt;
  const bool rev = d == 1;
  const float bn = (KIND == 2 && row && g >= 2 * H && bhn) ? bhn[d * H + (g - 2 * H)] : 0.f;
  __syncthreads();
  for (int s = 0; s < a.steps; ++s) {
    const int ts = rev ? a.steps - 1 - s : s;
    if (row) {
      f32x4 pre, acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < LS; ++e)
        pre[e] = q0 + e < a.Q ? grow[(size_t)(q0 + e) * a.q_stride + (size_t)ts * a.step_stride] : 0.f;
#pragma unroll 8
      for (int kk = 0; kk < H; ++kk) acc += wt[(size_t)kk * G] * hbuf[kk];
      if (KIND == 2 && g >= 2 * H) {
        abuf[g] = acc + bn;
        xbuf[g - 2 * H] = pre;
      } else {
        abuf[g] = acc + pre;
      }
    }
    __syncthreads();
    if (g < H * LS) {
      const float* ab = reinterpret_cast<const float*>(abuf);
      if constexpr (KIND == 2) {
        const float r = sigmoidf_(ab[j * LS + i]);
        const float z = sigmoidf_(ab[(H + j) * LS + i]);
        const float nn = tanhf(reinterpret_cast<const float*>(xbuf)[j * LS + i] + r * ab[(2 * H + j) * LS + i]);
        h = (1.f - z) * nn + z * h;
      } else {
        h = tanhf(ab[j * LS + i]);
      }
      reinterpret_cast<float*>(hbuf)[j * LS + i] = h;
      if (cell) hrow[(size_t)(q0 + i) * a.q_stride + (size_t)ts * a.step_stride] = h;
    }
    __syncthreads();
  }
  if (cell && a.h_last) a.h_last[((size_t)n * DH + d * H + j) * a.ldq + q0 + i] = h;
}

// ---- MFMA recurrence for H = 64 / 128 ----------------------------------------------------------------------------
// One workgroup owns 16 sequences (flat index b = n*Q + q, so a group may span two utterances) and H/16 waves; wave
// w owns hidden units [16w, 16w+16) and ALL FOUR gates of them, so the cell update needs no exchange:
//   D_g[unit][seq] = gx_g + W_hh,g[16 units x H] * h[H x 16 seqs]      g = i, f, g, o
// as 4 x H/4 v_mfma_f32_16x16x4_f32 per step with W_hh resident in VGPRs as A fragments.  C/D layout: lane (seq =
// lane&15, quad = lane>>4) holds units 4*quad + r of its wave in register r -- the same (unit, seq) position in all four
// gate accumulators -- so c, h stay in registers.  h' goes through LDS once per step ([k-block][quad][seq], written
// and read conflict-free) to become everybody's B fragments; the k axis is walked in the order (wave, r, quad) the
// producers hold it in, with W_hh's fragments loaded in the same order.
template <int H>
struct LstmMfma {
  static constexpr int NW = H / 16;  // waves
  static constexpr int KB = H / 4;   // k-blocks of 4
  static constexpr int PF = H == 64 ? 4 : 2;
};

__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.88539008177792681f * x)) - 1.f;
}

template <int H, bool CONTIG>
__global__ __launch_bounds__(H * 4) void lstm_mfma_kernel(LstmK k) {
  using P = LstmMfma<H>;
  constexpr int KB = P::KB, PF = CONTIG ? 4 : P::PF;
  __shared__ float hbuf[2][H * 16];
  const ps_lstm_args& a = k.a;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = lane & 15, quad = lane >> 4;
  const int d = blockIdx.z;
  const int b = blockIdx.x * 16 + col;
  const bool valid = b < a.N * a.Q;
  const int n = valid ? b / a.Q : 0, q = valid ? b % a.Q : 0;
  const int G = 4 * H;
  const int unit0 = 16 * w + 4 * quad;  // first of this lane's 4 units

  // A fragments: gate g, k-block kb = (w', r'): A[row = lane&15][k = 16w' + 4*quad + r']
  float wf[4][KB];
  {
    const float* wt = a.whh_t + (size_t)d * H * G;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int kk = 16 * (kb >> 2) + 4 * quad + (kb & 3);
        wf[g][kb] = wt[(size_t)kk * G + g * H + 16 * w + col];
      }
  }

  float c[4] = {0.f, 0.f, 0.f, 0.f}, h[4] = {0.f, 0.f, 0.f, 0.f};
  if (valid && (a.h0 || a.c0)) {
    const int bs = b - a.state_shift;
    if (bs >= 0) {
      const int nn = bs / a.Q, qq = bs % a.Q;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t off = ((size_t)(nn * a.D + d) * H + unit0 + r) * a.ldq + qq;
        if (a.h0) h[r] = a.h0[off];
        if (a.c0) c[r] = a.c0[off];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) hbuf[0][((w * 4 + r) * 4 + quad) * 16 + col] = h[r];

  const float* gp = a.gx + ((size_t)(n * a.D + d) * G + unit0) * a.ldt + (size_t)q * a.q_stride;
  float* hp = a.hout + ((size_t)(n * a.D + d) * H + unit0) * a.ldt + (size_t)q * a.q_stride;
  const int steps = a.steps;
  const bool rev = d == 1;
  const size_t ldt = a.ldt;

  // pre[u][g][r]: gate pre-activations of the step in ring slot u.  CONTIG (steps are consecutive frames, 16-byte
  // aligned groups of 4): one 16-byte load per (gate, unit) fetches a whole group of 4 steps, the next group is in
  // flight while this one is consumed, and h' leaves as 16-byte stores; otherwise scalar loads PF steps ahead.
  float pre[PF][4][4];
  f32x4 nxt[4][4];
  auto load_group = [&](int s0) {  // CONTIG: steps s0 .. s0+3 -> nxt
    const int f0 = rev ? steps - 4 - s0 : s0;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        nxt[g][r] = (valid && s0 < steps) ? *reinterpret_cast<const f32x4*>(gp + (size_t)(g * H + r) * ldt + f0)
                                          : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto take_group = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) pre[u][g][r] = rev ? nxt[g][r][3 - u] : nxt[g][r][u];
  };
  if constexpr (CONTIG) {
    load_group(0);
  } else {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int ts = rev ? steps - 1 - u : u;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pre[u][g][r] = (valid && u < steps) ? gp[(size_t)(g * H + r) * ldt + (size_t)ts * a.step_stride] : 0.f;
    }
  }
  __syncthreads();

  for (int s0 = 0; s0 < steps; s0 += PF) {
    float hst[4][4];  // CONTIG: h' of the group, [r][u]
    if constexpr (CONTIG) {
      take_group();
      load_group(s0 + 4);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int s = s0 + u;
      if (s < steps) {  // uniform
        const float* hb = hbuf[s & 1];
        float bf[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) bf[kb] = hb[kb * 64 + lane];
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{pre[u][g][0], pre[u][g][1], pre[u][g][2], pre[u][g][3]};
        if constexpr (!CONTIG) {  // refill the ring slot with step s + PF
          const int sn = s + PF;
          const int ts = rev ? steps - 1 - sn : sn;
          if (sn < steps) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int r = 0; r < 4; ++r)
                pre[u][g][r] = valid ? gp[(size_t)(g * H + r) * ldt + (size_t)ts * a.step_stride] : 0.f;
          }
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[g][kb], bf[kb], acc[g], 0, 0, 0);
        float* hn = hbuf[(s + 1) & 1];
        const int ts = rev ? steps - 1 - s : s;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gi = fast_sigmoid(acc[0][r]);
          const float gf = fast_sigmoid(acc[1][r]);
          const float gg = fast_tanh(acc[2][r]);
          const float go = fast_sigmoid(acc[3][r]);
          c[r] = gf * c[r] + gi * gg;
          h[r] = go * fast_tanh(c[r]);
          hn[((w * 4 + r) * 4 + quad) * 16 + col] = h[r];
          if constexpr (CONTIG) {
            hst[r][rev ? 3 - u : u] = h[r];
          } else {
            if (valid) hp[(size_t)r * ldt + (size_t)ts * a.step_stride] = h[r];
          }
        }
        __syncthreads();
      }
    }
    if constexpr (CONTIG) {
      if (valid) {
        const int f0 = rev ? steps - 4 - s0 : s0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<f32x4*>(hp + (size_t)r * ldt + f0) = f32x4{hst[r][0], hst[r][1], hst[r][2], hst[r][3]};
      }
    }
  }
  if (valid) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t off = ((size_t)(n * a.D + d) * H + unit0 + r) * a.ldq + q;
      if (a.h_last) a.h_last[off] = h[r];
      if (a.c_last) a.c_last[off] = c[r];
    }
  }
}

// ---- whole-segment MFMA recurrence (intra-segment pass: many short sequences of consecutive frames) ---------------
// The 16-sequence kernel above fetches the gate pre-activations in 16-byte groups of 4 steps, one group ahead: with
// q_stride = K frames (K = 20: 80 bytes) each 128-byte line of a gate row holds 1.6 whole sequences and is visited
// by all five groups, ~6 us apart; 64 resident workgroups per XCD keep 21 MB of such lines in flight against a 4 MB
// L2, so every visit misses: 645 MB fetched for 131 MB of pre-activations (profiles/r03_pmc_cfg4_traffic_group_kernel.txt) and the
// pass runs at the fabric's rate, not the recurrence's.  Here a workgroup fetches ALL the steps of its 16 sequences
// before the first one (the five groups of a line leave back to back and merge in the vector L1 / L2), keeps them
// in registers -- STEPS x 16 values per lane, the reason for one wave per SIMD (__launch_bounds__(256, 1): 512
// registers) -- and stages h' in LDS so that it leaves as whole 16-sequence x STEPS-frame rows of 16-byte stores.
template <int H, int STEPS, bool BIDIR>
__global__ __launch_bounds__(H * 4, 1) void lstm_seg_kernel(LstmK k) {
  static_assert(H == 64 && STEPS % 4 == 0, "16 units per wave, 4 waves; steps in 16-byte groups");
  constexpr int KB = H / 4, NG = STEPS / 4;
  constexpr int LDH = 16 * STEPS + 4;  // staged h' row: [unit][seq * STEPS + frame], 16-byte aligned rows
  __shared__ __attribute__((aligned(16))) float hbuf[2][H * 16];
  __shared__ __attribute__((aligned(16))) float hstage[H * LDH];
  __shared__ long long seq_off[16];
  const ps_lstm_args& a = k.a;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = lane & 15, quad = lane >> 4;
  const int d = blockIdx.z;
  const int b = blockIdx.x * 16 + col;
  const bool valid = b < a.N * a.Q;
  // sequences past the end (the last workgroup) replay the last real one: every load below is unconditional -- a guarded
  // load is an exec-mask branch each, and one of them made the compiler wait for ALL outstanding loads -- and nothing of
  // theirs is stored (seq_off < 0, `valid` around the final states)
  const int bb = valid ? b : a.N * a.Q - 1;
  const int n = bb / a.Q, q = bb % a.Q;
  const int G = 4 * H;
  const int unit0 = 16 * w + 4 * quad;
  const bool rev = BIDIR && d == 1;  // (one direction: the loads land in their final registers, no selects)
  const size_t ldt = a.ldt;

  if (threadIdx.x < 16) seq_off[threadIdx.x] = valid ? (long long)((size_t)(n * a.D + d) * H * ldt + (size_t)q * a.q_stride) : -1;

  // recurrent weights and initial states first: loads return in order, and these are needed before step 0
  float wf[4][KB];
  {
    const float* wt = a.whh_t + (size_t)d * H * G;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int kk = 16 * (kb >> 2) + 4 * quad + (kb & 3);
        wf[g][kb] = wt[(size_t)kk * G + g * H + 16 * w + col];
      }
  }

  float c[4] = {0.f, 0.f, 0.f, 0.f}, h[4] = {0.f, 0.f, 0.f, 0.f};
  if (valid && (a.h0 || a.c0)) {
    const int bs = b - a.state_shift;
    if (bs >= 0) {
      const int nn = bs / a.Q, qq = bs % a.Q;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t off = ((size_t)(nn * a.D + d) * H + unit0 + r) * a.ldq + qq;
        if (a.h0) h[r] = a.h0[off];
        if (a.c0) c[r] = a.c0[off];
      }
    }
  }

  // every step's pre-activations: pre[s][g][r], s in processing order (frame STEPS-1-s for the reverse direction)
  float pre[STEPS][4][4];
  {
    const float* gp = a.gx + ((size_t)(n * a.D + d) * G + unit0) * ldt + (size_t)q * a.q_stride;
    // (the five 16-byte groups of a row back to back: they share a 128-byte line and merge on the way to the L2; with the
    //  step groups outermost -- first steps startable earlier -- the kernel took 72 us instead of 61)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < NG; ++j) {
          const int jj = rev ? NG - 1 - j : j;
          const f32x4 v = *reinterpret_cast<const f32x4*>(gp + (size_t)(g * H + r) * ldt + 4 * jj);
#pragma unroll
          for (int e = 0; e < 4; ++e) pre[4 * j + e][g][r] = rev ? v[3 - e] : v[e];
        }
  }

#pragma unroll
  for (int r = 0; r < 4; ++r) hbuf[0][((w * 4 + r) * 4 + quad) * 16 + col] = h[r];
  __syncthreads();

#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const float* hb = hbuf[s & 1];
    float bf[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) bf[kb] = hb[kb * 64 + lane];
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{pre[s][g][0], pre[s][g][1], pre[s][g][2], pre[s][g][3]};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[g][kb], bf[kb], acc[g], 0, 0, 0);
    float* hn = hbuf[(s + 1) & 1];
    const int f = rev ? STEPS - 1 - s : s;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gi = fast_sigmoid(acc[0][r]);
      const float gf = fast_sigmoid(acc[1][r]);
      const float gg = fast_tanh(acc[2][r]);
      const float go = fast_sigmoid(acc[3][r]);
      c[r] = gf * c[r] + gi * gg;
      h[r] = go * fast_tanh(c[r]);
      hn[((w * 4 + r) * 4 + quad) * 16 + col] = h[r];
      hstage[(unit0 + r) * LDH + col * STEPS + f] = h[r];
    }
    __syncthreads();
  }

  // h' rows: 16 sequences x STEPS frames per unit, 16 bytes per store (the last barrier of the loop covers hstage)
  for (int p = threadIdx.x; p < H * 16 * NG; p += H * 4) {
    const int unit = p / (16 * NG), rem = p % (16 * NG);
    const int sq = rem / NG, gr = rem % NG;
    const long long off = seq_off[sq];
    if (off >= 0)
      *reinterpret_cast<f32x4*>(a.hout + off + (size_t)unit * ldt + 4 * gr) =
          *reinterpret_cast<const f32x4*>(&hstage[unit * LDH + sq * STEPS + 4 * gr]);
  }
  if (valid) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t off = ((size_t)(n * a.D + d) * H + unit0 + r) * a.ldq + q;
      if (a.h_last) a.h_last[off] = h[r];
      if (a.c_last) a.c_last[off] = c[r];
    }
  }
}

// ---- the same with the recurrent product W_hh h in two fp16 terms per operand (ps_lstm_f16x2_f32) -----------------
// 64 v_mfma_f32_16x16x4_f32 per step (2048 cycles of the matrix pipe, two thirds of the step) become 24
// v_mfma_f32_16x16x32_f16 (384 cycles): W_hh * 2^e = W_hi + W_lo once per workgroup (e puts the largest |weight| into
// [2^12, 2^13)), h * 2^10 = h_hi + h_lo every step (|h| < 1), three products hi*hi + hi*lo + lo*hi accumulated in
// fp32 on top of the (scaled) fp32 pre-activations; the scale comes out again, exactly, before the cell non-linearities.
// Dropped: the lo*lo term and the fp16 rounding of the two lo parts, together <= 2^-21 of sum_k |W_k| |h_k| per gate -- the
// class of the fp16x2 GEMM that produced the pre-activations.  C/D layout as the 16x16x4 instruction, so the cell and
// the staging are the fp32 kernel's; h' crosses LDS as [plane][sequence][k] halves (8-byte writes, 16-byte reads).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int H, int STEPS, bool BIDIR>
__global__ __launch_bounds__(H * 4, 1) void lstm_seg_f16x2_kernel(LstmK k) {
  static_assert(H == 64 && STEPS % 4 == 0, "16 units per wave, 4 waves; steps in 16-byte groups");
  constexpr int NG = STEPS / 4;
  constexpr int LDH = 16 * STEPS + 4;
  constexpr int LDK = H + 8;  // halves per (plane, sequence) row: 144 bytes, conflict-free 8-byte writes / 16-byte reads
  __shared__ __attribute__((aligned(16))) _Float16 hb[2][2][16][LDK];
  __shared__ __attribute__((aligned(16))) float hstage[H * LDH];
  __shared__ long long seq_off[16];
  __shared__ float wmax[4];
  const ps_lstm_args& a = k.a;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = lane & 15, quad = lane >> 4;
  const int d = blockIdx.z;
  const int b = blockIdx.x * 16 + col;
  const bool valid = b < a.N * a.Q;
  // sequences past the end (the last workgroup) replay the last real one: every load below is unconditional -- a guarded
  // load is an exec-mask branch each, and one of them made the compiler wait for ALL outstanding loads -- and nothing of
  // theirs is stored (seq_off < 0, `valid` around the final states)
  const int bb = valid ? b : a.N * a.Q - 1;
  const int n = bb / a.Q, q = bb % a.Q;
  const int G = 4 * H;
  const int unit0 = 16 * w + 4 * quad;
  const bool rev = BIDIR && d == 1;  // (one direction: the loads land in their final registers, no selects)
  const size_t ldt = a.ldt;

  if (threadIdx.x < 16) seq_off[threadIdx.x] = valid ? (long long)((size_t)(n * a.D + d) * H * ldt + (size_t)q * a.q_stride) : -1;

  // A fragments: gate g, k-step ks: lane (row = lane & 15 -> unit 16 w + row, k-group = lane >> 4) holds k = 32 ks + 8 kgroup + e
  // Issue order: the 64 recurrent weights of this lane, then all STEPS x 16 pre-activations, THEN the weights' maximum,
  // scale and split -- the weights return first (loads return in order) and are processed under the pre-activations'
  // flight instead of in front of it.
  const float* wt = a.whh_t + (size_t)d * H * G;
  float wv[4][2][8];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) wv[g][ks][e] = wt[(size_t)(32 * ks + 8 * quad + e) * G + g * H + 16 * w + col];

  // (initial states next: loads return in order, and these are waited for before step 0)
  float c[4] = {0.f, 0.f, 0.f, 0.f}, h[4] = {0.f, 0.f, 0.f, 0.f};
  if (valid && (a.h0 || a.c0)) {
    const int bs = b - a.state_shift;
    if (bs >= 0) {
      const int nn = bs / a.Q, qq = bs % a.Q;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const size_t off = ((size_t)(nn * a.D + d) * H + unit0 + r) * a.ldq + qq;
        if (a.h0) h[r] = a.h0[off];
        if (a.c0) c[r] = a.c0[off];
      }
    }
  }

  float pre[STEPS][4][4];
  {
    const float* gp = a.gx + ((size_t)(n * a.D + d) * G + unit0) * ldt + (size_t)q * a.q_stride;
    // (the five 16-byte groups of a row back to back: they share a 128-byte line and merge on the way to the L2; with the
    //  step groups outermost -- first steps startable earlier -- the kernel took 72 us instead of 61)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < NG; ++j) {
          const int jj = rev ? NG - 1 - j : j;
          const f32x4 v = *reinterpret_cast<const f32x4*>(gp + (size_t)(g * H + r) * ldt + 4 * jj);
#pragma unroll
          for (int e = 0; e < 4; ++e) pre[4 * j + e][g][r] = rev ? v[3 - e] : v[e];
        }
  }


  f16x8 whi[4][2], wlo[4][2];
  float inv;
  {
    float m = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(wv[g][ks][e]));
    m = wave_max(m);
    if (lane == 0) wmax[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    int ex = 0;
    if (m > 0.f) frexpf(m, &ex);  // m = f * 2^ex, f in [0.5, 1)
    ex = ex < -27 ? -27 : ex;     // (tiny matrices: keep the scaled products far from overflow)
    const float S = ldexpf(1.f, 13 - ex);
    inv = 1.f / (S * 1024.f);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float ws = wv[g][ks][e] * S;
          const _Float16 hi = (_Float16)ws;
          whi[g][ks][e] = hi;
          wlo[g][ks][e] = (_Float16)(ws - (float)hi);
        }
  }

  auto put_h = [&](int buf) {  // this lane's 4 units of sequence `col`: two 8-byte writes
    f16x4 hi4, lo4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float hs = h[r] * 1024.f;
      const _Float16 hi = (_Float16)hs;
      hi4[r] = hi;
      lo4[r] = (_Float16)(hs - (float)hi);
    }
    *reinterpret_cast<f16x4*>(&hb[buf][0][col][unit0]) = hi4;
    *reinterpret_cast<f16x4*>(&hb[buf][1][col][unit0]) = lo4;
  };
  put_h(0);
  __syncthreads();

#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    f16x8 bh[2][2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bh[pl][ks] = *reinterpret_cast<const f16x8*>(&hb[s & 1][pl][col][32 * ks + 8 * quad]);
    f32x4 acc[4];  // starts at zero: the fp32 pre-activation joins after the scale has come out (one fma per value)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[g][ks], bh[0][ks], acc[g], 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[g][ks], bh[1][ks], acc[g], 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[g][ks], bh[0][ks], acc[g], 0, 0, 0);
    const int f = rev ? STEPS - 1 - s : s;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float gi = fast_sigmoid(fmaf(acc[0][r], inv, pre[s][0][r]));
      const float gf = fast_sigmoid(fmaf(acc[1][r], inv, pre[s][1][r]));
      const float gg = fast_tanh(fmaf(acc[2][r], inv, pre[s][2][r]));
      const float go = fast_sigmoid(fmaf(acc[3][r], inv, pre[s][3][r]));
      c[r] = gf * c[r] + gi * gg;
      h[r] = go * fast_tanh(c[r]);
      hstage[(unit0 + r) * LDH + col * STEPS + f] = h[r];
    }
    put_h((s + 1) & 1);
    __syncthreads();
  }

  for (int p = threadIdx.x; p < H * 16 * NG; p += H * 4) {
    const int unit = p / (16 * NG), rem = p % (16 * NG);
    const int sq = rem / NG, gr = rem % NG;
    const long long off = seq_off[sq];
    if (off >= 0)
      *reinterpret_cast<f32x4*>(a.hout + off + (size_t)unit * ldt + 4 * gr) =
          *reinterpret_cast<const f32x4*>(&hstage[unit * LDH + sq * STEPS + 4 * gr]);
  }
  if (valid) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t off = ((size_t)(n * a.D + d) * H + unit0 + r) * a.ldq + q;
      if (a.h_last) a.h_last[off] = h[r];
      if (a.c_last) a.c_last[off] = c[r];
    }
  }
}

// ---- 4-sequence MFMA recurrence (v_mfma_f32_4x4x1_16B_f32) -------------------------------------------------------
// For passes with few, long sequences (the inter-segment pass: N*K sequences of S steps) the 16-sequence kernel above
// leaves most CUs idle and walks the steps at ~2 us each.  Here a workgroup owns only 4 sequences.  The instruction
// multiplies 16 independent 4x4 blocks: block = hidden unit (16 per wave), block row = gate (i, f, g, o), block column
// = sequence, K = 1 per issue, so lane (unit b, sequence j) receives the four gate pre-activations of ITS (unit,
// sequence) in the four result registers and does the cell update alone.  A = W_hh[gate][unit][k] resident (H VGPRs),
// B = h[k][sequence] broadcast from LDS.  Four independent accumulators hide the dependent-issue latency.
template <int H, bool CONTIG>
__global__ __launch_bounds__(H * 4) void lstm_m4_kernel(LstmK k) {
  constexpr int HS = H + 16;  // LDS row stride: conflict-free h' writes
  constexpr int PF = CONTIG ? 4 : 8;
  __shared__ __attribute__((aligned(16))) float hbuf[2][4 * HS];
  const ps_lstm_args& a = k.a;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = lane & 3, ub = lane >> 2;
  const int unit = 16 * w + ub;
  const int d = blockIdx.z;
  // XCD-aware group order: consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), but
  // neighbouring sequence groups share the 128-byte lines of the gate rows (4 sequences = 16 bytes of a line), so
  // groups are renumbered to put neighbours on ONE XCD: id -> (id % 8) * ceil(G/8) + id / 8.
  // (the grid is a multiple of 8 groups, so this is a bijection; groups past the last sequence are masked by `valid`)
  const int per = gridDim.x >> 3;
  const int grp = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int b = grp * 4 + j;
  const bool valid = b < a.N * a.Q;
  // sequences past the end replay the last real one: the gate loads of the step loop are unconditional (a guarded load is
  // an exec-mask branch in the dependent chain of every step); nothing of theirs is stored
  const int bb = valid ? b : a.N * a.Q - 1;
  const int n = bb / a.Q, q = bb % a.Q;
  const int G = 4 * H;

  // A operand: lane (unit, i = lane&3) holds W_hh[gate i][unit][k]
  float wf[H];
  {
    const float* wt = a.whh_t + (size_t)d * H * G + j * H + unit;
#pragma unroll
    for (int kk = 0; kk < H; ++kk) wf[kk] = wt[(size_t)kk * G];
  }

  float c = 0.f, h = 0.f;
  if (valid && (a.h0 || a.c0)) {
    const int bs = b - a.state_shift;
    if (bs >= 0) {
      const int nn = bs / a.Q, qq = bs % a.Q;
      const size_t off = ((size_t)(nn * a.D + d) * H + unit) * a.ldq + qq;
      if (a.h0) h = a.h0[off];
      if (a.c0) c = a.c0[off];
    }
  }
  hbuf[0][j * HS + unit] = h;

  const float* gp = a.gx + ((size_t)(n * a.D + d) * G + unit) * a.ldt + (size_t)q * a.q_stride;
  float* hp = a.hout + ((size_t)(n * a.D + d) * H + unit) * a.ldt + (size_t)q * a.q_stride;
  const int steps = a.steps;
  const bool rev = d == 1;
  const size_t gstride = (size_t)H * a.ldt;  // between the gates of one unit

  float pre[PF][4];
  f32x4 nxt[4];
  auto load_group = [&](int s0) {  // CONTIG: steps s0 .. s0+3 of every gate -> nxt
    const int f0 = rev ? steps - 4 - s0 : s0;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      nxt[g] = (valid && s0 < steps) ? *reinterpret_cast<const f32x4*>(gp + g * gstride + f0)
                                     : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  if constexpr (CONTIG) {
    load_group(0);
  } else {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int uc = u < steps ? u : steps - 1;
      const int ts = rev ? steps - 1 - uc : uc;
#pragma unroll
      for (int g = 0; g < 4; ++g) pre[u][g] = gp[g * gstride + (size_t)ts * a.step_stride];
    }
  }
  __syncthreads();

  for (int s0 = 0; s0 < steps; s0 += PF) {
    float hst[4];
    if constexpr (CONTIG) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[u][g] = rev ? nxt[g][3 - u] : nxt[g][u];
      load_group(s0 + 4);
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int s = s0 + u;
      if (s < steps) {  // uniform
        const float* hb = hbuf[s & 1] + j * HS;
        f32x4 acc[2];
        acc[0] = f32x4{pre[u][0], pre[u][1], pre[u][2], pre[u][3]};
        acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef PS_M4_ABL
#define PS_M4_ABL 0
#endif
#pragma unroll
        for (int kk = 0; kk < ((PS_M4_ABL & 1) ? 1 : H / 4); ++kk) {
          const f32x4 hv = *reinterpret_cast<const f32x4*>(hb + 4 * kk);
#pragma unroll
          for (int e = 0; e < 4; ++e)  // two accumulator chains (116 us per inter pass; four: 119, one: 130)
            acc[e & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wf[4 * kk + e], hv[e], acc[e & 1], 0, 0, 0);
        }
        if constexpr (!CONTIG) {
          // refill the ring slot with step s + PF -- behind the MFMAs (in front of them the four loads and their
          // address arithmetic sat between the barrier and the first LDS read of the step's dependent chain); steps past
          // the end re-read the last one
          const int sn = s + PF < steps ? s + PF : steps - 1;
          const int ts = rev ? steps - 1 - sn : sn;
#pragma unroll
          for (int g = 0; g < 4; ++g) pre[u][g] = gp[g * gstride + (size_t)ts * a.step_stride];
          // ... and the PREVIOUS step's h' leaves here too (same reason; the last one after the loop)
          if (s > 0 && valid) hp[(size_t)(rev ? steps - s : s - 1) * a.step_stride] = h;
        }
        const f32x4 t = acc[0] + acc[1];
#if PS_M4_ABL & 2
        c = t[1] * c + t[0] * t[2];
        h = t[3] * c;
#else
        const float gi = fast_sigmoid(t[0]);
        const float gf = fast_sigmoid(t[1]);
        const float gg = fast_tanh(t[2]);
        const float go = fast_sigmoid(t[3]);
        c = gf * c + gi * gg;
        h = go * fast_tanh(c);
#endif
        hbuf[(s + 1) & 1][j * HS + unit] = h;
        __syncthreads();
        if constexpr (CONTIG) {
          hst[rev ? 3 - u : u] = h;
        }
      }
    }
    if constexpr (CONTIG) {
      if (valid) {
        const int f0 = rev ? steps - 4 - s0 : s0;
        *reinterpret_cast<f32x4*>(hp + f0) = f32x4{hst[0], hst[1], hst[2], hst[3]};
      }
    }
  }
  if constexpr (!CONTIG) {
    if (valid) hp[(size_t)(rev ? 0 : steps - 1) * a.step_stride] = h;
  }
  if (valid) {
    const size_t off = ((size_t)(n * a.D + d) * H + unit) * a.ldq + q;
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
template <int PARTS>
__global__ __launch_bounds__(64 * PARTS) void chan_layernorm_kernel(ClnArgs a) {
  __shared__ float red[PARTS][64];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane, n = blockIdx.y;
  const bool live = t < a.T;
  const size_t base = (size_t)n * a.C * a.ldt + (live ? t : 0);
  float s = 0.f;
  for (int ch = part; ch < a.C; ch += PARTS) s += live ? a.x[base + (size_t)ch * a.ldt] : 0.f;
  red[part][lane] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int p = 0; p < PARTS; ++p) tot += red[p][lane];
  const float mean = tot / (float)a.C;
  __syncthreads();
  float q = 0.f;
  for (int ch = part; ch < a.C; ch += PARTS) {
    const float dv = live ? a.x[base + (size_t)ch * a.ldt] - mean : 0.f;
    q += dv * dv;
  }
  red[part][lane] = q;
  __syncthreads();
  tot = 0.f;
#pragma unroll
  for (int p = 0; p < PARTS; ++p) tot += red[p][lane];
  const float var = tot / (float)a.C;
  const float rstd = 1.f / sqrtf(var + a.eps);
  if (!live) return;
  const float slope = a.slope ? a.slope[0] : 1.f;
  for (int ch = part; ch < a.C; ch += PARTS) {
    const size_t off = base + (size_t)ch * a.ldt;
    float v = (a.x[off] - mean) * rstd * a.gamma[ch] + a.beta[ch];
    if (a.slope) v = prelu(v, slope);
    if (a.sigmoid) v = sigmoidf_(v);
    if (a.mul) v *= a.mul[off];
    if (a.res) v += a.res[off];
    a.y[off] = v;
  }
}

// The same for C <= 4 * CPT with the thread's channels held in registers: one pass over x instead of three (on the 2-D maps
// of DPCRN / DPARN -- 32 x 32,745 frames x 128 channels -- the three passes ran at 2 TB/s of useful traffic, 770 us).  Sums in the
// order of chan_layernorm_kernel<4>: identical results.
template <int CPT>
__global__ __launch_bounds__(256) void chan_layernorm_reg_kernel(ClnArgs a) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane, n = blockIdx.y;
  const bool live = t < a.T;
  const size_t base = (size_t)n * a.C * a.ldt + (live ? t : 0);
  float v[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int ch = part + 4 * i;
    v[i] = (live && ch < a.C) ? a.x[base + (size_t)ch * a.ldt] : 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) s += v[i];
  red[part][lane] = s;
  __syncthreads();
  const float mean = (((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]) / (float)a.C;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const float dv = (live && part + 4 * i < a.C) ? v[i] - mean : 0.f;
    q += dv * dv;
  }
  red[part][lane] = q;
  __syncthreads();
  const float var = (((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]) / (float)a.C;
  const float rstd = 1.f / sqrtf(var + a.eps);
  if (!live) return;
  const float slope = a.slope ? a.slope[0] : 1.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int ch = part + 4 * i;
    if (ch < a.C) {
      const size_t off = base + (size_t)ch * a.ldt;
      float o = (v[i] - mean) * rstd * a.gamma[ch] + a.beta[ch];
      if (a.slope) o = prelu(o, slope);
      if (a.sigmoid) o = sigmoidf_(o);
      if (a.mul) o *= a.mul[off];
      if (a.res) o += a.res[off];
      a.y[off] = o;
    }
  }
}

// One LSTM cell update per (unit, frame) from complete gate pre-activations (the streaming step: the recurrent
// product W_hh h is part of the gates GEMM there, its K axis being [x; h]).
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ gates, float* __restrict__ c,
                                                        float* __restrict__ h, int H, int T, int ldg, int lds_) {
  const int t = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int nd = blockIdx.z;  // utterance * directions + direction
  if (t >= T || j >= H) return;
  const float* g = gates + ((size_t)nd * 4 * H + j) * ldg + t;
  const size_t so = ((size_t)nd * H + j) * lds_ + t;
  const float gi = sigmoidf_(g[0]);
  const float gf = sigmoidf_(g[(size_t)H * ldg]);
  const float gg = tanhf(g[(size_t)2 * H * ldg]);
  const float go = sigmoidf_(g[(size_t)3 * H * ldg]);
  const float cn = gf * c[so] + gi * gg;
  c[so] = cn;
  h[so] = go * tanhf(cn);
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

// ---- the 4-sequence recurrence with the product W_hh h in two fp16 terms (ps_lstm_f16x2_f32, inter-segment pass) --------
// The step of lstm_m4_kernel is a dependent chain: 64 v_mfma_f32_4x4x1_f32 (512 cycles of issue), the cell, one LDS
// exchange.  v_mfma_f32_4x4x4_f16 takes four k per issue: with W_hh 2^e = hi + lo once per workgroup and h 2^10 = hi +
// lo every step (the scheme of lstm_seg_f16x2_kernel, same error class) the chain is 3 x 16 issues, 384 cycles.  Layout
// as the fp32 kernel: lane (unit, j): A = W_hh[gate j][unit][4 k], B = h[4 k][sequence j], the four result registers
// the four gates of (unit, sequence j).  h' crosses LDS as halves, [plane][sequence][k], read 16 bytes (8 k) at a time.
typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void lstm_m4_f16x2_kernel(LstmK k) {
  constexpr int H = 64, PF = 8;
  constexpr int LDK = H + 8;  // halves per (plane, sequence) row
  __shared__ __attribute__((aligned(16))) _Float16 hb[2][2][4][LDK];
  __shared__ float wmax[4];
  const ps_lstm_args& a = k.a;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = lane & 3, ub = lane >> 2;
  const int unit = 16 * w + ub;
  const int d = blockIdx.z;
  const int per = gridDim.x >> 3;  // XCD-aware group order, as lstm_m4_kernel
  const int grp = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const int b = grp * 4 + j;
  const bool valid = b < a.N * a.Q;
  const int bb = valid ? b : a.N * a.Q - 1;  // (sequences past the end replay the last one; nothing of theirs is stored)
  const int n = bb / a.Q, q = bb % a.Q;
  const int G = 4 * H;

  // A operand: lane (unit, i = lane & 3) holds W_hh[gate i][unit][k], k-step kk: k = 4 kk .. 4 kk + 3
  f16x4v whi[H / 4], wlo[H / 4];
  float inv;
  {
    const float* wt = a.whh_t + (size_t)d * H * G + j * H + unit;
    float wv[H];
    float m = 0.f;
#pragma unroll
    for (int kk = 0; kk < H; ++kk) {
      wv[kk] = wt[(size_t)kk * G];
      m = fmaxf(m, fabsf(wv[kk]));
    }
    m = wave_max(m);
    if (lane == 0) wmax[w] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    int ex = 0;
    if (m > 0.f) frexpf(m, &ex);
    ex = ex < -27 ? -27 : ex;
    const float S = ldexpf(1.f, 13 - ex);
    inv = 1.f / (S * 1024.f);
#pragma unroll
    for (int kk = 0; kk < H / 4; ++kk)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ws = wv[4 * kk + e] * S;
        const _Float16 hi = (_Float16)ws;
        whi[kk][e] = hi;
        wlo[kk][e] = (_Float16)(ws - (float)hi);
      }
  }

  float c = 0.f, h = 0.f;
  if (a.h0 || a.c0) {  // (replayed sequences take the replayed state: they must store the very same h')
    const int bs = bb - a.state_shift;
    if (bs >= 0) {
      const int nn = bs / a.Q, qq = bs % a.Q;
      const size_t off = ((size_t)(nn * a.D + d) * H + unit) * a.ldq + qq;
      if (a.h0) h = a.h0[off];
      if (a.c0) c = a.c0[off];
    }
  }
  auto put_h = [&](int buf) {
    const float hs = h * 1024.f;
    const _Float16 hi = (_Float16)hs;
    hb[buf][0][j][unit] = hi;
    hb[buf][1][j][unit] = (_Float16)(hs - (float)hi);
  };
  put_h(0);

  const float* gp = a.gx + ((size_t)(n * a.D + d) * G + unit) * a.ldt + (size_t)q * a.q_stride;
  float* hp = a.hout + ((size_t)(n * a.D + d) * H + unit) * a.ldt + (size_t)q * a.q_stride;
  const int steps = a.steps;
  const bool rev = d == 1;
  const size_t gstride = (size_t)H * a.ldt;

  float pre[PF][4];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int uc = u < steps ? u : steps - 1;
    const int ts = rev ? steps - 1 - uc : uc;
#pragma unroll
    for (int g = 0; g < 4; ++g) pre[u][g] = gp[g * gstride + (size_t)ts * a.step_stride];
  }
  __syncthreads();

  for (int s0 = 0; s0 < steps; s0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int s = s0 + u;
      if (s < steps) {  // uniform
        const _Float16* hh = &hb[s & 1][0][j][0];
        const _Float16* hl = &hb[s & 1][1][j][0];
        constexpr int NC = 2;  // accumulator chains: 105 us with two, 108 with four, 110 with one, 115 with six
        f32x4 acc[NC];
#pragma unroll
        for (int e = 0; e < NC; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
        const f32x4 p4 = f32x4{pre[u][0], pre[u][1], pre[u][2], pre[u][3]};
#pragma unroll
        for (int k8 = 0; k8 < H / 8; ++k8) {
          const f16x8 vh = *reinterpret_cast<const f16x8*>(hh + 8 * k8);
          const f16x8 vl = *reinterpret_cast<const f16x8*>(hl + 8 * k8);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int kk = 2 * k8 + e;
            const f16x4v bh = f16x4v{vh[4 * e], vh[4 * e + 1], vh[4 * e + 2], vh[4 * e + 3]};
            const f16x4v bl = f16x4v{vl[4 * e], vl[4 * e + 1], vl[4 * e + 2], vl[4 * e + 3]};
            const int c0 = (3 * kk) % NC, c1 = (3 * kk + 1) % NC, c2 = (3 * kk + 2) % NC;
            acc[c0] = __builtin_amdgcn_mfma_f32_4x4x4f16(whi[kk], bh, acc[c0], 0, 0, 0);
            acc[c1] = __builtin_amdgcn_mfma_f32_4x4x4f16(whi[kk], bl, acc[c1], 0, 0, 0);
            acc[c2] = __builtin_amdgcn_mfma_f32_4x4x4f16(wlo[kk], bh, acc[c2], 0, 0, 0);
          }
        }
        {  // refill the ring slot (step s + PF) and let the previous step's h' leave: in the shadow of the MFMAs
          const int sn = s + PF < steps ? s + PF : steps - 1;
          const int ts = rev ? steps - 1 - sn : sn;
#pragma unroll
          for (int g = 0; g < 4; ++g) pre[u][g] = gp[g * gstride + (size_t)ts * a.step_stride];
          // (guarded: made unconditional -- replayed sequences storing the same values -- the kernel took 134 us, not 107)
          if (s > 0 && valid) hp[(size_t)(rev ? steps - s : s - 1) * a.step_stride] = h;
        }
        f32x4 sum = acc[0];
#pragma unroll
        for (int e = 1; e < NC; ++e) sum += acc[e];
        const f32x4 t = sum * inv + p4;
        const float gi = fast_sigmoid(t[0]);
        const float gf = fast_sigmoid(t[1]);
        const float gg = fast_tanh(t[2]);
        const float go = fast_sigmoid(t[3]);
        c = gf * c + gi * gg;
        h = go * fast_tanh(c);
        put_h((s + 1) & 1);
        __syncthreads();
      }
    }
  }
  if (valid) {
    hp[(size_t)(rev ? 0 : steps - 1) * a.step_stride] = h;
    const size_t off = ((size_t)(n * a.D + d) * H + unit) * a.ldq + q;
    if (a.h_last) a.h_last[off] = h;
    if (a.c_last) a.c_last[off] = c;
  }
}

typedef unsigned u32x4l __attribute__((ext_vector_type(4)));
#include "lstm_fm.inc"
typedef float f32x2 __attribute__((ext_vector_type(2)));
#include "lstm_fm256.inc"
#include "lstm_coop.inc"

}  // namespace ps

using namespace ps;

static int lstm_launch(const ps_lstm_args* args, void* stream, bool f16x2);

// (no message: the probe of a caller choosing its path)
static bool lstm_fmajor_fits(const ps_lstm_args& a, int ldm) {
  if (!a.gx || !a.whh_t || !a.hout || a.h0 || a.c0 || a.h_last || a.c_last || a.H != 128 || a.D < 1 || a.D > 2 || a.N <= 0 ||
      a.Q <= 0 || a.steps <= 0 || a.q_stride < 0 || a.step_stride < 0 || a.ldt <= 0 || a.state_shift != 0)
    return false;
  if ((long long)(a.Q - 1) * a.q_stride + (long long)(a.steps - 1) * a.step_stride >= a.ldt) return false;
  if (ldm < a.D * 512 || ldm % 4 || (long long)a.ldt * ldm * 4 >= (1LL << 31) || (long long)a.ldt * 128 * 4 >= (1LL << 31)) return false;
  if (((uintptr_t)a.gx & 15) || ((uintptr_t)a.hout & 3)) return false;
  return (long long)a.N * ((a.Q + 15) / 16) < (1LL << 30);
}

static bool lstm_h256_fits(const ps_lstm_args& a, int ldm) {
  if (!a.gx || !a.hout || (a.H != 256 && a.H != 192) || a.D < 1 || a.D > 2 || a.N <= 0 || a.Q <= 0 || a.steps <= 0 || a.q_stride < 0 ||
      a.step_stride < 0 || a.ldt <= 0 || (a.state_shift != 0 && a.state_shift != 1))
    return false;
  if ((long long)(a.Q - 1) * a.q_stride + (long long)(a.steps - 1) * a.step_stride >= a.ldt) return false;
  if (ldm < a.D * 4 * a.H || ldm % 4 || ((uintptr_t)a.gx & 15) || ((uintptr_t)a.hout & 3)) return false;
  if ((a.h0 || a.c0 || a.h_last || a.c_last) && a.ldq < a.Q) return false;
  return (long long)a.N * a.Q < (1LL << 30);
}

extern "C" int ps_lstm_fmajor_h256_ok(const ps_lstm_args* args, int ldm) { return args && lstm_h256_fits(*args, ldm) ? 1 : 0; }

extern "C" int ps_lstm_fmajor_h256_f16x2_f32(const ps_lstm_args* args, int ldm, const void* whh_image, const float* acc_scale,
                                             void* stream) {
  if (!args || !whh_image || !acc_scale || ((uintptr_t)whh_image & 15)) {
    set_error("ps_lstm_fmajor_h256_f16x2_f32: null argument or unaligned weight image");
    return PS_E_INVALID;
  }
  if (!lstm_h256_fits(*args, ldm)) {
    set_error("ps_lstm_fmajor_h256_f16x2_f32: H = 256 or 192, D = 1 or 2, every frame inside the row, ldm >= D*4H (ps_lstm_fmajor_h256_ok)");
    return PS_E_UNSUPPORTED;
  }
  const ps_lstm_args& a = *args;
  LstmFm256 k{a, ldm, whh_image, {acc_scale[0], a.D > 1 ? acc_scale[1] : acc_scale[0]}};
  if (!(k.up[0] > 0.f) || !(k.up[1] > 0.f)) {
    set_error("ps_lstm_fmajor_h256_f16x2_f32: accumulator scales must be positive");
    return PS_E_INVALID;
  }
  const long long seqs = (long long)a.N * a.Q;
  dim3 grid((unsigned)((seqs + 15) / 16), 1, a.D);
  const bool pairs = a.step_stride == 1 && a.q_stride % 2 == 0 && a.ldt % 2 == 0 && !((uintptr_t)a.hout & 7) &&
                     (a.D == 1 || a.steps % 2 == 0) && !(g_debug_flags & (1 << 20));
  {
    LaunchTimer timer("lstm", (hipStream_t)stream);
    if (a.H == 256 && pairs)
      hipLaunchKernelGGL((lstm_fm_h256_kernel<256, 2>), grid, dim3(512), 0, (hipStream_t)stream, k);
    else if (a.H == 256)
      hipLaunchKernelGGL((lstm_fm_h256_kernel<256, 1>), grid, dim3(512), 0, (hipStream_t)stream, k);
    else if (pairs)
      hipLaunchKernelGGL((lstm_fm_h256_kernel<192, 2>), grid, dim3(384), 0, (hipStream_t)stream, k);
    else
      hipLaunchKernelGGL((lstm_fm_h256_kernel<192, 1>), grid, dim3(384), 0, (hipStream_t)stream, k);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_lstm_fmajor_h256_f16x2_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// the cooperative kernel: the streamed kernel's shapes, few enough sequence groups that every slice of every group gets a CU
// of its own at the same time (the group barrier spins: co-residency is a correctness condition, not a tuning choice)
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* p, int n) {
  for (int i = threadIdx.x; i < n; i += 256) p[i] = 0u;
}

// waves per workgroup (= 16-unit row blocks per slice): 2 when the launch then still fits the chip -- more, smaller slices
// measured faster per step with a handful of groups (the speaker LSTM: 9.5 ms against 13.6) -- else 4, else 0 (does not fit)
static int lstm_coop_waves(const ps_lstm_args& a, int ldm, int* groups_out) {
  if (!lstm_h256_fits(a, ldm) || a.steps < 2) return 0;
  const long long groups = ((long long)a.N * a.Q + 15) / 16;
  const long long rounds = (groups * a.D + 7) / 8 * 8;
  if (groups_out) *groups_out = (int)groups;
  if (rounds * (a.H / 32) <= device_cus()) return 2;
  if (rounds * (a.H / 64) <= device_cus()) return 4;
  // both directions do not fit at once: one launch per direction, if a direction does
  if (a.D == 2 && (groups + 7) / 8 * 8 * (a.H / 64) <= device_cus()) return -4;
  return 0;
}

extern "C" size_t ps_lstm_fmajor_coop_workspace_bytes(const ps_lstm_args* args, int ldm) {
  int groups = 0;
  int wv = args ? lstm_coop_waves(*args, ldm, &groups) : 0;
  if (!wv) return 0;
  wv = wv < 0 ? -wv : wv;
  const size_t hx = align_up((size_t)2 * args->D * groups * 2 * 16 * (args->H + 8) * sizeof(_Float16), 256);
  return hx + align_up((size_t)(args->D * groups * (2 + args->H / (16 * wv)) + 1) * sizeof(unsigned), 256);
}

extern "C" int ps_lstm_fmajor_coop_f16x2_f32(const ps_lstm_args* args, int ldm, const void* whh_image, const float* acc_scale,
                                             void* workspace, size_t workspace_bytes, void* stream) {
  if (!args || !whh_image || !acc_scale || !workspace || ((uintptr_t)whh_image & 15) || ((uintptr_t)workspace & 255)) {
    set_error("ps_lstm_fmajor_coop_f16x2_f32: null argument, unaligned weight image or workspace (256 bytes)");
    return PS_E_INVALID;
  }
  const size_t need = ps_lstm_fmajor_coop_workspace_bytes(args, ldm);
  if (!need) {
    set_error("ps_lstm_fmajor_coop_f16x2_f32: ps_lstm_fmajor_h256_f16x2_f32's shapes with at most %d / (D * H / 64) groups of 16 "
              "sequences and at least two steps (ps_lstm_fmajor_coop_workspace_bytes = 0)", device_cus());
    return PS_E_UNSUPPORTED;
  }
  if (workspace_bytes < need) {
    set_error("ps_lstm_fmajor_coop_f16x2_f32: workspace too small (%zu < %zu)", workspace_bytes, need);
    return PS_E_INVALID;
  }
  const ps_lstm_args& a = *args;
  int groups = 0;
  int wv = lstm_coop_waves(a, ldm, &groups);
  const bool per_direction = wv < 0;
  wv = wv < 0 ? -wv : wv;
  const size_t hx = align_up((size_t)2 * a.D * groups * 2 * 16 * (a.H + 8) * sizeof(_Float16), 256);
  LstmCoop k{a, ldm, whh_image, {acc_scale[0], a.D > 1 ? acc_scale[1] : acc_scale[0]}, (_Float16*)workspace,
             (unsigned*)((char*)workspace + hx), groups, (g_debug_flags & (1 << 19)) ? 0 : 1, (g_debug_flags & (1 << 18)) ? 1 : 0,
             (g_debug_flags & (1 << 17)) ? 1 : 0, 0, a.D, (int)((g_debug_flags >> 24) & 15)};
  if (!(k.up[0] > 0.f) || !(k.up[1] > 0.f)) {
    set_error("ps_lstm_fmajor_coop_f16x2_f32: accumulator scales must be positive");
    return PS_E_INVALID;
  }
  // (a kernel, not hipMemsetAsync: inside a replayed graph a memset node does not have to go through the L2 the counters'
  //  atomics work in -- the first graph replays of this launch computed with the previous replay's counts)
  hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, k.sync, (int)((need - hx) / sizeof(unsigned)));
  hipError_t e;
  {
    LaunchTimer timer("lstm", (hipStream_t)stream);
    // clusters (direction, group) in rounds of 8, one per XCD; H / 32 slices each
    for (int d0 = 0; d0 < (per_direction ? a.D : 1); ++d0) {
    if (per_direction) k.d0 = d0, k.nd = 1;
    dim3 grid((unsigned)((k.nd * groups + 7) / 8 * (a.H / (16 * wv)) * 8));
    // (the streamed kernel's rule for 8-byte h' stores; bit 20 keeps the 4-byte ones: tests run both)
    const bool pairs = a.step_stride == 1 && a.q_stride % 2 == 0 && a.ldt % 2 == 0 && !((uintptr_t)a.hout & 7) &&
                       (a.D == 1 || a.steps % 2 == 0) && !(g_debug_flags & (1 << 20));
#define PS_COOP(HH, WW)                                                                                            \
  if (pairs)                                                                                                       \
    hipLaunchKernelGGL((lstm_coop_kernel<HH, WW, 2>), grid, dim3(64 * WW), 0, (hipStream_t)stream, k);             \
  else                                                                                                             \
    hipLaunchKernelGGL((lstm_coop_kernel<HH, WW, 1>), grid, dim3(64 * WW), 0, (hipStream_t)stream, k);
    if (a.H == 256 && wv == 2) {
      PS_COOP(256, 2)
    } else if (a.H == 256) {
      PS_COOP(256, 4)
    } else if (wv == 2) {
      PS_COOP(192, 2)
    } else {
      PS_COOP(192, 4)
    }
    }
#undef PS_COOP
  }
  e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_lstm_fmajor_coop_f16x2_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int ps_lstm_fmajor_ok(const ps_lstm_args* args, int ldm) { return args && lstm_fmajor_fits(*args, ldm) ? 1 : 0; }

extern "C" int ps_lstm_fmajor_f16x2_f32(const ps_lstm_args* args, int ldm, void* stream) {
  if (!args || !lstm_fmajor_fits(*args, ldm)) {
    set_error("ps_lstm_fmajor_f16x2_f32: H = 128, D = 1 or 2, no states, every frame inside the row, slabs below 2 GiB "
              "(ps_lstm_fmajor_ok)");
    return args ? PS_E_UNSUPPORTED : PS_E_INVALID;
  }
  const ps_lstm_args& a = *args;
  LstmFm k{a, ldm, (g_debug_flags >> 24) & 15, (a.Q + 15) / 16, 0};
  k.total = a.N * k.nblk;
  // one workgroup per CU (148 KiB of LDS), the directions side by side; a multiple of 8 for the XCD-aware block order
  int cap = device_cus() / a.D / 8 * 8;
  cap = cap < 8 ? 8 : cap;
  const int want = (k.total + 7) / 8 * 8;
  dim3 grid((unsigned)(want < cap ? want : cap), 1, a.D);
  const int sp = (a.steps + 3) / 4 * 4;
  const bool contig = a.step_stride == 1 && a.q_stride % 4 == 0 && a.ldt % 4 == 0 && !((uintptr_t)a.hout & 15) &&
                      (a.steps % 4 == 0 || (a.D == 1 && (long long)(a.Q - 1) * a.q_stride + sp <= a.ldt)) &&
                      !(g_debug_flags & (1 << 20));  // (bit 20: 4-byte h' stores for consecutive frames too; tests run both)
  {
    LaunchTimer timer("lstm", (hipStream_t)stream);
    if (contig)
      hipLaunchKernelGGL((lstm_fm_f16x2_kernel<true>), grid, dim3(512), 0, (hipStream_t)stream, k);
    else
      hipLaunchKernelGGL((lstm_fm_f16x2_kernel<false>), grid, dim3(512), 0, (hipStream_t)stream, k);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_lstm_fmajor_f16x2_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int ps_rnn_f32(const ps_lstm_args* args, int kind, const float* bhn, void* stream) {
  if (!args || (kind != PS_RNN_TANH && kind != PS_RNN_GRU)) {
    set_error("ps_rnn_f32: null args or unknown cell kind %d", kind);
    return PS_E_INVALID;
  }
  const ps_lstm_args& a = *args;
  const int ng = kind == PS_RNN_GRU ? 3 : 1;
  if (!a.gx || !a.whh_t || !a.hout || a.c0 || a.c_last || a.N <= 0 || a.H <= 0 || a.D < 1 || a.D > 2 || a.Q <= 0 || a.steps <= 0 ||
      a.q_stride < 0 || a.step_stride < 0 || a.ldt <= 0 || a.N > 65535 || a.state_shift != 0 || (kind == PS_RNN_GRU && !bhn)) {
    set_error("ps_rnn_f32: bad argument (N=%d H=%d D=%d Q=%d steps=%d; no cell states, the GRU needs bhn)", a.N, a.H, a.D, a.Q, a.steps);
    return PS_E_INVALID;
  }
  if (ng * a.H > 1024 || a.H * LS > 1024) {
    set_error("ps_rnn_f32: hidden size %d is not supported (GRU: <= 256, RNN: <= 256)", a.H);
    return PS_E_UNSUPPORTED;
  }
  if ((long long)(a.Q - 1) * a.q_stride + (long long)(a.steps - 1) * a.step_stride >= a.ldt ||
      ((a.h0 || a.h_last) && a.ldq < a.Q)) {
    set_error("ps_rnn_f32: a frame lies outside the row (ldt=%d) or ldq=%d < Q=%d", a.ldt, a.ldq, a.Q);
    return PS_E_INVALID;
  }
  LstmK k{a};
  const int rows = ng * a.H > a.H * LS ? ng * a.H : a.H * LS;
  const int threads = (rows + 63) / 64 * 64;
  const size_t lds = (size_t)(a.H + ng * a.H + a.H) * sizeof(f32x4);
  dim3 grid((a.Q + LS - 1) / LS, a.N, a.D);
  {
    LaunchTimer timer("lstm", (hipStream_t)stream);
    if (kind == PS_RNN_GRU)
      hipLaunchKernelGGL((rnn_kernel<2>), grid, dim3(threads), lds, (hipStream_t)stream, k, bhn);
    else
      hipLaunchKernelGGL((rnn_kernel<0>), grid, dim3(threads), lds, (hipStream_t)stream, k, bhn);
  }
  return launch_status("ps_rnn_f32");
}

extern "C" int ps_lstm_f32(const ps_lstm_args* args, void* stream) { return lstm_launch(args, stream, false); }

extern "C" int ps_lstm_f16x2_f32(const ps_lstm_args* args, void* stream) { return lstm_launch(args, stream, true); }

static int lstm_launch(const ps_lstm_args* args, void* stream, bool f16x2) {
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
  if ((a.H == 64 || a.H == 128) && !(g_debug_flags & 2)) {
    const long long seqs = (long long)a.N * a.Q;
    // 16-byte step groups: steps are consecutive frames starting on a 16-byte boundary.  A forward-only pass may end
    // in a partial group: it reads / writes up to 3 frames past its last step, which must still lie inside the row
    // (pad frames; they are never read as data).  Without the groups a long pass over consecutive frames re-fetches
    // every 128-byte line of the gate pre-activations once per step (DPCRN's inter pass: 10.8 ms instead of ~3).
    const int steps4 = (a.steps + 3) / 4 * 4;
    const bool tail_ok = a.steps % 4 == 0 || (a.D == 1 && (long long)(a.Q - 1) * a.q_stride + steps4 <= a.ldt);
    const bool contig = a.step_stride == 1 && tail_ok && a.q_stride % 4 == 0 && a.ldt % 4 == 0 &&
                        !((uintptr_t)a.gx & 15) && !((uintptr_t)a.hout & 15);
    // 16 sequences per workgroup when there are enough sequences to fill the chip that way (debug bit 2 / 3 force one)
    const bool wide = (g_debug_flags & 4) ? true : (g_debug_flags & 8) ? false : (a.H == 64 && contig && seqs >= 16 * 256);
    LaunchTimer timer("lstm", (hipStream_t)stream);
    if (wide) {
      dim3 mgrid((unsigned)((seqs + 15) / 16), 1, a.D);
      // whole segments of 20 consecutive frames (DPRNN's intra pass at K = 20): all steps fetched up front
      const bool seg = a.H == 64 && contig && a.steps == 20 && !(g_debug_flags & (1 << 20));
      if (seg && f16x2 && a.D == 1)
        hipLaunchKernelGGL((lstm_seg_f16x2_kernel<64, 20, false>), mgrid, dim3(256), 0, (hipStream_t)stream, k);
      else if (seg && f16x2)
        hipLaunchKernelGGL((lstm_seg_f16x2_kernel<64, 20, true>), mgrid, dim3(256), 0, (hipStream_t)stream, k);
      else if (seg && a.D == 1)
        hipLaunchKernelGGL((lstm_seg_kernel<64, 20, false>), mgrid, dim3(256), 0, (hipStream_t)stream, k);
      else if (seg)
        hipLaunchKernelGGL((lstm_seg_kernel<64, 20, true>), mgrid, dim3(256), 0, (hipStream_t)stream, k);
      else if (a.H == 64 && contig)
        hipLaunchKernelGGL((lstm_mfma_kernel<64, true>), mgrid, dim3(256), 0, (hipStream_t)stream, k);
      else if (a.H == 64)
        hipLaunchKernelGGL((lstm_mfma_kernel<64, false>), mgrid, dim3(256), 0, (hipStream_t)stream, k);
      else  // H = 128: the 16-byte group path does not fit the 256-VGPR budget of 8 waves
        hipLaunchKernelGGL((lstm_mfma_kernel<128, false>), mgrid, dim3(512), 0, (hipStream_t)stream, k);
    } else {
      dim3 mgrid((unsigned)(((seqs + 3) / 4 + 7) / 8 * 8), 1, a.D);  // multiple of 8: see the XCD-aware group order
      if (a.H == 64 && contig)
        hipLaunchKernelGGL((lstm_m4_kernel<64, true>), mgrid, dim3(256), 0, (hipStream_t)stream, k);
      else if (a.H == 64 && f16x2 && !(g_debug_flags & (1 << 20)))
        hipLaunchKernelGGL(lstm_m4_f16x2_kernel, mgrid, dim3(256), 0, (hipStream_t)stream, k);
      else if (a.H == 64)
        hipLaunchKernelGGL((lstm_m4_kernel<64, false>), mgrid, dim3(256), 0, (hipStream_t)stream, k);
      else if (contig)
        hipLaunchKernelGGL((lstm_m4_kernel<128, true>), mgrid, dim3(512), 0, (hipStream_t)stream, k);
      else
        hipLaunchKernelGGL((lstm_m4_kernel<128, false>), mgrid, dim3(512), 0, (hipStream_t)stream, k);
    }
    return launch_status("ps_lstm_f32");
  }
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
    // few frames (streaming step, state rows): split the channels 16 ways instead of 4 to shorten the serial walk
    if ((long long)((T + 63) / 64) * N < 64)
      hipLaunchKernelGGL((chan_layernorm_kernel<16>), dim3((T + 63) / 64, N), dim3(1024), 0, (hipStream_t)stream, a);
    else if (C <= 64 && !(g_debug_flags & (1 << 23)))  // (bit 23: the three-pass kernel; tests run both)
      hipLaunchKernelGGL((chan_layernorm_reg_kernel<16>), dim3((T + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, a);
    else if (C <= 128 && !(g_debug_flags & (1 << 23)))
      hipLaunchKernelGGL((chan_layernorm_reg_kernel<32>), dim3((T + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, a);
    else if (C <= 256 && !(g_debug_flags & (1 << 23)))
      hipLaunchKernelGGL((chan_layernorm_reg_kernel<64>), dim3((T + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL((chan_layernorm_kernel<4>), dim3((T + 63) / 64, N), dim3(256), 0, (hipStream_t)stream, a);
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

extern "C" int ps_lstm_cell_f32(const float* gates, float* c, float* h, int N, int H, int D, int T, int ld_gates,
                                int ld_state, void* stream) {
  if (!gates || !c || !h || N <= 0 || H <= 0 || D < 1 || D > 2 || T <= 0 || ld_gates < T || ld_state < T ||
      (long long)N * D > 65535) {
    set_error("ps_lstm_cell_f32: bad argument (N=%d H=%d D=%d T=%d)", N, H, D, T);
    return PS_E_INVALID;
  }
  {
    LaunchTimer timer("lstm_cell", (hipStream_t)stream);
    hipLaunchKernelGGL(lstm_cell_kernel, dim3((T + 63) / 64, (H + 3) / 4, N * D), dim3(256), 0, (hipStream_t)stream,
                       gates, c, h, H, T, ld_gates, ld_state);
  }
  return launch_status("ps_lstm_cell_f32");
}

// ---- GatedTCN pieces (conv_tasnet.py:129-215) -----------------------------------------------------------------
namespace ps {

// Unfold a dense dilated convolution into a 1x1 one: row (j, k) of the output is input channel k shifted by tap j
// (zero outside [0, T)), so W[m][k][j] becomes a plain [M][P*Kc] matrix for ps_conv1x1_f32.  Optional per-(utterance,
// channel) FiLM scale/shift applied before the zero padding, and E constant embedding rows appended per tap (the
// reference concatenates the repeated embedding BEFORE F.conv1d pads, so its taps drop out at the edges too).
__global__ __launch_bounds__(256) void unfold_taps_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ embed, int K, int E, int T, int T_out,
                                                          int ldt, int P, int dilation, int left) {
  // T = valid input frames, T_out >= T = output frames (the causal gated block of the reference pads both sides and
  // trims only after its output conv: its norms see T + padding frames, conv_tasnet.py:203-211)
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int row = blockIdx.y;  // j * (K + E) + k
  const int n = blockIdx.z;
  const int Kc = K + E;
  const int j = row / Kc, k = row % Kc;
  if (t >= T_out) return;
  const int src = t + j * dilation - left;
  float v = 0.f;
  if (src >= 0 && src < T) {
    if (k < K) {
      v = x[((size_t)n * K + k) * ldt + src];
      if (scale) v = v * scale[(size_t)n * K + k] + shift[(size_t)n * K + k];
    } else {
      v = embed[(size_t)n * E + (k - K)];
    }
  }
  y[((size_t)n * P * Kc + row) * ldt + t] = v;
}

struct GateArgs {
  const float* l;
  const float* r;
  float* y;
  ps_prologue pl, pr;
  int H, T, ldt;
};

// y = PReLU(norm(l)) * sigmoid(PReLU(norm(r))): the two branch tails of the gated block, norms gLN / folded bN1d.
__global__ __launch_bounds__(256) void gated_product_kernel(GateArgs a) {
  __shared__ double red[8];
  const int n = blockIdx.z, ch = blockIdx.y;
  const NormScalars nl = load_norm_scalars(a.pl, n, red);
  const NormScalars nr = load_norm_scalars(a.pr, n, red);
  const int t = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t >= a.T) return;
  const float scl = a.pl.norm != PS_NORM_NONE ? a.pl.gamma[ch] * nl.rstd : 1.f;
  const float shl = (a.pl.norm != PS_NORM_NONE ? a.pl.beta[ch] : 0.f) - nl.mean * scl;
  const float scr = a.pr.norm != PS_NORM_NONE ? a.pr.gamma[ch] * nr.rstd : 1.f;
  const float shr = (a.pr.norm != PS_NORM_NONE ? a.pr.beta[ch] : 0.f) - nr.mean * scr;
  const float sl = a.pl.prelu ? a.pl.slope[0] : 1.f, sr = a.pr.prelu ? a.pr.slope[0] : 1.f;
  const size_t off = ((size_t)n * a.H + ch) * a.ldt + t;
  const f32x4 lv = *reinterpret_cast<const f32x4*>(a.l + off);
  const f32x4 rv = *reinterpret_cast<const f32x4*>(a.r + off);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float lf = prelu(lv[e] * scl + shl, sl);
    const float rf = prelu(rv[e] * scr + shr, sr);
    o[e] = lf * sigmoidf_(rf);
  }
  *reinterpret_cast<f32x4*>(a.y + off) = o;
}

}  // namespace ps

extern "C" int ps_unfold_taps_f32(const float* x, float* y, int N, int K, int T, int ldt, int P, int dilation, int left,
                                  const float* scale, const float* shift, const float* embed, int E, void* stream) {
  return ps_unfold_taps_out_f32(x, y, N, K, T, T, ldt, P, dilation, left, scale, shift, embed, E, stream);
}

extern "C" int ps_unfold_taps_out_f32(const float* x, float* y, int N, int K, int T, int T_out, int ldt, int P,
                                      int dilation, int left, const float* scale, const float* shift,
                                      const float* embed, int E, void* stream) {
  if (!x || !y || N <= 0 || K <= 0 || T <= 0 || T_out < T || ldt < T_out || P <= 0 || dilation <= 0 || left < 0 || E < 0 ||
      (E > 0 && !embed) || ((scale == nullptr) != (shift == nullptr)) || (long long)P * (K + E) > 65535 || N > 65535) {
    set_error("ps_unfold_taps_f32: bad argument (N=%d K=%d T=%d P=%d dilation=%d left=%d E=%d)", N, K, T, P, dilation,
              left, E);
    return PS_E_INVALID;
  }
  {
    LaunchTimer timer("unfold_taps", (hipStream_t)stream);
    hipLaunchKernelGGL(unfold_taps_kernel, dim3((T_out + 255) / 256, P * (K + E), N), dim3(256), 0, (hipStream_t)stream,
                       x, y, scale, shift, embed, K, E, T, T_out, ldt, P, dilation, left);
  }
  return launch_status("ps_unfold_taps_f32");
}

static int check_gate_prologue(const ps_prologue& p, const char* side) {
  if (p.norm != PS_NORM_NONE && (!p.gamma || !p.beta)) {
    set_error("ps_gated_product_f32: %s norm needs gamma/beta", side);
    return PS_E_INVALID;
  }
  if (p.norm == PS_NORM_GLOBAL && (!p.stats || p.parts <= 0 || p.count <= 0)) {
    set_error("ps_gated_product_f32: %s PS_NORM_GLOBAL needs stats/parts/count", side);
    return PS_E_INVALID;
  }
  if (p.prelu && !p.slope) {
    set_error("ps_gated_product_f32: %s prelu needs slope", side);
    return PS_E_INVALID;
  }
  return 0;
}

extern "C" int ps_gated_product_f32(const float* left, const float* right, float* y, int N, int H, int T, int ldt,
                                    const ps_prologue* pro_left, const ps_prologue* pro_right, void* stream) {
  if (!left || !right || !y || !pro_left || !pro_right || N <= 0 || H <= 0 || T <= 0 || ldt < T || H > 65535 ||
      N > 65535) {
    set_error("ps_gated_product_f32: bad argument (N=%d H=%d T=%d ldt=%d)", N, H, T, ldt);
    return PS_E_INVALID;
  }
  if (ldt % 4 || ((uintptr_t)left & 15) || ((uintptr_t)right & 15) || ((uintptr_t)y & 15)) {
    set_error("ps_gated_product_f32: rows must be 16-byte aligned");
    return PS_E_ALIGN;
  }
  int rc = check_gate_prologue(*pro_left, "left");
  if (rc) return rc;
  rc = check_gate_prologue(*pro_right, "right");
  if (rc) return rc;
  GateArgs a{left, right, y, *pro_left, *pro_right, H, T, ldt};
  {
    LaunchTimer timer("gated_product", (hipStream_t)stream);
    hipLaunchKernelGGL(gated_product_kernel, dim3((T + 1023) / 1024, H, N), dim3(256), 0, (hipStream_t)stream, a);
  }
  return launch_status("ps_gated_product_f32");
}

// ---- 50 % overlapped segmentation (SplitMerge.split / merge, lobe/trivial.py:178-241; SkiM.split / merge) ------------
namespace ps {

// mode 0 (split): dst frame s*K + k <- src frame (s/2)*K + k + (s&1)*K/2 - K/2 (zero outside [0, T_src))
// mode 1 (merge): dst frame t <- (src[even cover] + src[odd cover]) / 2
__global__ __launch_bounds__(256) void segment_overlap_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              int T_src, int ld_src, int T_dst, int ld_dst, int K,
                                                              int mode) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  const size_t row = blockIdx.y;
  if (f >= T_dst) return;
  const float* s = src + row * ld_src;
  const int stride = K / 2;
  float v;
  if (mode == 0) {
    const int seg = f / K, k = f % K;
    const int t = (seg >> 1) * K + k + (seg & 1) * stride - stride;
    v = (t >= 0 && t < T_src) ? s[t] : 0.f;
  } else {
    const int e = stride + f;
    const int fa = (2 * (e / K)) * K + e % K;
    const int fb = (2 * (f / K) + 1) * K + f % K;
    v = (s[fa] + s[fb]) * 0.5f;
  }
  dst[row * ld_dst + f] = v;
}

}  // namespace ps

extern "C" int ps_segment_overlap_f32(const float* src, float* dst, int64_t rows, int T_src, int ld_src, int T_dst,
                                      int ld_dst, int K, int merge, void* stream) {
  if (!src || !dst || rows <= 0 || rows > 65535 * 32768LL || T_src <= 0 || T_dst <= 0 || ld_src < T_src ||
      ld_dst < T_dst || K < 2) {
    set_error("ps_segment_overlap_f32: bad argument");
    return PS_E_INVALID;
  }
  const int stride = K / 2;
  if (!merge) {
    // every source index the split reads is checked in the kernel; the destination must be whole segment pairs
    if (T_dst % (2 * K)) {
      set_error("ps_segment_overlap_f32: split destination must hold an even number of %d-frame segments", K);
      return PS_E_INVALID;
    }
  } else {
    // the last merged frame reads even-stream index stride + T_dst - 1 and odd-stream index T_dst - 1
    const long long e = (long long)stride + T_dst - 1;
    const long long fa = (2 * (e / K)) * K + e % K, fb = (2LL * ((T_dst - 1) / K) + 1) * K + (T_dst - 1) % K;
    if (fa >= T_src || fb >= T_src) {
      set_error("ps_segment_overlap_f32: merge source is too short (%d frames)", T_src);
      return PS_E_INVALID;
    }
  }
  using namespace ps;
  LaunchTimer timer("segment_overlap", (hipStream_t)stream);
  const int64_t chunk = 65535;
  for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
    const int64_t nr = rows - r0 < chunk ? rows - r0 : chunk;
    hipLaunchKernelGGL(segment_overlap_kernel, dim3((T_dst + 255) / 256, (unsigned)nr), dim3(256), 0,
                       (hipStream_t)stream, src + r0 * ld_src, dst + r0 * ld_dst, T_src, ld_src, T_dst, ld_dst, K, merge);
  }
  return launch_status("ps_segment_overlap_f32");
}
