// Learned filterbank encoder / decoder (FreeEncDec, lobe/encoder.py:71-94 of mcw519/PureSound), the
// mask application + output constraint fused into the decoder (base_nn.py:81-95,146-159,414-424),
// the padded-layout helpers and the per-utterance embedding bias (conv_tasnet.py:80-85,348-349).
#include "ps_common.h"

namespace ps {

// ------------------------------------------------------------------------------------------------
// Encoder: feats[n][c][t] = act(sum_j w[c][j] * wav[n][t*hop + j]).
// One thread per frame (lanes along t -> coalesced 256-B stores per channel), the frame's window is
// held in registers (WIN known at compile time) and the filter taps are wave-uniform, so they come
// in through the scalar cache.  The write of feats (C*4 bytes per frame) dominates: HBM-bound.
// WIN == 0 selects the run-time window fallback (window re-read through L1).
// ------------------------------------------------------------------------------------------------
constexpr int ENC_CCHUNK = 64;

template <int WIN>
__global__ __launch_bounds__(256) void free_encode_kernel(const float* __restrict__ wav,
                                                          const float* __restrict__ w,
                                                          float* __restrict__ feats, int L, int C, int win,
                                                          int hop, int T, int ldt, int relu, int cchunk) {
  const int n = blockIdx.z;
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int c0 = blockIdx.y * cchunk;
  const int c1 = min(c0 + cchunk, C);
  const bool live = t < T;
  const float* xs = wav + (size_t)n * L + (size_t)(live ? t : 0) * hop;
  float* out = feats + ((size_t)n * C) * ldt + t;
  if (WIN > 0) {
    float xr[WIN > 0 ? WIN : 1];
#pragma unroll
    for (int j = 0; j < WIN; ++j) xr[j] = live ? xs[j] : 0.f;
    for (int c = c0; c < c1; ++c) {
      const float* wc = w + (size_t)c * WIN;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < WIN; ++j) s = fmaf(wc[j], xr[j], s);
      if (relu) s = relu_keep_nan(s);
      if (live) out[(size_t)c * ldt] = s;
    }
  } else {
    if (!live) return;
    for (int c = c0; c < c1; ++c) {
      const float* wc = w + (size_t)c * win;
      float s = 0.f;
      for (int j = 0; j < win; ++j) s = fmaf(wc[j], xs[j], s);
      if (relu) s = relu_keep_nan(s);
      out[(size_t)c * ldt] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Decoder: e = feats * act(mask); frame_out[t][j] = sum_c w[c][j] * e[c][t]; out = constrain(OLA-sum).
// One thread per frame accumulates its WIN outputs over all channels (coalesced reads of feats and
// mask, scalar-cache filter taps); the workgroup overlap-adds through LDS.  A workgroup computes
// R-1 = ceil(WIN/HOP)-1 halo frames on its left so every output sample is written by exactly one
// workgroup (no atomics, deterministic).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float mask_act(float m, int act) {
  if (act == PS_ACT_RELU) return relu_keep_nan(m);
  if (act == PS_ACT_SIGMOID) return 1.f / (1.f + expf(-m));
  return m;
}

__device__ __forceinline__ float out_constrain(float v, int mode) {
  if (mode == PS_OUT_CLAMP) return clamp1_keep_nan(v);
  if (mode == PS_OUT_SIGMOID) return 1.f / (1.f + expf(-v));
  return v;
}

template <int WIN, int HOP>
__global__ __launch_bounds__(256) void free_decode_kernel(const float* __restrict__ feats,
                                                          const float* __restrict__ mask, int mask_mode,
                                                          const float* __restrict__ w, float* __restrict__ out,
                                                          int C, int T, int ldt, int out_mode) {
  constexpr int R = (WIN + HOP - 1) / HOP;
  constexpr int BTF = 256 - (R - 1);  // frames owned per workgroup
  __shared__ float ola[(256 + R) * HOP];
  const int n = blockIdx.y;
  const int tl = threadIdx.x;
  const int tfirst = blockIdx.x * BTF - (R - 1);  // first (halo) frame of this workgroup
  const int t = tfirst + tl;
  const bool live = t >= 0 && t < T;
  const int Lout = (T - 1) * HOP + WIN;

  float acc[WIN];
#pragma unroll
  for (int j = 0; j < WIN; ++j) acc[j] = 0.f;
  if (live) {
    const float* f = feats + (size_t)n * C * ldt + t;
    const float* m = mask ? mask + (size_t)n * C * ldt + t : nullptr;
    // the walk over the channels is bandwidth work with one frame per thread: 8 channels (16 loads) are put in
    // flight together, otherwise each wave has two 256-byte requests outstanding and the kernel sits at 0.2 of HBM
    constexpr int UC = 8;
    int c = 0;
    for (; c + UC <= C; c += UC) {
      float e[UC], mv[UC];
#pragma unroll
      for (int u = 0; u < UC; ++u) e[u] = f[(size_t)(c + u) * ldt];
      if (m) {
#pragma unroll
        for (int u = 0; u < UC; ++u) mv[u] = m[(size_t)(c + u) * ldt];
#pragma unroll
        for (int u = 0; u < UC; ++u) e[u] *= mask_act(mv[u], mask_mode);
      }
#pragma unroll
      for (int u = 0; u < UC; ++u) {
        const float* wc = w + (size_t)(c + u) * WIN;
#pragma unroll
        for (int j = 0; j < WIN; ++j) acc[j] = fmaf(wc[j], e[u], acc[j]);
      }
    }
    for (; c < C; ++c) {
      float e = f[(size_t)c * ldt];
      if (m) e *= mask_act(m[(size_t)c * ldt], mask_mode);
      const float* wc = w + (size_t)c * WIN;
#pragma unroll
      for (int j = 0; j < WIN; ++j) acc[j] = fmaf(wc[j], e, acc[j]);
    }
  }
  // overlap-add in R conflict-free phases: phase r adds acc[r*HOP .. r*HOP+HOP) at slot (tl + r)
  for (int i = tl; i < (256 + R) * HOP; i += 256) ola[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int j = 0; j < HOP; ++j) {
      if (r * HOP + j < WIN) ola[(tl + r) * HOP + j] += acc[r * HOP + j];
    }
    __syncthreads();
  }
  // owned samples: [t_own0*HOP, t_own1*HOP), the last workgroup also owns the tail up to Lout
  const int t_own0 = blockIdx.x * BTF;
  const int last = (t_own0 + BTF >= T);
  const int s0 = t_own0 * HOP;
  const int s1 = last ? Lout : (t_own0 + BTF) * HOP;
  float* o = out + (size_t)n * Lout;
  for (int s = s0 + tl; s < s1; s += 256) {
    // slot index in ola: sample s lives at (s - tfirst*HOP)
    o[s] = out_constrain(ola[s - tfirst * HOP], out_mode);
  }
}

// ---- round 3: encoder and decoder on the matrix pipe (win = 32, hop = 16) -----------------------------------------------
// Both are thin GEMMs -- feats = W [C x 32] . windows [32 x T], frames = W^T [32 x C] . e [C x T] -- and the VALU kernels
// above spend 32 multiply-adds per 4 bytes moved: 61 us of FMA issue per launch at 32 x 512 x 3999 even at a perfect
// VALU rate (measured 113 / 187 us; 0.29 / 0.36 of 8 TB/s).  On v_mfma_f32_32x32x2_f32 (fp32 operands, fp32 accumulate:
// the products are exact, the result differs from the VALU kernels only in summation order) the arithmetic takes 31 us
// of matrix-pipe time per launch and the kernels are the HBM-bound streams they should be.
//
// Encoder: a wave owns a block of 32 channels -- its 16 weight fragments (k pairs of the 32-tap window) stay in registers
// -- and walks frame tiles of 32; the 528 samples a tile touches are staged once per workgroup in LDS (frame stride 17
// floats: the B fragment read x[16 (t0 + lane) + j] would otherwise be an 8-way bank conflict); a store instruction
// covers 32 consecutive frames of two channels (two 128-byte pieces).
#ifndef PS_EM_TILES
#define PS_EM_TILES 8
#endif
constexpr int EM_TILES = PS_EM_TILES;  // frame tiles per workgroup

__global__ __launch_bounds__(256) void free_encode_mfma_kernel(const float* __restrict__ wav, const float* __restrict__ w,
                                                               float* __restrict__ feats, int L, int C, int T, int ldt,
                                                               int relu) {
  __shared__ float xs[34 * 17];
  __shared__ float ws[128 * 33];  // the workgroup's 128 filter rows, row stride 33 floats (conflict-free fragment reads)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the store resource depends on it)
  const int lr = lane & 31, lh = lane >> 5;
  const int n = blockIdx.z;
  const int c_base = (blockIdx.y * 4 + wave) * 32;
  const bool cb_live = c_base < C;  // (C is a multiple of 32: launcher)
  // the 128 rows are 16 KiB contiguous in memory: one coalesced pass into LDS, then every lane picks its row's taps
  // (read straight from memory a lane's 16 taps are 16 loads that touch 32 cache lines each: a workgroup with one tile
  //  took 139 us per launch, with four 88)
  for (int i = tid; i < 128 * 8; i += 256) {
    const int row = i >> 3, q = i & 7;
    const f32x4 v = (blockIdx.y * 128 + row) < C ? *reinterpret_cast<const f32x4*>(w + (size_t)(blockIdx.y * 128 + row) * 32 + 4 * q)
                                                   : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) ws[row * 33 + 4 * q + e] = v[e];
  }
  __syncthreads();
  float wa[16];
#pragma unroll
  for (int kp = 0; kp < 16; ++kp) wa[kp] = ws[(wave * 32 + lr) * 33 + 2 * kp + lh];
  const float* x = wav + (size_t)n * L;
  const int tile0 = blockIdx.x * EM_TILES;
  float pre[3];
  auto fetch = [&](int tile) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int i = tid + 256 * q, sidx = 16 * 32 * tile + i;
      pre[q] = (i < 528 && sidx < L) ? x[sidx] : 0.f;
    }
  };
  fetch(tile0);
  for (int tile = tile0; tile < tile0 + EM_TILES; ++tile) {
    const int t0 = tile * 32;
    if (t0 >= T) break;
    __syncthreads();  // the previous tile's fragment reads are done
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int i = tid + 256 * q;
      if (i < 528) xs[(i >> 4) * 17 + (i & 15)] = pre[q];
    }
    __syncthreads();
    if (tile + 1 < tile0 + EM_TILES && (tile + 1) * 32 < T) fetch(tile + 1);
    if (!cb_live) continue;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
      const int j = 2 * kp + lh;  // tap of this lane's k
      const float b = xs[(lr + (j >> 4)) * 17 + (j & 15)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[kp], b, acc, 0, 0, 0);
    }
    const int t = t0 + lr;
#ifdef PS_ENC_FLAT_STORES
    if (t < T) {
      float* o = feats + ((size_t)n * C + c_base + 4 * lh) * ldt + t;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        if (relu) v = relu_keep_nan(v);
        o[(size_t)((r & 3) + 8 * (r >> 2)) * ldt] = v;
      }
    }
#else
    {  // buffer stores: the wave's 32-channel block as the resource, the row as a scalar offset, frames past T out of range
      const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
          feats + ((size_t)n * C + c_base) * ldt, 0, 32 * ldt * 4, 0x00020000);
      const unsigned vo = t < T ? (unsigned)(4 * lh * ldt + t) * 4u : 0x7ffffff0u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        if (relu) v = relu_keep_nan(v);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, vo, ((r & 3) + 8 * (r >> 2)) * ldt * 4, 0);
      }
    }
#endif
  }
}

// Decoder: a wave owns one tile of 32 frames and walks all channels (k pairs): the B fragment is feats * act(mask) of 32
// consecutive frames of two channels (two 128-byte pieces per load), the A fragment the two channels' 32 filter taps
// (256 contiguous bytes, L2-resident: 64 KiB in all).  The 32 x 32 result holds every frame's 32 output samples; the
// overlap-add of a frame's head with its left neighbour's tail is one lane shift inside the wave.  Only a tile's FIRST
// frame needs the previous tile's last tail: the tile stores its partial head there and its own last tail to a small
// side buffer, and free_decode_fixup_kernel adds the two and applies the output constraint (deterministic: one writer
// per sample, no atomics).
#ifndef PS_DM_UC
#define PS_DM_UC 8
#endif
constexpr int DM_UC = PS_DM_UC;  // k pairs whose loads are in flight together per wave

// Signal scores in the decoder's epilogue (SDRLoss / si_snr of loss/sdr.py:104-299 after _align_waveform,
// base_nn.py:398-412): with MOM the kernels also leave the five moments of (estimate, aligned reference) -- sum a, sum b,
// sum a^2, sum b^2, sum ab in fp64 -- of the samples they complete, one slot per tile (main kernel) and per workgroup (fixup
// kernel), written by one lane each: no atomics, the caller adds a row's slots up in order.  A reference shorter than the
// estimate counts as left-padded with zeros ("align from last"), a longer one is cut.
struct DecodeScore {
  const float* ref;  // [N][ldr]
  double* parts;     // [N][nparts][5]
  int ldr, ref_len, nparts;
};

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <bool MOM>
__global__ __launch_bounds__(256) void free_decode_mfma_kernel(const float* __restrict__ feats, const float* __restrict__ mask,
                                                               int mask_mode, const float* __restrict__ w,
                                                               float* __restrict__ out, float* __restrict__ tails, int C,
                                                               int T, int ldt, int ntiles, int out_mode, DecodeScore sc) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int n = blockIdx.y, tile = blockIdx.x * 4 + wave;
  if (tile >= ntiles) return;
  const int t = tile * 32 + lr;
  const bool live = t < T;
  const int Lout = (T - 1) * 16 + 32;
  // feature / mask rows through buffer loads: the utterance as the resource, the channel pair as a scalar offset, one lane
  // offset (frames past T out of range: they read 0); C is a multiple of 2 DM_UC (launcher), so no channel masks
  const int slab = C * ldt * 4;
  const __amdgpu_buffer_rsrc_t fr =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feats) + (size_t)n * C * ldt, 0, slab, 0x00020000);
  const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(mask ? mask : feats) + (size_t)n * C * ldt, 0, mask ? slab : 0, 0x00020000);
  const unsigned vo = live ? (unsigned)(lh * ldt + t) * 4u : 0x7ffffff0u;
  const float* wl = w + lh * 32 + lr;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int c0 = 0; c0 < C; c0 += 2 * DM_UC) {
    float e[DM_UC], mv[DM_UC], wv[DM_UC];
#pragma unroll
    for (int u = 0; u < DM_UC; ++u) {
      const int so = (c0 + 2 * u) * ldt * 4;
      e[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(fr, vo, so, 0));
      if (mask) mv[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(mr, vo, so, 0));
      wv[u] = wl[(c0 + 2 * u) * 32];
    }
#pragma unroll
    for (int u = 0; u < DM_UC; ++u) {
      const float ev = mask ? e[u] * mask_act(mv[u], mask_mode) : e[u];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u], ev, acc, 0, 0, 0);
    }
  }
  // lane (frame lr, half lh) holds output samples j = (r & 3) + 8 (r >> 2) + 4 lh of its frame: r < 8 the head (j < 16),
  // r >= 8 the tail (j - 16 in the same pattern)
  float o[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float left = __shfl_up(acc[r + 8], 1, 32);  // the left neighbour's tail sample
    o[r] = lr > 0 ? acc[r] + left : acc[r];
  }
  const bool final_here = lr > 0 || tile == 0;  // a tile's first frame still misses the previous tile's tail
  if (final_here) {
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = out_constrain(o[r], out_mode);
  }
  if (t <= T) {  // frame T (all zero) carries the tail of frame T - 1: samples [16 T, 16 T + 16) = the end of the output
    float* dst = out + (size_t)n * Lout + 16 * (size_t)t + 4 * lh;
    *reinterpret_cast<f32x4*>(dst) = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(dst + 8) = f32x4{o[4], o[5], o[6], o[7]};
  }
  if (lr == 31) {
    float* dst = tails + ((size_t)n * ntiles + tile) * 16 + 4 * lh;
    *reinterpret_cast<f32x4*>(dst) = f32x4{acc[8], acc[9], acc[10], acc[11]};
    *reinterpret_cast<f32x4*>(dst + 8) = f32x4{acc[12], acc[13], acc[14], acc[15]};
  }
  if constexpr (MOM) {
    double m[5] = {0., 0., 0., 0., 0.};
    if (t <= T && final_here) {  // the samples this lane completed (a tile's first 16 belong to the fixup kernel)
      const int pad = Lout > sc.ref_len ? Lout - sc.ref_len : 0;
      const float* rr = sc.ref + (size_t)n * sc.ldr;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int sidx = 16 * t + 4 * lh + (r & 3) + 8 * (r >> 2);
        const double a = (double)o[r], b = sidx >= pad ? (double)rr[sidx - pad] : 0.;
        m[0] += a, m[1] += b, m[2] += a * a, m[3] += b * b, m[4] += a * b;
      }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) m[k] = wave_sum_f64(m[k]);
    if (lane == 0) {
      double* dst = sc.parts + ((size_t)n * sc.nparts + tile) * 5;
#pragma unroll
      for (int k = 0; k < 5; ++k) dst[k] = m[k];
    }
  }
}

// MOM: grid (ceil(16 (ntiles - 1) / 256), N) so that a workgroup's moments belong to one utterance; slot ntiles + blockIdx.x
template <bool MOM>
__global__ __launch_bounds__(256) void free_decode_fixup_kernel(float* __restrict__ out, const float* __restrict__ tails,
                                                                int T, int ntiles, int N, int out_mode, DecodeScore sc) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // (n, tile >= 1, j < 16)
  const int j = idx & 15, rest = idx >> 4;
  const int tile = MOM ? rest + 1 : rest % (ntiles - 1) + 1, n = MOM ? (int)blockIdx.y : rest / (ntiles - 1);
  const int Lout = (T - 1) * 16 + 32;
  const size_t sidx = (size_t)16 * 32 * tile + j;
  const bool live = n < N && tile < ntiles && (int)sidx < Lout;
  if (!MOM && !live) return;
  float a = 0.f;
  if (live) {
    float* o = out + (size_t)n * Lout + sidx;
    a = out_constrain(*o + tails[((size_t)n * ntiles + tile - 1) * 16 + j], out_mode);
    *o = a;
  }
  if constexpr (MOM) {
    __shared__ double red[4][5];
    const int pad = Lout > sc.ref_len ? Lout - sc.ref_len : 0;
    const double ad = (double)a, b = live && (int)sidx >= pad ? (double)sc.ref[(size_t)n * sc.ldr + sidx - pad] : 0.;
    double m[5] = {ad, b, ad * ad, b * b, ad * b};
#pragma unroll
    for (int k = 0; k < 5; ++k) m[k] = wave_sum_f64(m[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int k = 0; k < 5; ++k) red[threadIdx.x >> 6][k] = m[k];
    }
    __syncthreads();
    if (threadIdx.x < 5)
      sc.parts[((size_t)n * sc.nparts + ntiles + blockIdx.x) * 5 + threadIdx.x] =
          ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
  }
}

// Run-time (win, hop) fallback: one thread per output sample.
__global__ __launch_bounds__(256) void free_decode_generic_kernel(const float* __restrict__ feats,
                                                                  const float* __restrict__ mask, int mask_mode,
                                                                  const float* __restrict__ w,
                                                                  float* __restrict__ out, int C, int T, int ldt,
                                                                  int win, int hop, int out_mode) {
  const int n = blockIdx.y;
  const int Lout = (T - 1) * hop + win;
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= Lout) return;
  int t_hi = s / hop;
  if (t_hi > T - 1) t_hi = T - 1;
  // smallest frame t with t*hop + win > s
  const int t_lo = (s - win + 1 <= 0) ? 0 : (s - win + 1 + hop - 1) / hop;
  float accv = 0.f;
  for (int t = t_lo; t <= t_hi; ++t) {
    const int j = s - t * hop;
    const float* f = feats + (size_t)n * C * ldt + t;
    const float* m = mask ? mask + (size_t)n * C * ldt + t : nullptr;
    // four independent partial sums keep several channel loads in flight (the walk is latency bound)
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    int c = 0;
    for (; c + 4 <= C; c += 4) {
      float e0 = f[(size_t)c * ldt], e1 = f[(size_t)(c + 1) * ldt], e2 = f[(size_t)(c + 2) * ldt],
            e3 = f[(size_t)(c + 3) * ldt];
      if (m) {
        e0 *= mask_act(m[(size_t)c * ldt], mask_mode);
        e1 *= mask_act(m[(size_t)(c + 1) * ldt], mask_mode);
        e2 *= mask_act(m[(size_t)(c + 2) * ldt], mask_mode);
        e3 *= mask_act(m[(size_t)(c + 3) * ldt], mask_mode);
      }
      p0 = fmaf(w[(size_t)c * win + j], e0, p0);
      p1 = fmaf(w[(size_t)(c + 1) * win + j], e1, p1);
      p2 = fmaf(w[(size_t)(c + 2) * win + j], e2, p2);
      p3 = fmaf(w[(size_t)(c + 3) * win + j], e3, p3);
    }
    for (; c < C; ++c) {
      float e = f[(size_t)c * ldt];
      if (m) e *= mask_act(m[(size_t)c * ldt], mask_mode);
      p0 = fmaf(w[(size_t)c * win + j], e, p0);
    }
    accv += (p0 + p1) + (p2 + p3);
  }
  out[(size_t)n * Lout + s] = out_constrain(accv, out_mode);
}

// hop == win (no overlap; the streaming harness decodes one frame per stream): one workgroup per frame, threads =
// win output samples x 256/win channel parts, partial sums meet in LDS.
__global__ __launch_bounds__(256) void free_decode_frame_kernel(const float* __restrict__ feats,
                                                                const float* __restrict__ mask, int mask_mode,
                                                                const float* __restrict__ w, float* __restrict__ out,
                                                                int C, int T, int ldt, int win, int out_mode) {
  __shared__ float red[256];
  const int t = blockIdx.x, n = blockIdx.y;
  const int parts = 256 / win;
  const int j = threadIdx.x % win, cp = threadIdx.x / win;
  const float* f = feats + (size_t)n * C * ldt + t;
  const float* m = mask ? mask + (size_t)n * C * ldt + t : nullptr;
  float s = 0.f;
  for (int c = cp; c < C; c += parts) {
    float e = f[(size_t)c * ldt];
    if (m) e *= mask_act(m[(size_t)c * ldt], mask_mode);
    s = fmaf(w[(size_t)c * win + j], e, s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (cp == 0) {
    float tot = 0.f;
    for (int p = 0; p < parts; ++p) tot += red[p * win + j];
    out[((size_t)n * T + t) * win + j] = out_constrain(tot, out_mode);
  }
}

// Averaging overlap-add of the streaming harness (egs/tse/demo/utils.py:121-128): the first `ov` samples of the new
// frame are averaged with the tail of the running output, the rest is copied.
__global__ __launch_bounds__(256) void overlap_average_kernel(const float* __restrict__ tail, int ld_tail,
                                                              const float* __restrict__ cur, float* __restrict__ out,
                                                              int win, int ov) {
  const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (j >= win) return;
  const float v = cur[(size_t)b * win + j];
  out[(size_t)b * win + j] = j < ov ? (tail[(size_t)b * ld_tail + j] + v) * 0.5f : v;
}

// The harness' sliding window for all hops of a chunk at once (egs/tse/demo/utils.py:100-118 runs it hop by hop):
// sig_b = queue[b][hop .. win) ++ chunk[b][0 .. hops * hop); wins[i][b * win + j] = sig_b[i * hop + j].
__global__ __launch_bounds__(256) void stream_windows_kernel(const float* __restrict__ queue, const float* __restrict__ chunk,
                                                             float* __restrict__ wins, int B, int hops, int win, int hop) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // (i, b, j)
  if (idx >= hops * B * win) return;
  const int j = idx % win, b = (idx / win) % B, i = idx / (win * B);
  const int s = i * hop + j;  // position in sig_b; its first win - hop samples are the tail of the previous window
  const int keep = win - hop;
  wins[idx] = s < keep ? queue[(size_t)b * win + hop + s] : chunk[(size_t)b * hops * hop + (s - keep)];
}

// Averaging overlap-add of all hops of a chunk (utils.py:121-128, win = 2 hop): block i of stream b is the mean of the
// previous frame's second half (the running tail for i = 0) and frame i's first half; then tail <- the last frame's
// second half and queue <- the last window.  One thread per (stream, sample of a hop) walks the hops in order, so the
// tail it reads is the one it later replaces.
__global__ __launch_bounds__(256) void stream_overlap_kernel(const float* __restrict__ frames, const float* __restrict__ wins,
                                                             float* __restrict__ tail, float* __restrict__ blocks,
                                                             float* __restrict__ queue, int B, int hops, int win, int hop) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * win) return;
  const int j = idx % win, b = idx / win;
  queue[idx] = wins[(size_t)(hops - 1) * B * win + idx];
  if (j >= hop) return;
  float prev = tail[(size_t)b * hop + j];
  for (int i = 0; i < hops; ++i) {
    const float* f = frames + ((size_t)i * B + b) * win;
    blocks[(size_t)b * hops * hop + i * hop + j] = (prev + f[j]) * 0.5f;
    prev = f[hop + j];
  }
  tail[(size_t)b * hop + j] = prev;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                       int64_t rows, int T, int ldt) {
  const int64_t row = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (row < rows && t < ldt) dst[row * ldt + t] = t < T ? src[row * T + t] : 0.f;
}

__global__ __launch_bounds__(256) void unpad_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         int64_t rows, int T, int ldt) {
  const int64_t row = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (row < rows && t < T) dst[row * T + t] = src[row * ldt + t];
}

// bias_n[n][m] = sum_e W[m][e] * dvec_hat[n][e]; one wave per (n, m).
__global__ __launch_bounds__(256) void embed_bias_kernel(const float* __restrict__ dvec,
                                                         const float* __restrict__ w, float* __restrict__ bias_n,
                                                         int E, int M, int normalize) {
  const int n = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  const float* d = dvec + (size_t)n * E;
  float nrm = 0.f, dot = 0.f;
  for (int e = lane; e < E; e += 64) {
    const float v = d[e];
    nrm = fmaf(v, v, nrm);
    if (m < M) dot = fmaf(w[(size_t)m * E + e], v, dot);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    nrm += __shfl_xor(nrm, off, 64);
    dot += __shfl_xor(dot, off, 64);
  }
  if (lane == 0 && m < M) {
    const float scale = normalize ? 1.f / fmaxf(sqrtf(nrm), 1e-12f) : 1.f;
    bias_n[(size_t)n * M + m] = dot * scale;
  }
}

static int check_launch(const char* who) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", who, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

}  // namespace ps

using namespace ps;

extern "C" int ps_pad_rows_f32(const float* src, float* dst, int64_t rows, int T, int ldt, void* stream) {
  if (!src || !dst || rows <= 0 || T <= 0 || ldt < T || rows > 0x7fffffff) {
    set_error("ps_pad_rows_f32: bad argument");
    return PS_E_INVALID;
  }
  // grid.y is limited to 65535: fold rows
  const int64_t chunk = 65535;
  LaunchTimer timer("pad_rows", (hipStream_t)stream);
  for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
    const int64_t nr = rows - r0 < chunk ? rows - r0 : chunk;
    hipLaunchKernelGGL(pad_rows_kernel, dim3((ldt + 255) / 256, (unsigned)nr), dim3(256), 0, (hipStream_t)stream,
                       src + r0 * T, dst + r0 * ldt, nr, T, ldt);
  }
  return check_launch("ps_pad_rows_f32");
}

extern "C" int ps_unpad_rows_f32(const float* src, float* dst, int64_t rows, int T, int ldt, void* stream) {
  if (!src || !dst || rows <= 0 || T <= 0 || ldt < T || rows > 0x7fffffff) {
    set_error("ps_unpad_rows_f32: bad argument");
    return PS_E_INVALID;
  }
  const int64_t chunk = 65535;
  LaunchTimer timer("unpad_rows", (hipStream_t)stream);
  for (int64_t r0 = 0; r0 < rows; r0 += chunk) {
    const int64_t nr = rows - r0 < chunk ? rows - r0 : chunk;
    hipLaunchKernelGGL(unpad_rows_kernel, dim3((T + 255) / 256, (unsigned)nr), dim3(256), 0, (hipStream_t)stream,
                       src + r0 * ldt, dst + r0 * T, nr, T, ldt);
  }
  return check_launch("ps_unpad_rows_f32");
}

extern "C" int ps_free_encode_f32(const float* wav, const float* w, float* feats, int N, int L, int C, int win,
                                  int hop, int T, int ldt, int relu, void* stream) {
  if (!wav || !w || !feats || N <= 0 || C <= 0 || win <= 0 || hop <= 0 || L < win) {
    set_error("ps_free_encode_f32: bad argument (N=%d L=%d C=%d win=%d hop=%d)", N, L, C, win, hop);
    return PS_E_INVALID;
  }
  if (T != (L - win) / hop + 1 || ldt < T || ldt % kTileT != 0) {
    set_error("ps_free_encode_f32: T=%d must equal floor((L-win)/hop)+1=%d, ldt=%d a multiple of %d >= T", T,
              (L - win) / hop + 1, ldt, kTileT);
    return PS_E_INVALID;
  }
  // few frames (streaming: one frame per stream): spread the channels over more workgroups
  const int cchunk = ((long long)N * ((T + 255) / 256) * ((C + ENC_CCHUNK - 1) / ENC_CCHUNK) < 64) ? 4 : ENC_CCHUNK;
  dim3 grid((T + 255) / 256, (C + cchunk - 1) / cchunk, N);
  hipStream_t s = (hipStream_t)stream;
  LaunchTimer timer("free_encode", s);
  // long rows of the benchmark's filterbank: the matrix-pipe kernel (ps_debug_flags bit 0 keeps the VALU kernels of
  // rounds 1-2: tests run both)
  if (win == 32 && hop == 16 && C % 32 == 0 && T >= 64 && !(g_debug_flags & 1)) {
    dim3 g((T + 32 * EM_TILES - 1) / (32 * EM_TILES), (C + 127) / 128, N);
    hipLaunchKernelGGL(free_encode_mfma_kernel, g, dim3(256), 0, s, wav, w, feats, L, C, T, ldt, relu);
    return check_launch("ps_free_encode_f32");
  }
  if (win == 32)
    hipLaunchKernelGGL(free_encode_kernel<32>, grid, dim3(256), 0, s, wav, w, feats, L, C, win, hop, T, ldt, relu,
                       cchunk);
  else if (win == 16)
    hipLaunchKernelGGL(free_encode_kernel<16>, grid, dim3(256), 0, s, wav, w, feats, L, C, win, hop, T, ldt, relu,
                       cchunk);
  else
    hipLaunchKernelGGL(free_encode_kernel<0>, grid, dim3(256), 0, s, wav, w, feats, L, C, win, hop, T, ldt, relu,
                       cchunk);
  return check_launch("ps_free_encode_f32");
}

extern "C" int ps_free_decode_f32(const float* feats, const float* mask, int mask_act, const float* w, float* out,
                                  int N, int C, int T, int ldt, int win, int hop, int out_mode, void* stream) {
  if (!feats || !w || !out || N <= 0 || C <= 0 || T <= 0 || win <= 0 || hop <= 0 || ldt < T) {
    set_error("ps_free_decode_f32: bad argument (N=%d C=%d T=%d win=%d hop=%d)", N, C, T, win, hop);
    return PS_E_INVALID;
  }
  if (mask_act < PS_ACT_LINEAR || mask_act > PS_ACT_SIGMOID || out_mode < PS_OUT_CLAMP || out_mode > PS_OUT_NONE) {
    set_error("ps_free_decode_f32: unknown mask_act=%d or out_mode=%d", mask_act, out_mode);
    return PS_E_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  LaunchTimer timer("free_decode", s);
  if (win == 32 && hop == 16) {
    constexpr int BTF = 255;
    hipLaunchKernelGGL((free_decode_kernel<32, 16>), dim3((T + BTF - 1) / BTF, N), dim3(256), 0, s, feats, mask,
                       mask_act, w, out, C, T, ldt, out_mode);
  } else if (win == 16 && hop == 8) {
    constexpr int BTF = 255;
    hipLaunchKernelGGL((free_decode_kernel<16, 8>), dim3((T + BTF - 1) / BTF, N), dim3(256), 0, s, feats, mask,
                       mask_act, w, out, C, T, ldt, out_mode);
  } else if (hop == win && win <= 256 && 256 % win == 0 && T <= 65535) {
    hipLaunchKernelGGL(free_decode_frame_kernel, dim3(T, N), dim3(256), 0, s, feats, mask, mask_act, w, out, C, T, ldt,
                       win, out_mode);
  } else {
    const int Lout = (T - 1) * hop + win;
    hipLaunchKernelGGL(free_decode_generic_kernel, dim3((Lout + 255) / 256, N), dim3(256), 0, s, feats, mask,
                       mask_act, w, out, C, T, ldt, win, hop, out_mode);
  }
  return check_launch("ps_free_decode_f32");
}

extern "C" size_t ps_free_decode_workspace_bytes(int N, int T, int win, int hop) {
  if (N <= 0 || T <= 0 || win != 32 || hop != 16) return 0;
  return (size_t)N * (T / 32 + 1) * 16 * sizeof(float);  // one 16-sample tail per 32-frame tile
}

extern "C" int ps_free_decode_ws_f32(const float* feats, const float* mask, int mask_act, const float* w, float* out,
                                     int N, int C, int T, int ldt, int win, int hop, int out_mode, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  const size_t need = ps_free_decode_workspace_bytes(N, T, win, hop);
  // the matrix-pipe kernel: the benchmark's filterbank on long rows, with the side buffer it needs (ps_debug_flags
  // bit 0 keeps the VALU kernel; so does a missing or short workspace)
  if (need == 0 || !workspace || workspace_bytes < need || T < 64 || N > 65535 || ((uintptr_t)out & 15) ||
      ((uintptr_t)workspace & 15) || C % (2 * ps::DM_UC) || (long long)C * ldt * 4 >= (1ll << 31) || (g_debug_flags & 1))
    return ps_free_decode_f32(feats, mask, mask_act, w, out, N, C, T, ldt, win, hop, out_mode, stream);
  if (!feats || !w || !out || C <= 0 || ldt < T) {
    set_error("ps_free_decode_ws_f32: bad argument (N=%d C=%d T=%d)", N, C, T);
    return PS_E_INVALID;
  }
  if (mask_act < PS_ACT_LINEAR || mask_act > PS_ACT_SIGMOID || out_mode < PS_OUT_CLAMP || out_mode > PS_OUT_NONE) {
    set_error("ps_free_decode_ws_f32: unknown mask_act=%d or out_mode=%d", mask_act, out_mode);
    return PS_E_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  LaunchTimer timer("free_decode", s);
  const int ntiles = T / 32 + 1;
  hipLaunchKernelGGL(free_decode_mfma_kernel<false>, dim3((ntiles + 3) / 4, N), dim3(256), 0, s, feats, mask, mask_act, w,
                     out, (float*)workspace, C, T, ldt, ntiles, out_mode, ps::DecodeScore{});
  const long long fix = (long long)N * (ntiles - 1) * 16;
  hipLaunchKernelGGL(free_decode_fixup_kernel<false>, dim3((unsigned)((fix + 255) / 256)), dim3(256), 0, s, out,
                     (const float*)workspace, T, ntiles, N, out_mode, ps::DecodeScore{});
  return check_launch("ps_free_decode_ws_f32");
}

extern "C" int ps_free_decode_moments_parts(int N, int C, int T, int ldt, int win, int hop) {
  if (N <= 0 || N > 65535 || C <= 0 || T < 64 || ldt < T || win != 32 || hop != 16 || C % (2 * ps::DM_UC) ||
      (long long)C * ldt * 4 >= (1ll << 31))
    return 0;
  const int ntiles = T / 32 + 1;
  return ntiles + ((ntiles - 1) * 16 + 255) / 256;
}

extern "C" int ps_free_decode_moments_f32(const float* feats, const float* mask, int mask_act, const float* w, float* out,
                                          int N, int C, int T, int ldt, int win, int hop, int out_mode, const float* ref,
                                          int ldr, int ref_len, double* partials, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  const int nparts = ps_free_decode_moments_parts(N, C, T, ldt, win, hop);
  if (nparts == 0) {
    set_error("ps_free_decode_moments_f32: shape outside the fused kernel (N=%d C=%d T=%d ldt=%d win=%d hop=%d): decode, "
              "then ps_wave_moments_f64", N, C, T, ldt, win, hop);
    return PS_E_UNSUPPORTED;
  }
  if (!feats || !w || !out || !ref || !partials || !workspace || ref_len <= 0 || ldr < ref_len ||
      workspace_bytes < ps_free_decode_workspace_bytes(N, T, win, hop) || ((uintptr_t)out & 15) ||
      ((uintptr_t)workspace & 15)) {
    set_error("ps_free_decode_moments_f32: null / unaligned pointer, short workspace or bad reference row (ref_len=%d "
              "ldr=%d)", ref_len, ldr);
    return PS_E_INVALID;
  }
  if (mask_act < PS_ACT_LINEAR || mask_act > PS_ACT_SIGMOID || out_mode < PS_OUT_CLAMP || out_mode > PS_OUT_NONE) {
    set_error("ps_free_decode_moments_f32: unknown mask_act=%d or out_mode=%d", mask_act, out_mode);
    return PS_E_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  LaunchTimer timer("free_decode_moments", s);
  const int ntiles = T / 32 + 1;
  const ps::DecodeScore sc{ref, partials, ldr, ref_len, nparts};
  hipLaunchKernelGGL(free_decode_mfma_kernel<true>, dim3((ntiles + 3) / 4, N), dim3(256), 0, s, feats, mask, mask_act, w,
                     out, (float*)workspace, C, T, ldt, ntiles, out_mode, sc);
  hipLaunchKernelGGL(free_decode_fixup_kernel<true>, dim3(((ntiles - 1) * 16 + 255) / 256, N), dim3(256), 0, s, out,
                     (const float*)workspace, T, ntiles, N, out_mode, sc);
  return check_launch("ps_free_decode_moments_f32");
}

extern "C" int ps_overlap_average_f32(const float* tail, int ld_tail, const float* cur, float* out, int B, int win,
                                      int overlap, void* stream) {
  if (!tail || !cur || !out || B <= 0 || B > 65535 || win <= 0 || overlap < 0 || overlap > win || ld_tail < overlap) {
    set_error("ps_overlap_average_f32: bad argument (B=%d win=%d overlap=%d)", B, win, overlap);
    return PS_E_INVALID;
  }
  LaunchTimer timer("overlap_average", (hipStream_t)stream);
  hipLaunchKernelGGL(overlap_average_kernel, dim3((win + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, tail,
                     ld_tail, cur, out, win, overlap);
  return check_launch("ps_overlap_average_f32");
}

extern "C" int ps_stream_windows_f32(const float* queue, const float* chunk, float* wins, int B, int hops, int win,
                                     int hop, void* stream) {
  if (!queue || !chunk || !wins || B <= 0 || hops <= 0 || hop <= 0 || win != 2 * hop ||
      (long long)B * hops * win > 0x7fffffffLL) {
    set_error("ps_stream_windows_f32: bad argument (B=%d hops=%d win=%d hop=%d; win must be 2 * hop)", B, hops, win, hop);
    return PS_E_INVALID;
  }
  LaunchTimer timer("stream_windows", (hipStream_t)stream);
  hipLaunchKernelGGL(stream_windows_kernel, dim3((hops * B * win + 255) / 256), dim3(256), 0, (hipStream_t)stream, queue,
                     chunk, wins, B, hops, win, hop);
  return check_launch("ps_stream_windows_f32");
}

extern "C" int ps_stream_overlap_f32(const float* frames, const float* wins, float* tail, float* blocks, float* queue,
                                     int B, int hops, int win, int hop, void* stream) {
  if (!frames || !wins || !tail || !blocks || !queue || B <= 0 || hops <= 0 || hop <= 0 || win != 2 * hop) {
    set_error("ps_stream_overlap_f32: bad argument (B=%d hops=%d win=%d hop=%d; win must be 2 * hop)", B, hops, win, hop);
    return PS_E_INVALID;
  }
  LaunchTimer timer("stream_overlap", (hipStream_t)stream);
  hipLaunchKernelGGL(stream_overlap_kernel, dim3((B * win + 255) / 256), dim3(256), 0, (hipStream_t)stream, frames, wins,
                     tail, blocks, queue, B, hops, win, hop);
  return check_launch("ps_stream_overlap_f32");
}

extern "C" int ps_embed_bias_f32(const float* dvec, const float* w_embed, float* bias_n, int N, int E, int M,
                                 int normalize, void* stream) {
  if (!dvec || !w_embed || !bias_n || N <= 0 || E <= 0 || M <= 0) {
    set_error("ps_embed_bias_f32: bad argument");
    return PS_E_INVALID;
  }
  LaunchTimer timer("embed_bias", (hipStream_t)stream);
  hipLaunchKernelGGL(embed_bias_kernel, dim3((M + 3) / 4, N), dim3(256), 0, (hipStream_t)stream, dvec, w_embed,
                     bias_n, E, M, normalize);
  return check_launch("ps_embed_bias_f32");
}
