// 1x1 convolution on the bf16 matrix pipe with fp32 in, fp32 accumulate, fp32 out:
//   y[n][m][t] = sum_k W[m][k] * f(x[n][k][t]) (+ bias, + bias_n, + res), optional partial statistics of y
// f = the same consumer-side prologue as ps_conv1x1_f32 (gLN / folded bN1d, PReLU, ReLU-before / tanh-after).
//
// PLANES = 1: operands rounded to bf16 (the "bf16" configurations of BASELINE.json: storage fp32, products bf16,
//             accumulation fp32).
// PLANES = 3: every fp32 operand is split into three bf16 terms x = x0 + x1 + x2 (x0 = bf16(x), x1 = bf16(x - x0),
//             x2 = bf16(x - x0 - x1)) and the six products whose magnitude reaches 2^-16 of the leading one are
//             accumulated: W0x0, W0x1, W1x0, W0x2, W2x0, W1x1.  The dropped terms are <= 2^-24 relative, i.e. the
//             result is fp32-accurate (measured on the full config-2 forward: the deviation from the reference equals
//             the reference's own fp32 reordering noise) at 6 bf16 MFMAs = 0.375 of the cost of the
//             fp32 MFMA per k.
//
// v_mfma_f32_32x32x16_bf16: A fragment = 8 consecutive k of one row, B fragment = 8 consecutive k of one column, so
// both operand tiles live in LDS k-innermost ([row][16 k] bf16 = 32 B per row, read with ds_read_b128, conflict free).
// Weights arrive already split and in exactly that image from the host packer ([m-tile][k-step][plane][256][16]), so
// a K-step of weights is a straight 8*PLANES KiB copy.  Activations are fp32 [k][t] in HBM: thread (t, k-half) loads
// its 8 k values of one frame (coalesced along t), applies the prologue, splits, and writes one 16-byte LDS row piece
// per plane.  Tile 256 (m) x 128 (t) per workgroup, 2 x 2 waves of 128 x 64 (the accumulator layout and the epilogue
// are those of the fp32 kernel), two LDS slots, one barrier per K-step, global loads of step s+1 in flight over the
// MFMAs of step s.
#include <type_traits>

#include "ps_common.h"

namespace ps {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

constexpr int XB_M = 256, XB_T = 128, XB_K = 16;

struct BfArgs {
  const float* x;
  const unsigned short* wt;  // [tiles_m][ksteps][PLANES][256][16] bf16
  float* y;
  const float* bias;
  const float* bias_n;
  const float* res;
  double* ostats;
  ps_prologue pro;
  int K, M, T, ldt, ksteps, tiles_t, tiles_m, N;
  unsigned long long* stamps;  // ps_debug_buffer(): 6 x u64 per workgroup (s_memtime buckets), diagnostics only
  int ablate;  // profiling only (ps_debug_flags bits 24..26): 1 = no MFMA, 2 = no epilogue, 4 = no activation split
};

template <int PLANES>
struct BfLds {
  static constexpr int A_BYTES = PLANES * XB_M * XB_K * 2;  // 8 KiB per plane
  static constexpr int B_BYTES = PLANES * XB_T * XB_K * 2;  // 4 KiB per plane
  static constexpr int SLOT = A_BYTES + B_BYTES;
  static constexpr int KTAB = 512;   // floats per table: sc / sh of the prologue (K <= 512)
  static constexpr int TOTAL = 2 * SLOT + 2 * KTAB * 4 + 64;  // PLANES = 3: 76 KiB -> two workgroups per CU
};

template <int PLANES, bool TR, bool STATS, bool RES>
__global__ __launch_bounds__(256, 2) void conv1x1_bf16_kernel(BfArgs a) {
  using L = BfLds<PLANES>;
  __shared__ __attribute__((aligned(16))) unsigned char smem[L::TOTAL];
  float* tab = reinterpret_cast<float*>(smem + 2 * L::SLOT);  // sc[512] | sh[512]
  double* red = reinterpret_cast<double*>(smem + 2 * L::SLOT + 2 * L::KTAB * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wt = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int t0 = blockIdx.x * XB_T, mt = blockIdx.y, n = blockIdx.z;
  const int m0 = mt * XB_M;

  // prologue tables (per utterance): u = x * sc[k] + sh[k]
  const bool has_norm = TR && a.pro.norm != PS_NORM_NONE;
  const float slope = (TR && a.pro.prelu) ? a.pro.slope[0] : 1.f;
  if constexpr (TR) {
    const NormScalars ns = load_norm_scalars(a.pro, n, red);
    for (int k = tid; k < a.ksteps * XB_K; k += 256) {
      float sc = 0.f, sh = 0.f;
      if (k < a.K) {
        sc = has_norm ? a.pro.gamma[k] * ns.rstd : 1.f;
        sh = has_norm ? a.pro.beta[k] - ns.mean * sc : 0.f;
      }
      tab[k] = sc;
      tab[512 + k] = sh;
    }
    __syncthreads();
  }

  // ---- staging ------------------------------------------------------------------------------------------------
  constexpr int A_PIECES = 2 * PLANES;  // 16-byte pieces per thread per K-step
  const u32x4v* wsrc = reinterpret_cast<const u32x4v*>(a.wt) + (size_t)mt * a.ksteps * (L::A_BYTES / 16);
  const int bt = tid & 127, bh = tid >> 7;  // activation staging: frame, k-half
  // weights (L2 resident) are fetched one K-step ahead, activations (HBM) two: breg is a two-deep register queue
  constexpr int A_DEPTH = PLANES == 1 ? 2 : 1;  // weight K-steps in flight (register budget)
  u32x4v areg[A_DEPTH][A_PIECES];
  float breg[2][8];
  const int a_last = a.ksteps - 1;
  auto load_a = [&](int ks, auto q_c) {  // K-steps past the end re-read the last one (never stored)
    constexpr int q = decltype(q_c)::value;
    const int kc = ks < a_last ? ks : a_last;
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) areg[q][i] = wsrc[(size_t)kc * (L::A_BYTES / 16) + tid + 256 * i];
  };
  // activation loads go through a buffer descriptor over this utterance's [K][ldt] slab: rows k >= K and K-steps
  // past the end read 0.0f, so the loads carry no predicate (a predicated load made hipcc wait on each one)
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x) + (size_t)n * a.K * a.ldt, 0, a.K * a.ldt * 4, 0x00020000);
  const int xb_voff = (8 * bh * a.ldt + t0 + bt) * 4;
  auto load_b = [&](int ks, auto q_c) {
    constexpr int q = decltype(q_c)::value;
    const int soff = ks * XB_K * a.ldt * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      breg[q][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, xb_voff, soff + j * a.ldt * 4, 0));
  };
  auto store_step = [&](int ks, int slot, auto q_c) {
    constexpr int q = decltype(q_c)::value;
    unsigned char* sa = smem + slot * L::SLOT;
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) reinterpret_cast<u32x4v*>(sa)[tid + 256 * i] = areg[A_DEPTH - 1][i];
    if (a.ablate & 4) return;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float u = breg[q][j];
      if constexpr (TR) {
        const int k = ks * XB_K + 8 * bh + j;
        if (a.pro.pre_relu) u = fmaxf(u, 0.f);
        u = u * tab[k] + tab[512 + k];
        if (a.pro.prelu) u = prelu(u, slope);
        if (a.pro.post_tanh) u = tanhf(u);
        if (k >= a.K) u = 0.f;
      }
      v[j] = u;
    }
    unsigned char* sb = sa + L::A_BYTES;
#pragma unroll
    for (int p = 0; p < PLANES; ++p) {
      bf16x8 piece;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const bf16x2 h = __builtin_convertvector(f32x2v{v[j], v[j + 1]}, bf16x2);
        piece[j] = h[0];
        piece[j + 1] = h[1];
        if (p + 1 < PLANES) {
          const f32x2v back = __builtin_convertvector(h, f32x2v);
          v[j] -= back[0];
          v[j + 1] -= back[1];
        }
      }
      *reinterpret_cast<bf16x8*>(sb + ((p * XB_T + bt) * XB_K + 8 * bh) * 2) = piece;
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ti][r] = 0.f;

  auto compute = [&](int slot) {
    if (a.ablate & 1) return;
    const unsigned char* sa = smem + slot * L::SLOT;
    const unsigned char* sb = sa + L::A_BYTES;
    bf16x8 bf[PLANES][2];
#pragma unroll
    for (int p = 0; p < PLANES; ++p)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
        bf[p][ti] = *reinterpret_cast<const bf16x8*>(sb + ((p * XB_T + wt * 64 + ti * 32 + lr) * XB_K + 8 * lh) * 2);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      bf16x8 af[PLANES];
#pragma unroll
      for (int p = 0; p < PLANES; ++p)
        af[p] = *reinterpret_cast<const bf16x8*>(sa + ((p * XB_M + wm * 128 + mi * 32 + lr) * XB_K + 8 * lh) * 2);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        if constexpr (PLANES == 3) {
          // smallest terms first
          acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1][ti], acc[mi][ti], 0, 0, 0);
          acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0][ti], acc[mi][ti], 0, 0, 0);
          acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2][ti], acc[mi][ti], 0, 0, 0);
          acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0][ti], acc[mi][ti], 0, 0, 0);
          acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1][ti], acc[mi][ti], 0, 0, 0);
        }
        acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0][ti], acc[mi][ti], 0, 0, 0);
      }
    }
  };

  // ---- K loop -----------------------------------------------------------------------------------------------------
  using q0 = std::integral_constant<int, 0>;
  using q1 = std::integral_constant<int, 1>;
  // queues: areg[A_DEPTH-1] / breg[1] hold the NEXT step to be written to LDS, the lower entries the ones after it
  load_a(0, std::integral_constant<int, A_DEPTH - 1>{});
  load_b(0, q1{});
  store_step(0, 0, q1{});
  load_a(1, std::integral_constant<int, A_DEPTH - 1>{});
  if constexpr (A_DEPTH == 2) load_a(2, q0{});
  load_b(1, q1{});
  load_b(2, q0{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // one loop body (a second copy of the MFMA block makes the register allocator duplicate the accumulators)
  unsigned long long st_c = 0, st_s = 0, st_b = 0, st_t0 = 0, st_prev = 0;
#define BF_STAMP(bucket)                                               \
  if (a.stamps) {                                                      \
    unsigned long long now;                                            \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory"); \
    bucket += now - st_prev;                                           \
    st_prev = now;                                                     \
  }
  if (a.stamps) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0)::"memory");
    st_prev = st_t0;
  }
  for (int ks = 0; ks < a.ksteps; ++ks) {
    const bool more = ks + 1 < a.ksteps;
    compute(ks & 1);
    BF_STAMP(st_c)
    // the staging phase (VALU, LDS writes, load issue) runs at raised priority: next to the partner workgroup's
    // MFMA stream a wave at equal priority gets an issue slot only every few tens of cycles
    __builtin_amdgcn_s_setprio(3);
    if (more) store_step(ks + 1, (ks + 1) & 1, q1{});
    // advance the queues and refill their tails: A(ks + 1 + A_DEPTH), B(ks + 3)
    if constexpr (A_DEPTH == 2) {
#pragma unroll
      for (int i = 0; i < A_PIECES; ++i) areg[1][i] = areg[0][i];
      load_a(ks + 3, q0{});
    } else {
      load_a(ks + 2, q0{});
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) breg[1][j] = breg[0][j];
    load_b(ks + 3, q0{});
    // raw barrier: __syncthreads() would also wait for the loads in flight (vmcnt(0)) and put the whole memory
    // latency into every step; only this wave's LDS writes have to be complete here
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    BF_STAMP(st_s)
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    BF_STAMP(st_b)
  }
  unsigned long long st_loop_end = st_prev;

  // ---- epilogue: element (mi, ti, r) = row m0 + 128 wm + 32 mi + (r&3) + 8 (r>>2) + 4 lh, column t0 + 64 wt + 32 ti + lr.
  // All accesses go through buffer descriptors (rows >= M read 0 / drop their stores), so there are no per-element
  // branches; the 32 residual values of a row block are in flight together before the first add.
  if (a.ablate & 2) return;
  float fsum = 0.f, fsq = 0.f;
  const int slab = a.M * a.ldt * 4;
  const __amdgpu_buffer_rsrc_t yr =
      __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)n * a.M * a.ldt, 0, slab, 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(RES ? a.res : a.y) + (size_t)n * a.M * a.ldt, 0, slab, 0x00020000);
  const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.bias ? a.bias : a.x), 0, a.bias ? a.M * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t bnr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.bias_n ? a.bias_n + (size_t)n * a.M : a.x), 0, a.bias_n ? a.M * 4 : 0, 0x00020000);
  const int lane_off = (4 * lh * a.ldt + lr) * 4;
  const int tile_off = ((m0 + wm * 128) * a.ldt + t0 + wt * 64) * 4;
  float cm[2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti) cm[ti] = (t0 + wt * 64 + ti * 32 + lr < a.T) ? 1.f : 0.f;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    float bsum[16], rv[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rc = mi * 32 + (r & 3) + 8 * (r >> 2);
      const int moff = (m0 + wm * 128 + rc + 4 * lh) * 4;
      bsum[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, moff, 0, 0)) +
                __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bnr, moff, 0, 0));
      if constexpr (RES) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
          rv[ti][r] = __builtin_bit_cast(
              float, __builtin_amdgcn_raw_buffer_load_b32(rr, lane_off, tile_off + rc * a.ldt * 4 + ti * 128, 0));
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rc = mi * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        float v = acc[mi][ti][r] + bsum[r];
        if constexpr (STATS) {
          const float vm = v * cm[ti];
          fsum += vm;
          fsq += vm * vm;
        }
        if constexpr (RES) v += rv[ti][r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, lane_off,
                                              tile_off + rc * a.ldt * 4 + ti * 128, 0);
      }
    }
  }
  if (a.stamps && tid == 0) {
    unsigned long long now;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
    unsigned long long* d = a.stamps + ((size_t)(n * a.tiles_m + mt) * a.tiles_t + blockIdx.x) * 6;
    d[0] = st_t0;
    d[1] = st_c;
    d[2] = st_s;
    d[3] = st_b;
    d[4] = now - st_loop_end;  // epilogue
    d[5] = now;
  }
  if constexpr (STATS) {
    const double s = wave_sum((double)fsum), q = wave_sum((double)fsq);
    if (lane == 0) {
      const int parts = a.tiles_m * a.tiles_t * 4;
      const int part = (mt * a.tiles_t + blockIdx.x) * 4 + wave;
      double* dst = a.ostats + ((size_t)n * parts + part) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

// ---- wave-specialised persistent variant ---------------------------------------------------------------------------
// Measured on the kernel above (tools/stamp_bf16.py): the two co-resident workgroups fall into phase -- both stream
// MFMAs, then both stage -- so a K-step costs compute (3047 cycles, two waves sharing a matrix pipe) PLUS staging
// (1563) PLUS barrier (624) instead of their maximum.  Here ONE 512-thread workgroup per CU splits the roles for good:
// waves 0-3 ("consumers") only read fragments, issue MFMAs and drain finished tiles; waves 4-7 ("producers") only load,
// transform, split and write LDS, two K-steps ahead through a 3-slot ring, with their own deep register queues (they
// hold no accumulators).  The workgroup is persistent over a contiguous run of tiles, so the producers keep
// streaming into the next tile while the consumers drain the previous one.
constexpr int WS_MAXU = 8;  // utterances one workgroup's tile run may touch (prologue scalars kept in LDS)

struct WsTile {
  int n, mt, tt;
};

__device__ __forceinline__ WsTile ws_tile(int idx, int tiles_t, int tiles_m) {
  WsTile t;
  t.tt = idx % tiles_t;
  const int r = idx / tiles_t;
  t.mt = r % tiles_m;
  t.n = r / tiles_m;
  return t;
}

template <int PLANES, bool TR, bool STATS, bool RES>
__global__ __launch_bounds__(512, 1) void conv1x1_bf16_ws_kernel(BfArgs a) {
  using L = BfLds<PLANES>;
  constexpr int NSLOT = 3;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * L::SLOT + 2 * 512 * 4 + WS_MAXU * 2 * 4];
  float* tab = reinterpret_cast<float*>(smem + NSLOT * L::SLOT);  // gamma[512] | beta[512]
  float* scal = tab + 1024;                                       // [WS_MAXU][2]: mean, rstd of utterance n_lo + u
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool consumer = wave < 4;  // wave-uniform

  const int ntiles = a.tiles_t * a.tiles_m * a.N;
  const int G = gridDim.x;
  const int lo = (int)((long long)blockIdx.x * ntiles / G), hi = (int)((long long)(blockIdx.x + 1) * ntiles / G);
  if (lo >= hi) return;
  const int total = (hi - lo) * a.ksteps;
  const int n_lo = ws_tile(lo, a.tiles_t, a.tiles_m).n;

  const float slope = (TR && a.pro.prelu) ? a.pro.slope[0] : 1.f;
  if constexpr (TR) {
    const bool has_norm = a.pro.norm != PS_NORM_NONE;
    for (int k = tid; k < 512; k += 512) {
      tab[k] = (has_norm && k < a.K) ? a.pro.gamma[k] : (k < a.K ? 1.f : 0.f);
      tab[512 + k] = (has_norm && k < a.K) ? a.pro.beta[k] : 0.f;
    }
    const int n_hi = ws_tile(hi - 1, a.tiles_t, a.tiles_m).n;
    // every wave reduces the producer's partial statistics itself (no cross-wave step); wave u % 8 stores
    for (int u = 0; u <= n_hi - n_lo; ++u) {
      float mean = 0.f, rstd = 1.f;
      if (a.pro.norm == PS_NORM_GLOBAL) {
        double sa = 0.0, sq = 0.0;
        const double* src = a.pro.stats + (size_t)(n_lo + u) * a.pro.parts * 2;
        for (int i = lane; i < a.pro.parts; i += 64) {
          sa += src[2 * i];
          sq += src[2 * i + 1];
        }
        sa = wave_sum(sa);
        sq = wave_sum(sq);
        const double m = sa / a.pro.count;
        double var = sq / a.pro.count - m * m;
        var = var > 0.0 ? var : 0.0;
        mean = (float)m;
        rstd = (float)(1.0 / sqrt(var + (double)a.pro.eps));
      }
      if (wave == (u & 7) && lane == 0) {
        scal[2 * u] = mean;
        scal[2 * u + 1] = rstd;
      }
    }
  }
  __syncthreads();

  if (!consumer) {
    // =============================== producers =====================================================================
    const int ptid = tid - 256;
    constexpr int A_PIECES = 2 * PLANES;
    constexpr int AD = 2, BD = 3;  // K-steps of weights / activations in flight
    const int bt = ptid & 127, bh = ptid >> 7;
    const int xb_voff = (8 * bh * a.ldt + bt) * 4;
    u32x4v areg[AD][A_PIECES];
    float breg[BD][8];
    // load side: global step -> (tile, ks); steps past the end re-read the last one
    auto issue_a = [&](int g, auto q_c) {
      constexpr int q = decltype(q_c)::value;
      const int gc = g < total ? g : total - 1;
      const WsTile t = ws_tile(lo + gc / a.ksteps, a.tiles_t, a.tiles_m);
      const int ks = gc % a.ksteps;
      const u32x4v* src = reinterpret_cast<const u32x4v*>(a.wt) + ((size_t)t.mt * a.ksteps + ks) * (L::A_BYTES / 16);
#pragma unroll
      for (int i = 0; i < A_PIECES; ++i) areg[q][i] = src[ptid + 256 * i];
    };
    auto issue_b = [&](int g, auto q_c) {
      constexpr int q = decltype(q_c)::value;
      const int gc = g < total ? g : total - 1;
      const WsTile t = ws_tile(lo + gc / a.ksteps, a.tiles_t, a.tiles_m);
      const int ks = gc % a.ksteps;
      const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(a.x) + (size_t)t.n * a.K * a.ldt, 0, a.K * a.ldt * 4, 0x00020000);
      const int soff = (ks * XB_K * a.ldt + t.tt * XB_T) * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        breg[q][j] =
            __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, xb_voff, soff + j * a.ldt * 4, 0));
    };
    auto stage = [&](int g) {  // write step g (queue heads) into its ring slot
      unsigned char* sa = smem + (g % NSLOT) * L::SLOT;
#pragma unroll
      for (int i = 0; i < A_PIECES; ++i) reinterpret_cast<u32x4v*>(sa)[ptid + 256 * i] = areg[AD - 1][i];
      float v[8];
      const int ks = g % a.ksteps;
      float mean = 0.f, rstd = 1.f;
      if constexpr (TR) {
        const int u = ws_tile(lo + g / a.ksteps, a.tiles_t, a.tiles_m).n - n_lo;
        mean = scal[2 * u];
        rstd = scal[2 * u + 1];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float x = breg[BD - 1][j];
        if constexpr (TR) {
          const int k = ks * XB_K + 8 * bh + j;
          if (a.pro.pre_relu) x = fmaxf(x, 0.f);
          const float sc = tab[k] * rstd;
          x = x * sc + (tab[512 + k] - mean * sc);
          if (a.pro.prelu) x = prelu(x, slope);
          if (a.pro.post_tanh) x = tanhf(x);
          if (k >= a.K) x = 0.f;
        }
        v[j] = x;
      }
      unsigned char* sb = sa + L::A_BYTES;
#pragma unroll
      for (int p = 0; p < PLANES; ++p) {
        bf16x8 piece;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const bf16x2 h = __builtin_convertvector(f32x2v{v[j], v[j + 1]}, bf16x2);
          piece[j] = h[0];
          piece[j + 1] = h[1];
          if (p + 1 < PLANES) {
            const f32x2v back = __builtin_convertvector(h, f32x2v);
            v[j] -= back[0];
            v[j + 1] -= back[1];
          }
        }
        *reinterpret_cast<bf16x8*>(sb + ((p * XB_T + bt) * XB_K + 8 * bh) * 2) = piece;
      }
    };
    using i0 = std::integral_constant<int, 0>;
    using i1 = std::integral_constant<int, 1>;
    using i2 = std::integral_constant<int, 2>;
    // fill the queues: heads (index AD-1 / BD-1) = step 0
    issue_a(0, i1{});
    issue_a(1, i0{});
    issue_b(0, i2{});
    issue_b(1, i1{});
    issue_b(2, i0{});
    unsigned long long p_stage = 0, p_bar = 0, p_prev = 0;
    if (a.stamps) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(p_prev)::"memory");
    for (int i = 0; i < total + 2; ++i) {
      if (i < total) {
        stage(i);
        // advance the queues, refill the tails
#pragma unroll
        for (int q = 0; q < A_PIECES; ++q) areg[1][q] = areg[0][q];
        issue_a(i + 2, i0{});
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          breg[2][j] = breg[1][j];
          breg[1][j] = breg[0][j];
        }
        issue_b(i + 3, i0{});
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (a.stamps) {
        unsigned long long now;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
        p_stage += now - p_prev;
        p_prev = now;
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (a.stamps) {
        unsigned long long now;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
        p_bar += now - p_prev;
        p_prev = now;
      }
    }
    if (a.stamps && tid == 256) {
      unsigned long long* d = a.stamps + (size_t)blockIdx.x * 6;
      d[2] = p_stage;
      d[3] = p_bar;
    }
    return;
  }

  // ================================= consumers ======================================================================
  const int wm = wave >> 1, wt = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  f32x16 acc[4][2];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = zero16;

  auto compute = [&](int slot) {
    const unsigned char* sa = smem + slot * L::SLOT;
    const unsigned char* sb = sa + L::A_BYTES;
    bf16x8 bf[PLANES][2];
#pragma unroll
    for (int p = 0; p < PLANES; ++p)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
        bf[p][ti] = *reinterpret_cast<const bf16x8*>(sb + ((p * XB_T + wt * 64 + ti * 32 + lr) * XB_K + 8 * lh) * 2);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      bf16x8 af[PLANES];
#pragma unroll
      for (int p = 0; p < PLANES; ++p)
        af[p] = *reinterpret_cast<const bf16x8*>(sa + ((p * XB_M + wm * 128 + mi * 32 + lr) * XB_K + 8 * lh) * 2);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) {
        if constexpr (PLANES == 3) {
          acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1][ti], acc[mi][ti], 0, 0, 0);
          acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0][ti], acc[mi][ti], 0, 0, 0);
          acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2][ti], acc[mi][ti], 0, 0, 0);
          acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0][ti], acc[mi][ti], 0, 0, 0);
          acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1][ti], acc[mi][ti], 0, 0, 0);
        }
        acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0][ti], acc[mi][ti], 0, 0, 0);
      }
    }
  };

  auto drain = [&](const WsTile& t) {
    const int m0 = t.mt * XB_M, t0 = t.tt * XB_T, n = t.n;
    float fsum = 0.f, fsq = 0.f;
    const int slab = a.M * a.ldt * 4;
    const __amdgpu_buffer_rsrc_t yr =
        __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)n * a.M * a.ldt, 0, slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(RES ? a.res : a.y) + (size_t)n * a.M * a.ldt, 0, slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias ? a.bias : a.x), 0, a.bias ? a.M * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t bnr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias_n ? a.bias_n + (size_t)n * a.M : a.x), 0, a.bias_n ? a.M * 4 : 0, 0x00020000);
    const int lane_off = (4 * lh * a.ldt + lr) * 4;
    const int tile_off = ((m0 + wm * 128) * a.ldt + t0 + wt * 64) * 4;
    float cm[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) cm[ti] = (t0 + wt * 64 + ti * 32 + lr < a.T) ? 1.f : 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      float bsum[16], rv[2][16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rc = mi * 32 + (r & 3) + 8 * (r >> 2);
        const int moff = (m0 + wm * 128 + rc + 4 * lh) * 4;
        bsum[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, moff, 0, 0)) +
                  __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bnr, moff, 0, 0));
        if constexpr (RES) {
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
            rv[ti][r] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rr, lane_off, tile_off + rc * a.ldt * 4 + ti * 128, 0));
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rc = mi * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
          float v = acc[mi][ti][r] + bsum[r];
          if constexpr (STATS) {
            const float vm = v * cm[ti];
            fsum += vm;
            fsq += vm * vm;
          }
          if constexpr (RES) v += rv[ti][r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, lane_off,
                                                tile_off + rc * a.ldt * 4 + ti * 128, 0);
        }
      }
    }
    if constexpr (STATS) {
      const double s = wave_sum((double)fsum), q = wave_sum((double)fsq);
      if (lane == 0) {
        const int parts = a.tiles_m * a.tiles_t * 4;
        const int part = (t.mt * a.tiles_t + t.tt) * 4 + wave;
        double* dst = a.ostats + ((size_t)n * parts + part) * 2;
        dst[0] = s;
        dst[1] = q;
      }
    }
  };

  int cks = 0, ctile = lo;
  unsigned long long c_comp = 0, c_bar = 0, c_drain = 0, c_prev = 0;
  if (a.stamps) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_prev)::"memory");
#define WS_STAMP(bucket)                                                          \
  if (a.stamps) {                                                                 \
    unsigned long long now;                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory"); \
    bucket += now - c_prev;                                                       \
    c_prev = now;                                                                 \
  }
  for (int i = 0; i < total + 2; ++i) {
    if (i >= 2) {
      compute((i - 2) % NSLOT);
      WS_STAMP(c_comp)
      if (++cks == a.ksteps) {
        drain(ws_tile(ctile, a.tiles_t, a.tiles_m));
        WS_STAMP(c_drain)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = zero16;
        cks = 0;
        ++ctile;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    WS_STAMP(c_bar)
  }
  if (a.stamps && tid == 0) {
    unsigned long long* d = a.stamps + (size_t)blockIdx.x * 6;
    d[0] = c_comp;
    d[1] = c_bar;
    d[4] = c_drain;
    d[5] = (unsigned long long)total;
  }
}

static int bf16_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return cus;
}

template <int PLANES>
static void bf16_launch(const BfArgs& a, int N, bool tr, hipStream_t stream) {
  const bool stats = a.ostats != nullptr, res = a.res != nullptr;
  // The wave-specialised persistent kernel is NOT the default yet: its producers need 3100-4200 cycles per K-step
  // (integer divisions of the tile walk, un-pipelined table reads, queue moves) against 2000 for the consumers, and
  // its drains are un-overlapped, so it is 25-40 % slower than the simple kernel (tools/stamp_bf16.py).  ps_debug_flags
  // bit 27 selects it (tests run both); it also needs >= one tile per CU and a bounded utterance span per workgroup.
  const long long ntiles = (long long)a.tiles_t * a.tiles_m * N;
  const int G = (int)(ntiles < bf16_cus() ? ntiles : bf16_cus());
  const long long per_wg = (ntiles + G - 1) / G, per_utt = (long long)a.tiles_t * a.tiles_m;
  const bool ws = (g_debug_flags & (1 << 27)) && ntiles >= bf16_cus() && (per_wg + per_utt - 1) / per_utt + 1 <= WS_MAXU;
  if (ws) {
#define PS_WS(TRV, STV, RSV) \
  hipLaunchKernelGGL((conv1x1_bf16_ws_kernel<PLANES, TRV, STV, RSV>), dim3(G, 1), dim3(512), 0, stream, a)
    if (tr) {
      if (stats) PS_WS(true, true, false);
      else if (res) PS_WS(true, false, true);
      else PS_WS(true, false, false);
    } else {
      if (stats) PS_WS(false, true, false);
      else if (res) PS_WS(false, false, true);
      else PS_WS(false, false, false);
    }
#undef PS_WS
    return;
  }
  dim3 grid(a.tiles_t, a.tiles_m, N);
#define PS_BF(TRV, STV, RSV) \
  hipLaunchKernelGGL((conv1x1_bf16_kernel<PLANES, TRV, STV, RSV>), grid, dim3(256), 0, stream, a)
  if (tr) {
    if (stats) PS_BF(true, true, false);
    else if (res) PS_BF(true, false, true);
    else PS_BF(true, false, false);
  } else {
    if (stats) PS_BF(false, true, false);
    else if (res) PS_BF(false, false, true);
    else PS_BF(false, false, false);
  }
#undef PS_BF
}

}  // namespace ps

using namespace ps;

extern "C" size_t ps_conv1x1_bf16_weight_bytes(int M, int K, int planes) {
  if (M <= 0 || K <= 0 || (planes != 1 && planes != 3)) return 0;
  const size_t tiles_m = (M + XB_M - 1) / XB_M, ksteps = (K + XB_K - 1) / XB_K;
  return tiles_m * ksteps * planes * XB_M * XB_K * 2;
}

extern "C" int ps_conv1x1_bf16_f32(const float* x, const void* wt_planes, float* y, int N, int K, int M, int T, int ldt,
                                   int planes, const ps_prologue* pro, const float* bias, const float* bias_n,
                                   const float* res, double* ostats, void* stream) {
  if (!x || !wt_planes || !y || N <= 0 || K <= 0 || M <= 0 || T <= 0 || N > 65535) {
    set_error("ps_conv1x1_bf16_f32: null pointer or non-positive size (N=%d K=%d M=%d T=%d)", N, K, M, T);
    return PS_E_INVALID;
  }
  if (planes != 1 && planes != 3) {
    set_error("ps_conv1x1_bf16_f32: planes must be 1 (bf16 products) or 3 (fp32-accurate 3-way split), got %d", planes);
    return PS_E_INVALID;
  }
  if (ldt < T || ldt % kTileT != 0 || ((uintptr_t)wt_planes & 15)) {
    set_error("ps_conv1x1_bf16_f32: ldt=%d must be a multiple of %d >= T=%d, weights 16-byte aligned", ldt, kTileT, T);
    return PS_E_ALIGN;
  }
  if (res && ostats) {
    set_error("ps_conv1x1_bf16_f32: residual and output statistics cannot be combined");
    return PS_E_UNSUPPORTED;
  }
  bool tr = false;
  BfArgs a{};
  if (pro) {
    a.pro = *pro;
    tr = pro->norm != PS_NORM_NONE || pro->prelu || pro->pre_relu || pro->post_tanh;
    if (pro->norm == PS_NORM_GLOBAL && (!pro->stats || pro->parts <= 0 || pro->count <= 0 || !pro->gamma || !pro->beta)) {
      set_error("ps_conv1x1_bf16_f32: PS_NORM_GLOBAL prologue needs stats/parts/count/gamma/beta");
      return PS_E_INVALID;
    }
    if (pro->norm == PS_NORM_AFFINE && (!pro->gamma || !pro->beta)) {
      set_error("ps_conv1x1_bf16_f32: PS_NORM_AFFINE prologue needs gamma/beta");
      return PS_E_INVALID;
    }
    if (pro->prelu && !pro->slope) {
      set_error("ps_conv1x1_bf16_f32: prelu prologue needs slope");
      return PS_E_INVALID;
    }
  }
  if (tr && K > 512) {
    set_error("ps_conv1x1_bf16_f32: K=%d exceeds the 512 input channels the prologue keeps tables for", K);
    return PS_E_UNSUPPORTED;
  }
  a.x = x;
  a.wt = (const unsigned short*)wt_planes;
  a.y = y;
  a.bias = bias;
  a.bias_n = bias_n;
  a.res = res;
  a.ostats = ostats;
  a.K = K;
  a.M = M;
  a.T = T;
  a.ldt = ldt;
  a.ksteps = (K + XB_K - 1) / XB_K;
  a.tiles_t = (T + XB_T - 1) / XB_T;
  a.tiles_m = (M + XB_M - 1) / XB_M;
  a.N = N;
  a.ablate = (g_debug_flags >> 24) & 15;
  a.stamps = (unsigned long long*)g_debug_buffer;
  {
    LaunchTimer timer("conv1x1_bf16", (hipStream_t)stream);
    if (planes == 1)
      bf16_launch<1>(a, N, tr, (hipStream_t)stream);
    else
      bf16_launch<3>(a, N, tr, (hipStream_t)stream);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_conv1x1_bf16_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
