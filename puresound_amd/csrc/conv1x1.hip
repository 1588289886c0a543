// Fused 1x1 convolution for the Conv-TasNet TCN blocks: exact-fp32 MFMA GEMM with the producer's
// normalisation + PReLU applied to the activation operand, and bias / residual / partial statistics in the
// epilogue.
//
//   y[n][m][t] = sum_k W[m][k] * pro(x[n][k][t]) + bias[m] (+ bias_n[n][m]) (+ res[n][m][t])
//
// Reference arithmetic replaced: conv_tasnet.py:43-49,65,85-88 and lobe/cnn.py:75-79 of mcw519/PureSound.
//
// Mapping to CDNA4
//   * D = A*B with A = W (rows = output channel m) and B = activations (cols = frame t),
//     v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  Both operands sit k-major in LDS ([k][m] and
//     [k][t]) so the 32 lanes of a half-wave read 32 consecutive dwords (conflict-free ds_read_b32); the
//     weight is kept pre-transposed (and m-tile-major) in HBM for that reason.
//   * Persistent workgroups (2 per CU).  A workgroup walks a contiguous run of 256(m) x 128(t) tiles; its 4
//     waves form a 2x2 grid, each owning 128(m) x 64(t) = 4x2 MFMA tiles (128 accumulator VGPRs): one k-pair
//     costs 4 A + 2 B fragment reads for 8 MFMAs.  K is consumed in steps of 16.
//   * Weight rows go L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) without touching VGPRs, through a
//     3-slot ring that runs straight across tile boundaries with TWO K-steps in flight (counted s_waitcnt
//     vmcnt + raw s_barrier; a __syncthreads() would drain the ring).  Activation rows take the same route
//     when the launch has no prologue (in_conv).
//   * With a prologue (producer's norm + PReLU) the activation tile is staged through registers one K-step
//     ahead and transformed exactly once per element on its way into LDS: per-utterance tables
//     sc[k] = gamma[k]*rstd, sh[k] = beta[k] - mean*sc[k] live in LDS, so an element costs one FMA plus the
//     PReLU select.  VALU instructions are the scarce resource here -- with two waves per SIMD feeding the
//     matrix pipe, a VALU op waits ~30 cycles for an issue slot -- so the transform is not done per
//     fragment read (each element is read by two waves) and the epilogue is kept to 3-4 VALU per element.
//   * The epilogue rides inside the last K-step of each tile.  That step runs sub-tile-major: the 8 MFMAs of
//     one 32x32 sub-tile issue back to back, and while the next sub-tile accumulates, the finished one is
//     drained -- bias, residual, statistics, buffer_store -- two elements per MFMA.  The residual is
//     prefetched through a rolling window of four sub-tiles (64 VGPRs): the first four during the K-step
//     before, each later one into the registers its predecessor just freed.  A v1 of this kernel did the
//     epilogue after the K loop: every CU then stored at the same time, HBM-write bound (35-52k cycles per
//     tile with the matrix pipe idle; profiles/r01).
//   * The per-step path carries no VALU work outside the MFMA stream and no SGPR spills: a v_readlane
//     there queues behind the co-resident wave's MFMAs (measured: 2k cycles per step for ~30 of them).
#include <initializer_list>
#include <type_traits>
#include <utility>

#include "ps_common.h"

namespace ps {

constexpr int BM = 256;
constexpr int BT = 128;
constexpr int BK = 16;
constexpr int KMAX = 512;  // per-utterance scale/shift tables live in LDS

// LDS carve-up (floats).  One array: a second __shared__ object next to an LDS-DMA target can make hipcc
// drain vmcnt before every ds_read (cdna_hip_programming.md, "Projection GEMM at M = 256", item 4a).
constexpr int SLOT_A = BK * BM;            // 4096
constexpr int SLOT_B = BK * BT;            // 2048
constexpr int SLOT = SLOT_A + SLOT_B;      // 6144 floats = 24 KiB
constexpr int NSLOT = 3;
constexpr int OFF_SC = NSLOT * SLOT;       // [KMAX] gamma*rstd
constexpr int OFF_SH = OFF_SC + KMAX;      // [KMAX] beta - mean*sc
constexpr int OFF_BIAS = OFF_SH + KMAX;    // [2][BM]: bias | bias_n of the current tile
constexpr int OFF_RED = OFF_BIAS + 2 * BM; // 8 doubles
constexpr int LDS_FLOATS = OFF_RED + 16;   // 19984 floats = 78.06 KiB -> two workgroups per CU (160 KiB)
static_assert(2 * LDS_FLOATS * 4 <= 160 * 1024, "two workgroups must fit one CU's LDS");

struct Conv1x1Args {
  const float* x;
  const float* wt;
  float* y;
  const float* bias;
  const float* bias_n;
  const float* res;
  double* ostats;
  ps_prologue pro;
  int K, Kp, M, T, ldt;
  int tiles_t, tiles_m, ntiles, nsteps;
  unsigned long long* stamps;  // ps_debug_buffer(): per-workgroup s_memtime stamps (diagnostic runs only)
};

struct Tile {
  int n, m0, t0;
};

// next tile in (n, m-tile, t-tile) order, t fastest -- adds and compares only
__device__ __forceinline__ void next_tile(Tile& t, int m_end, int t_end) {
  t.t0 += BT;
  if (t.t0 >= t_end) {
    t.t0 = 0;
    t.m0 += BM;
    if (t.m0 >= m_end) {
      t.m0 = 0;
      ++t.n;
    }
  }
}

template <int V>
using ic = std::integral_constant<int, V>;

// call f(ic<First>{}), f(ic<First+1>{}), ... f(ic<First+Count-1>{})
template <int First, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (void)std::initializer_list<int>{(f(ic<First + I>{}), 0)...};
}
template <int First, int Count, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<First>(f, std::make_integer_sequence<int, Count>{});
}

#define PS_RSRC_FLAGS 0x00020000

// TR: the activation operand carries a norm and/or PReLU prologue (compile-time, so the in_conv
// instantiation has no fragment VALU at all).
// W:  size of the rolling residual window in 32x32 sub-tiles (0 = the launch has no residual).  16 VGPRs per
//     sub-tile; W = 2 keeps accumulators (128) + window (32) + working set inside 256 VGPRs.
// STATS: the launch writes partial statistics of y.
template <bool TR, int W, bool STATS>
__device__ __forceinline__ void conv1x1_body(const Conv1x1Args& a, float* lds) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 1;  // row half of the workgroup tile: rows wm*128 ..
  const int wt = wave_u & 1;   // column half: columns wt*64 ..
  const int lr = lane & 31;
  const int lk = lane >> 5;

  const int G = gridDim.x;
  const int lo = (int)((long long)blockIdx.x * a.ntiles / G);
  const int hi = (int)((long long)(blockIdx.x + 1) * a.ntiles / G);
  if (lo >= hi) return;

  unsigned long long t_begin = 0, t_loop = 0;
  if (a.stamps) t_begin = __builtin_amdgcn_s_memtime();

  const bool has_norm = a.pro.norm != PS_NORM_NONE;  // kernel-uniform
  constexpr bool has_res = W > 0;
  const float slope = (TR && a.pro.prelu) ? a.pro.slope[0] : 1.f;
  const bool pre_relu = TR && a.pro.pre_relu, post_tanh = TR && a.pro.post_tanh;  // kernel-uniform
  const int m_end = a.tiles_m * BM, t_end = a.tiles_t * BT;

  // ---- descriptors: every global access is SGPR-base + 32-bit offsets; out-of-range reads return 0.0f and
  //      out-of-range stores are dropped, which is how padding rows (k >= K, m >= M) are handled ----------
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.wt), 0, a.tiles_m * a.Kp * BM * 4, PS_RSRC_FLAGS);
  const int x_slab_bytes = a.K * a.ldt * 4;
  const int y_slab_bytes = a.M * a.ldt * 4;
  auto x_rsrc_of = [&](int n) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)n * a.K * a.ldt, 0, x_slab_bytes,
                                             PS_RSRC_FLAGS);
  };
  auto slab_rsrc = [&](const float* p, int n) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p) + (size_t)n * a.M * a.ldt, 0, y_slab_bytes,
                                             PS_RSRC_FLAGS);
  };

  // ---- load side: which (tile, k-step) the next DMA step fetches ---------------------------------------
  Tile ld;
  {
    const int tt = lo % a.tiles_t, r = lo / a.tiles_t;
    ld.n = r / a.tiles_m;
    ld.m0 = (r % a.tiles_m) * BM;
    ld.t0 = tt * BT;
  }
  Tile cur = ld;
  int ld_ks = 0, ld_slot = 0;
  int ld_left = (hi - lo) * a.nsteps;  // DMA steps still to issue
  const int w_voff = lane * 16;
  const int x_voff = ((lane >> 5) * a.ldt + (lane & 31) * 4) * 4;
  const int w_tile_bytes = a.Kp * BM * 4;

  // Per K-step each wave DMAs four 1-KiB weight rows (k = k0+4*wave .. +3; the packed weight is m-tile-major,
  // so consecutive k rows of one tile are 1 KiB apart in HBM exactly as in LDS and one M0 / one SGPR offset
  // serve all four through the immediate offset).  Without a prologue it also DMAs two 1-KiB blocks of two raw
  // activation rows [k][0..127] each.  LDS destinations are wave-uniform; lanes land 16 B apart.
  constexpr int DMA_PER_STEP = TR ? 4 : 6;
  auto dma_step = [&]() {
    if (ld_left <= 0) return;  // uniform
    --ld_left;
    const int k0 = ld_ks * BK + 4 * wave_u;
    float* as = lds + ld_slot * SLOT + 4 * wave_u * BM;
    const int w_soff = (ld.m0 / BM) * w_tile_bytes + k0 * (BM * 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, as, 16, w_voff, w_soff, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, as, 16, w_voff, w_soff, 1024, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, as, 16, w_voff, w_soff, 2048, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, as, 16, w_voff, w_soff, 3072, 0);
    if constexpr (!TR) {
      float* bs = lds + ld_slot * SLOT + SLOT_A + 4 * wave_u * BT;
      const __amdgpu_buffer_rsrc_t xr = x_rsrc_of(ld.n);
      const int x_soff = (k0 * a.ldt + ld.t0) * 4;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, bs, 16, x_voff, x_soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, bs + 2 * BT, 16, x_voff, x_soff + 2 * a.ldt * 4, 0, 0);
    }
    if (++ld_ks == a.nsteps) {
      ld_ks = 0;
      next_tile(ld, m_end, t_end);
    }
    ld_slot = ld_slot == NSLOT - 1 ? 0 : ld_slot + 1;
  };

  // Register-staged activation tile (TR only), one K-step ahead of the compute side: thread -> row tid>>4,
  // eight consecutive frames.  Channels k >= K read 0.0f through the descriptor and meet sc = sh = 0.
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  Tile bl = ld;       // (tile, k-step) of the next activation load: always the compute side's next step
  int bl_ks = 0, bl_left = ld_left;
  u32x4 rb0, rb1;
  float b_sc = 0.f, b_sh = 0.f;
  const int b_row = tid >> 4, b_col = (tid & 15) * 8;
  const int xb_voff = (b_row * a.ldt + b_col) * 4;
  bool b_deferred = false;  // the next step belongs to another utterance: stage it after the tables change
  auto b_load = [&]() {
    if (bl_left <= 0) return;  // uniform
    const int k0 = bl_ks * BK;
    const __amdgpu_buffer_rsrc_t xr = x_rsrc_of(bl.n);
    const int soff = (k0 * a.ldt + bl.t0) * 4;
    rb0 = __builtin_amdgcn_raw_buffer_load_b128(xr, xb_voff, soff, 0);
    rb1 = __builtin_amdgcn_raw_buffer_load_b128(xr, xb_voff + 16, soff, 0);
    b_sc = lds[OFF_SC + k0 + b_row];
    b_sh = lds[OFF_SH + k0 + b_row];
  };
  auto b_store = [&](int slot_next) {
    if (bl_left <= 0) return;  // uniform
    --bl_left;
    f32x4 v0 = __builtin_bit_cast(f32x4, rb0), v1 = __builtin_bit_cast(f32x4, rb1);
    if (pre_relu) {  // kernel-uniform
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v0[e] = relu_keep_nan(v0[e]);
        v1[e] = relu_keep_nan(v1[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float u0 = v0[e] * b_sc + b_sh, u1 = v1[e] * b_sc + b_sh;
      v0[e] = u0 >= 0.f ? u0 : slope * u0;
      v1[e] = u1 >= 0.f ? u1 : slope * u1;
    }
    if (post_tanh) {  // kernel-uniform; channels >= K have sc = sh = 0 -> tanh(0) = 0
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v0[e] = tanhf(v0[e]);
        v1[e] = tanhf(v1[e]);
      }
    }
    float* dst = lds + slot_next * SLOT + SLOT_A + b_row * BT + b_col;
    *reinterpret_cast<f32x4*>(dst) = v0;
    *reinterpret_cast<f32x4*>(dst + 4) = v1;
    if (++bl_ks == a.nsteps) {
      bl_ks = 0;
      next_tile(bl, m_end, t_end);
    }
  };

  // ---- compute side --------------------------------------------------------------------------------
  // element (mi, ti, r) of a wave's 128x64 tile sits at row 128*wm + 32*mi + (r&3) + 8*(r>>2) + 4*lk,
  // column 64*wt + 32*ti + lr: one per-lane VGPR offset that never changes plus wave-uniform SGPR offsets.
  f32x16 acc[4][2];  // [mi][ti]
  f32x16 side[W > 0 ? W : 1];  // rolling residual window: sub-tile s = mi*2+ti lives in side[s % W]
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int lane_off = (4 * lk * a.ldt + lr) * 4;
  auto tile_soff = [&](const Tile& t) { return ((t.m0 + wm * 128) * a.ldt + t.t0 + wt * 64) * 4; };

  // (`ldt4` = 4*ldt arrives through an opaque asm so that hipcc cannot hoist the row offsets rc*ldt4 out of
  //  the tile loop: kept live as loop invariants they spill SGPRs into VGPR lanes.)
  auto preload_elem = [&](auto sub_c, auto r_c, __amdgpu_buffer_rsrc_t rr, int soff, int ldt4) {
    constexpr int sub = decltype(sub_c)::value, r = decltype(r_c)::value;
    constexpr int mi = sub >> 1, ti = sub & 1;
    constexpr int rc = mi * 32 + (r & 3) + 8 * (r >> 2);
    side[sub % (W > 0 ? W : 1)][r] = __builtin_bit_cast(
        float, __builtin_amdgcn_raw_buffer_load_b32(rr, lane_off, soff + rc * ldt4 + ti * 128, 0));
  };

  // finish two vertically adjacent elements (registers r, r+1 of sub-tile `sub`: rows rc, rc+1): bias
  // (+ residual), statistics, store.  Written on float2 so that hipcc emits packed VALU (v_pk_add/mul/fma):
  // VALU issue slots are the scarce resource next to a co-resident wave's MFMAs.  Rows >= M need no mask:
  // their weights and (out-of-descriptor) bias rows are zero, so they add exactly 0 to the statistics; pad
  // columns are masked by a per-lane select per column block.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto drain_pair = [&](auto sub_c, auto r_c, f32x2 bias2, const bool (&cm2)[2], __amdgpu_buffer_rsrc_t yr, int soff,
                        int ldt4, f32x2& fsum2, f32x2& fsq2) {
    constexpr int sub = decltype(sub_c)::value, r = decltype(r_c)::value;  // r even
    constexpr int mi = sub >> 1, ti = sub & 1;
    constexpr int rc = mi * 32 + (r & 3) + 8 * (r >> 2);
    float v0 = acc[mi][ti][r] + bias2[0];
    float v1 = acc[mi][ti][r + 1] + bias2[1];
    if constexpr (STATS) {
      // (a select, not a 0/1 factor: a pad column may hold anything -- the depthwise kernel leaves the pad frames of
      //  its output untouched -- and 0 * NaN would poison the statistics of the whole utterance)
      const f32x2 vm = cm2[ti] ? f32x2{v0, v1} : f32x2{0.f, 0.f};
      fsum2 += vm;
      fsq2 += vm * vm;
    }
    if constexpr (W > 0) {
      v0 += side[sub % W][r];
      v1 += side[sub % W][r + 1];
    }
    // pad columns (>= T) are written too (never read as data); rows >= M lie outside the descriptor
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), yr, lane_off, soff + rc * ldt4 + ti * 128,
                                          0);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), yr, lane_off,
                                          soff + (rc + 1) * ldt4 + ti * 128, 0);
  };
  // summed bias rows (bias + bias_n) of registers r, r+1 of sub-tile `sub`
  auto bias_pair = [&](auto sub_c, auto r_c) {
    constexpr int sub = decltype(sub_c)::value, r = decltype(r_c)::value;
    constexpr int rc = (sub >> 1) * 32 + (r & 3) + 8 * (r >> 2);
    const float* bp = lds + OFF_BIAS + wm * 128 + rc + 4 * lk;
    return *reinterpret_cast<const f32x2*>(bp) + *reinterpret_cast<const f32x2*>(bp + BM);
  };

  auto write_stats = [&](const Tile& t, float fsum, float fsq) {
    const double s = wave_sum((double)fsum), q = wave_sum((double)fsq);
    if (lane == 0) {
      const int parts = a.tiles_m * a.tiles_t * 4;
      const int part = ((t.m0 / BM) * a.tiles_t + t.t0 / BT) * 4 + wave_u;
      double* dst = a.ostats + ((size_t)t.n * parts + part) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  };

  // B fragment of k-row `k`, column block ti (already transformed when the launch has a prologue)
  auto b_frag = [&](const float* bs, int k, int ti) { return bs[k * BT + wt * 64 + ti * 32 + lr]; };

  // K-step, k-major: 8 k-pairs x (4 A + 2 B fragment reads, 8 MFMAs).  With `pre` (wave-uniform; the K-step
  // before the last, when there is a residual) the residual of sub-tiles 0..W-1 is prefetched, one element
  // after each of the first 16*W MFMAs.  Rows >= M need no special casing (zero weights, out-of-descriptor stores/loads,
  // masked statistics).  One code copy serves every non-final K-step, so the accumulators keep one register set.
  auto kstep = [&](bool pre, int slot, const Tile& t) {
    const float* as = lds + slot * SLOT;
    const float* bs = as + SLOT_A;
    const __amdgpu_buffer_rsrc_t rr = slab_rsrc(has_res ? a.res : a.y, t.n);
    const int rsoff = tile_soff(t);
    int ldt4 = a.ldt * 4;
    asm volatile("" : "+s"(ldt4));
    static_for<0, BK / 2>([&](auto kk_c) {
      constexpr int kk = decltype(kk_c)::value;
      const int k = 2 * kk + lk;
      float av[4], bv[2];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) av[mi] = as[k * BM + wm * 128 + mi * 32 + lr];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) bv[ti] = b_frag(bs, k, ti);
      static_for<0, 8>([&](auto j_c) {
        constexpr int j = decltype(j_c)::value;
        acc[j >> 1][j & 1] =
            __builtin_amdgcn_mfma_f32_32x32x2f32(av[j >> 1], bv[j & 1], acc[j >> 1][j & 1], 0, 0, 0);
        if constexpr (W > 0) {
          constexpr int e = kk * 8 + j;  // MFMA number 0..63 of the step: element e of the first W sub-tiles
          if constexpr (e < W * 16) {
            if (pre) preload_elem(ic<(e >> 4)>{}, ic<(e & 15)>{}, rr, rsoff, ldt4);
          }
        }
      });
    });
  };

  // Last K-step of a tile, sub-tile-major (see the header): sub-tile s accumulates its 8 MFMAs while
  // sub-tile s-1 is drained one register pair per MFMA; each drained pair's window registers are refilled at
  // once with the residual of sub-tile s-1+W.  Software-pipelined one MFMA slot deep: the A/B fragments and
  // the bias pair of slot i+1 are read from LDS before the MFMA of slot i issues, so neither the MFMA nor
  // the drain waits on an LDS round trip.  Only the eighth sub-tile's 16 stores go without MFMA cover.
  auto kstep_last = [&](int slot, const Tile& t) {
    const float* as = lds + slot * SLOT;
    const float* bs = as + SLOT_A;
    const __amdgpu_buffer_rsrc_t yr = slab_rsrc(a.y, t.n);
    const __amdgpu_buffer_rsrc_t rr = slab_rsrc(has_res ? a.res : a.y, t.n);
    const int soff = tile_soff(t);
    int ldt4 = a.ldt * 4;
    asm volatile("" : "+s"(ldt4));
    f32x2 fsum2 = {0.f, 0.f}, fsq2 = {0.f, 0.f};
    bool cm2[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) cm2[ti] = t.t0 + wt * 64 + ti * 32 + lr < a.T;
    // slot i = sub*8 + kk: MFMA (mi, ti) = (sub>>1, sub&1), k-pair kk; drains pair kk of sub-tile sub-1
    float av[2], bv[2];
    f32x2 bp[2];
    auto frag = [&](auto i_c, auto buf_c) {
      constexpr int i = decltype(i_c)::value, b = decltype(buf_c)::value;
      constexpr int sub = i >> 3, kk = i & 7;
      const int k = 2 * kk + lk;
      av[b] = as[k * BM + wm * 128 + (sub >> 1) * 32 + lr];
      bv[b] = b_frag(bs, k, sub & 1);
      if constexpr (sub > 0) bp[b] = bias_pair(ic<sub - 1>{}, ic<2 * kk>{});
    };
    frag(ic<0>{}, ic<0>{});
    static_for<0, 64>([&](auto i_c) {
      constexpr int i = decltype(i_c)::value;
      constexpr int sub = i >> 3, kk = i & 7, cb = i & 1;
      if constexpr (i + 1 < 64) frag(ic<i + 1>{}, ic<cb ^ 1>{});
      acc[sub >> 1][sub & 1] =
          __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb], bv[cb], acc[sub >> 1][sub & 1], 0, 0, 0);
      if constexpr (sub > 0) {
        drain_pair(ic<sub - 1>{}, ic<2 * kk>{}, bp[cb], cm2, yr, soff, ldt4, fsum2, fsq2);
        if constexpr (W > 0 && sub - 1 + W < 8) {
          preload_elem(ic<sub - 1 + W>{}, ic<2 * kk>{}, rr, soff, ldt4);
          preload_elem(ic<sub - 1 + W>{}, ic<2 * kk + 1>{}, rr, soff, ldt4);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // pin the slot order: prefetch, MFMA, drain
    });
    static_for<0, 8>([&](auto q_c) {
      constexpr int r = 2 * decltype(q_c)::value;
      drain_pair(ic<7>{}, ic<r>{}, bias_pair(ic<7>{}, ic<r>{}), cm2, yr, soff, ldt4, fsum2, fsq2);
    });
    if constexpr (STATS) write_stats(t, fsum2[0] + fsum2[1], fsq2[0] + fsq2[1]);
  };

  // ---- pipeline fill: two K-steps in flight --------------------------------------------------------------
  dma_step();
  dma_step();
  const int total = (hi - lo) * a.nsteps;
  bool b_primed = false;  // TR: activation rows of step 0 staged (needs the tables of the first utterance)
  int g = 0;         // global K-step index of this workgroup
  int slot = 0;      // ring slot of step g
  int after_sb = 0;  // previous step issued a long side-band burst (residual loads and/or stores)
  int cur_n = -1;
  [[maybe_unused]] bool is_last_step = false;  // (phase-stamp builds only)

#ifdef PS_PHASE_STAMPS
  unsigned long long ph_wait = 0, ph_bar = 0, ph_dma = 0, ph_body = 0, ph_last = 0;
#define PS_STAMP(var) \
  unsigned long long var; \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#else
#define PS_STAMP(var)
#endif
  // One pipeline step: wait until this wave's LDS writes (staged activations) and its DMA of step g have landed (leaving the younger step's DMA --
  // and at most the side-band burst of the previous step -- in flight), meet the other waves (everyone's
  // step-g data is in LDS; everyone is done reading step g-1, whose slot is refilled next), issue the DMA
  // of step g+2, then run the body on slot g.
  // `new_tile`: first step of a tile -- wave 0 also DMAs the tile's bias rows (two 1-KiB pieces).  They go
  // out right after the barrier (every wave has finished the previous tile's drain, the only reader) and
  // BEFORE this step's ring DMA, so the next step's counted wait covers them.
  auto step = [&](bool new_tile, auto&& body) {
    PS_STAMP(s0);
    if (g + 1 < total) {
      if (after_sb)
        asm volatile("s_waitcnt vmcnt(63) lgkmcnt(0)" ::: "memory");  // >= 64 younger side-band ops may stay in flight
      else if (DMA_PER_STEP == 4)
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    PS_STAMP(s1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PS_STAMP(s2);
    if (new_tile && wave_u == 0) {
      const __amdgpu_buffer_rsrc_t bias_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(a.bias ? a.bias : a.wt), 0, a.bias ? a.M * 4 : 0, PS_RSRC_FLAGS);
      const __amdgpu_buffer_rsrc_t bn_rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(a.bias_n ? a.bias_n + (size_t)cur.n * a.M : a.wt), 0, a.bias_n ? a.M * 4 : 0,
          PS_RSRC_FLAGS);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bias_rsrc, lds + OFF_BIAS, 16, w_voff, cur.m0 * 4, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(bn_rsrc, lds + OFF_BIAS + BM, 16, w_voff, cur.m0 * 4, 0, 0);
    }
    if constexpr (TR) {
      // stage the next step's activations -- unless they belong to another utterance, whose tables are
      // not in LDS yet (done synchronously at that tile's start; once per utterance change)
      b_deferred = bl_left > 0 && bl.n != cur_n;
      if (!b_deferred) b_load();  // older than this step's ring DMA: its use below waits vmcnt(4), not 0
    }
    dma_step();
    PS_STAMP(s3);
    body(slot);
    if constexpr (TR) {
      if (!b_deferred) b_store(slot == NSLOT - 1 ? 0 : slot + 1);
    }
    PS_STAMP(s4);
#ifdef PS_PHASE_STAMPS
    ph_wait += s1 - s0;
    ph_bar += s2 - s1;
    ph_dma += s3 - s2;
    if (is_last_step)
      ph_last += s4 - s3;
    else
      ph_body += s4 - s3;
#endif
    slot = slot == NSLOT - 1 ? 0 : slot + 1;
    ++g;
  };

  for (int tile = lo; tile < hi; ++tile) {
    if (TR && cur.n != cur_n) {
      // per-utterance tables of the prologue norm.  The block reduce and the barriers drain the DMA ring;
      // this happens once per utterance change (a workgroup's run covers one or two utterances).
      const NormScalars ns = load_norm_scalars(a.pro, cur.n, reinterpret_cast<double*>(lds + OFF_RED));
      __syncthreads();  // nobody still reads the old tables
      for (int k = tid; k < a.nsteps * BK; k += 256) {  // nsteps*BK can exceed Kp when K <= 16
        float sc = 0.f, sh = 0.f;  // channels >= K: activation 0 (and the weight rows are zero padding)
        if (k < a.K) {
          sc = has_norm ? a.pro.gamma[k] * ns.rstd : 1.f;
          sh = has_norm ? a.pro.beta[k] - ns.mean * sc : 0.f;
        }
        lds[OFF_SC + k] = sc;
        lds[OFF_SH + k] = sh;
      }
      __syncthreads();
      cur_n = cur.n;
    }
    if (TR && (!b_primed || b_deferred)) {
      b_load();
      b_store(slot);  // `slot` is the ring slot of the step about to run; the barrier in step() publishes it
      b_primed = true;
      b_deferred = false;
    }
    if (tile == lo && a.stamps) t_loop = __builtin_amdgcn_s_memtime();
    // clear the accumulators (whole-tuple assignment: 128 v_mov per tile, < 0.5 % of a tile's MFMA time)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = zero16;
    for (int ks = 0; ks < a.nsteps - 1; ++ks) {
      const bool pre = has_res && ks == a.nsteps - 2;
      step(ks == 0, [&](int s) { kstep(pre, s, cur); });
      after_sb = pre;
    }
#ifdef PS_PHASE_STAMPS
    { const int keep = after_sb; (void)keep; }
#endif
    is_last_step = true;
    step(false, [&](int s) { kstep_last(s, cur); });
    is_last_step = false;
    after_sb = 1;
    next_tile(cur, m_end, t_end);
  }

  if (a.stamps && tid == 0) {
    const unsigned long long t_end_s = __builtin_amdgcn_s_memtime();
    unsigned long long* d = a.stamps + (size_t)blockIdx.x * 6;
    d[3] = (unsigned long long)(hi - lo);
#ifdef PS_PHASE_STAMPS
    d[0] = t_end_s - t_begin;
    d[1] = ph_wait;
    d[2] = ph_body;
    d[4] = ph_bar + ph_dma;
    d[5] = ph_last;
    (void)t_loop;
#else
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    d[0] = t_begin;
    d[1] = t_loop;
    d[2] = t_end_s;
    d[4] = hwid;
    d[5] = xcc;
#endif
  }
}

#define PS_CONV1X1_KERNEL(name, TR, W, STATS)                               \
  __global__ __launch_bounds__(256, 2) void name(Conv1x1Args a) {             \
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];            \
    conv1x1_body<TR, W, STATS>(a, lds);                                       \
  }
PS_CONV1X1_KERNEL(conv1x1_raw_stats, false, 0, true)   // in_conv: raw input, statistics out
PS_CONV1X1_KERNEL(conv1x1_tr_stats, true, 0, true)     // pointwise: norm+PReLU prologue, statistics out
PS_CONV1X1_KERNEL(conv1x1_tr_res, true, 2, false)      // out_conv: norm+PReLU prologue, residual
PS_CONV1X1_KERNEL(conv1x1_raw, false, 0, false)        // generic variants
PS_CONV1X1_KERNEL(conv1x1_tr, true, 0, false)
PS_CONV1X1_KERNEL(conv1x1_raw_res, false, 2, false)

static int persistent_grid() {
  return 2 * device_cus();  // two resident workgroups per CU (<= 256 VGPRs, 72 KiB LDS each)
}

}  // namespace ps

extern "C" int ps_conv1x1_stats_parts(int M, int T) {
  if (M <= 0 || T <= 0) return 0;
  return ((M + ps::BM - 1) / ps::BM) * ((T + ps::BT - 1) / ps::BT) * 4;
}

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
  if (((uintptr_t)bias & 3) || ((uintptr_t)bias_n & 3)) {
    set_error("ps_conv1x1_f32: bias pointers must be 4-byte aligned");
    return PS_E_ALIGN;
  }
  if (K > KMAX && pro && (pro->norm != PS_NORM_NONE || pro->prelu || pro->pre_relu || pro->post_tanh)) {
    set_error("ps_conv1x1_f32: K=%d exceeds the %d input channels the prologue keeps scale/shift tables for", K,
              KMAX);
    return PS_E_UNSUPPORTED;
  }
  if ((long long)K * ldt * 4 > 0x7fffffffLL || (long long)M * ldt * 4 > 0x7fffffffLL) {
    set_error("ps_conv1x1_f32: one utterance slab exceeds 2 GiB (32-bit buffer offsets)");
    return PS_E_UNSUPPORTED;
  }
  if (res && ostats) {
    set_error("ps_conv1x1_f32: residual and output statistics cannot be combined (no Conv-TasNet stage needs both)");
    return PS_E_UNSUPPORTED;
  }
  if (pro) {
    if (pro->norm == PS_NORM_GLOBAL && (!pro->stats || pro->parts <= 0 || pro->count <= 0 || !pro->gamma ||
                                        !pro->beta)) {
      set_error("ps_conv1x1_f32: PS_NORM_GLOBAL prologue needs stats/parts/count/gamma/beta");
      return PS_E_INVALID;
    }
    if (pro->norm == PS_NORM_AFFINE && (!pro->gamma || !pro->beta)) {
      set_error("ps_conv1x1_f32: PS_NORM_AFFINE prologue needs gamma/beta");
      return PS_E_INVALID;
    }
    if (pro->prelu && !pro->slope) {
      set_error("ps_conv1x1_f32: prelu prologue needs slope");
      return PS_E_INVALID;
    }
  }
  // short rows (streaming step, state rows): channel-split kernel instead of 256 x 128 tiles (debug bit 4 = off)
  if (T <= 64 && !ostats && !(pro && pro->norm == PS_NORM_GLOBAL) && !(g_debug_flags & 16)) {
    conv1x1_small_launch(x, wt, y, N, K, M, T, ldt, pro, bias, bias_n, res, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      set_error("ps_conv1x1_f32: launch failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    return 0;
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
  a.Kp = (K + BK - 1) / BK * BK;
  a.tiles_t = (T + BT - 1) / BT;
  a.tiles_m = (M + BM - 1) / BM;
  const long long ntiles = (long long)N * a.tiles_m * a.tiles_t;
  if (ntiles > 0x7fffffffLL) {
    set_error("ps_conv1x1_f32: too many tiles");
    return PS_E_INVALID;
  }
  a.ntiles = (int)ntiles;
  a.nsteps = (K + BK - 1) / BK;
  if (a.nsteps < 2) a.nsteps = 2;  // the residual prefetch needs a K-step before the draining one
  a.stamps = (unsigned long long*)g_debug_buffer;
  int grid = a.ntiles < persistent_grid() ? a.ntiles : persistent_grid();
  // test hook (ps_debug_flags bits 8..23): cap the persistent grid so that a workgroup's run spans many
  // tiles / utterances even on small problems
  if ((g_debug_flags >> 8) & 0xfff) grid = grid < ((g_debug_flags >> 8) & 0xfff) ? grid : ((g_debug_flags >> 8) & 0xfff);
  const bool tr = a.pro.norm != PS_NORM_NONE || a.pro.prelu || a.pro.pre_relu || a.pro.post_tanh;
  {
    LaunchTimer timer("conv1x1", (hipStream_t)stream);
    const dim3 gr(grid), bl(256);
    hipStream_t st = (hipStream_t)stream;
    if (tr && res)
      hipLaunchKernelGGL(conv1x1_tr_res, gr, bl, 0, st, a);
    else if (res)
      hipLaunchKernelGGL(conv1x1_raw_res, gr, bl, 0, st, a);
    else if (tr && ostats)
      hipLaunchKernelGGL(conv1x1_tr_stats, gr, bl, 0, st, a);
    else if (tr)
      hipLaunchKernelGGL(conv1x1_tr, gr, bl, 0, st, a);
    else if (ostats)
      hipLaunchKernelGGL(conv1x1_raw_stats, gr, bl, 0, st, a);
    else
      hipLaunchKernelGGL(conv1x1_raw, gr, bl, 0, st, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_conv1x1_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
