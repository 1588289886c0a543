// Fused 1x1 convolution for the Conv-TasNet TCN blocks: exact-fp32 MFMA GEMM with the producer's
// normalisation + PReLU applied while staging the activation tile, and bias / residual / partial
// statistics in the epilogue.
//
//   y[n][m][t] = sum_k W[m][k] * pro(x[n][k][t]) + bias[m] (+ bias_n[n][m]) (+ res[n][m][t])
//
// Reference arithmetic replaced: conv_tasnet.py:43-49,65,85-88 and lobe/cnn.py:75-79 of mcw519/PureSound.
//
// Mapping to CDNA4
//   * D = A*B with A = W (rows = output channel m) and B = activations (cols = frame t),
//     v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).  Both operands are staged k-major in LDS
//     ([k][m] and [k][t]) so the 32 lanes of a half-wave read 32 consecutive dwords (conflict-free
//     ds_read_b32); the weight is kept pre-transposed in HBM for that reason.
//   * Persistent workgroups (2 per CU).  A workgroup walks a contiguous run of 256(m) x 64(t) tiles; each of
//     its 4 waves owns 64(m) x 64(t) = 2x2 MFMA tiles.  K is consumed in steps of 16 through a 2-deep LDS
//     ring that runs straight across tile boundaries: the global loads of step g+1 are issued before the
//     MFMAs of step g and written to LDS after them (one barrier per step, no per-tile pipeline refill).
//   * The epilogue rides inside the last K-step of each tile.  That step runs sub-tile-major: the 8 MFMAs of
//     one 32x32 sub-tile issue back to back, and while the next sub-tile accumulates, the finished one is
//     drained -- bias, residual, statistics, buffer_store -- two elements per MFMA.  The residual tile is
//     prefetched into a side register set during the K-step before (two loads per MFMA).  A v1 of this
//     kernel did the epilogue after the K loop: every CU then stored at the same time, HBM-write bound
//     (35-52k cycles per tile with the matrix pipe idle; profiles/r01).
#include <initializer_list>
#include <type_traits>
#include <utility>

#include "ps_common.h"

namespace ps {

constexpr int BM = 256;
constexpr int BT = 64;
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
  int K, Kp, M, T, ldt, Mp;
  int tiles_t, tiles_m, ntiles, nsteps;
  unsigned long long* stamps;  // ps_debug_buffer(): per-workgroup s_memtime stamps (diagnostic runs only)
};

struct Tile {
  int n, m0, t0;
};

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

__global__ __launch_bounds__(256, 2) void conv1x1_kernel(Conv1x1Args a) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BT];
  __shared__ float biasS[2][BM];
  __shared__ double red[8];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int lr = lane & 31;
  const int lk = lane >> 5;

  const int G = gridDim.x;
  const int lo = (int)((long long)blockIdx.x * a.ntiles / G);
  const int hi = (int)((long long)(blockIdx.x + 1) * a.ntiles / G);
  if (lo >= hi) return;

  unsigned long long t_begin = 0, t_loop = 0;
  if (a.stamps) t_begin = __builtin_amdgcn_s_memtime();

  auto decode = [&](int id) {
    Tile t;
    const int tt = id % a.tiles_t;
    const int r = id / a.tiles_t;
    t.n = r / a.tiles_m;
    t.m0 = (r % a.tiles_m) * BM;
    t.t0 = tt * BT;
    return t;
  };

  const bool has_norm = a.pro.norm != PS_NORM_NONE;  // kernel-uniform
  const bool transform = has_norm || a.pro.prelu;
  const float slope = a.pro.prelu ? a.pro.slope[0] : 1.f;

  // ---- load side: which (tile, k-step) the next staging step fetches ---------------------------------
  int ld_tile = lo, ld_ks = 0;
  Tile ld = decode(lo);
  NormScalars ld_ns = load_norm_scalars(a.pro, ld.n, red);

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int b_row = tid >> 4;        // 0..15
  const int b_col = (tid & 15) * 4;  // 0..60

  // All global traffic goes through buffer descriptors (SGPR base + 32-bit offsets): no 64-bit per-lane
  // addresses, and out-of-range rows read as 0.0f -- a k >= K row gets x = 0, gamma = beta = 0, so the
  // staged activation is PReLU(0) = 0 without a predicate.
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t w_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wt), 0, a.Kp * a.Mp * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t g_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_norm ? a.pro.gamma : a.wt), 0, has_norm ? a.K * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(has_norm ? a.pro.beta : a.wt), 0, has_norm ? a.K * 4 : 0, 0x00020000);
  const int x_slab_bytes = a.K * a.ldt * 4;
  const int x_voff = (b_row * a.ldt + b_col) * 4;
  const int w_voff = lane * 16;

  u32x4 rb;
  float gm = 0.f, bt = 0.f;

  // Issue the global loads of one K-step.  The weight tile needs no transform, so it goes L2 -> LDS
  // directly (buffer_load_dwordx4 ... lds: each wave fills whole 1-KiB rows [k][0..255], lane-linear) into
  // the ring slot the previous step has finished reading; the activation row goes to registers because the
  // producer's norm + PReLU is applied on the way.  Nothing here consumes a loaded value, so no s_waitcnt
  // lands between these loads and the MFMAs that follow.
  auto load_step = [&](int buf) {
    const int k0 = ld_ks * BK;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x) + (size_t)ld.n * a.K * a.ldt, 0, x_slab_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // rows past Kp (only when nsteps was rounded up to 2) fall outside the descriptor: LDS gets zeros
      const int kr = k0 + wave_u + 4 * j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, &As[buf][wave_u + 4 * j][0], 16, w_voff,
                                               (kr * a.Mp + ld.m0) * 4, 0, 0);
    }
    rb = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, x_voff, (k0 * a.ldt + ld.t0) * 4, 0);
    gm = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_rsrc, b_row * 4, k0 * 4, 0));
    bt = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b_rsrc, b_row * 4, k0 * 4, 0));
  };

  auto store_step = [&](int buf) {
    f32x4 v = __builtin_bit_cast(f32x4, rb);
    if (transform) {
      const float sc = has_norm ? gm * ld_ns.rstd : 1.f;
      const float sh = has_norm ? bt : 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = prelu((v[e] - ld_ns.mean) * sc + sh, slope);
    }
    *reinterpret_cast<f32x4*>(&Bs[buf][b_row][b_col]) = v;
  };

  // move the load side to the next (tile, k-step); workgroup-uniform, may re-derive the norm scalars
  auto advance_load = [&]() {
    if (++ld_ks == a.nsteps) {
      ld_ks = 0;
      if (++ld_tile < hi) {
        const Tile nt = decode(ld_tile);
        if (nt.n != ld.n) ld_ns = load_norm_scalars(a.pro, nt.n, red);
        ld = nt;
      }
    }
  };

  // ---- compute side --------------------------------------------------------------------------------
  f32x16 acc[2][2];   // [mi][ti]: the wave's 64x64 accumulator
  f32x16 side[2][2];  // residual of the current tile, prefetched one K-step ahead (only with a.res)

  // element (mi, ti, r) of a wave's 64x64 tile sits at row 64*wave + 32*mi + (r&3) + 8*(r>>2) + 4*lk,
  // column 32*ti + lr.  Side-band stores/loads use buffer addressing: one 4-SGPR descriptor per utterance
  // slab [M][ldt], a per-lane VGPR offset that never changes and a wave-uniform SGPR offset per element --
  // no 64-bit per-element addresses in VGPRs, and rows >= M fall outside the descriptor (hardware drops
  // the store / returns 0.0f).
  const int lane_off = (4 * lk * a.ldt + lr) * 4;  // byte offset of this lane inside any 32x32 sub-tile
  const int slab_bytes = a.M * a.ldt * 4;
  auto slab_rsrc = [&](const float* p, const Tile& t) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p) + (size_t)t.n * a.M * a.ldt, 0, slab_bytes,
                                             0x00020000);
  };
  auto tile_soff = [&](const Tile& t) { return ((t.m0 + wave_u * 64) * a.ldt + t.t0) * 4; };

  // prefetch one residual element of the current tile
  auto preload_elem = [&](auto mi_c, auto ti_c, auto r_c, __amdgpu_buffer_rsrc_t rr, int soff) {
    constexpr int mi = decltype(mi_c)::value, ti = decltype(ti_c)::value, r = decltype(r_c)::value;
    constexpr int rc = mi * 32 + (r & 3) + 8 * (r >> 2);
    side[mi][ti][r] = __builtin_bit_cast(
        float, __builtin_amdgcn_raw_buffer_load_b32(rr, lane_off, soff + (rc * a.ldt + ti * 32) * 4, 0));
  };

  // finish one element: bias (+ residual), statistics, store
  auto drain_elem = [&](auto res_c, auto mi_c, auto ti_c, auto r_c, const Tile& t, int bsel, __amdgpu_buffer_rsrc_t yr,
                        int soff, float& fsum, float& fsq) {
    constexpr bool RES = decltype(res_c)::value != 0;
    constexpr int mi = decltype(mi_c)::value, ti = decltype(ti_c)::value, r = decltype(r_c)::value;
    constexpr int rc = mi * 32 + (r & 3) + 8 * (r >> 2);  // row inside the wave's block, before the 4*lk lane term
    const int rl = wave_u * 64 + rc + 4 * lk;
    float v = acc[mi][ti][r] + biasS[bsel][rl];
    const float vm = (t.m0 + rl < a.M && t.t0 + ti * 32 + lr < a.T) ? v : 0.f;  // statistics never see a residual
    fsum += vm;
    fsq += vm * vm;
    if constexpr (RES) v += side[mi][ti][r];
    // pad columns (>= T) are written too (never read as data); rows >= M lie outside the descriptor
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, lane_off,
                                          soff + (rc * a.ldt + ti * 32) * 4, 0);
  };

  auto write_stats = [&](const Tile& t, float fsum, float fsq) {
    const double s = wave_sum((double)fsum), q = wave_sum((double)fsq);
    if (lane == 0) {
      const int parts = a.tiles_m * a.tiles_t * 4;
      const int part = ((t.m0 / BM) * a.tiles_t + t.t0 / BT) * 4 + wave;
      double* dst = a.ostats + ((size_t)t.n * parts + part) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  };

  // K-step, k-major: 8 k-pairs x (2 A + 2 B fragment reads, 4 MFMAs).  PRE: two residual elements of the
  // current tile are prefetched after every MFMA (64 in all), one K-step before they are needed.
  // Rows >= M need no special casing (zero weights, out-of-descriptor stores/loads, masked statistics),
  // so the MFMA stream carries no branches.
  // FIRST: first K-step of a tile -- the first k-pair's MFMAs take a zero C operand instead of the old
  // accumulator (no separate clearing pass, and the previous tile's values die at their store).
  auto kstep = [&](auto first_c, auto pre_c, int buf, const Tile& t) {
    constexpr bool FIRST = decltype(first_c)::value != 0;
    constexpr bool PRE = decltype(pre_c)::value != 0;
    const __amdgpu_buffer_rsrc_t rr = slab_rsrc(PRE ? a.res : a.y, t);
    const int rsoff = tile_soff(t);
    static_for<0, BK / 2>([&](auto kk_c) {
      constexpr int kk = decltype(kk_c)::value;
      const int k = 2 * kk + lk;
      float av[2], bv[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) av[mi] = As[buf][k][wave_u * 64 + mi * 32 + lr];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) bv[ti] = Bs[buf][k][ti * 32 + lr];
      static_for<0, 4>([&](auto j_c) {
        constexpr int j = decltype(j_c)::value;
        if constexpr (FIRST && kk == 0) {
          const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc[j >> 1][j & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j >> 1], bv[j & 1], zero, 0, 0, 0);
        } else {
          acc[j >> 1][j & 1] =
              __builtin_amdgcn_mfma_f32_32x32x2f32(av[j >> 1], bv[j & 1], acc[j >> 1][j & 1], 0, 0, 0);
        }
        if constexpr (PRE) {
          constexpr int e0 = (kk * 4 + j) * 2;  // elements e0, e0+1 of 64: (mi, ti, r) = (e>>5, (e>>4)&1, e&15)
          preload_elem(ic<(e0 >> 5)>{}, ic<((e0 >> 4) & 1)>{}, ic<(e0 & 15)>{}, rr, rsoff);
          preload_elem(ic<((e0 + 1) >> 5)>{}, ic<(((e0 + 1) >> 4) & 1)>{}, ic<((e0 + 1) & 15)>{}, rr, rsoff);
          __builtin_amdgcn_sched_barrier(0);  // pin the interleave: one MFMA, two loads
        }
      });
    });
  };

  // Last K-step of a tile, sub-tile-major: the 8 MFMAs of one 32x32 sub-tile run back to back, so that
  // sub-tile is final while the next one is still accumulating; its 16 elements per lane are drained (bias,
  // residual, statistics, store) two per MFMA of the following sub-tile.  Only the fourth sub-tile's 16
  // stores are issued without MFMA cover.
  auto kstep_last = [&](auto res_c, int buf, const Tile& t, int bsel) {
    const __amdgpu_buffer_rsrc_t yr = slab_rsrc(a.y, t);
    const int ysoff = tile_soff(t);
    float fsum = 0.f, fsq = 0.f;
    static_for<0, 4>([&](auto sub_c) {
      constexpr int sub = decltype(sub_c)::value;
      constexpr int mi = sub >> 1, ti = sub & 1;
      static_for<0, BK / 2>([&](auto kk_c) {
        constexpr int kk = decltype(kk_c)::value;
        const int k = 2 * kk + lk;
        const float av = As[buf][k][wave_u * 64 + mi * 32 + lr];
        const float bv = Bs[buf][k][ti * 32 + lr];
        acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi][ti], 0, 0, 0);
        if constexpr (sub > 0) {
          constexpr int pm = (sub - 1) >> 1, pti = (sub - 1) & 1;
          drain_elem(res_c, ic<pm>{}, ic<pti>{}, ic<2 * kk>{}, t, bsel, yr, ysoff, fsum, fsq);
          drain_elem(res_c, ic<pm>{}, ic<pti>{}, ic<2 * kk + 1>{}, t, bsel, yr, ysoff, fsum, fsq);
          __builtin_amdgcn_sched_barrier(0);  // pin the interleave: one MFMA, two stores
        }
      });
    });
    static_for<0, 16>([&](auto r_c) { drain_elem(res_c, ic<1>{}, ic<1>{}, r_c, t, bsel, yr, ysoff, fsum, fsq); });
    if (a.ostats) write_stats(t, fsum, fsq);
  };

  const int total = (hi - lo) * a.nsteps;
  const bool has_res = a.res != nullptr;  // kernel-uniform
  int g = 0;

  // pipeline fill
  load_step(0);
  store_step(0);
  advance_load();
  __syncthreads();  // staging buffer 0 written by every thread
  if (a.stamps) t_loop = __builtin_amdgcn_s_memtime();

  for (int tile = lo; tile < hi; ++tile) {
    const Tile cur = decode(tile);
    const int bsel = (tile - lo) & 1;
    {  // bias of this tile's rows, read in its last K-step (at least one barrier lies between)
      const int row = cur.m0 + tid;
      float b = 0.f;
      if (row < a.M) {
        if (a.bias) b = a.bias[row];
        if (a.bias_n) b += a.bias_n[(size_t)cur.n * a.M + row];
      }
      biasS[bsel][tid] = b;
    }
    // One pipeline step around a K-step body: issue the next staging loads, run the body on the current
    // LDS slot, then write the staged activations into the other slot and meet at the barrier.
    auto step = [&](auto&& body) {
      const int buf = g & 1;
      const bool more = g + 1 < total;
      if (more) load_step(buf ^ 1);
      body(buf);
      if (more) {
        store_step(buf ^ 1);
        advance_load();
      }
      __syncthreads();
      ++g;
    };
    // The first K-step is peeled so that the accumulators are (re)defined before any use in every tile:
    // nothing is carried across the tile loop and the register allocator keeps a single set.
    const bool pre_first = has_res && a.nsteps == 2;
    if (pre_first)
      step([&](int buf) { kstep(ic<1>{}, ic<1>{}, buf, cur); });
    else
      step([&](int buf) { kstep(ic<1>{}, ic<0>{}, buf, cur); });
    for (int ks = 1; ks < a.nsteps - 1; ++ks) {
      if (has_res && ks == a.nsteps - 2)
        step([&](int buf) { kstep(ic<0>{}, ic<1>{}, buf, cur); });
      else
        step([&](int buf) { kstep(ic<0>{}, ic<0>{}, buf, cur); });
    }
    if (has_res)
      step([&](int buf) { kstep_last(ic<1>{}, buf, cur, bsel); });
    else
      step([&](int buf) { kstep_last(ic<0>{}, buf, cur, bsel); });
  }

  if (a.stamps && tid == 0) {
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* d = a.stamps + (size_t)blockIdx.x * 6;
    d[0] = t_begin;
    d[1] = t_loop;
    d[2] = t_end;
    d[3] = (unsigned long long)(hi - lo);
    d[4] = hwid;
    d[5] = xcc;
  }
}

static int persistent_grid() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  return 2 * cus;  // two resident workgroups per CU (<= 256 VGPRs, 42 KB LDS each)
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
  if (res && ostats) {
    set_error("ps_conv1x1_f32: residual and output statistics cannot be combined (no Conv-TasNet stage needs both)");
    return PS_E_UNSUPPORTED;
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
  if (a.nsteps < 2) a.nsteps = 2;  // a tile needs a first (zero-C) K-step and a last (draining) K-step
  a.stamps = (unsigned long long*)g_debug_buffer;
  const int grid = a.ntiles < persistent_grid() ? a.ntiles : persistent_grid();
  {
    LaunchTimer timer("conv1x1", (hipStream_t)stream);
    hipLaunchKernelGGL(conv1x1_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_conv1x1_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
