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
// PLANES = 2: "fp16x2": two fp16 terms per operand (x = x0 + x1, 22 significant bits: x0 = fp16(x), x1 = fp16(x - x0))
//             and three products W0x0, W0x1, W1x0 on v_mfma_f32_32x32x16_f16 -- half the matrix-pipe work of PLANES = 3
//             at <= 3 * 2^-22 relative error per product (PLANES = 3 / fp32: 2^-24).  fp16 has 5 exponent bits, so both
//             operands are brought into its range by powers of two: the packer scales the weight matrix to max |w| in
//             [2^13, 2^14) (BfArgs::oscale undoes it in the epilogue), activations are scaled by 2^-2 behind a
//             normalising prologue (|u| <= 2.6e5 representable; below 0.5 the low term is subnormal: absolute
//             resolution 2^-23 of a unit-variance row) and by 2^-4 when raw (|x| <= 1e6).  Beyond that range the
//             product overflows to inf / NaN in the output -- loudly, not silently.
//
// v_mfma_f32_32x32x16_bf16: A fragment = 8 consecutive k of one row, B fragment = 8 consecutive k of one column, so
// both operand tiles live in LDS k-innermost ([row][16 k] bf16 = 32 B per row, read with ds_read_b128, conflict free).
// Weights arrive already split and in exactly that image from the host packer ([m-tile][k-step][plane][256][16]), so
// a K-step of weights is a straight 8*PLANES KiB copy.  Activations are fp32 [k][t] in HBM: they are transformed
// (prologue), split and written as one 16-byte LDS row piece per plane and (frame, k-half).
//
// Two kernels:
//   * conv1x1_bf16_pp_kernel ("ping-pong", the one the Conv-TasNet shapes run on): one persistent 512-thread workgroup
//     per CU whose two halves alternate between an MFMA phase and a staging phase; all operand traffic by LDS-DMA.
//   * conv1x1_bf16_kernel: one tile per workgroup (256 x 128, or 256 x 32 when even those cannot fill the chip), two
//     LDS slots, operands staged through registers, one barrier per K-step -- for everything the persistent kernel
//     does not take (few tiles, K < 64, workgroup runs that would span more than two utterances).
#include <type_traits>

#include "ps_common.h"

namespace ps {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

// operand element type and MFMA of an arithmetic (PLANES = 1 / 3: bf16 terms, PLANES = 2: fp16 terms)
template <int PLANES>
struct Arith {
  using x8 = bf16x8;
  using x2 = bf16x2;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Arith<2> {
  using x8 = f16x8;
  using x2 = f16x2;
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

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
  int x_bf16, y_bf16;  // the rows of x / y are bf16 in HBM (PLANES = 1 only; bias, residual, statistics stay fp32 / fp64)
  int delay, groups;   // interleaved kernel: start offset (cycles) between the `groups` phases of workgroups
  int fm_ld;           // register-B kernel, frame-major output: floats between consecutive frames (>= M)
  const float* ln_gamma;  // register-B kernel, LayerNorm epilogue (ps_conv1x1_f16x2_ln_f32): 128 channels
  const float* ln_beta;
  float ln_eps;
  int ln_inside;  // 1: y = LN(W f(x) + b + res) (post-norm transformer blocks) instead of res + LN(W f(x) + b)
  int pair_r;          // interleaved kernel, two m-tiles: > 0 = workgroups per (utterance, m-tile) row; the two
                       // workgroups that read the same activation tiles are placed on the same XCD (see bf16_launch)
  // PLANES = 2: activations are multiplied by a power of two before the fp16 split, accumulators by
  // winv / (that power) in the epilogue (winv = 2^-w_exp undoes the packer's weight scale).  The power is `xscale` (the
  // host's choice: from a bound on |f(x)|, or a default), or -- x_amax given -- per utterance from the producer's partial
  // maxima of |x| ([N][x_amax_parts]).  y_amax: [N][tiles_m * tiles_t * 4] partial maxima of |y| for the next consumer.
  float xscale, winv;
  const float* x_amax;
  int x_amax_parts;
  float* y_amax;
  // the maxima are those of a producer whose output reaches the GEMM through a per-channel affine map + PReLU (a folded
  // BatchNorm prologue): the range of f(x) is bounded by amax_mul * max |x| + amax_add (amax_mul = 0: the maxima as they are)
  float amax_mul, amax_add;
};

// PLANES = 2: the scale pair of utterance n (every lane of the calling wave gets the same values)
__device__ __forceinline__ void f16_scales(const BfArgs& a, int n, int lane, float& xs, float& os) {
  xs = a.xscale;
  if (a.x_amax) {
    float m = 0.f;
    const float* src = a.x_amax + (size_t)n * a.x_amax_parts;
    for (int i = lane; i < a.x_amax_parts; i += 64) m = fmaxf(m, src[i]);
    m = wave_max(m);
    if (a.amax_mul > 0.f) m = m * a.amax_mul + a.amax_add;
    // max |x| in [2^e, 2^(e+1)) goes to [2^14, 2^15) (fp16 holds up to 65504); an all-zero utterance keeps 1
    int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 255) - 127;
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    xs = m > 0.f ? ldexpf(1.f, 14 - e) : 1.f;
  }
  os = a.winv / xs;  // (powers of two: exact)
}

// TT = frames per workgroup tile: 128 (2 x 2 waves of 128 x 64) or 32 (4 x 1 waves of 64 x 32 -- the small-grid
// variant: four times the workgroups for launches that cannot fill the chip, e.g. one utterance at a time)
template <int PLANES, int TT = XB_T>
struct BfLds {
  static constexpr int A_BYTES = PLANES * XB_M * XB_K * 2;  // 8 KiB per plane
  static constexpr int B_BYTES = PLANES * TT * XB_K * 2;    // 4 KiB per plane at TT = 128
  static constexpr int SLOT = A_BYTES + B_BYTES;
  static constexpr int KTAB = 512;   // floats per table: sc / sh of the prologue (K <= 512)
  static constexpr int TOTAL = 2 * SLOT + 2 * KTAB * 4 + 64;  // PLANES = 3: 76 KiB -> two workgroups per CU
};

template <int PLANES, bool TR, bool STATS, bool RES, int TT = XB_T, bool XB = false, bool YB = false>
__global__ __launch_bounds__(256, 2) void conv1x1_bf16_kernel(BfArgs a) {
  static_assert(!(XB || YB) || PLANES == 1, "bf16 rows go with bf16 products");
  constexpr int XE = XB ? 2 : 4, YE = YB ? 2 : 4;
  using L = BfLds<PLANES, TT>;
  using AR = Arith<PLANES>;
  using x8 = typename AR::x8;
  using x2 = typename AR::x2;
  constexpr bool NARROW = TT != XB_T;
  constexpr int MI = NARROW ? 2 : 4, TI = NARROW ? 1 : 2;  // 32 x 32 MFMA tiles per wave
  constexpr int WROWS = MI * 32;                            // rows per wave
  __shared__ __attribute__((aligned(16))) unsigned char smem[L::TOTAL];
  float* tab = reinterpret_cast<float*>(smem + 2 * L::SLOT);  // sc[512] | sh[512]
  double* red = reinterpret_cast<double*>(smem + 2 * L::SLOT + 2 * L::KTAB * 4);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = NARROW ? wave : wave >> 1, wt = NARROW ? 0 : wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int t0 = blockIdx.x * TT, mt = blockIdx.y, n = blockIdx.z;
  const int m0 = mt * XB_M;
  // (narrow tiles: the grid covers whole 128-frame tiles, pad frames included -- every kernel of the path writes the
  //  pad frames of its output with finite values so that no later stage can read uninitialised memory as data)

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

  [[maybe_unused]] float xs = 1.f, os = 1.f;
  if constexpr (PLANES == 2) f16_scales(a, n, lane, xs, os);

  // ---- staging ------------------------------------------------------------------------------------------------
  constexpr int A_PIECES = 2 * PLANES;  // 16-byte pieces per thread per K-step
  const u32x4v* wsrc = reinterpret_cast<const u32x4v*>(a.wt) + (size_t)mt * a.ksteps * (L::A_BYTES / 16);
  const int bt = tid % TT, bh = (tid / TT) & 1;  // activation staging: frame, k-half
  const bool b_active = tid < 2 * TT;           // (narrow tiles: 64 of the 256 threads)
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
      reinterpret_cast<unsigned char*>(const_cast<float*>(a.x)) + (size_t)n * a.K * a.ldt * XE, 0, a.K * a.ldt * XE,
      0x00020000);
  const int xb_voff = (8 * bh * a.ldt + t0 + bt) * XE;
  auto load_b = [&](int ks, auto q_c) {
    constexpr int q = decltype(q_c)::value;
    const int soff = ks * XB_K * a.ldt * XE;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (XB)
        breg[q][j] = __builtin_bit_cast(
            float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(xr, xb_voff, soff + j * a.ldt * XE, 0) << 16);
      else
        breg[q][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, xb_voff, soff + j * a.ldt * 4, 0));
    }
    (void)b_active;  // idle threads load in-range duplicates (tid % TT) and skip the LDS write
  };
  auto store_step = [&](int ks, int slot, auto q_c) {
    constexpr int q = decltype(q_c)::value;
    unsigned char* sa = smem + slot * L::SLOT;
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) reinterpret_cast<u32x4v*>(sa)[tid + 256 * i] = areg[A_DEPTH - 1][i];
    if ((a.ablate & 4) || !b_active) return;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float u = breg[q][j];
      if constexpr (TR) {
        const int k = ks * XB_K + 8 * bh + j;
        if (a.pro.pre_relu) u = relu_keep_nan(u);
        u = u * tab[k] + tab[512 + k];
        if (a.pro.prelu) u = prelu(u, slope);
        if (a.pro.post_tanh) u = tanhf(u);
        if (k >= a.K) u = 0.f;
      }
      if constexpr (PLANES == 2) u *= xs;
      v[j] = u;
    }
    unsigned char* sb = sa + L::A_BYTES;
#pragma unroll
    for (int p = 0; p < PLANES; ++p) {
      x8 piece;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const x2 h = __builtin_convertvector(f32x2v{v[j], v[j + 1]}, x2);
        piece[j] = h[0];
        piece[j + 1] = h[1];
        if (p + 1 < PLANES) {
          const f32x2v back = __builtin_convertvector(h, f32x2v);
          v[j] -= back[0];
          v[j + 1] -= back[1];
        }
      }
      *reinterpret_cast<x8*>(sb + ((p * TT + bt) * XB_K + 8 * bh) * 2) = piece;
    }
  };

  f32x16 acc[MI][TI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ti][r] = 0.f;

  auto compute = [&](int slot) {
    if (a.ablate & 1) return;
    const unsigned char* sa = smem + slot * L::SLOT;
    const unsigned char* sb = sa + L::A_BYTES;
    x8 bf[PLANES][TI];
#pragma unroll
    for (int p = 0; p < PLANES; ++p)
#pragma unroll
      for (int ti = 0; ti < TI; ++ti)
        bf[p][ti] = *reinterpret_cast<const x8*>(sb + ((p * TT + wt * 64 + ti * 32 + lr) * XB_K + 8 * lh) * 2);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      x8 af[PLANES];
#pragma unroll
      for (int p = 0; p < PLANES; ++p)
        af[p] = *reinterpret_cast<const x8*>(sa + ((p * XB_M + wm * WROWS + mi * 32 + lr) * XB_K + 8 * lh) * 2);
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        if constexpr (PLANES == 3) {
          // smallest terms first
          acc[mi][ti] = AR::mfma(af[1], bf[1][ti], acc[mi][ti]);
          acc[mi][ti] = AR::mfma(af[2], bf[0][ti], acc[mi][ti]);
          acc[mi][ti] = AR::mfma(af[0], bf[2][ti], acc[mi][ti]);
          acc[mi][ti] = AR::mfma(af[1], bf[0][ti], acc[mi][ti]);
          acc[mi][ti] = AR::mfma(af[0], bf[1][ti], acc[mi][ti]);
        }
        if constexpr (PLANES == 2) {
          acc[mi][ti] = AR::mfma(af[1], bf[0][ti], acc[mi][ti]);
          acc[mi][ti] = AR::mfma(af[0], bf[1][ti], acc[mi][ti]);
        }
        acc[mi][ti] = AR::mfma(af[0], bf[0][ti], acc[mi][ti]);
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
  [[maybe_unused]] float amx2[TI] = {};  // per column block; the pad-frame mask is applied once at the end
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<unsigned char*>(a.y) + (size_t)n * a.M * a.ldt * YE, 0, a.M * a.ldt * YE, 0x00020000);
  // (bf16 output rows with a residual: the residual stream is bf16 too -- the "bf16 stream" arithmetic of BASELINE config 3)
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<unsigned char*>(const_cast<float*>(RES ? a.res : a.x)) + (size_t)n * a.M * a.ldt * YE, 0,
      RES ? a.M * a.ldt * YE : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.bias ? a.bias : a.x), 0, a.bias ? a.M * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t bnr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.bias_n ? a.bias_n + (size_t)n * a.M : a.x), 0, a.bias_n ? a.M * 4 : 0, 0x00020000);
  const int lane_off = (4 * lh * a.ldt + lr) * 4;
  const int tile_off = ((m0 + wm * WROWS) * a.ldt + t0 + wt * 64) * 4;
  bool cm[TI];  // (a select, not a 0/1 factor: a pad frame may hold anything and 0 * NaN would poison the statistics)
#pragma unroll
  for (int ti = 0; ti < TI; ++ti) cm[ti] = t0 + wt * 64 + ti * 32 + lr < a.T;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    float bsum[16], rv[TI][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rc = mi * 32 + (r & 3) + 8 * (r >> 2);
      const int moff = (m0 + wm * WROWS + rc + 4 * lh) * 4;
      bsum[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(br, moff, 0, 0)) +
                __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bnr, moff, 0, 0));
      if constexpr (RES) {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti) {
          if constexpr (YB)
            rv[ti][r] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(
                                                      rr, lane_off >> 1, (tile_off + rc * a.ldt * 4 + ti * 128) >> 1, 0) << 16);
          else
            rv[ti][r] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rr, lane_off, tile_off + rc * a.ldt * 4 + ti * 128, 0));
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rc = mi * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) {
        float v = PLANES == 2 ? acc[mi][ti][r] * os + bsum[r] : acc[mi][ti][r] + bsum[r];
        if constexpr (STATS) {
          const float vm = cm[ti] ? v : 0.f;
          fsum += vm;
          fsq += vm * vm;
        }
        if constexpr (RES) v += rv[ti][r];
        if constexpr (PLANES == 2) amx2[ti] = fmaxf(amx2[ti], fabsf(v));
        if constexpr (YB)
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)v), yr, lane_off >> 1,
                                                (tile_off + rc * a.ldt * 4 + ti * 128) >> 1, 0);
        else
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, lane_off,
                                                tile_off + rc * a.ldt * 4 + ti * 128, 0);
      }
    }
  }
  if (a.stamps && tid == 0 && !NARROW) {
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
  if constexpr (PLANES == 2) {
    if (a.y_amax) {  // partial maxima of |y|: same slots as the statistics (narrow tiles: the waves of a slot take turns,
      float amx = 0.f;      // which is why narrow launches go through LDS below)
#pragma unroll
      for (int ti = 0; ti < TI; ++ti) amx = fmaxf(amx, cm[ti] ? amx2[ti] : 0.f);
      amx = wave_max(amx);
      const int parts = a.tiles_m * a.tiles_t * 4;
      if constexpr (NARROW) {
        float* redf = reinterpret_cast<float*>(red);
        __syncthreads();
        if (lane == 0) redf[wave] = amx;
        __syncthreads();
        if (tid == 0)
          a.y_amax[(size_t)n * parts + (mt * a.tiles_t + (blockIdx.x >> 2)) * 4 + (blockIdx.x & 3)] =
              fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
        __syncthreads();
      } else if (lane == 0) {
        a.y_amax[(size_t)n * parts + (mt * a.tiles_t + blockIdx.x) * 4 + wave] = amx;
      }
    }
  }
  if constexpr (STATS) {
    double s = wave_sum((double)fsum), q = wave_sum((double)fsq);
    const int parts = a.tiles_m * a.tiles_t * 4;  // four slots per 256 x 128 tile: its waves, or its narrow workgroups
    if constexpr (NARROW) {
      __syncthreads();  // every wave is done with the operand slots: `red` may be anywhere in LDS
      if (lane == 0) {
        red[wave * 2] = s;
        red[wave * 2 + 1] = q;
      }
      __syncthreads();
      if (tid == 0) {
        s = red[0] + red[2] + red[4] + red[6];
        q = red[1] + red[3] + red[5] + red[7];
        double* dst = a.ostats + ((size_t)n * parts + (mt * a.tiles_t + (blockIdx.x >> 2)) * 4 + (blockIdx.x & 3)) * 2;
        dst[0] = s;
        dst[1] = q;
      }
    } else if (lane == 0) {
      const int part = (mt * a.tiles_t + blockIdx.x) * 4 + wave;
      double* dst = a.ostats + ((size_t)n * parts + part) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

// ---- ping-pong persistent variant ------------------------------------------------------------------------------------
// Measured on the kernel above (tools/stamp_bf16.py): the two co-resident workgroups fall into phase -- both stream
// MFMAs, then both stage -- so a K-step costs compute (3047 cycles, two waves sharing a matrix pipe) PLUS staging
// (1563) PLUS barrier (624) instead of their maximum.  Here ONE 512-thread workgroup per CU holds both tiles and the
// alternation is explicit: the two halves (waves 0-3 / 4-7; one wave of each per SIMD) own the two 256 x 128 tiles of a
// 256 (m) x 256 (t) supertile and run the same loop one phase apart, with two workgroup barriers per K-step:
//
//     phase 2s   : half 0 issues the MFMAs of step s      | half 1 stages (transform, split, LDS writes, load issue)
//     phase 2s+1 : half 0 stages                          | half 1 issues the MFMAs of step s
//
// so the matrix pipe always has exactly one wave per SIMD feeding it and the VALU/LDS-write work of the other half runs
// beside it.  Consequences of the fixed alternation:
//   * both halves use the same weight tile, so each stages only half of it (LDS ring of two weight slots: slot s&1 is
//     read in phases 2s, 2s+1 and rewritten in phases 2s+2 (half 1) and 2s+3 (half 0));
//   * a half never stages and computes at the same time, so it needs ONE activation slot;
//   * the register queues of loads in flight are consumed and refilled in place (static rotation by a wave-uniform
//     switch): a register move out of a queue slot would make hipcc wait for the load that is landing in it;
//   * a tile's drain falls into its half's staging phase; the accumulators are re-initialised right there with the
//     residual tile of the half's NEXT tile (out_conv) so that no load round trip sits between the last MFMA and the
//     stores -- the residual streams in behind the stores while the other half computes.
// The workgroup is persistent over a contiguous run of supertiles (t fastest, then m-tile, then utterance).
constexpr int PP_MAXU = 2;  // utterances one workgroup's run may touch (prologue tables kept in LDS)

template <int PLANES>
struct PpLds {
  static constexpr int A_SLOT = PLANES * XB_M * XB_K * 2;  // 8 KiB per plane
  static constexpr int B_SLOT = PLANES * XB_T * XB_K * 2;  // 4 KiB per plane
  static constexpr int NA = 3;                             // weight ring slots
  static constexpr int D = 3;                              // raw activation ring slots per half
  static constexpr int RAW = XB_K * XB_T * 4;              // one K-step of raw fp32 activations [16][128]
  static constexpr int OFF_RAW = NA * A_SLOT;              // [half][D][RAW]
  static constexpr int OFF_B = OFF_RAW + 2 * D * RAW;      // split activation slot of half 0 | half 1
  static constexpr int OFF_TAB = OFF_B + 2 * B_SLOT;       // [PP_MAXU][sc[512] | sh[512]] floats
  static constexpr int OFF_BIAS = OFF_TAB + PP_MAXU * 1024 * 4;  // [2 supertile parities][bias[256] | bias_n[256]] floats
  static constexpr int TOTAL = OFF_BIAS + 4 * XB_M * 4;    // PLANES = 3: 156 KiB
};
static_assert(PpLds<3>::TOTAL <= 160 * 1024, "one workgroup per CU");

// All operand traffic of the K loop is LDS-DMA (buffer_load ... lds): weights straight into a 3-slot ring in their
// final image, raw fp32 activations into a per-half 3-slot ring from which the owning half transforms / splits them
// into its bf16 slot one step later.  No operand waits in registers, so there are no register queues to rotate and
// hipcc's load bookkeeping cannot serialise the pipeline; the waits are counted by hand (every wave issues the same
// DMA sequence every iteration -- steps past the end go through an empty descriptor -- so "vmcnt(n)" always means
// "everything but my n youngest operations").
// The scalar unit is shared by the eight waves of the CU, so the per-iteration scalar path is kept to counters and
// adds: each pipeline stage (activation DMA, weight DMA, staging, compute) walks the K-steps with its own (supertile,
// step) counter and only decodes (utterance, m-tile, t-pair), rebuilds descriptors etc. when it crosses a supertile.
template <int V>
using bic = std::integral_constant<int, V>;

// Sum over the 64 lanes on DPP row rotations (every lane of a 16-lane row ends up with the row's total) and four
// v_readlane: ~30 instructions with no LDS round trips.  (The fp64 butterfly of wave_sum() is 24 ds_bpermute with a
// wait each -- 2.5 k cycles per tile drain next to the partner half's MFMA stream.)  fp32 is enough here: each lane's
// value is already an fp32 sum over its 128 elements.
__device__ __forceinline__ float wave_sum_dpp(float v) {
  auto ror = [](float x, auto ctl_c) {
    constexpr int ctl = decltype(ctl_c)::value;  // 0x120 + n = row_ror:n
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctl, 0xf, 0xf, false));
  };
  v += ror(v, bic<0x128>{});
  v += ror(v, bic<0x124>{});
  v += ror(v, bic<0x122>{});
  v += ror(v, bic<0x121>{});
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}

template <int PLANES, bool TR, bool STATS, bool RES>
__global__ __launch_bounds__(512, 1) void conv1x1_bf16_pp_kernel(BfArgs a) {
  using L = PpLds<PLANES>;
  constexpr int D = L::D, NA = L::NA;
  constexpr int P = 2 + PLANES;                   // DMA operations per wave per iteration: 2 activation, PLANES weight
  // Iteration i issues x(i+D) and A(i+2) at the start of its compute phase (2 + PLANES operations, in that order).
  constexpr int W_X = PLANES + (D - 1) * P;       // youngest operations allowed in flight when x(i+1) must have landed
  constexpr int W_A = P;                          // ... when this wave's share of A(i+1) must have landed
  __shared__ __attribute__((aligned(16))) unsigned char smem[L::TOTAL];
  float* tab = reinterpret_cast<float*>(smem + L::OFF_TAB);
  float* bias_lds = reinterpret_cast<float*>(smem + L::OFF_BIAS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave >> 2;  // half
  const int hw = wave & 3;
  const int ptid = tid & 255;
  const int wm = hw >> 1, wt = hw & 1;
  const int lr = lane & 31, lh = lane >> 5;

  const int S = a.ksteps;
  const int st_per = (a.tiles_t + 1) >> 1;
  const int nsuper = st_per * a.tiles_m * a.N;
  const int G = gridDim.x;
  const int lo = (int)((long long)blockIdx.x * nsuper / G), hi = (int)((long long)(blockIdx.x + 1) * nsuper / G);
  if (lo >= hi) return;
  const int total = (hi - lo) * S;
  auto decode = [&](int idx, int& n, int& mt, int& t2) {  // supertile index -> utterance, m-tile, pair of t-tiles
    t2 = idx % st_per;
    const int r = idx / st_per;
    mt = r % a.tiles_m;
    n = r / a.tiles_m;
  };
  int n_lo, mt_lo, t2_lo;
  decode(lo, n_lo, mt_lo, t2_lo);

  const float slope = (TR && a.pro.prelu) ? a.pro.slope[0] : 1.f;
  const bool plain_tr = !a.pro.pre_relu && !a.pro.post_tanh;  // kernel-uniform: scale/shift + PReLU only
  if constexpr (TR) {
    const bool has_norm = a.pro.norm != PS_NORM_NONE;
    const int n_hi = ((hi - 1) / st_per) / a.tiles_m;
    for (int u = 0; u <= n_hi - n_lo; ++u) {  // every wave reduces the producer's partial statistics itself
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
      const int k = tid;  // 512 threads, 512 table rows
      float sc = 0.f, sh = 0.f;
      if (k < a.K) {
        sc = has_norm ? a.pro.gamma[k] * rstd : 1.f;
        sh = has_norm ? a.pro.beta[k] - mean * sc : 0.f;
      }
      tab[u * 1024 + k] = sc;
      tab[u * 1024 + 512 + k] = sh;
    }
  }

  // ---- activation DMA: raw fp32 [16][128] of one K-step -> ring slot --------------------------------------------------
  // piece (16 B) p = i*256 + hw*64 + lane = row p>>5, frames 4*(p&31)..; a wave's DMA covers two rows of 128 frames
  const int x_voff = ((lane >> 5) * a.ldt + (lane & 31) * 4) * 4;
  const int x_step = XB_K * a.ldt * 4, x_half = 8 * a.ldt * 4;
  float* const raw_w = reinterpret_cast<float*>(smem + L::OFF_RAW + h * D * L::RAW) + hw * 256;
  int xi = lo, xk = 0, xso = 0;  // supertile, step, scalar offset of the step's first row of this wave
  __amdgpu_buffer_rsrc_t xr;
  auto x_setup = [&]() {
    int n, mt, t2;
    decode(xi, n, mt, t2);
    const bool ok = xi < hi && 2 * t2 + h < a.tiles_t;  // past the end / tile outside the row: empty descriptor
    xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)n * a.K * a.ldt, 0,
                                           ok ? a.K * a.ldt * 4 : 0, 0x00020000);
    xso = (hw * 2 * a.ldt + (2 * t2 + h) * XB_T) * 4;
  };
  auto dma_x = [&](int slot) {
    float* dst = raw_w + slot * (L::RAW / 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, x_voff, xso, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst + 1024, 16, x_voff, xso + x_half, 0, 0);
  };
  auto x_advance = [&]() {
    xso += x_step;
    if (++xk == S) {
      xk = 0;
      ++xi;
      x_setup();
    }
  };

  // ---- weight DMA: this half's share (half a slot) of one K-step, already in its LDS image ----------------------------
  const int w_voff = lane * 16;
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(a.wt), 0, a.tiles_m * S * L::A_SLOT, 0x00020000);
  const int a_share = (h * PLANES * 256 + hw * 64) * 16;  // byte offset of this wave's pieces inside a slot
  int ai = lo, ak = 0, aso = 0;
  auto a_setup = [&]() {
    int n, mt, t2;
    decode(ai, n, mt, t2);
    // past the end: an offset outside the descriptor (reads nothing, writes zeros)
    aso = ai < hi ? mt * S * L::A_SLOT + a_share : 0x7f000000;
  };
  auto dma_a = [&](int slot) {
    float* dst = reinterpret_cast<float*>(smem + slot * L::A_SLOT + a_share);
#pragma unroll
    for (int i = 0; i < PLANES; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst + i * 1024, 16, w_voff, aso + i * 4096, 0, 0);
  };
  auto a_advance = [&]() {
    aso += L::A_SLOT;
    if (++ak == S) {
      ak = 0;
      ++ai;
      a_setup();
    }
  };

  // bias rows of supertile (n, mt) -> LDS (one 1-KiB DMA each for bias and bias_n, wave 0 only; its extra operations
  // only make its counted waits stricter).  Issued while the supertile's step 0 is staged, read by the drains S - 1
  // or more iterations later, after this wave has passed a counted wait and a barrier.
  auto bias_dma = [&](int n, int mt, int par) {
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias ? a.bias : a.x), 0, a.bias ? a.M * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t bnr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias_n ? a.bias_n + (size_t)n * a.M : a.x), 0, a.bias_n ? a.M * 4 : 0, 0x00020000);
    float* dst = bias_lds + par * 2 * XB_M;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(br, dst, 16, w_voff, mt * XB_M * 4, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(bnr, dst + XB_M, 16, w_voff, mt * XB_M * 4, 0, 0);
  };

  // ---- staging: raw ring slot -> this half's bf16 slot (prologue transform, split into PLANES bf16 terms) -------------
  const int bt = ptid & 127, bh = ptid >> 7;  // frame, k-half
  const float* const raw_r = reinterpret_cast<const float*>(smem + L::OFF_RAW + h * D * L::RAW) + 8 * bh * XB_T + bt;
  unsigned char* const bx_w = smem + L::OFF_B + h * L::B_SLOT + (bt * XB_K + 8 * bh) * 2;
  int si = lo, sk = 0, spar = 0;  // supertile, step, bias-table parity of the step to stage next
  const float* stab = tab + 8 * bh;  // this thread's table rows of the current utterance and step
  auto s_setup = [&]() {
    int n, mt, t2;
    decode(si, n, mt, t2);
    stab = tab + (n - n_lo) * 1024 + 8 * bh;
    if (wave == 0 && si < hi) bias_dma(n, mt, spar);
  };
  auto stage_b = [&](int slot) {
    if (si >= hi) return;
#ifdef PS_PP_STAMPS
    if (a.ablate & 4) return;
#endif
    const float* raw = raw_r + slot * (L::RAW / 4);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = raw[j * XB_T];
    if constexpr (TR) {
      const f32x4 sc0 = *reinterpret_cast<const f32x4*>(stab), sc1 = *reinterpret_cast<const f32x4*>(stab + 4);
      const f32x4 sh0 = *reinterpret_cast<const f32x4*>(stab + 512), sh1 = *reinterpret_cast<const f32x4*>(stab + 516);
      if (plain_tr) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = prelu(v[j] * (j < 4 ? sc0[j & 3] : sc1[j & 3]) + (j < 4 ? sh0[j & 3] : sh1[j & 3]), slope);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float u = v[j];
          if (a.pro.pre_relu) u = relu_keep_nan(u);
          u = u * (j < 4 ? sc0[j & 3] : sc1[j & 3]) + (j < 4 ? sh0[j & 3] : sh1[j & 3]);
          u = prelu(u, slope);
          if (a.pro.post_tanh) u = tanhf(u);
          v[j] = u;
        }
      }
    }
#pragma unroll
    for (int p = 0; p < PLANES; ++p) {
      bf16x8 piece;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const bf16x2 hh = __builtin_convertvector(f32x2v{v[j], v[j + 1]}, bf16x2);
        piece[j] = hh[0];
        piece[j + 1] = hh[1];
        if (p + 1 < PLANES) {
          const f32x2v back = __builtin_convertvector(hh, f32x2v);
          v[j] -= back[0];
          v[j + 1] -= back[1];
        }
      }
      *reinterpret_cast<bf16x8*>(bx_w + p * XB_T * XB_K * 2) = piece;
    }
  };
  auto s_advance = [&]() {
    stab += XB_K;
    if (++sk == S) {
      sk = 0;
      ++si;
      spar ^= 1;
      s_setup();
    }
  };

  // ---- compute side ---------------------------------------------------------------------------------------------------
  f32x16 acc[4][2];
  const int lane_off = (4 * lh * a.ldt + lr) * 4;
  // this half's tile of supertile idx exists?
  auto tile_ok = [&](int idx, int t2) { return idx < hi && 2 * t2 + h < a.tiles_t; };
  // accumulators <- residual tile of this half's tile of supertile idx (zeros when there is no residual / no such tile)
  auto init_acc = [&](int idx) {
    if constexpr (RES) {
      int n, mt, t2;
      decode(idx, n, mt, t2);
      const int slab = a.M * a.ldt * 4;
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(a.res) + (size_t)n * a.M * a.ldt, 0, tile_ok(idx, t2) ? slab : 0, 0x00020000);
      const int tile_off = ((mt * XB_M + wm * 128) * a.ldt + (2 * t2 + h) * XB_T + wt * 64) * 4;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rc = mi * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
            acc[mi][ti][r] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rr, lane_off, tile_off + rc * a.ldt * 4 + ti * 128, 0));
        }
    } else {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][ti][r] = 0.f;
    }
  };
  // hipcc waits for pending loads at the first use of a register; this empty use pins that wait (and keeps hipcc from
  // restructuring the accumulators around the drain)
  auto acc_fence = [&]() {
    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]),
                 "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
  };

  const unsigned char* const a_frag = smem + ((wm * 128 + lr) * XB_K + 8 * lh) * 2;
  const unsigned char* const b_frag = smem + L::OFF_B + h * L::B_SLOT + ((wt * 64 + lr) * XB_K + 8 * lh) * 2;
#ifdef PS_PP_STAMPS
  unsigned long long st_f = 0, st_m = 0, st_prev = 0;
#define PP_STAMP(bucket)                                                          \
  if (a.stamps) {                                                                 \
    unsigned long long now;                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory"); \
    bucket += now - st_prev;                                                      \
    st_prev = now;                                                                \
  }
#else
#define PP_STAMP(bucket)
#endif
  // `after(mi)` runs behind the MFMAs of row block mi (mi = 0, 1, 2): the DMA pieces and the scalar bookkeeping of the
  // iteration ride between MFMA blocks, where an LDS-DMA piece costs the stream ~60 cycles instead of several hundred
  // in front of the first MFMA
  auto compute = [&](int aslot, auto&& after) {
    const unsigned char* sa = a_frag + aslot * L::A_SLOT;
    bf16x8 bf[PLANES][2], af[2][PLANES];
    // fragment reads in the order the MFMAs use them: (W1, x1), (W2, x0), (W0, x2), ...
    constexpr int ORD_A[3] = {1, 2, 0}, ORD_B[3] = {1, 0, 2};
#pragma unroll
    for (int q = 0; q < PLANES; ++q) {
      const int pa = PLANES == 3 ? ORD_A[q] : 0, pb = PLANES == 3 ? ORD_B[q] : 0;
      af[0][pa] = *reinterpret_cast<const bf16x8*>(sa + pa * XB_M * XB_K * 2);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
        bf[pb][ti] = *reinterpret_cast<const bf16x8*>(b_frag + (pb * XB_T + ti * 32) * XB_K * 2);
    }
    PP_STAMP(st_f)
#ifdef PS_PP_STAMPS
    if (a.ablate & 1) {
      after(bic<0>{});
      after(bic<1>{});
      after(bic<2>{});
      return;
    }
#endif
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      // the next row block's weight fragments are requested before this block's MFMAs issue
      if (mi < 3) {
#pragma unroll
        for (int p = 0; p < PLANES; ++p)
          af[(mi + 1) & 1][p] = *reinterpret_cast<const bf16x8*>(sa + ((p * XB_M + (mi + 1) * 32) * XB_K) * 2);
      }
      const bf16x8* f = af[mi & 1];
      // smallest terms first; the two column blocks alternate so that consecutive MFMAs never share an accumulator
      if constexpr (PLANES == 3) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], bf[1][ti], acc[mi][ti], 0, 0, 0);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2], bf[0][ti], acc[mi][ti], 0, 0, 0);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], bf[2][ti], acc[mi][ti], 0, 0, 0);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], bf[0][ti], acc[mi][ti], 0, 0, 0);
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], bf[1][ti], acc[mi][ti], 0, 0, 0);
      }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], bf[0][ti], acc[mi][ti], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (mi == 0) after(bic<0>{});
      if (mi == 1) after(bic<1>{});
      if (mi == 2) after(bic<2>{});
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // finished tile of supertile idx: bias, statistics, stores; then the accumulators take the residual of this half's
  // next tile.  Element (mi, ti, r) = row 128 wm + 32 mi + (r&3) + 8 (r>>2) + 4 lh, column 64 wt + 32 ti + lr.
  auto drain = [&](int idx, int parity) {
    int n, mt, t2, nn, nmt, nt2;
    decode(idx, n, mt, t2);
    decode(idx + 1, nn, nmt, nt2);
    const bool ok = 2 * t2 + h < a.tiles_t;
    const int m0 = mt * XB_M, t0 = (2 * t2 + h) * XB_T;
    float fsum = 0.f, fsq = 0.f;
    const int slab = a.M * a.ldt * 4;
    const __amdgpu_buffer_rsrc_t yr =
        __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)n * a.M * a.ldt, 0, ok ? slab : 0, 0x00020000);
    const int tile_off = ((m0 + wm * 128) * a.ldt + t0 + wt * 64) * 4;
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(RES ? a.res : a.y) + (size_t)nn * a.M * a.ldt, 0, (RES && tile_ok(idx + 1, nt2)) ? slab : 0,
        0x00020000);
    [[maybe_unused]] const int ntile_off = ((nmt * XB_M + wm * 128) * a.ldt + (2 * nt2 + h) * XB_T + wt * 64) * 4;
    bool cm[2];  // (a select, not a 0/1 factor: 0 * NaN would poison the statistics)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) cm[ti] = t0 + wt * 64 + ti * 32 + lr < a.T;
    const float* bl = bias_lds + parity * 2 * XB_M + wm * 128 + 4 * lh;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + mi * 32 + 8 * rq) +
                         *reinterpret_cast<const f32x4*>(bl + XB_M + mi * 32 + 8 * rq);
#pragma unroll
        for (int r3 = 0; r3 < 4; ++r3) {
          const int r = rq * 4 + r3;
          const int rc = mi * 32 + r3 + 8 * rq;
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) {
            const float v = acc[mi][ti][r] + b4[r3];
            if constexpr (STATS) {
              const float vm = cm[ti] ? v : 0.f;
              fsum += vm;
              fsq += vm * vm;
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, lane_off,
                                                  tile_off + rc * a.ldt * 4 + ti * 128, 0);
            if constexpr (RES)
              acc[mi][ti][r] = __builtin_bit_cast(
                  float, __builtin_amdgcn_raw_buffer_load_b32(rr, lane_off, ntile_off + rc * a.ldt * 4 + ti * 128, 0));
            else
              acc[mi][ti][r] = 0.f;
          }
          // keep (store, reload) pairs in program order: hoisted reloads (or 128 pre-computed store values) would
          // double the live accumulators
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (STATS) {
      const double s = wave_sum_dpp(fsum), q = wave_sum_dpp(fsq);
      if (lane == 0 && ok) {
        const int parts = a.tiles_m * a.tiles_t * 4;
        const int part = (mt * a.tiles_t + 2 * t2 + h) * 4 + hw;
        double* dst = a.ostats + ((size_t)n * parts + part) * 2;
        dst[0] = s;
        dst[1] = q;
      }
    }
  };

  // ---- prologue: activations of steps 0 .. D-1, weights of steps 0 and 1, bias, residual of the first tile --------------
  x_setup();
  a_setup();
#pragma unroll
  for (int j = 0; j < D; ++j) {
    dma_x(j);
    x_advance();
  }
  dma_a(0);
  a_advance();
  dma_a(1);
  a_advance();
  if (wave == 0) bias_dma(n_lo, mt_lo, 0);
  init_acc(lo);
  acc_fence();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // tables, raw step 0 and the first weight slots are in LDS
  stage_b(0);
  stab += XB_K;  // (step 1 of the first supertile: S >= 2)
  sk = 1;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (h == 1) {  // half 1 runs one phase behind
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

#ifdef PS_PP_STAMPS
  unsigned long long st_c = 0, st_s = 0, st_b = 0, st_d = 0, st_t0 = 0;
  if (a.stamps) {
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0)::"memory");
    st_prev = st_t0;
  }
#endif

  // ---- main loop: [DMA issue] compute(g) | barrier X | [drain] stage(g+1) | barrier Y -------------------------------------
  // Counted waits and drains: a drain puts >= 128 operations behind everything issued before it, so whatever was
  // issued before a drain has landed once the drain's operations have been ISSUED (at most 63 stay in flight) and
  // needs no wait; counting through it would wait for the drain's own traffic.  `sd` = iterations since the last
  // drain; only operations issued after that drain are waited for, which gives its stores (and residual loads) two
  // iterations in the background before anything queues behind them.
  int ci = lo, ck = 0;  // compute: supertile, step
  int parity = 0;       // bias-table parity of ci
  int rx = 0;           // g % D: raw ring slot refilled this iteration (step g + D); step g + 1 is read from slot rx + 1
  int ra = 0;           // g % NA: weight ring slot of step g
  int sd = 1000;
  for (int g = 0; g < total; ++g) {
    int sl = ra + 2;  // weight ring slot of step g + 2 (its previous tenant, step g - 1, was last read a phase ago)
    sl = sl >= NA ? sl - NA : sl;
    compute(ra, [&](auto blk_c) {
      constexpr int blk = decltype(blk_c)::value;
      if constexpr (blk == 0) dma_x(rx);
      if constexpr (blk == 1) dma_a(sl);
      if constexpr (blk == 2) {
        x_advance();
        a_advance();
      }
    });
    PP_STAMP(st_m)
    // before barrier X: this wave's pieces of the raw activations of step g+1 (issued in iteration g-2) have landed;
    // half 1 also its share of the weights of step g+1 (issued in iteration g-1; read by half 0 right after barrier Y,
    // which half 1 meets straight from its MFMAs)
    // (`sd` still holds last iteration's count here: 0 = the drain ran in iteration g-1, 1 = in iteration g-2, both
    //  AFTER the operations waited for were issued; a drain in g-3 or earlier lies before them and hides nothing)
    if (h == 0) {
      if (sd <= 1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W_X) : "memory");
    } else {
      if (sd == 0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W_A) : "memory");
    }
    PP_STAMP(st_c)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PP_STAMP(st_b)
    __builtin_amdgcn_s_setprio(3);
    const int rs = rx + 1 == D ? 0 : rx + 1;
    ++sd;
    if (ck == S - 1) {
      drain(ci, parity);
      parity ^= 1;
      sd = 0;
      PP_STAMP(st_d)
      stage_b(rs);
      acc_fence();  // the wait for the residual loads belongs on THIS path, not in front of every iteration's MFMAs
    } else {
      stage_b(rs);
    }
    s_advance();
    if (++ck == S) {
      ck = 0;
      ++ci;
    }
    rx = rs;
    ra = ra + 1 == NA ? 0 : ra + 1;
    // before barrier Y: half 0's share of the weights of step g+1 (issued in iteration g-1) has landed (sd is up to
    // date here: 0 = drained just now, 1 = in iteration g-1)
    if (h == 0 && sd >= 2)
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(W_A) : "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(0);
    PP_STAMP(st_s)
    if (h == 0 || g + 1 < total) {  // half 1 started one barrier late
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    PP_STAMP(st_b)
  }
#ifdef PS_PP_STAMPS
  if (a.stamps && (tid & 255) == 0) {
    unsigned long long now;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
    unsigned long long* d = a.stamps + ((size_t)blockIdx.x * 2 + h) * 6;
    d[0] = now - st_t0;
    d[1] = st_c + (st_f << 20) + (st_m << 42);  // < 2^20 / 2^22 / 2^22 ticks each
    d[2] = st_s;
    d[3] = st_b;
    d[4] = st_d;
    d[5] = (unsigned long long)total;
  }
#endif
}

// ---- single-wave-per-SIMD variant (experiment, ps_debug_flags bit 30) ------------------------------------------------
// One 256-thread workgroup per CU, one wave per SIMD with the whole 512-register file: the fragments of K-step g+1 are
// read from LDS while the MFMAs of step g issue (two fragment sets in registers), the activations of step g+2 are
// split in the same stream, and there is ONE barrier per K-step.  Same rings, same counted waits as the ping-pong
// kernel; tiles are walked one at a time.  State: parity-green, 5-10 % slower than the ping-pong kernel per launch
// (220 / 123 / 255 us against 204 / 107 / 244 for in / pointwise / out at 32 utterances) with the tile drain still at
// the end of the tile; what it is for is the register budget to hold a finished tile (a second accumulator set) and
// spread its stores over the next tile's K-steps, which the 256-register waves of the ping-pong kernel cannot.
// (Every lambda is force-inlined: called from the two copies of the loop body, hipcc otherwise keeps the captured
// accumulators and fragment sets in scratch memory -- 9 x slower.)
template <int PLANES>
struct SoloLds {
  static constexpr int A_SLOT = PLANES * XB_M * XB_K * 2;
  static constexpr int B_SLOT = PLANES * XB_T * XB_K * 2;
  static constexpr int NA = 3, D = 3;
  static constexpr int RAW = XB_K * XB_T * 4;
  static constexpr int OFF_RAW = NA * A_SLOT;
  static constexpr int OFF_B = OFF_RAW + D * RAW;
  static constexpr int OFF_TAB = OFF_B + 2 * B_SLOT;
  static constexpr int OFF_BIAS = OFF_TAB + PP_MAXU * 1024 * 4;
  static constexpr int TOTAL = OFF_BIAS + 4 * XB_M * 4;  // PLANES = 3: 132 KiB
};

template <int PLANES, bool TR, bool STATS, bool RES>
__global__ __launch_bounds__(256, 1) void conv1x1_bf16_solo_kernel(BfArgs a) {
  using L = SoloLds<PLANES>;
  constexpr int P = 2 + 2 * PLANES;  // DMA operations per wave per K-step: 2 activation pieces, 2 * PLANES weight pieces
  __shared__ __attribute__((aligned(16))) unsigned char smem[L::TOTAL];
  float* tab = reinterpret_cast<float*>(smem + L::OFF_TAB);
  float* bias_lds = reinterpret_cast<float*>(smem + L::OFF_BIAS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wt = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  const int S = a.ksteps;  // even (launcher)
  const int ntiles = a.tiles_t * a.tiles_m * a.N;
  const int G = gridDim.x;
  const int lo = (int)((long long)blockIdx.x * ntiles / G), hi = (int)((long long)(blockIdx.x + 1) * ntiles / G);
  if (lo >= hi) return;
  const int total = (hi - lo) * S;
  auto decode = [&](int idx, int& n, int& mt, int& tt) __attribute__((always_inline)) {
    tt = idx % a.tiles_t;
    const int r = idx / a.tiles_t;
    mt = r % a.tiles_m;
    n = r / a.tiles_m;
  };
  int n_lo, mt_lo, tt_lo;
  decode(lo, n_lo, mt_lo, tt_lo);

  const float slope = (TR && a.pro.prelu) ? a.pro.slope[0] : 1.f;
  const bool plain_tr = !a.pro.pre_relu && !a.pro.post_tanh;
  if constexpr (TR) {
    const bool has_norm = a.pro.norm != PS_NORM_NONE;
    const int n_hi = ((hi - 1) / a.tiles_t) / a.tiles_m;
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
      for (int k = tid; k < 512; k += 256) {
        float sc = 0.f, sh = 0.f;
        if (k < a.K) {
          sc = has_norm ? a.pro.gamma[k] * rstd : 1.f;
          sh = has_norm ? a.pro.beta[k] - mean * sc : 0.f;
        }
        tab[u * 1024 + k] = sc;
        tab[u * 1024 + 512 + k] = sh;
      }
    }
  }

  // ---- activation DMA -------------------------------------------------------------------------------------------------
  const int x_voff = ((lane >> 5) * a.ldt + (lane & 31) * 4) * 4;
  const int x_step = XB_K * a.ldt * 4, x_half = 8 * a.ldt * 4;
  float* const raw_w = reinterpret_cast<float*>(smem + L::OFF_RAW) + wave * 256;
  int xi = lo, xk = 0, xso = 0;
  __amdgpu_buffer_rsrc_t xr;
  auto x_setup = [&]() __attribute__((always_inline)) {
    int n, mt, tt;
    decode(xi, n, mt, tt);
    xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (size_t)n * a.K * a.ldt, 0,
                                           xi < hi ? a.K * a.ldt * 4 : 0, 0x00020000);
    xso = (wave * 2 * a.ldt + tt * XB_T) * 4;
  };
  auto dma_x = [&](int slot) __attribute__((always_inline)) {
    float* dst = raw_w + slot * (L::RAW / 4);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, x_voff, xso, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst + 1024, 16, x_voff, xso + x_half, 0, 0);
  };
  auto x_advance = [&]() __attribute__((always_inline)) {
    xso += x_step;
    if (++xk == S) {
      xk = 0;
      ++xi;
      x_setup();
    }
  };
  // ---- weight DMA -----------------------------------------------------------------------------------------------------
  const int w_voff = lane * 16;
  const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned short*>(a.wt), 0, a.tiles_m * S * L::A_SLOT, 0x00020000);
  const int a_share = wave * 64 * 16;
  int ai = lo, ak = 0, aso = 0;
  auto a_setup = [&]() __attribute__((always_inline)) {
    int n, mt, tt;
    decode(ai, n, mt, tt);
    aso = ai < hi ? mt * S * L::A_SLOT + a_share : 0x7f000000;
  };
  auto dma_a = [&](int slot) __attribute__((always_inline)) {
    float* dst = reinterpret_cast<float*>(smem + slot * L::A_SLOT + a_share);
#pragma unroll
    for (int i = 0; i < 2 * PLANES; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, dst + i * 1024, 16, w_voff, aso + i * 4096, 0, 0);
  };
  auto a_advance = [&]() __attribute__((always_inline)) {
    aso += L::A_SLOT;
    if (++ak == S) {
      ak = 0;
      ++ai;
      a_setup();
    }
  };
  auto bias_dma = [&](int n, int mt, int par) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias ? a.bias : a.x), 0, a.bias ? a.M * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t bnr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias_n ? a.bias_n + (size_t)n * a.M : a.x), 0, a.bias_n ? a.M * 4 : 0, 0x00020000);
    float* dst = bias_lds + par * 2 * XB_M;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(br, dst, 16, w_voff, mt * XB_M * 4, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(bnr, dst + XB_M, 16, w_voff, mt * XB_M * 4, 0, 0);
  };

  // ---- staging --------------------------------------------------------------------------------------------------------
  const int bt = tid & 127, bh = tid >> 7;
  const float* const raw_r = reinterpret_cast<const float*>(smem + L::OFF_RAW) + 8 * bh * XB_T + bt;
  unsigned char* const bx_w = smem + L::OFF_B + (bt * XB_K + 8 * bh) * 2;
  int si = lo, sk = 0, spar = 0;
  const float* stab = tab + 8 * bh;
  auto s_setup = [&]() __attribute__((always_inline)) {
    int n, mt, tt;
    decode(si, n, mt, tt);
    stab = tab + (n - n_lo) * 1024 + 8 * bh;
    if (wave == 0 && si < hi) bias_dma(n, mt, spar);
  };
  auto stage_b = [&](int rslot, int bslot) __attribute__((always_inline)) {
    if (si >= hi) return;
    const float* raw = raw_r + rslot * (L::RAW / 4);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = raw[j * XB_T];
    if constexpr (TR) {
      const f32x4 sc0 = *reinterpret_cast<const f32x4*>(stab), sc1 = *reinterpret_cast<const f32x4*>(stab + 4);
      const f32x4 sh0 = *reinterpret_cast<const f32x4*>(stab + 512), sh1 = *reinterpret_cast<const f32x4*>(stab + 516);
      if (plain_tr) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = prelu(v[j] * (j < 4 ? sc0[j & 3] : sc1[j & 3]) + (j < 4 ? sh0[j & 3] : sh1[j & 3]), slope);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float u = v[j];
          if (a.pro.pre_relu) u = relu_keep_nan(u);
          u = u * (j < 4 ? sc0[j & 3] : sc1[j & 3]) + (j < 4 ? sh0[j & 3] : sh1[j & 3]);
          u = prelu(u, slope);
          if (a.pro.post_tanh) u = tanhf(u);
          v[j] = u;
        }
      }
    }
    unsigned char* dstb = bx_w + bslot * L::B_SLOT;
#pragma unroll
    for (int p = 0; p < PLANES; ++p) {
      bf16x8 piece;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const bf16x2 hh = __builtin_convertvector(f32x2v{v[j], v[j + 1]}, bf16x2);
        piece[j] = hh[0];
        piece[j + 1] = hh[1];
        if (p + 1 < PLANES) {
          const f32x2v back = __builtin_convertvector(hh, f32x2v);
          v[j] -= back[0];
          v[j + 1] -= back[1];
        }
      }
      *reinterpret_cast<bf16x8*>(dstb + p * XB_T * XB_K * 2) = piece;
    }
  };
  auto s_advance = [&]() __attribute__((always_inline)) {
    stab += XB_K;
    if (++sk == S) {
      sk = 0;
      ++si;
      spar ^= 1;
      s_setup();
    }
  };

  // ---- compute side ---------------------------------------------------------------------------------------------------
  f32x16 acc[4][2];
  const int lane_off = (4 * lh * a.ldt + lr) * 4;
  auto init_acc = [&](int idx) __attribute__((always_inline)) {
    if constexpr (RES) {
      int n, mt, tt;
      decode(idx, n, mt, tt);
      const int slab = a.M * a.ldt * 4;
      const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(a.res) + (size_t)n * a.M * a.ldt, 0, idx < hi ? slab : 0, 0x00020000);
      const int tile_off = ((mt * XB_M + wm * 128) * a.ldt + tt * XB_T + wt * 64) * 4;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rc = mi * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
            acc[mi][ti][r] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rr, lane_off, tile_off + rc * a.ldt * 4 + ti * 128, 0));
        }
    } else {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mi][ti][r] = 0.f;
    }
  };
  auto acc_fence = [&]() __attribute__((always_inline)) {
    asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[2][0]),
                 "+v"(acc[2][1]), "+v"(acc[3][0]), "+v"(acc[3][1]));
  };
  const unsigned char* const a_frag = smem + ((wm * 128 + lr) * XB_K + 8 * lh) * 2;
  const unsigned char* const b_frag = smem + L::OFF_B + ((wt * 64 + lr) * XB_K + 8 * lh) * 2;
  bf16x8 fb[2][PLANES][2], fa[2][4][PLANES];  // two fragment sets: the one the MFMAs read, the one being prefetched
  auto read_frags = [&](auto q_c, int aslot, int bslot) __attribute__((always_inline)) {
    constexpr int q = decltype(q_c)::value;
    const unsigned char* sa = a_frag + aslot * L::A_SLOT;
    const unsigned char* sb = b_frag + bslot * L::B_SLOT;
#pragma unroll
    for (int p = 0; p < PLANES; ++p)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) fb[q][p][ti] = *reinterpret_cast<const bf16x8*>(sb + (p * XB_T + ti * 32) * XB_K * 2);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int p = 0; p < PLANES; ++p)
        fa[q][mi][p] = *reinterpret_cast<const bf16x8*>(sa + ((p * XB_M + mi * 32) * XB_K) * 2);
  };
  auto mfma_block = [&](auto q_c, auto mi_c) __attribute__((always_inline)) {
    constexpr int q = decltype(q_c)::value, mi = decltype(mi_c)::value;
    if constexpr (PLANES == 3) {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q][mi][1], fb[q][1][ti], acc[mi][ti], 0, 0, 0);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q][mi][2], fb[q][0][ti], acc[mi][ti], 0, 0, 0);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q][mi][0], fb[q][2][ti], acc[mi][ti], 0, 0, 0);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q][mi][1], fb[q][0][ti], acc[mi][ti], 0, 0, 0);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q][mi][0], fb[q][1][ti], acc[mi][ti], 0, 0, 0);
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[q][mi][0], fb[q][0][ti], acc[mi][ti], 0, 0, 0);
  };

  auto drain = [&](int idx, int parity) __attribute__((always_inline)) {
    int n, mt, tt;
    decode(idx, n, mt, tt);
    const int m0 = mt * XB_M, t0 = tt * XB_T;
    float fsum = 0.f, fsq = 0.f;
    const int slab = a.M * a.ldt * 4;
    const __amdgpu_buffer_rsrc_t yr =
        __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)n * a.M * a.ldt, 0, slab, 0x00020000);
    const int tile_off = ((m0 + wm * 128) * a.ldt + t0 + wt * 64) * 4;
    int nn, nmt, ntt;
    decode(idx + 1, nn, nmt, ntt);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(RES ? a.res : a.y) + (size_t)nn * a.M * a.ldt, 0, (RES && idx + 1 < hi) ? slab : 0, 0x00020000);
    [[maybe_unused]] const int ntile_off = ((nmt * XB_M + wm * 128) * a.ldt + ntt * XB_T + wt * 64) * 4;
    bool cm[2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) cm[ti] = t0 + wt * 64 + ti * 32 + lr < a.T;
    const float* bl = bias_lds + parity * 2 * XB_M + wm * 128 + 4 * lh;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bl + mi * 32 + 8 * rq) +
                         *reinterpret_cast<const f32x4*>(bl + XB_M + mi * 32 + 8 * rq);
#pragma unroll
        for (int r3 = 0; r3 < 4; ++r3) {
          const int r = rq * 4 + r3;
          const int rc = mi * 32 + r3 + 8 * rq;
#pragma unroll
          for (int ti = 0; ti < 2; ++ti) {
            const float v = acc[mi][ti][r] + b4[r3];
            if constexpr (STATS) {
              const float vm = cm[ti] ? v : 0.f;
              fsum += vm;
              fsq += vm * vm;
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, lane_off,
                                                  tile_off + rc * a.ldt * 4 + ti * 128, 0);
            if constexpr (RES)
              acc[mi][ti][r] = __builtin_bit_cast(
                  float, __builtin_amdgcn_raw_buffer_load_b32(rr, lane_off, ntile_off + rc * a.ldt * 4 + ti * 128, 0));
            else
              acc[mi][ti][r] = 0.f;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (STATS) {
      const double s = wave_sum_dpp(fsum), q = wave_sum_dpp(fsq);
      if (lane == 0) {
        const int parts = a.tiles_m * a.tiles_t * 4;
        double* dst = a.ostats + ((size_t)n * parts + (mt * a.tiles_t + tt) * 4 + wave) * 2;
        dst[0] = s;
        dst[1] = q;
      }
    }
  };

  // ---- prologue -------------------------------------------------------------------------------------------------------
  x_setup();
  a_setup();
  for (int j = 0; j < 3; ++j) {
    dma_x(j);
    x_advance();
    dma_a(j);
    a_advance();
  }
  if (wave == 0) bias_dma(n_lo, mt_lo, 0);
  init_acc(lo);
  acc_fence();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  stage_b(0, 0);  // x(0) -> Bx[0]
  s_advance();
  stage_b(1, 1);  // x(1) -> Bx[1]
  s_advance();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  dma_x(0);  // x(3) -> raw slot 0 (x(0) has been staged)
  x_advance();
  read_frags(bic<0>{}, 0, 0);

  int ci = lo, ck = 0, parity = 0;
  int r3 = 0;  // g % 3
  auto body = [&](int g, auto q_c) __attribute__((always_inline)) {
    constexpr int q = decltype(q_c)::value;
    const int r1 = r3 + 1 == 3 ? 0 : r3 + 1, r2 = r1 + 1 == 3 ? 0 : r1 + 1;
    read_frags(bic<q ^ 1>{}, r1, (g + 1) & 1);  // fragments of step g+1
    mfma_block(q_c, bic<0>{});
    __builtin_amdgcn_sched_barrier(0);
    dma_x(r1);  // x(g+4) -> raw slot (g+4) % 3
    x_advance();
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(q_c, bic<1>{});
    __builtin_amdgcn_sched_barrier(0);
    dma_a(r3);  // A(g+3) -> slot g % 3 (its fragments are in registers)
    a_advance();
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(q_c, bic<2>{});
    mfma_block(q_c, bic<3>{});
    if (ck == S - 1) {
      drain(ci, parity);
      parity ^= 1;
      stage_b(r2, g & 1);  // x(g+2) -> Bx[g & 1]
      acc_fence();
    } else {
      stage_b(r2, g & 1);
    }
    s_advance();
    if (++ck == S) {
      ck = 0;
      ++ci;
    }
    r3 = r1;
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(P) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  for (int g = 0; g < total; g += 2) {
    body(g, bic<0>{});
    body(g + 1, bic<1>{});
  }
}

#include "conv1x1_bf16_il.inc"
#include "conv1x1_f16x2_w1.inc"
#include "conv1x1_f16x2_rb.inc"

static int bf16_cus() { return device_cus(); }

// Can this launch run on the register-B kernel (conv1x1_f16x2_rb.inc)?  K a multiple of 32, M a multiple of 256, a
// scale / shift + PReLU prologue at most, enough tiles for the chip and a workgroup's run inside PP_MAXU utterances.
// Gr receives the grid.
static bool rb_ok(const BfArgs& a, int N, int* Gr_out) {
  if (a.K % 32 != 0 || a.M % 256 != 0 || a.ksteps < 4 || a.pro.pre_relu || a.pro.post_tanh) return false;
  const int cus = bf16_cus();
  const long long ntiles = (long long)a.tiles_t * a.tiles_m * N;
  const long long st_per = (a.tiles_t + 1) / 2, nsuper = st_per * a.tiles_m * N;
  if (!(2 * nsuper >= cus || (g_debug_flags & (1 << 28)))) return false;
  const int Gr = (int)(ntiles < 2 * cus ? ntiles : 2 * cus);
  const long long pw = (ntiles + Gr - 1) / Gr, pu = (long long)a.tiles_t * a.tiles_m;
  if ((pw + pu - 2) / pu + 1 > PP_MAXU) return false;
  if (Gr_out) *Gr_out = Gr;
  return true;
}

template <bool R16, bool FM = false, bool LNE = false>
static void rb_launch(const BfArgs& a, int N, bool tr, int Gr, hipStream_t stream) {
  const bool stats = a.ostats != nullptr, res = a.res != nullptr;
  BfArgs& b = const_cast<BfArgs&>(a);
  b.groups = Gr / 2;
  b.delay = 0;
  // two m-tiles: workgroups 8 apart share a run of frame tiles, one m-tile each (see the kernel); needs an even grid
  // whose halves get whole, equal runs
  b.pair_r = ((a.tiles_m == 2 || a.tiles_m == 4) && Gr % (8 * a.tiles_m) == 0 && !(g_debug_flags & 64)) ? 1 : 0;
  if (b.pair_r) {  // (a pair's / quad's run of frame tiles must stay within PP_MAXU utterances too)
    const int gp = Gr / a.tiles_m;
    const long long fr = (long long)a.tiles_t * N, pw2 = (fr + gp - 1) / gp;
    if ((pw2 + a.tiles_t - 2) / a.tiles_t + 1 > PP_MAXU) b.pair_r = 0;
  }
#define PS_RB(TRV, STV, RSV) \
  hipLaunchKernelGGL((conv1x1_f16x2_rb_kernel<TRV, STV, RSV, R16>), dim3(Gr, 1), dim3(256), 0, stream, a)
  if constexpr (FM) {  // (split_gemm admits no prologue, residual or statistics here)
    hipLaunchKernelGGL((conv1x1_f16x2_rb_kernel<false, false, false, false, true>), dim3(Gr, 1), dim3(256), 0, stream, a);
    return;
  }
  if constexpr (LNE) {
    if (tr)
      hipLaunchKernelGGL((conv1x1_f16x2_rb_kernel<true, false, false, false, false, true>), dim3(Gr, 1), dim3(256), 0, stream, a);
    else
      hipLaunchKernelGGL((conv1x1_f16x2_rb_kernel<false, false, false, false, false, true>), dim3(Gr, 1), dim3(256), 0, stream, a);
    return;
  }
  if (tr) {
    if (stats) PS_RB(true, true, false);
    else if (res) PS_RB(true, false, true);
    else PS_RB(true, false, false);
  } else {
    if (stats) PS_RB(false, true, false);
    else if (res) PS_RB(false, false, true);
    else PS_RB(false, false, false);
  }
#undef PS_RB
}

template <int PLANES, bool XB = false, bool YB = false>
static void bf16_launch(const BfArgs& a, int N, bool tr, hipStream_t stream) {
  const bool stats = a.ostats != nullptr, res = a.res != nullptr;
  // The ping-pong kernel needs enough supertiles to give every CU work, K-steps to pipeline over, and a bounded
  // utterance span per workgroup (prologue tables in LDS); everything else takes the simple kernel.  ps_debug_flags
  // bit 27 forces the simple kernel (tests run both).
  const int st_per = (a.tiles_t + 1) / 2;
  const long long nsuper = (long long)st_per * a.tiles_m * N;
  const int cus = bf16_cus();
  int G = (int)(nsuper < cus ? nsuper : cus);
  const int cap = (g_debug_flags >> 8) & 0xfff;  // grid cap (tests, experiments)
  if (cap && G > cap) G = cap;
  const long long per_wg = (nsuper + G - 1) / G, per_utt = (long long)st_per * a.tiles_m;
  const bool big = 2 * nsuper >= cus || (g_debug_flags & (1 << 28));  // bit 28: ping-pong kernel at any size (tests)
  const bool pp = !(g_debug_flags & (1 << 27)) && a.ksteps >= 4 && big && (per_wg + per_utt - 2) / per_utt + 1 <= PP_MAXU;
  const long long ntiles_all = (long long)a.tiles_t * a.tiles_m * N;
  const int Gs = (int)(ntiles_all < cus ? ntiles_all : cus);
  const long long per_wg_s = (ntiles_all + Gs - 1) / Gs, per_utt_s = (long long)a.tiles_t * a.tiles_m;
  const bool solo = PLANES != 2 && !XB && !YB && (g_debug_flags & (1 << 30)) && a.ksteps >= 4 && a.ksteps % 2 == 0 && ntiles_all >= cus &&
                    (per_wg_s + per_utt_s - 2) / per_utt_s + 1 <= PP_MAXU;
  if constexpr (PLANES != 2) if (solo) {
#define PS_SOLO(TRV, STV, RSV) \
  hipLaunchKernelGGL((conv1x1_bf16_solo_kernel<PLANES, TRV, STV, RSV>), dim3(Gs, 1), dim3(256), 0, stream, a)
    if (tr) {
      if (stats) PS_SOLO(true, true, false);
      else if (res) PS_SOLO(true, false, true);
      else PS_SOLO(true, false, false);
    } else {
      if (stats) PS_SOLO(false, true, false);
      else if (res) PS_SOLO(false, false, true);
      else PS_SOLO(false, false, false);
    }
#undef PS_SOLO
    return;
  }
  // bit 5: keep the two-barrier ping-pong kernel (tests run both).  The interleaved kernel's chunked prologue is
  // scale / shift + PReLU only.
  const bool plain_tr = !a.pro.pre_relu && !a.pro.post_tanh;
  if (pp && plain_tr && !(g_debug_flags & 32)) {
    // out_conv at a full batch: every CU reaches a tile's drain (128 KiB of residual in, 128 KiB out per half) at the
    // same time and the drains run at what HBM gives the whole chip at once (39 k cycles); four phases of workgroups
    // 16 k cycles apart shorten them by more than the offset costs at the end of the launch (248 -> 221 us at 32
    // utterances; the in / pointwise drains are issue-bound and gain nothing)
    const_cast<BfArgs&>(a).delay = (PLANES >= 2 && res && per_wg >= 4) ? 16000 : 0;
    const_cast<BfArgs&>(a).groups = 4;
#ifdef PS_TUNE  // tuning builds only: phase offset / phase count from the environment (res / stats / plain launches)
    {
      const char* key = res ? "PS_IL_DELAY_RES" : stats ? "PS_IL_DELAY_STATS" : "PS_IL_DELAY_PLAIN";
      if (const char* v = getenv(key)) const_cast<BfArgs&>(a).delay = atoi(v);
      if (const char* v = getenv("PS_IL_GROUPS")) const_cast<BfArgs&>(a).groups = atoi(v);
    }
#endif
    // out_conv (two m-tiles) reads every activation tile twice, once per m-tile.  Workgroups go to the XCDs round robin
    // (blockIdx & 7), each XCD has its own L2: with the plain numbering the two readers of a tile sit on different XCDs
    // and both reads come from HBM.  Renumbered so that they are 8 apart -- same XCD, dispatched together, running the
    // same K-steps at the same time -- the second read is an L2 hit.
    const int per = (int)(nsuper / (G > 0 ? G : 1));
    const bool pairable = G == 256 && a.tiles_m == 2 && nsuper % 256 == 0 && per > 0 && st_per % per == 0 &&
                          !(g_debug_flags & 64);
    const_cast<BfArgs&>(a).pair_r = pairable ? st_per / per : 0;
    // fp16x2 on fp32 rows: ps_debug_flags bit 7 selects the one-wave-per-SIMD kernel (conv1x1_f16x2_w1.inc; tests run
    // both).  Measured in the same process on the benchmark's step: 126-130 us per launch against 124 for the interleaved
    // kernel -- two unrelated schedules land within 3 % of each other (profiles/r03_gemm_two_designs.txt).
    if constexpr (PLANES == 2 && !XB && !YB) if (g_debug_flags & 128) {
#define PS_W1(TRV, STV, RSV) \
  hipLaunchKernelGGL((conv1x1_f16x2_w1_kernel<TRV, STV, RSV>), dim3(G, 1), dim3(256), 0, stream, a)
      const_cast<BfArgs&>(a).delay = 0;
      if (tr) {
        if (stats) PS_W1(true, true, false);
        else if (res) PS_W1(true, false, true);
        else PS_W1(true, false, false);
      } else {
        if (stats) PS_W1(false, true, false);
        else if (res) PS_W1(false, false, true);
        else PS_W1(false, false, false);
      }
#undef PS_W1
      return;
    }
    // fp16x2 on fp32 rows, K a multiple of 32, M a multiple of 256: the register-B kernel (conv1x1_f16x2_rb.inc) --
    // activations go from HBM to the MFMA operand registers without touching LDS, v_mfma_f32_16x16x32_f16, two workgroups
    // per CU.  Same-box comparison in the benchmark's step (profiles/r04_gemm_kernels_same_box.txt), out / in / pointwise:
    // 176.0 / 108.0 / 70.7 us against 179.0 / 111.7 / 76.6 for the interleaved kernel.  ps_debug_flags bit 22 keeps the
    // interleaved kernel (tests run both).
    if constexpr (PLANES == 2 && !XB && !YB) if (!(g_debug_flags & ((1 << 22) | 128))) {
      int Gr = 0;
      if (rb_ok(a, N, &Gr)) {
        rb_launch<false>(a, N, tr, Gr, stream);
        return;
      }
    }
#define PS_IL(TRV, STV, RSV) \
  hipLaunchKernelGGL((conv1x1_bf16_il_kernel<PLANES, TRV, STV, RSV, XB, YB>), dim3(G, 1), dim3(512), 0, stream, a)
    if (tr) {
      if (stats) PS_IL(true, true, false);
      else if (res) PS_IL(true, false, true);
      else PS_IL(true, false, false);
    } else {
      if (stats) PS_IL(false, true, false);
      else if (res) PS_IL(false, false, true);
      else PS_IL(false, false, false);
    }
#undef PS_IL
    return;
  }
  if constexpr (PLANES != 2) if (pp && !XB && !YB) {  // (bf16 rows, fp16 terms: the interleaved kernel or the simple one)
#define PS_PP(TRV, STV, RSV) \
  hipLaunchKernelGGL((conv1x1_bf16_pp_kernel<PLANES, TRV, STV, RSV>), dim3(G, 1), dim3(512), 0, stream, a)
    if (tr) {
      if (stats) PS_PP(true, true, false);
      else if (res) PS_PP(true, false, true);
      else PS_PP(true, false, false);
    } else {
      if (stats) PS_PP(false, true, false);
      else if (res) PS_PP(false, false, true);
      else PS_PP(false, false, false);
    }
#undef PS_PP
    return;
  }
  // launches that cannot fill the chip with 256 x 128 tiles (one or a few utterances) take 256 x 32 tiles: four times
  // the workgroups, a quarter of the staging and MFMA work each (ps_debug_flags bit 29 keeps the wide tile)
  const bool narrow = 2 * (long long)a.tiles_t * a.tiles_m * N <= cus && !(g_debug_flags & (1 << 29));
  dim3 grid(narrow ? a.tiles_t * 4 : a.tiles_t, a.tiles_m, N);
#define PS_BF(TRV, STV, RSV)                                                                                          \
  do {                                                                                                                \
    if (narrow)                                                                                                       \
      hipLaunchKernelGGL((conv1x1_bf16_kernel<PLANES, TRV, STV, RSV, 32, XB, YB>), grid, dim3(256), 0, stream, a);    \
    else                                                                                                              \
      hipLaunchKernelGGL((conv1x1_bf16_kernel<PLANES, TRV, STV, RSV, XB_T, XB, YB>), grid, dim3(256), 0, stream, a);  \
  } while (0)
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
  if (M <= 0 || K <= 0 || planes < 1 || planes > 3) return 0;
  const size_t tiles_m = (M + XB_M - 1) / XB_M, ksteps = (K + XB_K - 1) / XB_K;
  return tiles_m * ksteps * planes * XB_M * XB_K * 2;
}

extern "C" int ps_conv1x1_bf16_f32(const float* x, const void* wt_planes, float* y, int N, int K, int M, int T, int ldt,
                                   int planes, const ps_prologue* pro, const float* bias, const float* bias_n,
                                   const float* res, double* ostats, void* stream) {
  return ps_conv1x1_bf16_io(x, 0, wt_planes, y, 0, N, K, M, T, ldt, planes, pro, bias, bias_n, res, ostats, stream);
}

static int split_gemm(const void* x_any, int x_bf16, const void* wt_planes, const ps_f16x2_range* rng, void* y_any,
                      int y_bf16, int N, int K, int M, int T, int ldt, int planes, const ps_prologue* pro,
                      const float* bias, const float* bias_n, const float* res, double* ostats, void* stream,
                      int fm_ld = 0);

extern "C" int ps_conv1x1_bf16_io(const void* x_any, int x_bf16, const void* wt_planes, void* y_any, int y_bf16, int N,
                                  int K, int M, int T, int ldt, int planes, const ps_prologue* pro, const float* bias,
                                  const float* bias_n, const float* res, double* ostats, void* stream) {
  if (planes != 1 && planes != 3) {
    set_error("ps_conv1x1_bf16_f32: planes must be 1 (bf16 products) or 3 (fp32-accurate 3-way split), got %d", planes);
    return PS_E_INVALID;
  }
  return split_gemm(x_any, x_bf16, wt_planes, nullptr, y_any, y_bf16, N, K, M, T, ldt, planes, pro, bias, bias_n, res,
                    ostats, stream);
}

extern "C" int ps_conv1x1_f16x2_f32(const float* x, const void* wt_planes, const ps_f16x2_range* rng, float* y, int N,
                                    int K, int M, int T, int ldt, const ps_prologue* pro, const float* bias,
                                    const float* bias_n, const float* res, double* ostats, void* stream) {
  if (!rng || rng->w_exp < -100 || rng->w_exp > 100 || rng->x_bound < 0.f || (rng->x_amax && rng->x_amax_parts <= 0)) {
    set_error("ps_conv1x1_f16x2_f32: range descriptor missing or out of range (w_exp within +-100, x_bound >= 0)");
    return PS_E_INVALID;
  }
  return split_gemm(x, 0, wt_planes, rng, y, 0, N, K, M, T, ldt, 2, pro, bias, bias_n, res, ostats, stream);
}

extern "C" int ps_conv1x1_f16x2_fmajor_ok(int N, int K, int M, int T, int ldt, int ldm) {
  return ps_conv1x1_f16_rows_ok(N, K, M, T) && ldm >= M && ldm % 4 == 0 && (long long)ldm * ldt * 4 < (1LL << 31);
}

extern "C" int ps_conv1x1_f16x2_fmajor_f32(const float* x, const void* wt_planes, const ps_f16x2_range* rng, float* y, int N,
                                           int K, int M, int T, int ldt, int ldm, const float* bias, void* stream) {
  if (!rng || rng->w_exp < -100 || rng->w_exp > 100 || rng->x_bound < 0.f || (rng->x_amax && rng->x_amax_parts <= 0)) {
    set_error("ps_conv1x1_f16x2_fmajor_f32: range descriptor missing or out of range (w_exp within +-100, x_bound >= 0)");
    return PS_E_INVALID;
  }
  if (ldm < M || ldm % 4) {
    set_error("ps_conv1x1_f16x2_fmajor_f32: ldm=%d must be a multiple of 4 >= M=%d", ldm, M);
    return PS_E_INVALID;
  }
  return split_gemm(x, 0, wt_planes, rng, y, 0, N, K, M, T, ldt, 2, nullptr, bias, nullptr, nullptr, nullptr, stream, ldm);
}

extern "C" int ps_conv1x1_f16x2_ln_ok(int N, int K, int C, int T) { return C == 128 && ps_conv1x1_f16_rows_ok(N, K, 256, T); }

extern "C" int ps_conv1x1_f16x2_ln_f32(const float* x, const void* wt_planes, const ps_f16x2_range* rng, float* y, int N, int K,
                                       int C, int T, int ldt, const ps_prologue* pro, const float* bias, const float* gamma,
                                       const float* beta, float eps, const float* res, int res_inside, void* stream) {
  if (!rng || rng->w_exp < -100 || rng->w_exp > 100 || rng->x_bound < 0.f || (rng->x_amax && rng->x_amax_parts <= 0)) {
    set_error("ps_conv1x1_f16x2_ln_f32: range descriptor missing or out of range (w_exp within +-100, x_bound >= 0)");
    return PS_E_INVALID;
  }
  if (!x || !wt_planes || !y || !gamma || !beta || N <= 0 || K <= 0 || T <= 0 || N > 65535 || !(eps >= 0.f)) {
    set_error("ps_conv1x1_f16x2_ln_f32: null pointer or non-positive size (N=%d K=%d C=%d T=%d)", N, K, C, T);
    return PS_E_INVALID;
  }
  if (C != 128 || ldt < T || ldt % kTileT != 0 || ((uintptr_t)wt_planes & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15) ||
      (long long)128 * ldt * 4 >= (1LL << 31)) {
    set_error("ps_conv1x1_f16x2_ln_f32: C = 128 channels, ldt a multiple of %d >= T, 16-byte aligned parameters (ps_conv1x1_f16x2_ln_ok)",
              kTileT);
    return PS_E_UNSUPPORTED;
  }
  BfArgs a{};
  bool tr = false;
  if (pro) {
    a.pro = *pro;
    tr = pro->norm != PS_NORM_NONE || pro->prelu;
    if (pro->pre_relu || pro->post_tanh || pro->norm == PS_NORM_GLOBAL || (pro->norm == PS_NORM_AFFINE && (!pro->gamma || !pro->beta)) ||
        (pro->prelu && !pro->slope) || K > 512) {
      set_error("ps_conv1x1_f16x2_ln_f32: the prologue may be a per-channel affine map and / or a PReLU (a ReLU is the PReLU of slope 0)");
      return PS_E_UNSUPPORTED;
    }
  }
  a.x = x, a.wt = (const unsigned short*)wt_planes, a.y = y, a.bias = bias, a.res = res;
  a.K = K, a.M = 256, a.T = T, a.ldt = ldt, a.N = N;  // (the weight image has 256 rows, the upper 128 of them zero)
  a.ksteps = (K + XB_K - 1) / XB_K, a.tiles_t = (T + XB_T - 1) / XB_T, a.tiles_m = 1;
  a.ablate = (g_debug_flags >> 24) & 15;
  a.stamps = (unsigned long long*)g_debug_buffer;
  a.ln_gamma = gamma, a.ln_beta = beta, a.ln_eps = eps, a.ln_inside = res_inside && res ? 1 : 0;
  int x_exp = (tr && a.pro.norm != PS_NORM_NONE) ? -2 : -4;
  if (rng->x_bound > 0.f) {
    int e;
    frexpf(rng->x_bound, &e);
    x_exp = 15 - e;
    x_exp = x_exp < -100 ? -100 : (x_exp > 100 ? 100 : x_exp);
  }
  a.xscale = ldexpf(1.f, x_exp);
  a.winv = ldexpf(1.f, -rng->w_exp);
  a.x_amax = rng->x_bound > 0.f ? nullptr : rng->x_amax;
  a.x_amax_parts = rng->x_amax_parts;
  a.amax_mul = rng->amax_mul > 0.f ? rng->amax_mul : 0.f;
  a.amax_add = rng->amax_add > 0.f ? rng->amax_add : 0.f;
  a.y_amax = rng->y_amax;  // [N][ps_conv1x1_stats_parts(256, T)] partial maxima of |y| (the next GEMM's input range) or NULL
  int Gr = 0;
  if (!rb_ok(a, N, &Gr)) {
    set_error("ps_conv1x1_f16x2_ln_f32: this launch cannot run on the register-B kernel (ps_conv1x1_f16x2_ln_ok)");
    return PS_E_UNSUPPORTED;
  }
  {
    LaunchTimer timer("conv1x1_bf16", (hipStream_t)stream);
    rb_launch<false, false, true>(a, N, tr, Gr, (hipStream_t)stream);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_conv1x1_f16x2_ln_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int ps_conv1x1_f16_rows_ok(int N, int K, int M, int T) {
  if (N <= 0 || K <= 0 || M <= 0 || T <= 0) return 0;
  BfArgs a{};
  a.K = K, a.M = M, a.T = T;
  a.ksteps = (K + XB_K - 1) / XB_K;
  a.tiles_t = (T + XB_T - 1) / XB_T;
  a.tiles_m = (M + XB_M - 1) / XB_M;
  return rb_ok(a, N, nullptr) ? 1 : 0;
}

extern "C" int ps_conv1x1_f16_rows(const void* x, const void* wt_planes, const ps_f16x2_range* rng, void* y, int N, int K,
                                   int M, int T, int ldt, const ps_prologue* pro, const float* bias, const float* bias_n,
                                   const void* res, double* ostats, void* stream) {
  if (!rng || rng->w_exp < -100 || rng->w_exp > 100 || rng->x_bound < 0.f || (rng->x_amax && rng->x_amax_parts <= 0) ||
      (!(rng->x_bound > 0.f) && !rng->x_amax)) {
    set_error("ps_conv1x1_f16_rows: range descriptor missing or incomplete (w_exp within +-100 and x_bound > 0 or x_amax)");
    return PS_E_INVALID;
  }
  return split_gemm(x, 1, wt_planes, rng, y, 1, N, K, M, T, ldt, 2, pro, bias, bias_n, (const float*)res, ostats, stream);
}

// partial maxima of |x| per utterance for the fp16x2 GEMM's range: [N][PS_ABSMAX_PARTS]
constexpr int kAbsmaxParts = 64;
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int T,
                                                     int ldt) {
  const int p = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const float* base = x + (size_t)n * C * ldt;
  float m = 0.f;
  for (int c = p; c < C; c += kAbsmaxParts) {
    const float* row = base + (size_t)c * ldt;
    for (int t = tid * 4; t < T; t += 1024) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + t);  // (rows are padded to a multiple of 128 frames)
      m = fmaxf(m, fabsf(v[0]));
      if (t + 1 < T) m = fmaxf(m, fabsf(v[1]));
      if (t + 2 < T) m = fmaxf(m, fabsf(v[2]));
      if (t + 3 < T) m = fmaxf(m, fabsf(v[3]));
    }
  }
  __shared__ float red[4];
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) out[(size_t)n * kAbsmaxParts + p] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

extern "C" int ps_absmax_parts(void) { return kAbsmaxParts; }

extern "C" int ps_absmax_f32(const float* x, float* amax, int N, int C, int T, int ldt, void* stream) {
  if (!x || !amax || N <= 0 || C <= 0 || T <= 0 || N > 65535 || ldt < T || ldt % kTileT != 0) {
    set_error("ps_absmax_f32: null pointer, non-positive size or ldt=%d not a multiple of %d >= T=%d", ldt, kTileT, T);
    return PS_E_INVALID;
  }
  LaunchTimer timer("absmax", (hipStream_t)stream);
  hipLaunchKernelGGL(absmax_kernel, dim3(kAbsmaxParts, N), dim3(256), 0, (hipStream_t)stream, x, amax, C, T, ldt);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_absmax_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

static int split_gemm(const void* x_any, int x_bf16, const void* wt_planes, const ps_f16x2_range* rng, void* y_any,
                      int y_bf16, int N, int K, int M, int T, int ldt, int planes, const ps_prologue* pro,
                      const float* bias, const float* bias_n, const float* res, double* ostats, void* stream,
                      int fm_ld) {
  const bool fmajor = fm_ld > 0;
  const float* x = (const float*)x_any;
  float* y = (float*)y_any;
  const bool rows16 = planes == 2 && x_bf16 && y_bf16;  // ps_conv1x1_f16_rows
  if ((x_bf16 || y_bf16) && planes != 1 && !rows16) {
    set_error("ps_conv1x1_bf16_io: bf16 activation rows go with planes = 1 (got %d)", planes);
    return PS_E_UNSUPPORTED;
  }
  if (!x || !wt_planes || !y || N <= 0 || K <= 0 || M <= 0 || T <= 0 || N > 65535) {
    set_error("ps_conv1x1_bf16_f32: null pointer or non-positive size (N=%d K=%d M=%d T=%d)", N, K, M, T);
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
  a.x_bf16 = x_bf16 != 0;
  a.y_bf16 = y_bf16 != 0;
  {
    LaunchTimer timer("conv1x1_bf16", (hipStream_t)stream);
    hipStream_t st = (hipStream_t)stream;
    if (planes == 2) {
      // range of the activations (see the header): a host-side bound, the producer's maxima, or the defaults
      int x_exp = (tr && a.pro.norm != PS_NORM_NONE) ? -2 : -4;
      if (rng->x_bound > 0.f) {
        int e;
        frexpf(rng->x_bound, &e);  // bound < 2^e  ->  scaled below 2^15
        x_exp = 15 - e;
        x_exp = x_exp < -100 ? -100 : (x_exp > 100 ? 100 : x_exp);
      }
      a.xscale = ldexpf(1.f, x_exp);
      a.winv = ldexpf(1.f, -rng->w_exp);
      a.x_amax = rng->x_bound > 0.f ? nullptr : rng->x_amax;
      a.x_amax_parts = rng->x_amax_parts;
      a.y_amax = rng->y_amax;
      a.amax_mul = rng->amax_mul > 0.f ? rng->amax_mul : 0.f;
      a.amax_add = rng->amax_add > 0.f ? rng->amax_add : 0.f;
      if (fmajor) {
        int Gr = 0;
        a.fm_ld = fm_ld;
        if (tr || res || ostats || !rb_ok(a, N, &Gr) || fm_ld < M || fm_ld % 4 || (long long)fm_ld * ldt * 4 >= (1LL << 31)) {
          set_error("ps_conv1x1_f16x2_fmajor_f32: this launch cannot run on the register-B kernel (ps_conv1x1_f16x2_fmajor_ok)");
          return PS_E_UNSUPPORTED;
        }
        rb_launch<false, true>(a, N, false, Gr, st);
      } else if (rows16) {
        int Gr = 0;
        if (!rb_ok(a, N, &Gr)) {
          set_error("ps_conv1x1_f16_rows: this launch cannot run on the register-B kernel (ps_conv1x1_f16_rows_ok)");
          return PS_E_UNSUPPORTED;
        }
        rb_launch<true>(a, N, tr, Gr, st);
      } else {
        bf16_launch<2>(a, N, tr, st);
      }
    } else if (planes == 3)
      bf16_launch<3>(a, N, tr, st);
    else if (a.x_bf16 && a.y_bf16)
      bf16_launch<1, true, true>(a, N, tr, st);
    else if (a.x_bf16)
      bf16_launch<1, true, false>(a, N, tr, st);
    else if (a.y_bf16)
      bf16_launch<1, false, true>(a, N, tr, st);
    else
      bf16_launch<1>(a, N, tr, st);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_conv1x1_bf16_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
