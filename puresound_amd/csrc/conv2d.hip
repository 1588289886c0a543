// Conv2d / ConvTranspose2d of the U-Net family as an implicit GEMM (unet.py:100-165 of mcw519/PureSound): the tap
// matrix that ps_unfold2d_f32 would materialise (K = Cin*kf*kt rows per output frequency row, 36 GB per forward of
// ns_dpcrn_v0 at 32 x 4 s) is never written; the B operand of the MFMA is gathered straight from the input rows.
//
//   out[n][co][fo][t] = act( bias[co] + sum_k W[co][k] * B_fo[k][t] ),   k = (ci, jf, jt)
//   conv:        B_fo[k][t] = x[n][ci][fo*sf + jf*df - pf][t + jt*dt - pt]
//   transposed:  B_fo[k][t] = x[n][ci][(fo + pf - jf*df)/sf][t + pt - jt*dt]   (when divisible)       0 outside the input
//
// A workgroup owns one output frequency row fo of one utterance, 128 frames and MB*32 output channels; wave w owns
// frames [32w, 32w+32) of the tile and all its channels (MB accumulators of v_mfma_f32_32x32x2_f32), so every B element is
// needed by exactly one wave and goes from global memory to its lane without LDS.  The k -> (source row, time shift)
// table of this fo lives in LDS.  Weights come from the packed transposed layout of ps_conv1x1_f32 (L2 resident).
// Eval BatchNorm2d is folded into W / bias by the caller; the activation is the epilogue.
#include <type_traits>

#include "ps_common.h"

namespace ps {

struct Conv2dArgs {
  const float* x1;
  const float* x2;
  const float* wt;
  const float* bias;
  const float* slope;
  float* y;
  int C1, C2, Fin, T, Tin, ld, kf, kt, sf, df, dt, pf, pt, Fout, M, K, Kp, transposed, act;
  double* stats;  // conv2d_lds_kernel only: [N][parts][2] partial (sum, sum of squares) of the outputs over the valid frames
};

constexpr int C2D_MAXK = 4096;  // table entries (Cin*kf*kt rounded up to 16)
constexpr int C2D_UN = 8;       // k-pairs whose loads are in flight together

__device__ __forceinline__ float act_apply(float u, int kind, float s) {
  switch (kind) {
    case 1: return relu_keep_nan(u);
    case 2: return u >= 0.f ? u : s * u;
    case 3: return u * tanhf(u > 20.f ? u : log1pf(expf(u)));
    case 4: return 1.f / (1.f + expf(-u));
    case 5: return tanhf(u);
    default: return u;
  }
}

template <int MB>
__global__ __launch_bounds__(256) void conv2d_kernel(Conv2dArgs a) {
  // dynamic LDS, 2 * Kp ints (round 4: the static 32 KiB tables held a CU to five workgroups whatever K was)
  extern __shared__ int c2d_tab[];
  int* const tab_off = c2d_tab;            // element offset of the source row inside the utterance (bit 30: second source), -1 = zero row
  int* const tab_shift = c2d_tab + a.Kp;  // frame shift
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 31, lk = lane >> 5;
  const int t0 = blockIdx.x * 128, fo = blockIdx.y;
  const int mtiles = (a.M + 32 * MB - 1) / (32 * MB);
  const int n = blockIdx.z / mtiles, m0 = (blockIdx.z % mtiles) * 32 * MB;

  for (int k = tid; k < a.Kp; k += 256) {
    int off = -1, sh = 0;
    if (k < a.K) {
      const int jt = k % a.kt, jf = (k / a.kt) % a.kf, ci = k / (a.kt * a.kf);
      int fi;
      bool ok = true;
      if (!a.transposed) {
        fi = fo * a.sf + jf * a.df - a.pf;
        sh = jt * a.dt - a.pt;
      } else {
        const int num = fo + a.pf - jf * a.df;
        ok = num >= 0 && num % a.sf == 0;
        fi = num / a.sf;
        sh = a.pt - jt * a.dt;
      }
      if (ok && fi >= 0 && fi < a.Fin)
        off = ci < a.C1 ? (ci * a.Fin + fi) * a.ld : (((ci - a.C1) * a.Fin + fi) * a.ld) | (1 << 30);
    }
    tab_off[k] = off;
    tab_shift[k] = sh;
  }
  __syncthreads();

  const float* x1n = a.x1 + (size_t)n * a.C1 * a.Fin * a.ld;
  const float* x2n = a.x2 ? a.x2 + (size_t)n * a.C2 * a.Fin * a.ld : a.x1;
  const int tcol = t0 + 32 * w + lr;
  f32x16 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

  // Operands of C2D_UN k-pairs per batch; the loads of batch i + 1 are issued before the MFMAs of batch i (round 4: with
  // load -> wait -> MFMA in sequence the kernel ran at a third of the fp32 MFMA rate, every wave waiting out an L2 round
  // trip per batch).
  const int npairs = a.Kp / 2;
  float av[2][C2D_UN][MB], bv[2][C2D_UN];
  auto fetch = [&](int p0, auto buf_c) {
    constexpr int buf = decltype(buf_c)::value;
#pragma unroll
    for (int u = 0; u < C2D_UN; ++u) {
      const int k = 2 * (p0 + u) + lk;  // Kp is a multiple of 16 = 2 * C2D_UN: always < Kp
      const int off = tab_off[k];
      const int ti = tcol + tab_shift[k];
      const bool ok = off >= 0 && ti >= 0 && ti < a.Tin;
      const float* src = (off & (1 << 30)) ? x2n : x1n;
      const int idx = ok ? (off & ((1 << 30) - 1)) + ti : 0;
      const float v = src[idx];   // unconditional load of a valid address, masked afterwards
      bv[buf][u] = ok ? v : 0.f;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int m = m0 + 32 * mb + lr;  // the packed weight is zero padded to 256 rows per tile
        av[buf][u][mb] = a.wt[((size_t)(m >> 8) * a.Kp + k) * 256 + (m & 255)];
      }
    }
  };
  auto multiply = [&](auto buf_c) {
    constexpr int buf = decltype(buf_c)::value;
#pragma unroll
    for (int u = 0; u < C2D_UN; ++u)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[buf][u][mb], bv[buf][u], acc[mb], 0, 0, 0);
  };
  using b0 = std::integral_constant<int, 0>;
  using b1 = std::integral_constant<int, 1>;
  fetch(0, b0{});
  for (int p0 = 0; p0 < npairs; p0 += 2 * C2D_UN) {
    if (p0 + C2D_UN < npairs) fetch(p0 + C2D_UN, b1{});
    multiply(b0{});
    if (p0 + C2D_UN < npairs) {
      if (p0 + 2 * C2D_UN < npairs) fetch(p0 + 2 * C2D_UN, b0{});
      multiply(b1{});
    }
  }

  const float s = a.slope ? a.slope[0] : 0.f;
  if (tcol < a.ld) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < a.M) {
          float v = acc[mb][r] + (a.bias ? a.bias[m] : 0.f);
          v = act_apply(v, a.act, s);
          a.y[(((size_t)n * a.M + m) * a.Fout + fo) * a.ld + tcol] = tcol < a.T ? v : 0.f;
        }
      }
  }
}

// ---- round 4: weights through LDS, valid taps only ------------------------------------------------------------------------
// conv2d_kernel above reads its A operand (weights) from global memory in every wave: 4 waves x MB x 256 bytes per k-pair and
// workgroup, every workgroup of the launch the same matrix -- at 32 x 4 s of ns_dpcrn_v0 that is ~30 TB/s asked of the L2s,
// and the ten convolutions ran at 40-60 TFLOP/s of the 157 the fp32 matrix pipe has (profiles/r04_dpcrn_conv2d_before.txt).
// Here a workgroup stages 32 k of its channel tile in LDS (double buffered, one barrier per chunk) and the four waves read
// their fragments from there.  The k -> (source row, shift) table keeps only the taps that exist for this output row: a
// transposed convolution with stride 2 uses every second frequency tap, and rows at the edge lose the taps outside the
// input -- the table is compacted in k order by one wave (ballot prefix: deterministic summation order), zero-filled to a
// multiple of 32.
constexpr int C2D_KC = 32;

__device__ __forceinline__ void c2d_tap(const Conv2dArgs& a, int k, int fo, int& off, int& sh) {
  off = -1, sh = 0;
  if (k < a.K) {
    const int jt = k % a.kt, jf = (k / a.kt) % a.kf, ci = k / (a.kt * a.kf);
    int fi;
    bool ok = true;
    if (!a.transposed) {
      fi = fo * a.sf + jf * a.df - a.pf;
      sh = jt * a.dt - a.pt;
    } else {
      const int num = fo + a.pf - jf * a.df;
      ok = num >= 0 && num % a.sf == 0;
      fi = num / a.sf;
      sh = a.pt - jt * a.dt;
    }
    if (ok && fi >= 0 && fi < a.Fin)
      off = ci < a.C1 ? (ci * a.Fin + fi) * a.ld : (((ci - a.C1) * a.Fin + fi) * a.ld) | (1 << 30);
  }
}

// Workgroup -> (frame tile, output row, utterance x channel tile), XCD-aware.  Workgroups go to the 8 XCDs round robin in
// dispatch order, each XCD has its own L2, and neighbouring output rows read the same input rows (kf = 3: two of three):
// with the plain numbering a row's neighbours sit on other XCDs and every workgroup's B operand comes from HBM (u1 of
// ns_dpcrn_v0: 8 GB per launch for a 1 GB input).  Here XCD x owns the rows [x Fout/8, (x+1) Fout/8) of a (frame tile,
// utterance) and walks them in order.
__device__ __forceinline__ void c2d_block(int& bx, int& fo, int& bz) {
  bx = blockIdx.x, fo = blockIdx.y, bz = blockIdx.z;
  if (gridDim.y % 8 == 0) {
    const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int fpc = gridDim.y / 8, xcd = L & 7, s = L >> 3;
    fo = xcd * fpc + s % fpc;
    const int rest = s / fpc;
    bx = rest % gridDim.x;
    bz = rest / gridDim.x;
  }
}

template <int MB>
__global__ __launch_bounds__(256) void conv2d_lds_kernel(Conv2dArgs a) {
  constexpr int AS = 32 * MB + (MB > 1 ? 32 : 0);  // floats per k row: rows k, k + 1 (the two half-waves) 32 banks apart
  extern __shared__ int c2d_tab[];
  __shared__ int nk_s;
  const int Kq = (a.Kp + 31) / 32 * 32;
  int* const tab_off = c2d_tab;
  int* const tab_shift = c2d_tab + Kq;
  int* const tab_k = c2d_tab + 2 * Kq;
  float* const As = reinterpret_cast<float*>(c2d_tab + 3 * Kq);  // [2][32][AS]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 31, lk = lane >> 5;
  int bx, fo, bz;
  c2d_block(bx, fo, bz);
  const int t0 = bx * 128;
  const int mtiles = (a.M + 32 * MB - 1) / (32 * MB);
  const int n = bz / mtiles, m0 = (bz % mtiles) * 32 * MB;

  if (w == 0) {
    int count = 0;
    for (int k0 = 0; k0 < a.K; k0 += 64) {
      int off, sh;
      c2d_tap(a, k0 + lane, fo, off, sh);
      const unsigned long long mask = __ballot(off >= 0);
      const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
      if (off >= 0) {
        tab_off[pos] = off;
        tab_shift[pos] = sh;
        tab_k[pos] = k0 + lane;
      }
      count += __popcll(mask);
    }
    const int kc = (count + C2D_KC - 1) / C2D_KC * C2D_KC;
    for (int i = count + lane; i < kc; i += 64) tab_off[i] = -1, tab_shift[i] = 0, tab_k[i] = 0;
    if (lane == 0) nk_s = kc;
  }
  __syncthreads();
  const int nch = nk_s / C2D_KC;

  const float* x1n = a.x1 + (size_t)n * a.C1 * a.Fin * a.ld;
  const float* x2n = a.x2 ? a.x2 + (size_t)n * a.C2 * a.Fin * a.ld : a.x1;
  const int tcol = t0 + 32 * w + lr;
  f32x16 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;

  // A chunk: thread (k row = tid / 8, segment = tid % 8) moves MB x 16 bytes of its row
  const int kr = tid >> 3, seg = tid & 7;
  const float* wbase = a.wt + (size_t)(m0 >> 8) * a.Kp * 256 + (m0 & 255) + seg * 4;
  f32x4 areg[MB];
  auto a_fetch = [&](int c) {
    const int ks = tab_k[c * C2D_KC + kr];
#pragma unroll
    for (int j = 0; j < MB; ++j) areg[j] = *reinterpret_cast<const f32x4*>(wbase + (size_t)ks * 256 + 32 * j);
  };
  auto a_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < MB; ++j) *reinterpret_cast<f32x4*>(As + (buf * C2D_KC + kr) * AS + seg * 4 + 32 * j) = areg[j];
  };

  // B operand: batches of 8 k-pairs, the next batch's loads in flight under this batch's MFMAs (a whole chunk of 16 pairs
  // ahead measured slower: 3.13 ms against 3.10 for the 256 -> 64 layer, 1.79 against 1.47 for 64 -> 128)
  float bv[2][C2D_UN];
  auto fetch = [&](int bi, auto buf_c) {  // batch bi: pairs 8 bi .. 8 bi + 7 of the compacted table
    constexpr int buf = decltype(buf_c)::value;
#pragma unroll
    for (int u = 0; u < C2D_UN; ++u) {
      const int k = 2 * (bi * C2D_UN + u) + lk;
      const int off = tab_off[k];
      const int ti = tcol + tab_shift[k];
      const bool ok = off >= 0 && ti >= 0 && ti < a.Tin;
      const float* src = (off & (1 << 30)) ? x2n : x1n;
      const int idx = ok ? (off & ((1 << 30) - 1)) + ti : 0;
      const float v = src[idx];  // unconditional load of a valid address, masked afterwards
      bv[buf][u] = ok ? v : 0.f;
    }
  };
  auto multiply = [&](int c, auto half_c) {
    constexpr int half = decltype(half_c)::value;
    const float* ab = As + ((c & 1) * C2D_KC + half * 16 + lk) * AS + lr;
#pragma unroll
    for (int u = 0; u < C2D_UN; ++u)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[2 * u * AS + 32 * mb], bv[half][u], acc[mb], 0, 0, 0);
  };
  using b0 = std::integral_constant<int, 0>;
  using b1 = std::integral_constant<int, 1>;
  if (nch > 0) {
    a_fetch(0);
    a_store(0);
    fetch(0, b0{});
  }
  __syncthreads();
  for (int c = 0; c < nch; ++c) {
    if (c + 1 < nch) a_fetch(c + 1);
    fetch(2 * c + 1, b1{});
    multiply(c, b0{});
    if (c + 1 < nch) fetch(2 * c + 2, b0{});
    multiply(c, b1{});
    if (c + 1 < nch) a_store((c + 1) & 1);
    __syncthreads();
  }

  const float s = a.slope ? a.slope[0] : 0.f;
  float fsum = 0.f, fsq = 0.f;
  if (tcol < a.ld) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * mb + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < a.M) {
          float v = acc[mb][r] + (a.bias ? a.bias[m] : 0.f);
          if (tcol < a.T) fsum += v, fsq += v * v;  // (statistics of the pre-activation output: the gLN that follows)
          v = act_apply(v, a.act, s);
          a.y[(((size_t)n * a.M + m) * a.Fout + fo) * a.ld + tcol] = tcol < a.T ? v : 0.f;
        }
      }
  }
  if (a.stats) {  // one partial per workgroup, in the workgroup's own slot: deterministic
    __shared__ double red[2][4];
    const double ws = wave_sum((double)fsum), wq = wave_sum((double)fsq);
    __syncthreads();
    if (lane == 0) red[0][w] = ws, red[1][w] = wq;
    __syncthreads();
    if (tid == 0) {
      const int parts = gridDim.x * gridDim.y * mtiles;
      const int part = ((bz % mtiles) * gridDim.y + fo) * gridDim.x + bx;
      double* dst = a.stats + ((size_t)n * parts + part) * 2;
      dst[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
      dst[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
  }
}

// ---- the same implicit GEMM in the fp16x2 arithmetic (ps_conv2d_f16x2_f32) ------------------------------------------------
// conv2d_lds_kernel is bound by the fp32 matrix pipe: 2048 / 4096 matrix-pipe cycles per wave and 32-k chunk at 64 / 128
// channels (the 256 -> 64 layer of ns_dpcrn_v0: 3.1 ms, 65 TFLOP/s).  Here the products are v_mfma_f32_16x16x32_f16 on two
// fp16 terms per operand (W 2^e = W_hi + W_lo from the host's image, x S = x_hi + x_lo split in registers; three products,
// fp32 accumulation: the error class of the fp16x2 GEMMs): 24 / 48 issues of 16 cycles per chunk.  What remains is the
// gather and the split of the B operand on the vector ALU (~170 instructions per chunk), which the wave's other work hides.
//   * B: lane (col c = lane & 15, k-group kg = lane >> 4) loads x[k = 32 chunk + 8 kg + e][frame] for its two frames (column
//     blocks cb = 0, 1: frames t0 + 32 w + 16 cb + c) -- exactly the MFMA's B fragment (8 consecutive k of one column).
//   * The range of x is not known to the caller (PReLU outputs): every wave scales by a power of two S of its own, chosen
//     from the maximum |x| it has seen so far (wave_max per chunk); when a chunk exceeds the current range the accumulators
//     are multiplied by the ratio (a power of two: exact) and S drops.  Smaller values later keep S -- fp16 is floating
//     point, their two terms still carry 22 bits -- and S only ever falls, so nothing overflows (|W_hi| < 2^14, |x S| < 2^15,
//     K <= 4096: sums below 2^41).
//   * A: the host packs W 2^e into MFMA-fragment order per (channel tile, chunk): [plane][row block][kg][row][8 k] halves;
//     a chunk is MT x 128 bytes, copied to LDS by all threads (double buffered, one barrier per chunk); fragment reads are
//     conflict-free 16-byte reads.  k runs in natural order (the table marks rows outside the input): layers whose taps
//     need compaction (transposed with stride 2) stay on conv2d_lds_kernel.
typedef _Float16 c2d_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned c2d_u32x4 __attribute__((ext_vector_type(4)));

struct Conv2dF16Args {
  Conv2dArgs g;       // (wt unused)
  const void* wimg;   // [mtiles][Kq / 32][2][RB][4][16][8] halves
  float winv;         // 2^-e
};

template <int MB>
__global__ __launch_bounds__(256) void conv2d_f16x2_kernel(Conv2dF16Args fa) {
  const Conv2dArgs& a = fa.g;
  constexpr int MT = 32 * MB, RB = 2 * MB, CHB = MT * 128;  // channels, row blocks, bytes of an A chunk
  extern __shared__ int c2d_tab[];
  const int Kq = (a.Kp + 31) / 32 * 32;
  int* const tab_off = c2d_tab;
  int* const tab_shift = c2d_tab + Kq;
  unsigned char* const As = reinterpret_cast<unsigned char*>(c2d_tab + 2 * Kq);  // [2][CHB]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, kg = lane >> 4;
  int bx, fo, bz;
  c2d_block(bx, fo, bz);
  const int t0 = bx * 128;
  const int mtiles = (a.M + MT - 1) / MT;
  const int n = bz / mtiles, mt = bz % mtiles, m0 = mt * MT;
  for (int k = tid; k < Kq; k += 256) {
    int off, sh;
    c2d_tap(a, k, fo, off, sh);
    tab_off[k] = off;
    tab_shift[k] = sh;
  }
  const int nch = Kq / 32;
  const c2d_u32x4* wsrc = reinterpret_cast<const c2d_u32x4*>(fa.wimg) + (size_t)mt * nch * (CHB / 16);
  c2d_u32x4 areg[MB];
  auto a_fetch = [&](int ch) {
#pragma unroll
    for (int j = 0; j < MB; ++j) areg[j] = wsrc[(size_t)ch * (CHB / 16) + j * 256 + tid];
  };
  auto a_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < MB; ++j) reinterpret_cast<c2d_u32x4*>(As + buf * CHB)[j * 256 + tid] = areg[j];
  };
  a_fetch(0);
  a_store(0);
  __syncthreads();

  const float* x1n = a.x1 + (size_t)n * a.C1 * a.Fin * a.ld;
  const float* x2n = a.x2 ? a.x2 + (size_t)n * a.C2 * a.Fin * a.ld : a.x1;
  const int tc0 = t0 + 32 * w + c, tc1 = tc0 + 16;
  f32x4 acc[RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb][0] = acc[rb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  float S = 1.0995116e12f;  // 2^40: the first chunk with a non-zero value sets the scale
  int se = 40;              // S = 2^se

  float bv[2][8];
  auto fetch = [&](int ch) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = ch * 32 + 8 * kg + e;
      const int off = tab_off[k];
      const int sh = tab_shift[k];
      const float* src = (off & (1 << 30)) ? x2n : x1n;
      const int o = off & ((1 << 30) - 1);
      const int i0 = tc0 + sh, i1 = tc1 + sh;
      const bool ok0 = off >= 0 && (unsigned)i0 < (unsigned)a.Tin, ok1 = off >= 0 && (unsigned)i1 < (unsigned)a.Tin;
      const float v0 = src[ok0 ? o + i0 : 0], v1 = src[ok1 ? o + i1 : 0];  // unconditional loads of valid addresses
      bv[0][e] = ok0 ? v0 : 0.f;
      bv[1][e] = ok1 ? v1 : 0.f;
    }
  };
  if (nch > 0) fetch(0);
  for (int ch = 0; ch < nch; ++ch) {
    // range of this chunk's values in the wave; the scale follows downwards
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, fmaxf(fabsf(bv[0][e]), fabsf(bv[1][e])));
    m = wave_max(m);
    {
      int ex = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 255) - 127;  // |x| < 2^(ex + 1)
      ex = ex > 100 ? 100 : ex;
      const int want = 14 - ex;  // x 2^want < 2^15
      if (m > 0.f && want < se) {  // uniform
        const float r = ldexpf(1.f, want - se);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb][0] *= r, acc[rb][1] *= r;
        se = want;
        S = ldexpf(1.f, se);
      }
    }
    c2d_f16x8 bh[2], bl[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xs = bv[cb][e] * S;
        const _Float16 hi = (_Float16)xs;
        bh[cb][e] = hi;
        bl[cb][e] = (_Float16)(xs - (float)hi);
      }
    if (ch + 1 < nch) {
      a_fetch(ch + 1);
      fetch(ch + 1);
    }
    const c2d_f16x8* ab = reinterpret_cast<const c2d_f16x8*>(As + (ch & 1) * CHB) + lane;  // + (plane * RB + rb) * 64
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const c2d_f16x8 ahi = ab[rb * 64], alo = ab[(RB + rb) * 64];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bh[cb], acc[rb][cb], 0, 0, 0);
        acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo, bh[cb], acc[rb][cb], 0, 0, 0);
        acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi, bl[cb], acc[rb][cb], 0, 0, 0);
      }
    }
    if (ch + 1 < nch) a_store((ch + 1) & 1);
    __syncthreads();
  }

  const float osc = fa.winv * ldexpf(1.f, -se);
  const float s = a.slope ? a.slope[0] : 0.f;
  float fsum = 0.f, fsq = 0.f;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int mrow = m0 + 16 * rb + 4 * kg + r;
      if (mrow < a.M) {
        const float b = a.bias ? a.bias[mrow] : 0.f;
        float* row = a.y + (((size_t)n * a.M + mrow) * a.Fout + fo) * a.ld;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const int t = cb ? tc1 : tc0;
          float v = acc[rb][cb][r] * osc + b;
          if (t < a.T) fsum += v, fsq += v * v;
          v = act_apply(v, a.act, s);
          if (t < a.ld) row[t] = t < a.T ? v : 0.f;
        }
      }
    }
  if (a.stats) {
    __shared__ double red[2][4];
    const double ws = wave_sum((double)fsum), wq = wave_sum((double)fsq);
    __syncthreads();
    if (lane == 0) red[0][w] = ws, red[1][w] = wq;
    __syncthreads();
    if (tid == 0) {
      const int parts = gridDim.x * gridDim.y * mtiles;
      const int part = (mt * gridDim.y + fo) * gridDim.x + bx;
      double* dst = a.stats + ((size_t)n * parts + part) * 2;
      dst[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
      dst[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
  }
}

// ---- four output channels at most (the mask layer of a U-Net decoder: 64 -> 2 channels, 5 x 2 taps, 256 output rows) -------
// The MFMA tile is 32 channels wide: 2 useful rows of 32 (168 GFLOP issued for 10 at 32 x 4 s, 3.3 ms of the forward).  Here
// the matrix pipe is left alone: a thread owns two frames (t, t + 256) of R = 8 consecutive output rows and all M channels
// (2 x 8 x MM accumulators) and walks the input rows (ci, fi, jt) that feed the block -- every input value is loaded once
// and applied to the R x MM outputs with the weights W[m][ci][jf(r, fi)][jt] of a table in LDS (zero where row r has no tap
// on fi: stride-2 transposed layers use 5 of 8), one 16-byte broadcast read per 4 weights.  Summation order: (ci, fi, jt) ascending -- not the MFMA kernel's pairs;
// the results differ from it in the last bits only.
constexpr int C2D_R = 8;

template <int MM>
__global__ __launch_bounds__(256) void conv2d_rows_kernel(Conv2dArgs a, int span, int nent) {
  extern __shared__ int c2d_tab[];
  int* const ent_off = c2d_tab;           // element offset of the input row (bit 30: second source), -1: outside the input
  int* const ent_shift = c2d_tab + nent;  // frame shift of the tap
  f32x4* const ent_w = reinterpret_cast<f32x4*>(c2d_tab + 2 * ((nent + 3) / 4 * 4));  // [nent][R * MM / 4]
  constexpr int W4 = C2D_R * MM / 4;
  const int tid = threadIdx.x;
  const int fo0 = blockIdx.y * C2D_R, n = blockIdx.z;
  int fi_lo;
  if (!a.transposed) {
    fi_lo = fo0 * a.sf - a.pf;
  } else {
    const int num = fo0 + a.pf - (a.kf - 1) * a.df;  // fi * sf >= num
    fi_lo = num >= 0 ? (num + a.sf - 1) / a.sf : -((-num) / a.sf);
  }
  for (int e = tid; e < nent; e += 256) {
    const int jt = e % a.kt, fidx = (e / a.kt) % span, ci = e / (a.kt * span);
    const int fi = fi_lo + fidx;
    const bool row_ok = fi >= 0 && fi < a.Fin;
    float wv[C2D_R * MM];
    bool any = false;
#pragma unroll
    for (int r = 0; r < C2D_R; ++r) {
      const int fo = fo0 + r;
      const int num = a.transposed ? fo + a.pf - fi * a.sf : fi - fo * a.sf + a.pf;
      const int jf = num / a.df;
      const bool tap = row_ok && fo < a.Fout && num >= 0 && num % a.df == 0 && jf < a.kf;
      const int k = tap ? (ci * a.kf + jf) * a.kt + jt : 0;
#pragma unroll
      for (int m = 0; m < MM; ++m) {
        const float w = (tap && m < a.M) ? a.wt[(size_t)k * 256 + m] : 0.f;
        wv[r * MM + m] = w;
        any = any || tap;
      }
    }
    ent_off[e] = (row_ok && any) ? (ci < a.C1 ? (ci * a.Fin + fi) * a.ld : (((ci - a.C1) * a.Fin + fi) * a.ld) | (1 << 30)) : -1;
    ent_shift[e] = a.transposed ? a.pt - jt * a.dt : jt * a.dt - a.pt;
#pragma unroll
    for (int i = 0; i < W4; ++i) ent_w[e * W4 + i] = f32x4{wv[4 * i], wv[4 * i + 1], wv[4 * i + 2], wv[4 * i + 3]};
  }
  __syncthreads();

  const float* x1n = a.x1 + (size_t)n * a.C1 * a.Fin * a.ld;
  const float* x2n = a.x2 ? a.x2 + (size_t)n * a.C2 * a.Fin * a.ld : a.x1;
  const int ta = blockIdx.x * 512 + tid, tb = ta + 256;
  float acc0[C2D_R * MM], acc1[C2D_R * MM];
#pragma unroll
  for (int i = 0; i < C2D_R * MM; ++i) acc0[i] = 0.f, acc1[i] = 0.f;
  // four entries per round: their eight loads leave together (two waves per SIMD at 64 KiB of LDS per workgroup: nothing
  // else hides the latency), then 4 x 2 x R x MM fmas; an entry outside the input contributes zeros
  constexpr int EU = 4;
  for (int e0 = 0; e0 < nent; e0 += EU) {
    float va[EU], vb[EU];
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int e = e0 + u < nent ? e0 + u : nent - 1;
      const int off = e0 + u < nent ? ent_off[e] : -1;
      const int sh = ent_shift[e];
      const float* src = (off & (1 << 30)) ? x2n : x1n;
      const int o = off & ((1 << 30) - 1);
      const int ia = ta + sh, ib = tb + sh;
      const bool oka = off >= 0 && ia >= 0 && ia < a.Tin, okb = off >= 0 && ib >= 0 && ib < a.Tin;
      const float xa = src[oka ? o + ia : 0], xb = src[okb ? o + ib : 0];
      va[u] = oka ? xa : 0.f;
      vb[u] = okb ? xb : 0.f;
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int e = e0 + u < nent ? e0 + u : nent - 1;
#pragma unroll
      for (int i = 0; i < W4; ++i) {
        const f32x4 w4 = ent_w[e * W4 + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc0[4 * i + j] = fmaf(w4[j], va[u], acc0[4 * i + j]);
          acc1[4 * i + j] = fmaf(w4[j], vb[u], acc1[4 * i + j]);
        }
      }
    }
  }
  const float s = a.slope ? a.slope[0] : 0.f;
#pragma unroll
  for (int r = 0; r < C2D_R; ++r) {
    const int fo = fo0 + r;
    if (fo < a.Fout) {
#pragma unroll
      for (int m = 0; m < MM; ++m)
        if (m < a.M) {
          const float b = a.bias ? a.bias[m] : 0.f;
          float* row = a.y + (((size_t)n * a.M + m) * a.Fout + fo) * a.ld;
          if (ta < a.ld) row[ta] = ta < a.T ? act_apply(acc0[r * MM + m] + b, a.act, s) : 0.f;
          if (tb < a.ld) row[tb] = tb < a.T ? act_apply(acc1[r * MM + m] + b, a.act, s) : 0.f;
        }
    }
  }
}

}  // namespace ps

using namespace ps;

extern "C" int ps_conv2d_f16x2_f32(const float* x1, int C1, const float* x2, int C2, const void* wimg, int w_exp,
                                   const float* bias, float* y, int N, int M, int Fin, int T_in, int T, int ld, int kf, int kt,
                                   int stride_f, int dil_f, int dil_t, int pad_f, int pad_t, int Fout, int transposed, int act,
                                   const float* slope, double* ostats, void* stream) {
  if (!x1 || !wimg || !y || N <= 0 || M <= 0 || C1 <= 0 || C2 < 0 || (C2 > 0 && !x2) || Fin <= 0 || Fout <= 0 || T <= 0 ||
      T_in <= 0 || ld < T || ld < T_in || ld % 128 || kf <= 0 || kt <= 0 || stride_f <= 0 || dil_f <= 0 || dil_t <= 0 ||
      Fout > 65535 || act < 0 || act > 5 || (act == 2 && !slope) || w_exp < -100 || w_exp > 100 || ((uintptr_t)wimg & 15)) {
    set_error("ps_conv2d_f16x2_f32: bad argument (N=%d M=%d C=%d+%d F=%d->%d T=%d k=%dx%d act=%d)", N, M, C1, C2, Fin, Fout, T,
              kf, kt, act);
    return PS_E_INVALID;
  }
  const long long K = (long long)(C1 + C2) * kf * kt;
  const int Kp = (int)((K + 15) / 16 * 16);
  if (Kp > C2D_MAXK || (long long)(C1 > C2 ? C1 : C2) * Fin * ld >= (1LL << 30)) {
    set_error("ps_conv2d_f16x2_f32: Cin*kf*kt = %lld exceeds %d, or an utterance of the input exceeds 2^30 elements", K, C2D_MAXK);
    return PS_E_UNSUPPORTED;
  }
  const int mb = M <= 32 ? 1 : M <= 64 ? 2 : 4;
  const int mtiles = (M + 32 * mb - 1) / (32 * mb);
  if ((long long)N * mtiles > 65535) {
    set_error("ps_conv2d_f16x2_f32: N * channel tiles exceeds the grid limit");
    return PS_E_UNSUPPORTED;
  }
  Conv2dF16Args fa{{x1, x2, nullptr, bias, slope, y, C1, C2, Fin, T, T_in, ld, kf, kt, stride_f, dil_f, dil_t, pad_f, pad_t,
                    Fout, M, (int)K, Kp, transposed, act, ostats},
                   wimg, ldexpf(1.f, -w_exp)};
  dim3 grid(ld / 128, Fout, N * mtiles);
  const int Kq = (Kp + 31) / 32 * 32;
  const size_t lds = (size_t)2 * Kq * sizeof(int) + (size_t)2 * 32 * mb * 128;
  {
    LaunchTimer timer("conv2d", (hipStream_t)stream);
    if (mb == 1)
      hipLaunchKernelGGL((conv2d_f16x2_kernel<1>), grid, dim3(256), lds, (hipStream_t)stream, fa);
    else if (mb == 2)
      hipLaunchKernelGGL((conv2d_f16x2_kernel<2>), grid, dim3(256), lds, (hipStream_t)stream, fa);
    else
      hipLaunchKernelGGL((conv2d_f16x2_kernel<4>), grid, dim3(256), lds, (hipStream_t)stream, fa);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_conv2d_f16x2_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

extern "C" int ps_conv2d_stats_parts(int M, int Fout, int ld) {
  if (M <= 0 || Fout <= 0 || ld <= 0 || ld % 128) return 0;
  const int mb = M <= 32 ? 1 : M <= 64 ? 2 : 4;
  return (ld / 128) * Fout * ((M + 32 * mb - 1) / (32 * mb));
}

static int conv2d_launch(const float* x1, int C1, const float* x2, int C2, const float* wt, const float* bias, float* y, int N,
                         int M, int Fin, int T_in, int T, int ld, int kf, int kt, int stride_f, int dil_f, int dil_t, int pad_f,
                         int pad_t, int Fout, int transposed, int act, const float* slope, double* ostats, void* stream);

extern "C" int ps_conv2d_f32(const float* x1, int C1, const float* x2, int C2, const float* wt, const float* bias,
                             float* y, int N, int M, int Fin, int T_in, int T, int ld, int kf, int kt, int stride_f,
                             int dil_f, int dil_t, int pad_f, int pad_t, int Fout, int transposed, int act,
                             const float* slope, void* stream) {
  return conv2d_launch(x1, C1, x2, C2, wt, bias, y, N, M, Fin, T_in, T, ld, kf, kt, stride_f, dil_f, dil_t, pad_f, pad_t, Fout,
                       transposed, act, slope, nullptr, stream);
}

extern "C" int ps_conv2d_stats_f32(const float* x1, int C1, const float* x2, int C2, const float* wt, const float* bias,
                                   float* y, int N, int M, int Fin, int T_in, int T, int ld, int kf, int kt, int stride_f,
                                   int dil_f, int dil_t, int pad_f, int pad_t, int Fout, int transposed, double* ostats,
                                   void* stream) {
  if (!ostats) {
    set_error("ps_conv2d_stats_f32: ostats is NULL");
    return PS_E_INVALID;
  }
  return conv2d_launch(x1, C1, x2, C2, wt, bias, y, N, M, Fin, T_in, T, ld, kf, kt, stride_f, dil_f, dil_t, pad_f, pad_t, Fout,
                       transposed, 0, nullptr, ostats, stream);
}

static int conv2d_launch(const float* x1, int C1, const float* x2, int C2, const float* wt, const float* bias, float* y, int N,
                         int M, int Fin, int T_in, int T, int ld, int kf, int kt, int stride_f, int dil_f, int dil_t, int pad_f,
                         int pad_t, int Fout, int transposed, int act, const float* slope, double* ostats, void* stream) {
  if (!x1 || !wt || !y || N <= 0 || M <= 0 || C1 <= 0 || C2 < 0 || (C2 > 0 && !x2) || Fin <= 0 || Fout <= 0 || T <= 0 ||
      T_in <= 0 || ld < T || ld < T_in || ld % 128 || kf <= 0 || kt <= 0 || stride_f <= 0 || dil_f <= 0 || dil_t <= 0 ||
      Fout > 65535 || act < 0 || act > 5 || (act == 2 && !slope)) {
    set_error("ps_conv2d_f32: bad argument (N=%d M=%d C=%d+%d F=%d->%d T=%d k=%dx%d act=%d)", N, M, C1, C2, Fin, Fout, T, kf,
              kt, act);
    return PS_E_INVALID;
  }
  const long long K = (long long)(C1 + C2) * kf * kt;
  const int Kp = (int)((K + 15) / 16 * 16);
  if (Kp > C2D_MAXK) {
    set_error("ps_conv2d_f32: Cin*kf*kt = %lld exceeds %d", K, C2D_MAXK);
    return PS_E_UNSUPPORTED;
  }
  if ((long long)(C1 > C2 ? C1 : C2) * Fin * ld >= (1LL << 30)) {
    set_error("ps_conv2d_f32: one utterance of the input exceeds 2^30 elements");
    return PS_E_UNSUPPORTED;
  }
  const int mb = M <= 32 ? 1 : M <= 64 ? 2 : 4;
  const int mtiles = (M + 32 * mb - 1) / (32 * mb);
  if ((long long)N * mtiles > 65535) {
    set_error("ps_conv2d_f32: N * channel tiles exceeds the grid limit");
    return PS_E_UNSUPPORTED;
  }
  Conv2dArgs a{x1, x2, wt, bias, slope, y, C1, C2, Fin, T, T_in, ld, kf, kt, stride_f, dil_f, dil_t, pad_f, pad_t,
               Fout, M, (int)K, Kp, transposed, act, ostats};
  dim3 grid(ld / 128, Fout, N * mtiles);
  {
    LaunchTimer timer("conv2d", (hipStream_t)stream);
    const bool old_kernel = (g_debug_flags & (1 << 23)) != 0 && !ostats;  // (bit 23: the round-3 kernel; tests run both)
    // <= 4 output channels: the row-block kernel when its table fits (span = input rows feeding 8 output rows)
    const int span = transposed ? ((C2D_R - 1) + (kf - 1) * dil_f) / stride_f + 2 : (C2D_R - 1) * stride_f + (kf - 1) * dil_f + 1;
    const long long nent = (long long)(C1 + C2) * span * kt;
    const int mm = M <= 2 ? 2 : 4;
    const size_t rows_lds = (size_t)((nent + 3) / 4 * 4) * 2 * sizeof(int) + (size_t)nent * C2D_R * mm * sizeof(float);
    if (M <= 4 && !old_kernel && !ostats && N <= 65535 && rows_lds <= 150 * 1024 && (Fout + C2D_R - 1) / C2D_R <= 65535) {
      dim3 rgrid((ld + 511) / 512, (Fout + C2D_R - 1) / C2D_R, N);
      static bool lds_raised = false;  // (dynamic LDS above 64 KiB has to be allowed per kernel, once)
      if (!lds_raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_rows_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_rows_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        lds_raised = true;
      }
      if (mm == 2)
        hipLaunchKernelGGL((conv2d_rows_kernel<2>), rgrid, dim3(256), rows_lds, (hipStream_t)stream, a, span, (int)nent);
      else
        hipLaunchKernelGGL((conv2d_rows_kernel<4>), rgrid, dim3(256), rows_lds, (hipStream_t)stream, a, span, (int)nent);
    } else if (!old_kernel) {
      const int Kq = (Kp + 31) / 32 * 32;
      const size_t as = 32 * mb + (mb > 1 ? 32 : 0);
      const size_t lds = (size_t)3 * Kq * sizeof(int) + 2 * C2D_KC * as * sizeof(float);
      if (mb == 1)
        hipLaunchKernelGGL((conv2d_lds_kernel<1>), grid, dim3(256), lds, (hipStream_t)stream, a);
      else if (mb == 2)
        hipLaunchKernelGGL((conv2d_lds_kernel<2>), grid, dim3(256), lds, (hipStream_t)stream, a);
      else
        hipLaunchKernelGGL((conv2d_lds_kernel<4>), grid, dim3(256), lds, (hipStream_t)stream, a);
    } else {
      const size_t lds = (size_t)2 * Kp * sizeof(int);
      if (mb == 1)
        hipLaunchKernelGGL((conv2d_kernel<1>), grid, dim3(256), lds, (hipStream_t)stream, a);
      else if (mb == 2)
        hipLaunchKernelGGL((conv2d_kernel<2>), grid, dim3(256), lds, (hipStream_t)stream, a);
      else
        hipLaunchKernelGGL((conv2d_kernel<4>), grid, dim3(256), lds, (hipStream_t)stream, a);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_conv2d_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
