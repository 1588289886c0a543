// Fused depthwise dilated convolution of the TCN block (lobe/cnn.py:62-74 of mcw519/PureSound):
//   a = PReLU(norm(x))   (the in_conv's gLN/bN1d + PReLU, applied on load)
//   y[n][h][t] = b[h] + sum_j w[h][j] * a[n][h][t + j*dilation - left]      (a == 0 outside [0,T))
// plus per-workgroup partial (sum, sumsq) of y for the next global norm.
//
// HBM-bound streaming kernel with LDS-staged input tiles.  A workgroup owns DW_ROWS channels x 1024 frames
// and walks them DW_RB rows at a time: each row segment [t0 - left, t0 + 1024 + right) is loaded ONCE with
// coalesced 16-byte loads (the halo by the first threads), normalised + activated ONCE per element, zeroed
// outside [0, T) and written to LDS; the taps are then plain LDS reads (16-byte when the tap offset is a
// multiple of 4 frames, dword otherwise), so neither the tap loop nor the zero padding costs masks or repeated
// transforms.  (A register-only version re-loaded and re-transformed every element once per tap: 26 VALU per
// output frame and 0.43-0.6 of the achievable bandwidth.)  DW_RB rows are in flight per thread to cover the
// HBM latency at the per-CU bandwidth share.
#include <type_traits>

#include "ps_common.h"

namespace ps {

constexpr int DW_ROWS = 16;
constexpr int DW_FRAMES = 1024;  // 256 threads x 4
constexpr int DW_MAXP = 8;
constexpr int DW_RB = 4;          // rows staged together
constexpr int DW_MAXHALO = 1024;  // (P-1)*dilation frames of halo the LDS image can hold ...
constexpr int DW_SMALLHALO = 288;  // ... and in the small-halo build (P = 3, dilation <= 128: 21 KiB of LDS instead of
                                   // 32, i.e. seven resident workgroups per CU instead of four)

struct DwArgs {
  const float* x;
  const float* w;
  const float* b;
  float* y;
  double* ostats;
  ps_prologue pro;
  int H, T, ldt, P, dilation, left;
  float* amax;  // wave kernel, AMAX build: [N][ps_dwconv_stats_parts] partial maxima of |y|
};

// P > 0: compile-time tap count; P == 0: run-time taps.  ALIGNED: every tap offset is a multiple of 4 frames.
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

template <bool B16>
__device__ __forceinline__ f32x4 dw_load4(const void* base, size_t elem) {
  if constexpr (B16) {
    const u16x4 h = *reinterpret_cast<const u16x4*>(reinterpret_cast<const unsigned short*>(base) + elem);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_bit_cast(float, (unsigned)h[e] << 16);
    return v;
  } else {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + elem);
  }
}

// XB / YB: the rows of x / y are bf16 in HBM (hidden maps of a TCN block in the "bf16" arithmetic); LDS image,
// taps, bias and statistics stay fp32.
template <int P, bool ALIGNED, int HALO = DW_MAXHALO, bool XB = false, bool YB = false>
__global__ __launch_bounds__(256) void dwconv_kernel(DwArgs a) {
  constexpr int DW_SEG = DW_FRAMES + HALO;
  __shared__ __attribute__((aligned(16))) float seg[DW_RB][DW_SEG];
  __shared__ double red[8];
  const int tid = threadIdx.x;
  const int n = blockIdx.z;
  const int h0 = blockIdx.y * DW_ROWS;
  const int t0 = blockIdx.x * DW_FRAMES;
  const int np = P > 0 ? P : a.P;
  const int halo = (np - 1) * a.dilation;  // left + right
  const int halo4 = (halo + 3) / 4;        // halo vectors, loaded by the first threads
  // segment origin in frames, rounded down to a multiple of 4 so that every staging load is 16-byte aligned
  const int lpad = (a.left + 3) / 4 * 4;
  const int org = t0 - lpad;               // frame of seg[.][0]
  const int shift = lpad - a.left;         // tap j of output frame f reads seg[f - t0 + j*dilation + shift]
  const int nvec = DW_FRAMES / 4 + (lpad + halo - a.left + 3) / 4 + 0;  // vectors that can be touched
  (void)halo4;

  const bool has_norm = a.pro.norm != PS_NORM_NONE;
  const bool has_prelu = a.pro.prelu != 0;
  const float slope = has_prelu ? a.pro.slope[0] : 1.f;
  const bool slope01 = slope >= 0.f && slope <= 1.f;

  float fsum = 0.f, fsq = 0.f;
  const int t = t0 + tid * 4;
  const int i1 = tid + 256;  // second vector index (halo part): only the first (nvec - 256) threads
  // The row batches are software pipelined: the loads of batch b+1 are issued before batch b is transformed, written
  // to LDS and consumed, so a workgroup always has a batch of HBM loads in flight behind its LDS phase (without it
  // every batch paid the full load latency between its two barriers: 4.0 TB/s).
  f32x4 v[2][DW_RB][2];
  auto load_batch = [&](int r0, auto q_c) {
    constexpr int q = decltype(q_c)::value;
#pragma unroll
    for (int r = 0; r < DW_RB; ++r) {
      const int h = h0 + r0 + r;
      const size_t row = ((size_t)n * a.H + (h < a.H ? h : a.H - 1)) * a.ldt;
      const int f0 = org + tid * 4, f1 = org + i1 * 4;
      v[q][r][0] = (f0 >= 0 && f0 < a.T) ? dw_load4<XB>(a.x, row + f0) : f32x4{0.f, 0.f, 0.f, 0.f};
      v[q][r][1] = (i1 < nvec && f1 >= 0 && f1 < a.T) ? dw_load4<XB>(a.x, row + f1) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  static_assert(DW_ROWS / DW_RB == 4, "four row batches, unrolled below");
  using q0 = std::integral_constant<int, 0>;
  using q1 = std::integral_constant<int, 1>;
  const int nb = (a.H - h0 + DW_RB - 1) / DW_RB;  // batches that hold rows (uniform)
  load_batch(0, q0{});
  if (nb > 1) load_batch(DW_RB, q1{});
  // (the reduction of the producer's partial statistics -- two barriers and an fp64 butterfly -- runs behind the first
  //  loads instead of in front of them: it cost 11 % of the launch)
  // per-row scale / shift of all DW_ROWS rows up front: their gamma / beta loads are in flight together (fetched per
  // batch, each batch stalled on them between its barriers: 8-10 us of a 62 us launch)
  float gam[DW_ROWS], bet[DW_ROWS];
#pragma unroll
  for (int r = 0; r < DW_ROWS; ++r) {
    const int hc = h0 + r < a.H ? h0 + r : a.H - 1;
    gam[r] = has_norm ? a.pro.gamma[hc] : 1.f;
    bet[r] = has_norm ? a.pro.beta[hc] : 0.f;
  }
  const NormScalars ns = load_norm_scalars(a.pro, n, red);
  auto process_batch = [&](auto r0_c, auto q_c) {
    constexpr int r0 = decltype(r0_c)::value;
    constexpr int qb = decltype(q_c)::value;
    // ---- transform once, zero outside [0,T), write LDS ------------------------------------------------
#pragma unroll
    for (int r = 0; r < DW_RB; ++r) {
      const float sc = has_norm ? gam[r0 + r] * ns.rstd : 1.f;
      const float sh = bet[r0 + r] - ns.mean * sc;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int idx = q == 0 ? tid : i1;
        if (q == 1 && idx >= nvec) continue;
        const int f = org + idx * 4;
        f32x4 u;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float z = v[qb][r][q][e] * sc + sh;
          // PReLU as max(z, slope z) when 0 <= slope <= 1 (two instructions instead of compare / select / multiply)
          if (has_prelu) z = slope01 ? fmaxf(z, slope * z) : prelu(z, slope);
          u[e] = z;
        }
        if (!(f >= 0 && f + 3 < a.T)) {  // edge vectors only: the conv's zero padding, applied AFTER norm + PReLU
#pragma unroll
          for (int e = 0; e < 4; ++e) u[e] = (f + e >= 0 && f + e < a.T) ? u[e] : 0.f;
        }
        *reinterpret_cast<f32x4*>(&seg[r][idx * 4]) = u;
      }
    }
    __syncthreads();
    // ---- taps from LDS, bias, statistics, store --------------------------------------------------------
    if (t < a.T) {
#pragma unroll
      for (int r = 0; r < DW_RB; ++r) {
        const int h = h0 + r0 + r;
        if (h < a.H) {
          const float bias = a.b ? a.b[h] : 0.f;
          f32x4 out{bias, bias, bias, bias};
          const float* sp = &seg[r][tid * 4 + shift];
          if constexpr (P > 0) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
              const float wj = a.w[h * P + j];
              f32x4 s;
              if constexpr (ALIGNED) {
                s = *reinterpret_cast<const f32x4*>(sp + j * a.dilation);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] = sp[j * a.dilation + e];
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) out[e] += wj * s[e];
            }
          } else {
            for (int j = 0; j < np; ++j) {
              const float wj = a.w[h * np + j];
#pragma unroll
              for (int e = 0; e < 4; ++e) out[e] += wj * sp[j * a.dilation + e];
            }
          }
          if (t + 3 < a.T) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              fsum += out[e];
              fsq += out[e] * out[e];
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (t + e < a.T) {
                fsum += out[e];
                fsq += out[e] * out[e];
              }
          }
          if constexpr (YB) {
            u16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __builtin_bit_cast(unsigned short, (__bf16)out[e]);
            *reinterpret_cast<u16x4*>(reinterpret_cast<unsigned short*>(a.y) + ((size_t)n * a.H + h) * a.ldt + t) = o;
          } else {
            *reinterpret_cast<f32x4*>(a.y + ((size_t)n * a.H + h) * a.ldt + t) = out;
          }
        }
      }
    }
    __syncthreads();  // the next batch overwrites the LDS image
  };
  using ic = std::integral_constant<int, 0>;
  process_batch(ic{}, q0{});
  if (nb > 1) {
    if (nb > 2) load_batch(2 * DW_RB, q0{});
    process_batch(std::integral_constant<int, DW_RB>{}, q1{});
    if (nb > 2) {
      if (nb > 3) load_batch(3 * DW_RB, q1{});
      process_batch(std::integral_constant<int, 2 * DW_RB>{}, q0{});
      if (nb > 3) process_batch(std::integral_constant<int, 3 * DW_RB>{}, q1{});
    }
  }
  if (a.ostats) {
    double s = fsum, q = fsq;
    block_sum2(s, q, red);
    if (tid == 0) {
      const int parts = gridDim.x * gridDim.y;
      double* dst = a.ostats + ((size_t)n * parts + blockIdx.y * gridDim.x + blockIdx.x) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

// ---- wave-private variant (round 3): P = 3, fp32 rows, halo <= 256 frames -------------------------------------------------
// The kernel above synchronises a 256-thread workgroup twice per batch of rows; with HBM under load every batch then waits
// for the slowest of its 256 threads' loads, and the launch ran at 4.0 TB/s where a device copy of the same bytes reaches
// 6.1 (tools/copy_ceiling.py).  Here a WAVE owns a row segment: 64 lanes x 16-byte loads cover 1024 frames in four
// coalesced 1 KiB pieces (+ the halo piece), the wave transforms them, writes them to its own LDS strip and reads the taps
// back -- LDS operations of one wave execute in order, so no workgroup barrier is needed, and the loads of the wave's
// next row are in flight while it works on the current one.  Same grid, same partial-statistics slots as the kernel
// above (one (sum, sumsq) per workgroup = 16 rows x 1024 frames).
constexpr int DWW_SEG = DW_FRAMES + 256;  // frames per LDS strip (64 * DWW_NV pieces): 4 x 5 KiB = 20 KiB per workgroup, so
                                          // that EIGHT workgroups -- all 2048 of a 32 x 256 x 3999 launch at once -- fit a CU
constexpr int DWW_NV = 5;                     // 16-byte pieces per lane: 4 + halo

// B16: the rows of x AND y are bf16 in HBM (round 4: the hidden maps of a block whose activation rows are all bf16); the
// LDS strip and the arithmetic stay fp32, a piece is 8 bytes per lane instead of 16.
// AMAX: instead of the (sum, sumsq) partials the workgroup leaves the maximum of |y| over its valid frames in a.amax (same
// slots): what a following fp16x2 GEMM needs when the norm between them is a folded BatchNorm, not a global norm.
template <bool ALIGNED, bool B16 = false, bool AMAX = false>
__global__ __launch_bounds__(256, 8) void dwconv_wave_kernel(DwArgs a) {
  __shared__ __attribute__((aligned(16))) float strip[4][DWW_SEG];
  double* const red = reinterpret_cast<double*>(&strip[0][0]);  // the two reductions run before / after the strips are in use
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.z;
  const int h0 = blockIdx.y * DW_ROWS;
  const int t0 = blockIdx.x * DW_FRAMES;
  const int halo = 2 * a.dilation;
  const int lpad = (a.left + 3) / 4 * 4;
  const int org = t0 - lpad;
  const int shift = lpad - a.left;
  const int nvec = DW_FRAMES / 4 + (lpad + halo - a.left + 3) / 4;  // <= 320 (launcher)
  const bool has_norm = a.pro.norm != PS_NORM_NONE;
  const bool has_prelu = a.pro.prelu != 0;
  const float slope = has_prelu ? a.pro.slope[0] : 1.f;
  const bool slope01 = slope >= 0.f && slope <= 1.f;
  float* const seg = strip[wave];

  // (one row per wave in flight: at eight waves per SIMD the other waves cover the latency, and a second row in registers
  //  would push the kernel past 64 VGPRs -- below eight workgroups per CU the launch needs a second, ragged round)
  f32x4 v[1][DWW_NV];
  auto load_row = [&](int r, auto q_c) {
    constexpr int q = decltype(q_c)::value;
    const int h = h0 + wave + 4 * r;
    const size_t row = ((size_t)n * a.H + (h < a.H ? h : a.H - 1)) * a.ldt;
#pragma unroll
    for (int k = 0; k < DWW_NV; ++k) {
      const int idx = k * 64 + lane, f = org + idx * 4;
      v[q][k] = (idx < nvec && f >= 0 && f < a.T) ? dw_load4<B16>(a.x, row + f) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  using q0 = std::integral_constant<int, 0>;
  load_row(0, q0{});
  // gamma / beta / taps / bias of the wave's four rows (their loads are in flight behind the row loads)
  float gam[4], bet[4], w0[4], w1[4], w2[4], bia[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int h = h0 + wave + 4 * r, hc = h < a.H ? h : a.H - 1;
    gam[r] = has_norm ? a.pro.gamma[hc] : 1.f;
    bet[r] = has_norm ? a.pro.beta[hc] : 0.f;
    w0[r] = a.w[hc * 3], w1[r] = a.w[hc * 3 + 1], w2[r] = a.w[hc * 3 + 2];
    bia[r] = a.b ? a.b[hc] : 0.f;
  }
  const NormScalars ns = load_norm_scalars(a.pro, n, red);  // (the only workgroup barriers before the final reduction)
  __syncthreads();                                          // (red is strip 0: every thread has read its totals)
  [[maybe_unused]] float fsum = 0.f, fsq = 0.f, famax = 0.f;
  auto process_row = [&](auto r_c, auto q_c) {
    constexpr int r = decltype(r_c)::value, q = decltype(q_c)::value;
    const int h = h0 + wave + 4 * r;
    const float sc = has_norm ? gam[r] * ns.rstd : 1.f;
    const float sh = bet[r] - ns.mean * sc;
#pragma unroll
    for (int k = 0; k < DWW_NV; ++k) {
      const int idx = k * 64 + lane, f = org + idx * 4;
      if (k == DWW_NV - 1 && idx >= nvec) continue;
      f32x4 u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float z = v[q][k][e] * sc + sh;
        if (has_prelu) z = slope01 ? fmaxf(z, slope * z) : prelu(z, slope);
        u[e] = z;
      }
      if (!(f >= 0 && f + 3 < a.T)) {  // edge pieces only: the conv's zero padding, applied AFTER norm + PReLU
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = (f + e >= 0 && f + e < a.T) ? u[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(seg + idx * 4) = u;
    }
    __builtin_amdgcn_wave_barrier();  // (the wave's own writes above, its own reads below: in order in the LDS queue)
    asm volatile("" ::: "memory");
    if (h < a.H) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = t0 + (k * 64 + lane) * 4;
        if (t < a.T) {
          const float* sp = seg + (k * 64 + lane) * 4 + shift;
          f32x4 s0, s1, s2;
          if constexpr (ALIGNED) {
            s0 = *reinterpret_cast<const f32x4*>(sp);
            s1 = *reinterpret_cast<const f32x4*>(sp + a.dilation);
            s2 = *reinterpret_cast<const f32x4*>(sp + 2 * a.dilation);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) s0[e] = sp[e], s1[e] = sp[a.dilation + e], s2[e] = sp[2 * a.dilation + e];
          }
          f32x4 out;
#pragma unroll
          for (int e = 0; e < 4; ++e) out[e] = ((bia[r] + w0[r] * s0[e]) + w1[r] * s1[e]) + w2[r] * s2[e];
          if constexpr (AMAX) {
#pragma unroll
            for (int e = 0; e < 4; ++e) famax = fmaxf(famax, t + e < a.T ? fabsf(out[e]) : 0.f);
          } else if (t + 3 < a.T) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              fsum += out[e];
              fsq += out[e] * out[e];
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (t + e < a.T) {
                fsum += out[e];
                fsq += out[e] * out[e];
              }
          }
          if constexpr (B16) {
            u16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __builtin_bit_cast(unsigned short, (__bf16)out[e]);
            *reinterpret_cast<u16x4*>(reinterpret_cast<unsigned short*>(a.y) + ((size_t)n * a.H + h) * a.ldt + t) = o;
          } else {
            *reinterpret_cast<f32x4*>(a.y + ((size_t)n * a.H + h) * a.ldt + t) = out;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
  };
  process_row(std::integral_constant<int, 0>{}, q0{});
  load_row(1, q0{});
  process_row(std::integral_constant<int, 1>{}, q0{});
  load_row(2, q0{});
  process_row(std::integral_constant<int, 2>{}, q0{});
  load_row(3, q0{});
  process_row(std::integral_constant<int, 3>{}, q0{});
  if constexpr (AMAX) {
    if (a.amax) {
      famax = wave_max(famax);
      __syncthreads();  // (every wave is done with its strip)
      float* const mx = &strip[0][0];
      if (lane == 0) mx[wave] = famax;
      __syncthreads();
      if (tid == 0)
        a.amax[(size_t)n * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x] =
            fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3]));
    }
  } else if (a.ostats) {
    double s = fsum, q = fsq;
    __syncthreads();  // (every wave is done with its strip)
    block_sum2(s, q, red);
    if (tid == 0) {
      const int parts = gridDim.x * gridDim.y;
      double* dst = a.ostats + ((size_t)n * parts + blockIdx.y * gridDim.x + blockIdx.x) * 2;
      dst[0] = s;
      dst[1] = q;
    }
  }
}

}  // namespace ps

extern "C" int ps_dwconv_stats_parts(int H, int T) {
  if (H <= 0 || T <= 0) return 0;
  return ((T + ps::DW_FRAMES - 1) / ps::DW_FRAMES) * ((H + ps::DW_ROWS - 1) / ps::DW_ROWS);
}

extern "C" int ps_dwconv_f32(const float* x, const float* w, const float* b, float* y, int N, int H, int T,
                             int ldt, int P, int dilation, int left, const ps_prologue* pro, double* ostats,
                             void* stream) {
  return ps_dwconv_io(x, 0, w, b, y, 0, N, H, T, ldt, P, dilation, left, pro, ostats, stream);
}

static int dwconv_any(const void* x_any, int x_bf16, const float* w, const float* b, void* y_any, int y_bf16, int N, int H, int T,
                      int ldt, int P, int dilation, int left, const ps_prologue* pro, double* ostats, float* amax, void* stream);

extern "C" int ps_dwconv_io(const void* x_any, int x_bf16, const float* w, const float* b, void* y_any, int y_bf16, int N,
                            int H, int T, int ldt, int P, int dilation, int left, const ps_prologue* pro, double* ostats,
                            void* stream) {
  return dwconv_any(x_any, x_bf16, w, b, y_any, y_bf16, N, H, T, ldt, P, dilation, left, pro, ostats, nullptr, stream);
}

extern "C" int ps_dwconv_amax_ok(int P, int dilation, int left) {
  return P == 3 && dilation > 0 && left >= 0 && left <= 2 * dilation && 2 * dilation <= 256 &&
         ps::DW_FRAMES / 4 + ((left + 3) / 4 * 4 + 2 * dilation - left + 3) / 4 <= 64 * ps::DWW_NV;
}

extern "C" int ps_dwconv_amax_f32(const float* x, const float* w, const float* b, float* y, int N, int H, int T, int ldt, int P,
                                  int dilation, int left, const ps_prologue* pro, float* y_amax, void* stream) {
  using namespace ps;
  if (!y_amax || !ps_dwconv_amax_ok(P, dilation, left)) {
    set_error("ps_dwconv_amax_f32: the maxima are an output of the wave-private kernel only (P = 3, 2 * dilation <= 256; "
              "ps_dwconv_amax_ok); got P=%d dilation=%d left=%d", P, dilation, left);
    return PS_E_UNSUPPORTED;
  }
  return dwconv_any(x, 0, w, b, y, 0, N, H, T, ldt, P, dilation, left, pro, nullptr, y_amax, stream);
}

static int dwconv_any(const void* x_any, int x_bf16, const float* w, const float* b, void* y_any, int y_bf16, int N, int H, int T,
                      int ldt, int P, int dilation, int left, const ps_prologue* pro, double* ostats, float* amax, void* stream) {
  using namespace ps;
  const float* x = (const float*)x_any;
  float* y = (float*)y_any;
  if (!x || !w || !y || N <= 0 || H <= 0 || T <= 0 || P <= 0 || P > DW_MAXP || dilation <= 0 || left < 0) {
    set_error("ps_dwconv_f32: bad argument (N=%d H=%d T=%d P=%d dilation=%d left=%d)", N, H, T, P, dilation, left);
    return PS_E_INVALID;
  }
  if (left > (P - 1) * dilation) {
    set_error("ps_dwconv_f32: left=%d exceeds the receptive field (P-1)*dilation=%d", left, (P - 1) * dilation);
    return PS_E_INVALID;
  }
  if ((P - 1) * dilation + 8 > DW_MAXHALO) {
    set_error("ps_dwconv_f32: (P-1)*dilation=%d exceeds the %d-frame halo the LDS tile holds", (P - 1) * dilation,
              DW_MAXHALO - 8);
    return PS_E_UNSUPPORTED;
  }
  if (ldt < T || ldt % kTileT != 0 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) {
    set_error("ps_dwconv_f32: ldt=%d must be a multiple of %d >= T=%d and pointers 16-byte aligned", ldt, kTileT, T);
    return PS_E_ALIGN;
  }
  DwArgs a{};
  a.x = x;
  a.w = w;
  a.b = b;
  a.y = y;
  a.ostats = ostats;
  a.amax = amax;
  if (pro) {
    a.pro = *pro;
    if (a.pro.norm != PS_NORM_NONE && (!a.pro.gamma || !a.pro.beta)) {
      set_error("ps_dwconv_f32: norm prologue needs gamma/beta");
      return PS_E_INVALID;
    }
    if (a.pro.norm == PS_NORM_GLOBAL && (!a.pro.stats || a.pro.parts <= 0 || a.pro.count <= 0)) {
      set_error("ps_dwconv_f32: PS_NORM_GLOBAL prologue needs stats/parts/count");
      return PS_E_INVALID;
    }
    if (a.pro.prelu && !a.pro.slope) {
      set_error("ps_dwconv_f32: prelu prologue needs slope");
      return PS_E_INVALID;
    }
  } else {
    a.pro.norm = PS_NORM_NONE;
  }
  a.H = H;
  a.T = T;
  a.ldt = ldt;
  a.P = P;
  a.dilation = dilation;
  a.left = left;
  const bool aligned = (dilation % 4 == 0) && (left % 4 == 0);
  dim3 grid((T + DW_FRAMES - 1) / DW_FRAMES, (H + DW_ROWS - 1) / DW_ROWS, N);
  {
    LaunchTimer timer("dwconv", (hipStream_t)stream);
    hipStream_t st = (hipStream_t)stream;
    const bool small = (P - 1) * dilation + 8 <= DW_SMALLHALO;
    const bool wave_ok = P == 3 && 2 * dilation <= 256 &&
                         DW_FRAMES / 4 + ((left + 3) / 4 * 4 + 2 * dilation - left + 3) / 4 <= 64 * DWW_NV && !(g_debug_flags & 1);
    if (amax) {  // (ps_dwconv_amax_f32: fp32 rows, the wave-private kernel's shapes)
      if (aligned) hipLaunchKernelGGL((dwconv_wave_kernel<true, false, true>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((dwconv_wave_kernel<false, false, true>), grid, dim3(256), 0, st, a);
    } else if (x_bf16 && y_bf16 && wave_ok) {  // (bit 0 keeps the workgroup-synchronised kernel: tests run both)
      if (aligned) hipLaunchKernelGGL((dwconv_wave_kernel<true, true>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((dwconv_wave_kernel<false, true>), grid, dim3(256), 0, st, a);
    } else if (x_bf16 || y_bf16) {
      if (P != 3 || !small) {
        set_error("ps_dwconv_io: bf16 rows are built for P = 3 with (P-1)*dilation <= %d", DW_SMALLHALO - 8);
        return PS_E_UNSUPPORTED;
      }
#define PS_DW(AL, XBV, YBV) hipLaunchKernelGGL((dwconv_kernel<3, AL, DW_SMALLHALO, XBV, YBV>), grid, dim3(256), 0, st, a)
      if (aligned) {
        if (x_bf16 && y_bf16) PS_DW(true, true, true);
        else if (x_bf16) PS_DW(true, true, false);
        else PS_DW(true, false, true);
      } else {
        if (x_bf16 && y_bf16) PS_DW(false, true, true);
        else if (x_bf16) PS_DW(false, true, false);
        else PS_DW(false, false, true);
      }
#undef PS_DW
    } else if (P == 3 && 2 * dilation <= 256 && DW_FRAMES / 4 + ((left + 3) / 4 * 4 + 2 * dilation - left + 3) / 4 <= 64 * DWW_NV &&
               !(g_debug_flags & 1)) {  // (ps_debug_flags bit 0 keeps the workgroup-synchronised kernel: tests run both)
      if (aligned) hipLaunchKernelGGL((dwconv_wave_kernel<true>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((dwconv_wave_kernel<false>), grid, dim3(256), 0, st, a);
    } else if (P == 3 && aligned && small)
      hipLaunchKernelGGL((dwconv_kernel<3, true, DW_SMALLHALO>), grid, dim3(256), 0, st, a);
    else if (P == 3 && small)
      hipLaunchKernelGGL((dwconv_kernel<3, false, DW_SMALLHALO>), grid, dim3(256), 0, st, a);
    else if (P == 3 && aligned)
      hipLaunchKernelGGL((dwconv_kernel<3, true>), grid, dim3(256), 0, st, a);
    else if (P == 3)
      hipLaunchKernelGGL((dwconv_kernel<3, false>), grid, dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((dwconv_kernel<0, false>), grid, dim3(256), 0, st, a);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_dwconv_f32: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
