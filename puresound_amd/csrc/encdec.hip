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
