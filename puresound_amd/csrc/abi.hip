// C-ABI glue: error reporting, sizing helpers and the network-level drivers that enqueue one TCN block
// (conv_tasnet.py:67-90 of mcw519/PureSound) and the whole Conv-TasNet masker (conv_tasnet.py:338-359).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <mutex>
#include <vector>

#include <atomic>

#include "ps_common.h"

namespace ps {

static thread_local char g_err[512] = "";
int g_debug_flags = 0;
void* g_debug_buffer = nullptr;

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct ProfRecord {
  const char* kernel;
  hipEvent_t start, stop;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRecord> g_prof;

LaunchTimer::LaunchTimer(const char* kernel, hipStream_t stream) : slot_(-1), stream_(stream) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRecord r{kernel, nullptr, nullptr};
  if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
  (void)hipEventRecord(r.start, stream);
  g_prof.push_back(r);
  slot_ = (int)g_prof.size() - 1;
}

LaunchTimer::~LaunchTimer() {
  if (slot_ < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_prof[slot_].stop, stream_);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace ps

using namespace ps;

namespace ps {
int device_cus() {
  static std::atomic<int> cache[64];  // 0 = not read yet
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int cus = cache[dev].load(std::memory_order_relaxed);
  if (cus == 0) {
    hipDeviceProp_t prop;
    cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
    if (cus <= 0) cus = 256;
    cache[dev].store(cus, std::memory_order_relaxed);
  }
  return cus;
}
}  // namespace ps

extern "C" int ps_abi_version(void) { return PS_ABI_VERSION; }
extern "C" const char* ps_last_error(void) { return g_err; }

extern "C" int ps_debug_flags(int flags) {
  const int old = g_debug_flags;
  if (flags >= 0) g_debug_flags = flags;
  return old;
}

extern "C" int ps_debug_buffer(void* device_buffer) {
  g_debug_buffer = device_buffer;
  return 0;
}

extern "C" int ps_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (on) {
    for (auto& r : g_prof) {
      (void)hipEventDestroy(r.start);
      (void)hipEventDestroy(r.stop);
    }
    g_prof.clear();
  }
  g_prof_on = on != 0;
  return 0;
}

extern "C" int ps_profile_read(const char* kernel, double* total_ms, int* launches) {
  if (!kernel || !total_ms || !launches) {
    set_error("ps_profile_read: null argument");
    return PS_E_INVALID;
  }
  std::lock_guard<std::mutex> lk(g_prof_mu);
  double tot = 0.0;
  int cnt = 0;
  for (auto& r : g_prof) {
    if (strcmp(r.kernel, kernel) != 0) continue;
    hipError_t e = hipEventSynchronize(r.stop);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, r.start, r.stop);
    if (e != hipSuccess) {
      set_error("ps_profile_read: %s", hipGetErrorString(e));
      return (int)e;
    }
    tot += ms;
    ++cnt;
  }
  *total_ms = tot;
  *launches = cnt;
  return 0;
}

// Rows are padded to an ODD multiple of 128 frames (512 B): a power-of-two row stride (T=3999 -> 16 KiB)
// would put the same column of every channel row on the same HBM channels ("channel camping").
extern "C" int ps_padded_frames(int frames) {
  if (frames <= 0) return 0;
  int tiles = ceil_div(frames, kTileT);
  if ((tiles & 1) == 0) ++tiles;
  return tiles * kTileT;
}

extern "C" int ps_stats_parts(int channels, int frames) {
  if (channels <= 0 || frames <= 0) return 0;
  const int gemm = ps_conv1x1_stats_parts(channels, frames);
  const int dw = ps_dwconv_stats_parts(channels, frames);
  return gemm > dw ? gemm : dw;
}

// Workspace carve-up for ps_conv_tasnet_f32:
//   y1, y2, y3 : 3 x [N][H][ldt] fp32
//   stats      : 3 x [N][parts][2] fp64
//   bias_n     : [N][H] fp32 (only with an embedding)
//   amax       : 2 x [N][amax_parts] fp32 partial maxima of the residual stream (gemm_planes = 2: the range the next
//                block's in_conv scales its input by; written by out_conv's epilogue, or by ps_absmax_f32 for block 0)
struct TasnetWs {
  float *y1, *y2, *y3, *bias_n;
  float* amax[2];
  double *s1, *s2, *s3;
  int parts, amax_parts;
  size_t bytes;
};

static TasnetWs carve(void* base, int N, int C, int H, int T) {
  TasnetWs w{};
  const int ldt = ps_padded_frames(T);
  // parts must cover producers with up to max(C,H) channels; H is what every stats producer emits
  w.parts = ps_stats_parts(H, T);
  const size_t map = align_up((size_t)N * H * ldt * sizeof(float), 256);
  const size_t st = align_up((size_t)N * w.parts * 2 * sizeof(double), 256);
  const size_t bn = align_up((size_t)N * H * sizeof(float), 256);
  w.amax_parts = ps_conv1x1_stats_parts(C > 0 ? C : H, T);
  if (w.amax_parts < ps_absmax_parts()) w.amax_parts = ps_absmax_parts();
  const size_t am = align_up((size_t)N * w.amax_parts * sizeof(float), 256);
  char* p = (char*)base;
  w.y1 = (float*)p; p += map;
  w.y2 = (float*)p; p += map;
  w.y3 = (float*)p; p += map;
  w.s1 = (double*)p; p += st;
  w.s2 = (double*)p; p += st;
  w.s3 = (double*)p; p += st;
  w.bias_n = (float*)p; p += bn;
  w.amax[0] = (float*)p; p += am;
  w.amax[1] = (float*)p; p += am;
  w.bytes = (size_t)(p - (char*)base);
  return w;
}

extern "C" size_t ps_conv_tasnet_workspace_bytes(int N, int C, int H, int T) {
  if (N <= 0 || C <= 0 || H <= 0 || T <= 0) return 0;
  return carve(nullptr, N, C, H, T).bytes;
}

// x_amax / x_amax_parts: partial maxima of |x_in| (gemm_planes = 2); y_amax: where out_conv leaves those of x_out
// sb: the residual stream (x_in, x_out) is bf16 rows too (ps_conv_tasnet_bf16_rows; needs gemm_planes = 1 with hidden_bf16)
static int run_block(const ps_tcn_block& b, const float* x_in, float* x_out, const float* dvec, int embed_norm,
                     int N, int T, int ldt, const TasnetWs& w, const float* x_amax, int x_amax_parts, float* y_amax,
                     void* stream, int sb = 0) {
  int rc;
  const float eps = 1e-8f;  // GlobLN.eps and gGN's eps (lobe/norm.py:10,96); folded BN carries its own
  const double count = (double)b.H * (double)T;
  const int gemm_parts_h = ps_conv1x1_stats_parts(b.H, T);
  const int dw_parts = ps_dwconv_stats_parts(b.H, T);

  // 1) in_conv (no bias) [+ per-utterance embedding bias]; stats of y1
  const float* bias_n = nullptr;
  if (b.in_embed_w) {
    if (!dvec) {
      set_error("ps_conv_tasnet_f32: block expects an embedding (E=%d) but dvec is NULL", b.E);
      return PS_E_INVALID;
    }
    rc = ps_embed_bias_f32(dvec, b.in_embed_w, w.bias_n, N, b.E, b.H, embed_norm, stream);
    if (rc) return rc;
    bias_n = w.bias_n;
  }
  // the three GEMMs: exact fp32 MFMA, or the bf16 pipe with 1 / 3 operand planes
  // hidden_bf16 (with gemm_planes = 1): y1, y2, y3 -- which never leave this workspace -- are bf16 rows; the block's
  // input and output (the residual stream) stay fp32
  const int hb = (b.hidden_bf16 && b.gemm_planes == 1) ? 1 : 0;
  // gemm_planes = 2 behind a folded BatchNorm (bN1d blocks: no bound on the normalised values exists): the producers leave
  // the maxima of |y2| / |y3| behind -- in the statistics slots such a norm has no use for -- and the consumer maps them
  // through the norm's largest scale and shift
  float* const a2 = (b.gemm_planes == 2 && b.dw_norm == PS_NORM_AFFINE) ? reinterpret_cast<float*>(w.s2) : nullptr;
  float* const a3 = (b.gemm_planes == 2 && b.pw_norm == PS_NORM_AFFINE) ? reinterpret_cast<float*>(w.s3) : nullptr;
  int a2_parts = 0;
  // `which` = 0 / 1 / 2: in_conv, pointwise, out_conv (gemm_planes = 2: each has its own range descriptor)
  auto gemm = [&](int which, const float* x, int xb, const float* wt, const void* wb, float* y, int yb, int K, int M,
                  const ps_prologue* pro, const float* bias, const float* bn, const float* res, double* st) {
    if (b.gemm_planes == 0) return ps_conv1x1_f32(x, wt, y, N, K, M, T, ldt, pro, bias, bn, res, st, stream);
    if (b.gemm_planes == 2) {
      ps_f16x2_range rng{};
      rng.w_exp = b.w_exp[which];
      const float gmax = which == 1 ? b.dw_gmax : b.pw_gmax, bmax = which == 1 ? b.dw_bmax : b.pw_bmax;
      const float* measured = which == 1 ? a2 : a3;  // maxima of the producer's output in front of a folded BatchNorm
      if (which == 0) {
        rng.x_amax = x_amax;
        rng.x_amax_parts = x_amax_parts;
      } else if (measured && gmax > 0.f) {
        // behind a per-channel affine map: |scale_c v + shift_c| <= max|scale| max|v| + max|shift| (times max(1, |slope|),
        // folded into gmax / bmax by the planner)
        rng.x_amax = measured;
        rng.x_amax_parts = which == 1 ? a2_parts : gemm_parts_h;
        rng.amax_mul = gmax;
        rng.amax_add = bmax;
      } else if (measured) {
        rng.x_bound = bmax > 0.f ? bmax : 1.f;  // (scale = 0: the map is the constant shift)
      } else {
        // behind a global norm |z| <= sqrt(count - 1), so |gamma z + beta| <= max|gamma| sqrt(count) + max|beta|
        rng.x_bound = gmax * (float)sqrt(count) + bmax;
        if (!(rng.x_bound > 0.f)) rng.x_bound = 1.f;  // (gamma = beta = 0: every value is 0)
      }
      if (which == 1) rng.y_amax = a3;
      if (which == 2) rng.y_amax = y_amax;
      return ps_conv1x1_f16x2_f32(x, wb, &rng, y, N, K, M, T, ldt, pro, bias, bn, res, st, stream);
    }
    // bf16 residual stream: launches the register-B kernel takes run with bf16 rows and ONE fp16 product per multiply-add
    // (ps_conv1x1_f16_rows) whenever the block carries the fp16 weight images and the input's range is known
    const void* wf = which == 0 ? b.in_wf : which == 1 ? b.pw_wf : b.out_wf;
    if (sb && wf && xb && yb && ps_conv1x1_f16_rows_ok(N, K, M, T)) {
      ps_f16x2_range rng{};
      rng.w_exp = b.w_exp[which];
      bool have = true;
      if (which == 0) {
        rng.x_amax = x_amax;
        rng.x_amax_parts = x_amax_parts;
        have = x_amax != nullptr;
      } else {
        const float gmax = which == 1 ? b.dw_gmax : b.pw_gmax, bmax = which == 1 ? b.dw_bmax : b.pw_bmax;
        rng.x_bound = gmax * (float)sqrt(count) + bmax;
        if (!(rng.x_bound > 0.f)) rng.x_bound = 1.f;
        have = (which == 1 ? b.dw_norm : b.pw_norm) == PS_NORM_GLOBAL;
      }
      if (which == 2) rng.y_amax = y_amax;
      if (have) return ps_conv1x1_f16_rows(x, wf, &rng, y, N, K, M, T, ldt, pro, bias, bn, res, st, stream);
    }
    return ps_conv1x1_bf16_io(x, xb, wb, y, yb, N, K, M, T, ldt, b.gemm_planes, pro, bias, bn, res, st, stream);
  };
  auto ranged = [](int norm) { return norm == PS_NORM_GLOBAL || norm == PS_NORM_AFFINE; };
  if (b.gemm_planes == 2 && (!ranged(b.dw_norm) || !ranged(b.pw_norm) || !x_amax)) {
    set_error("ps_conv_tasnet_f32: gemm_planes=2 (fp16x2) needs a global norm (a bound on the normalised values) or a "
              "per-channel affine norm (the producer's measured maxima) in front of the pointwise and output convs, and the "
              "range of the block's input");
    return PS_E_UNSUPPORTED;
  }
  if (b.gemm_planes != 0 && (!b.in_wb || !b.pw_wb || !b.out_wb)) {
    set_error("ps_conv_tasnet_f32: gemm_planes=%d needs the plane-packed weights in_wb / pw_wb / out_wb", b.gemm_planes);
    return PS_E_INVALID;
  }
  if (sb && !hb) {
    set_error("ps_conv_tasnet_bf16_rows: block needs gemm_planes = 1 with hidden_bf16 (got planes=%d hidden_bf16=%d)",
              b.gemm_planes, b.hidden_bf16);
    return PS_E_UNSUPPORTED;
  }
  rc = gemm(0, x_in, sb, b.in_wt, b.in_wb, w.y1, hb, b.C, b.H, nullptr, nullptr, bias_n, nullptr,
            b.in_norm == PS_NORM_GLOBAL ? w.s1 : nullptr);
  if (rc) return rc;

  // 2) depthwise: prologue = in_conv's norm + PReLU; stats of y2
  ps_prologue p1{};
  p1.norm = b.in_norm;
  p1.prelu = 1;
  p1.stats = w.s1;
  p1.parts = gemm_parts_h;
  p1.count = count;
  p1.eps = eps;
  p1.gamma = b.in_gamma;
  p1.beta = b.in_beta;
  p1.slope = b.in_slope;
  const int left = b.causal ? (b.P - 1) * b.dilation : ((b.P - 1) / 2) * b.dilation;
  if (a2 && ps_dwconv_amax_ok(b.P, b.dilation, left)) {
    rc = ps_dwconv_amax_f32(w.y1, b.dw_w, b.dw_b, w.y2, N, b.H, T, ldt, b.P, b.dilation, left, &p1, a2, stream);
    a2_parts = dw_parts;
  } else {
    rc = ps_dwconv_io(w.y1, hb, b.dw_w, b.dw_b, w.y2, hb, N, b.H, T, ldt, b.P, b.dilation, left, &p1,
                      b.dw_norm == PS_NORM_GLOBAL ? w.s2 : nullptr, stream);
    if (!rc && a2) {  // (a shape outside the kernel that measures while it writes: one more pass over y2)
      if ((size_t)ps_absmax_parts() * sizeof(float) > (size_t)w.parts * 2 * sizeof(double)) {
        set_error("ps_conv_tasnet_f32: no room for the maxima of the depthwise output (T=%d)", T);
        return PS_E_UNSUPPORTED;
      }
      rc = ps_absmax_f32(w.y2, a2, N, b.H, T, ldt, stream);
      a2_parts = ps_absmax_parts();
    }
  }
  if (rc) return rc;

  // 3) pointwise: prologue = depthwise norm + PReLU; stats of y3
  ps_prologue p2{};
  p2.norm = b.dw_norm;
  p2.prelu = 1;
  p2.stats = w.s2;
  p2.parts = dw_parts;
  p2.count = count;
  p2.eps = eps;
  p2.gamma = b.dw_gamma;
  p2.beta = b.dw_beta;
  p2.slope = b.dw_slope;
  rc = gemm(1, w.y2, hb, b.pw_wt, b.pw_wb, w.y3, hb, b.H, b.H, &p2, b.pw_b, nullptr, nullptr,
            b.pw_norm == PS_NORM_GLOBAL ? w.s3 : nullptr);
  if (rc) return rc;

  // 4) out_conv + bias + residual: prologue = pointwise norm + PReLU
  ps_prologue p3{};
  p3.norm = b.pw_norm;
  p3.prelu = 1;
  p3.stats = w.s3;
  p3.parts = gemm_parts_h;
  p3.count = count;
  p3.eps = eps;
  p3.gamma = b.pw_gamma;
  p3.beta = b.pw_beta;
  p3.slope = b.pw_slope;
  return gemm(2, w.y3, hb, b.out_wt, b.out_wb, x_out, sb, b.H, b.C, &p3, b.out_b, nullptr, x_in, nullptr);
}

extern "C" int ps_conv_tasnet_f32(const ps_tcn_block* blocks, int n_blocks, const float* x_in, float* x_out,
                                  const float* dvec, int embed_norm, int N, int T, int ldt, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  return ps_conv_tasnet_ranged_f32(blocks, n_blocks, x_in, x_out, dvec, embed_norm, N, T, ldt, workspace, workspace_bytes,
                                   nullptr, 0, stream);
}

static int conv_tasnet_rows(const ps_tcn_block* blocks, int n_blocks, const float* x_in, float* x_out, const float* dvec,
                            int embed_norm, int N, int T, int ldt, void* workspace, size_t workspace_bytes,
                            const float* x_amax, int x_amax_parts, void* stream, int sb);

extern "C" int ps_conv_tasnet_ranged_f32(const ps_tcn_block* blocks, int n_blocks, const float* x_in, float* x_out,
                                         const float* dvec, int embed_norm, int N, int T, int ldt, void* workspace,
                                         size_t workspace_bytes, const float* x_amax, int x_amax_parts, void* stream) {
  return conv_tasnet_rows(blocks, n_blocks, x_in, x_out, dvec, embed_norm, N, T, ldt, workspace, workspace_bytes, x_amax,
                          x_amax_parts, stream, 0);
}

// BASELINE config 3's arithmetic ("bf16 storage / fp32 accumulate"): the residual stream is bf16 rows as well
extern "C" int ps_conv_tasnet_bf16_rows(const ps_tcn_block* blocks, int n_blocks, const void* x_in, void* x_out,
                                        const float* dvec, int embed_norm, int N, int T, int ldt, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  return conv_tasnet_rows(blocks, n_blocks, (const float*)x_in, (float*)x_out, dvec, embed_norm, N, T, ldt, workspace,
                          workspace_bytes, nullptr, 0, stream, 1);
}

static int conv_tasnet_rows(const ps_tcn_block* blocks, int n_blocks, const float* x_in, float* x_out, const float* dvec,
                            int embed_norm, int N, int T, int ldt, void* workspace, size_t workspace_bytes,
                            const float* x_amax, int x_amax_parts, void* stream, int sb) {
  if (x_amax && x_amax_parts <= 0) {
    set_error("ps_conv_tasnet_ranged_f32: x_amax needs x_amax_parts > 0");
    return PS_E_INVALID;
  }
  if (!blocks || n_blocks <= 0 || !x_in || !x_out || !workspace || N <= 0 || T <= 0) {
    set_error("ps_conv_tasnet_f32: null pointer or non-positive size");
    return PS_E_INVALID;
  }
  if (x_in == x_out) {
    set_error("ps_conv_tasnet_f32: x_in must not alias x_out (the input is never modified)");
    return PS_E_INVALID;
  }
  if (ldt != ps_padded_frames(T)) {
    set_error("ps_conv_tasnet_f32: ldt=%d must be ps_padded_frames(T=%d)=%d", ldt, T, ps_padded_frames(T));
    return PS_E_ALIGN;
  }
  const int C = blocks[0].C, H = blocks[0].H;
  for (int i = 0; i < n_blocks; ++i) {
    const ps_tcn_block& b = blocks[i];
    if (b.C != C || b.H != H || b.P <= 0 || b.dilation <= 0) {
      set_error("ps_conv_tasnet_f32: block %d has inconsistent sizes (C=%d H=%d P=%d dilation=%d)", i, b.C, b.H, b.P,
                b.dilation);
      return PS_E_INVALID;
    }
    if (b.causal && (b.in_norm == PS_NORM_GLOBAL || b.dw_norm == PS_NORM_GLOBAL || b.pw_norm == PS_NORM_GLOBAL)) {
      // reference: AssertionError in DepthwiseSeparableConv1d (lobe/cnn.py:40-44)
      set_error("ps_conv_tasnet_f32: block %d: global norms conflict with causal=1", i);
      return PS_E_INVALID;
    }
  }
  if (workspace_bytes < ps_conv_tasnet_workspace_bytes(N, C, H, T)) {
    set_error("ps_conv_tasnet_f32: workspace too small (%zu < %zu)", workspace_bytes,
              ps_conv_tasnet_workspace_bytes(N, C, H, T));
    return PS_E_INVALID;
  }
  if ((uintptr_t)workspace & 255) {
    set_error("ps_conv_tasnet_f32: workspace must be 256-byte aligned");
    return PS_E_ALIGN;
  }
  const TasnetWs w = carve(workspace, N, C, H, T);
  // block 0 reads the caller's input and writes x_out; later blocks update x_out in place (each
  // workgroup reads exactly the residual elements it overwrites).
  int have_parts = 0;  // partial maxima of the current residual stream in w.amax[i & 1] (fp16x2 blocks only)
  const int out_parts = ps_conv1x1_stats_parts(C, T);
  for (int i = 0; i < n_blocks; ++i) {
    const float* xi = i == 0 ? x_in : x_out;
    // (bf16 rows: out_conv leaves the maxima of the stream for the next in_conv when it runs on the register-B kernel)
    const bool rows16 = sb && blocks[i].out_wf && blocks[i].in_wf && ps_conv1x1_f16_rows_ok(N, H, C, T) &&
                        blocks[i].pw_norm == PS_NORM_GLOBAL;
    const bool f16 = blocks[i].gemm_planes == 2 || rows16;
    const float* range = w.amax[i & 1];
    if (f16 && i == 0 && x_amax) {  // the caller knows the range of x_in (maxima, or any upper bound per utterance)
      range = x_amax;
      have_parts = x_amax_parts;
    } else if (f16 && !have_parts && sb) {
      range = nullptr;  // (no pass over bf16 rows: this block's in_conv takes the planes = 1 kernel)
    } else if (f16 && !have_parts) {  // first fp16x2 block (or one behind another arithmetic): one pass over its input
      const int rc = ps_absmax_f32(xi, w.amax[i & 1], N, C, T, ldt, stream);
      if (rc) return rc;
      have_parts = ps_absmax_parts();
    }
    if (sb && (blocks[i].gemm_planes != 1 || !blocks[i].hidden_bf16)) {
      set_error("ps_conv_tasnet_bf16_rows: block %d is not in the bf16 arithmetic (gemm_planes = 1, hidden_bf16)", i);
      return PS_E_UNSUPPORTED;
    }
    const int rc = run_block(blocks[i], xi, x_out, dvec, embed_norm, N, T, ldt, w, (f16 && have_parts) ? range : nullptr,
                             have_parts, f16 ? w.amax[(i + 1) & 1] : nullptr, stream, sb);
    if (rc) return rc;
    have_parts = f16 ? out_parts : 0;
  }
  return 0;
}
