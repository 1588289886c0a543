// Signal-level scores next to the path (loss/sdr.py:104-215, 263-330 of mcw519/PureSound): every SDR / SI-SNR variant
// of the reference is a function of five moments of the (estimate, reference) pair -- sum a, sum b, sum a^2, sum b^2,
// sum ab over the time axis -- so one streaming pass over the two waveforms replaces the reference's chain of
// mean / subtract / multiply / sum passes (zero-mean, projection and error signal are algebra on the moments, done in
// fp64 by the caller).  HBM-bound: 8 bytes per sample pair; fp64 accumulation, partials per workgroup (no atomics,
// deterministic), reduced by the caller.
#include "ps_common.h"

namespace ps {

constexpr int WM_CHUNK = 8192;  // samples per workgroup

__global__ __launch_bounds__(256) void wave_moments_kernel(const float* a, const float* b, double* out, int L, int lda,
                                                           int ldb, int chunks) {
  __shared__ double red[8 * 5];
  const int n = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
  const float* pa = a + (size_t)n * lda;
  const float* pb = b + (size_t)n * ldb;
  const int lo = c * WM_CHUNK, hi = lo + WM_CHUNK < L ? lo + WM_CHUNK : L;
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  // fp32 products are exact in fp64; four independent chains per thread keep the loads in flight
  for (int i = lo + tid; i < hi; i += 256) {
    const double x = pa[i], y = pb[i];
    s[0] += x;
    s[1] += y;
    s[2] += x * x;
    s[3] += y * y;
    s[4] += x * y;
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const double v = wave_sum(s[k]);
    if (lane == 0) red[wave * 5 + k] = v;
  }
  __syncthreads();
  if (tid < 5) out[((size_t)n * chunks + c) * 5 + tid] = red[tid] + red[5 + tid] + red[10 + tid] + red[15 + tid];
}

}  // namespace ps

extern "C" int ps_wave_moments_chunks(int L) { return L <= 0 ? 0 : (L + ps::WM_CHUNK - 1) / ps::WM_CHUNK; }

extern "C" int ps_wave_moments_f64(const float* a, const float* b, double* partials, int N, int L, int lda, int ldb,
                                   void* stream) {
  using namespace ps;
  if (!a || !b || !partials || N <= 0 || L <= 0 || lda < L || ldb < L || N > 65535) {
    set_error("ps_wave_moments_f64: null pointer or bad size (N=%d L=%d lda=%d ldb=%d)", N, L, lda, ldb);
    return PS_E_INVALID;
  }
  const int chunks = ps_wave_moments_chunks(L);
  {
    LaunchTimer timer("wave_moments", (hipStream_t)stream);
    hipLaunchKernelGGL(wave_moments_kernel, dim3(chunks, N), dim3(256), 0, (hipStream_t)stream, a, b, partials, L, lda,
                       ldb, chunks);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("ps_wave_moments_f64: launch failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
