// Probe: one wave per SIMD, 48 x v_mfma_f32_32x32x16_bf16 per iteration with the staging work of one K-step
// (VALU split + LDS traffic) interleaved.  Prints cycles per iteration for: MFMA only, natural order, grouped.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* out, unsigned long long* cyc, int iters, const float* src) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[96 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int i = tid; i < 96 * 1024 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = (float)(i & 1023) * 1e-3f;
  __syncthreads();
  const unsigned char* a_frag = smem + lane * 32;
  const unsigned char* b_frag = smem + 48 * 1024 + lane * 32;
  const float* raw = reinterpret_cast<const float*>(smem + 72 * 1024) + tid;
  unsigned char* bx = smem + 80 * 1024 + tid * 16;
  unsigned long long t0 = 0;
  bf16x8 bfq[2][3][2], afq[2][4][3];
  auto load_frags = [&](int q, int it) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) bfq[q][p][ti] = *reinterpret_cast<const bf16x8*>(b_frag + (p * 2 + ti) * 2048 + (it & 1) * 64);
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int p = 0; p < 3; ++p) afq[q][mi][p] = *reinterpret_cast<const bf16x8*>(a_frag + (mi * 3 + p) * 2048 + (it & 1) * 64);
  };
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 64 << 20, 0x00020000);
  auto body = [&](int it, auto q_c) {
    constexpr int q = decltype(q_c)::value;
    constexpr int BASE = MODE >= 7 ? 4 : (MODE >= 5 ? 3 : MODE);
    if (BASE < 3) load_frags(q, it);
    float v[8];
    if (BASE >= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = raw[j * 128 + (it & 3)];
    }
    if (BASE >= 3) load_frags(q ^ 1, it + 1);
    if (BASE >= 4) {
      float* dst = reinterpret_cast<float*>(smem + 84 * 1024) + (tid >> 6) * 256;
      const int so = ((blockIdx.x * 2048 + (it & 1023)) * 8) * 1024;
#pragma unroll
      for (int i = 0; i < 8; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst + (i & 1) * 1024, 16, lane * 16, so + i * 1024, 0, 0);
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const bf16x8* f = afq[q][mi];
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], bfq[q][1][ti], acc[mi][ti], 0, 0, 0);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2], bfq[q][0][ti], acc[mi][ti], 0, 0, 0);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], bfq[q][2][ti], acc[mi][ti], 0, 0, 0);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], bfq[q][0][ti], acc[mi][ti], 0, 0, 0);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], bfq[q][1][ti], acc[mi][ti], 0, 0, 0);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], bfq[q][0][ti], acc[mi][ti], 0, 0, 0);
    }
    if (BASE >= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float u = v[j] * 1.01f + 0.5f;
        v[j] = u >= 0.f ? u : 0.25f * u;
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        bf16x8 piece;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const bf16x2 hh = __builtin_convertvector(f32x2v{v[j], v[j + 1]}, bf16x2);
          piece[j] = hh[0];
          piece[j + 1] = hh[1];
          if (p < 2) {
            const f32x2v back = __builtin_convertvector(hh, f32x2v);
            v[j] -= back[0];
            v[j + 1] -= back[1];
          }
        }
        *reinterpret_cast<bf16x8*>(bx + p * 4096) = piece;
      }
    }
    if (MODE == 5 || MODE == 7) __builtin_amdgcn_iglp_opt(0);
    if (MODE == 6 || MODE == 8) __builtin_amdgcn_iglp_opt(1);
    if (BASE >= 4)
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  if (MODE >= 3) load_frags(0, 0);
  for (int it = 0; it < iters + 2; it += 2) {
    if (it == 2) t0 = __builtin_amdgcn_s_memtime();
    body(it, std::integral_constant<int, 0>{});
    body(it + 1, std::integral_constant<int, 1>{});
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out;
  float* src;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&src, 80 << 20);
  hipMemset(src, 0, 80 << 20);
  hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  unsigned long long h[256];
  for (int mode : {3, 4, 5, 6, 7, 8}) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters, src);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters, src);
      if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters, src);
      if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(256), dim3(256), 0, 0, out, cyc, iters, src);
      if (mode == 5) hipLaunchKernelGGL(probe<5>, dim3(256), dim3(256), 0, 0, out, cyc, iters, src);
      if (mode == 6) hipLaunchKernelGGL(probe<6>, dim3(256), dim3(256), 0, 0, out, cyc, iters, src);
      if (mode == 7) hipLaunchKernelGGL(probe<7>, dim3(256), dim3(256), 0, 0, out, cyc, iters, src);
      if (mode == 8) hipLaunchKernelGGL(probe<8>, dim3(256), dim3(256), 0, 0, out, cyc, iters, src);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("mode %d: %.0f cycles per iteration (48 MFMAs = 1536 ideal)\n", mode, s / 256 / iters);
  }
  return 0;
}
