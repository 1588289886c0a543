// Probe: what does feeding LDS cost next to an MFMA stream?  512 threads per CU (two waves per SIMD), every wave
// issues 12 groups of four v_mfma_f32_32x32x16_bf16 per iteration (= one K-step of the split GEMM); five of the
// groups are followed by one 1 KiB operand fetch in the form under test.
//   mode 0: no fetch                     mode 1: buffer_load_dwordx4 ... lds (LDS-DMA)
//   mode 2: buffer_load_dwordx4 -> VGPRs, ds_write_b128 one iteration later (counted vmcnt)
//   mode 3: global_load_dwordx4 -> VGPRs, ds_write_b128 one iteration later
//   mode 4: as 1, two pieces back to back in groups 0 / 1 and one in group 2
// build: hipcc --offload-arch=gfx950 -O3 -o dma_cost_probe dma_cost_probe.hip ; run: ./dma_cost_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(float* out, unsigned long long* cyc, int iters, const float* src, int span_kib, int rnd) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[64 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 af[3], bf[3];
  for (int p = 0; p < 3; ++p)
    for (int e = 0; e < 8; ++e) {
      if (rnd) {  // random sign / mantissa, exponents within 2^-3 .. 2^0 (what split planes of real data look like)
        unsigned hsh = (tid * 977u + p * 131u + e * 31u + blockIdx.x * 7919u) * 2654435761u;
        unsigned short ua = (unsigned short)((hsh & 0x807f) | ((124 + ((hsh >> 8) & 3)) << 7));
        hsh = hsh * 1664525u + 1013904223u;
        unsigned short ub = (unsigned short)((hsh & 0x807f) | ((124 + ((hsh >> 8) & 3)) << 7));
        af[p][e] = __builtin_bit_cast(__bf16, ua), bf[p][e] = __builtin_bit_cast(__bf16, ub);
      } else {
        af[p][e] = (__bf16)(0.001f * (lane + p + e)), bf[p][e] = (__bf16)(0.002f * (lane - p + e));
      }
    }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 0x7fffffff, 0x00020000);
  float* dst = reinterpret_cast<float*>(smem) + wave * 5 * 256;  // 5 KiB per wave
  f32x4 q[5];
  for (int i = 0; i < 5; ++i) q[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned long long t0;
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  const int span = span_kib * 1024;
  int so = (blockIdx.x * 40 * 1024) % span;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 12; ++g) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc[g & 3][e & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[(g + e) % 3], bf[e % 3], acc[g & 3][e & 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (g < 5) {
        const int off = so + (wave * 5 + g) * 1024;
        if constexpr (MODE == 1) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst + g * 256, 16, lane * 16, off, 0, 0);
        } else if constexpr (MODE == 2) {
          // the piece fetched an iteration ago goes to LDS, then its registers take the next one
          asm volatile("s_waitcnt vmcnt(4)" : "+v"(q[g])::"memory");
          *reinterpret_cast<f32x4*>(dst + g * 256 + lane * 4) = q[g];
          asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(q[g]) : "v"(lane * 16), "s"(rs), "s"(off) : "memory");
        } else if constexpr (MODE == 3) {
          asm volatile("s_waitcnt vmcnt(4)" : "+v"(q[g])::"memory");
          *reinterpret_cast<f32x4*>(dst + g * 256 + lane * 4) = q[g];
          const float* p = src + (off >> 2) + lane * 4;
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[g]) : "v"(p) : "memory");
        }
      }
      if constexpr (MODE == 5 || MODE == 6) {  // both waves idle a while after every group: the pipe at ~2/3 (5) or ~1/2 (6) duty
#pragma unroll
        for (int z = 0; z < (MODE == 5 ? 8 : 16); ++z) asm volatile("s_nop 15");
      }
      if constexpr (MODE == 4) {
        const int off = so + wave * 5 * 1024;
        if (g == 0) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, lane * 16, off, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst + 256, 16, lane * 16, off + 1024, 0, 0);
        }
        if (g == 1) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst + 512, 16, lane * 16, off + 2048, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst + 768, 16, lane * 16, off + 3072, 0, 0);
        }
        if (g == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst + 1024, 16, lane * 16, off + 4096, 0, 0);
      }
    }
    so += 40 * 1024 * 256;
    if (so >= span) so -= span;
    if (MODE == 1 || MODE == 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  for (int i = 0; i < 5; ++i) s += q[i][0];
  s += reinterpret_cast<float*>(smem)[tid];
  out[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
static void run(const char* name, float* out, unsigned long long* cyc, const float* src, int span_kib, int rnd = 0) {
  const int iters = 2000, G = 256;
  hipLaunchKernelGGL(probe<MODE>, dim3(G), dim3(512), 0, 0, out, cyc, iters, src, span_kib, rnd);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<MODE>, dim3(G), dim3(512), 0, 0, out, cyc, iters, src, span_kib, rnd);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(G * 8);
  hipMemcpy(h.data(), cyc, G * 64, hipMemcpyDeviceToHost);
  double m = 0, mn = 0;
  for (int b = 0; b < G; ++b) {
    unsigned long long hi = 0, lo = ~0ull;
    for (int w = 0; w < 8; ++w) hi = h[b * 8 + w] > hi ? h[b * 8 + w] : hi, lo = h[b * 8 + w] < lo ? h[b * 8 + w] : lo;
    m += (double)hi, mn += (double)lo;
  }
  printf("%-46s span %7d KiB: slowest wave %5.0f cycles / iteration (first done %5.0f); %.2f us / iteration, clock %.2f GHz\n",
         name, span_kib, m / G / iters, mn / G / iters, ms * 1e3 / iters, m / G / (ms * 1e6));
}

int main() {
  float *out, *src;
  unsigned long long* cyc;
  const size_t bytes = (size_t)1 << 31;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 64);
  hipMalloc(&src, bytes);
  hipMemset(src, 0, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    run<0>("0 no fetch", out, cyc, src, 10240);
    run<5>("5 no fetch, 128 idle cycles after each group", out, cyc, src, 10240);
    run<6>("6 no fetch, 256 idle cycles after each group", out, cyc, src, 10240);
    run<0>("0 no fetch", out, cyc, src, 10240);
    run<0>("0 no fetch, random operands", out, cyc, src, 10240, 1);
    run<5>("5 random operands, 128 idle cycles per group", out, cyc, src, 10240, 1);
    run<1>("1 LDS-DMA from HBM, random operands", out, cyc, src, 1 << 20, 1);
  }
  for (int span : {10240, 1 << 20}) {  // 10 MiB window (L2 / MALL resident after the first pass) and 1 GiB (HBM)
    run<0>("0 no fetch", out, cyc, src, span);
    run<1>("1 LDS-DMA, one piece per group", out, cyc, src, span);
    run<4>("4 LDS-DMA, pieces back to back (2, 2, 1)", out, cyc, src, span);
    run<2>("2 buffer_load -> VGPR, ds_write next iteration", out, cyc, src, span);
    run<3>("3 global_load -> VGPR, ds_write next iteration", out, cyc, src, span);
  }
  return 0;
}
