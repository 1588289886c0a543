// Probe (round 4): does the chip hold a higher clock on v_mfma_f32_16x16x32_f16 than on v_mfma_f32_32x32x16_f16 for the
// same work?  Same output tile per wave (256 rows x 32 columns), same K per iteration (32), the fp16x2 GEMM's three
// products per multiply-add, operands in registers (random fp16 values, a different register set per row block), two
// waves per SIMD, one 512-thread workgroup per CU.  Reports wall time per iteration and the in-kernel clock.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_shape_probe.hip -o tools/probe/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE, bool LDSA>
__global__ __launch_bounds__(512, 1) void probe(float* out, int iters, const f16x8* in, unsigned long long* stamps) {
  __shared__ f16x8 As[2 * 2 * 256 * 2];  // [slot][plane][k-half][256 rows] x 16 B = 32 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2 * 2 * 256 * 2; i += 512) As[i] = in[(i * 5 + 1) & 4095];
  __syncthreads();
  f16x8 a0[8], a1[8], b0[2], b1[2];
  for (int i = 0; i < 8; ++i) a0[i] = in[(tid * 8 + i) & 4095], a1[i] = in[(tid * 8 + i + 2048) & 4095];
  for (int i = 0; i < 2; ++i) b0[i] = in[(tid * 2 + i + 512) & 4095], b1[i] = in[(tid * 2 + i + 1536) & 4095];
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[8];
    for (int a = 0; a < 8; ++a)
      for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const f16x8* ap = As + ((it & 1) * 2) * 512 + kh * 256 + (lane >> 5) * 0 + (lane & 31);
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) {
          f16x8 w0 = a0[rb], w1 = a1[rb];
          if constexpr (LDSA) w0 = ap[rb * 32], w1 = ap[512 + rb * 32];
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b0[kh], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, b1[kh], acc[rb], 0, 0, 0);
          acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, b0[kh], acc[rb], 0, 0, 0);
        }
      }
      if constexpr (LDSA) __builtin_amdgcn_s_barrier();
    }
    for (int a = 0; a < 8; ++a)
      for (int r = 0; r < 16; ++r) s += acc[a][r];
  } else {
    f32x4 acc[16][2];
    for (int a = 0; a < 16; ++a)
      for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 4; ++r) acc[a][c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const f16x8* ap = As + ((it & 1) * 2) * 512 + (lane & 15) + (lane >> 4) * 64;
#pragma unroll
      for (int rb = 0; rb < 16; ++rb) {
        f16x8 w0 = a0[rb & 7], w1 = a1[rb & 7];
        if constexpr (LDSA) w0 = ap[rb * 16], w1 = ap[512 + rb * 16];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, b0[c], acc[rb][c], 0, 0, 0);
          acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, b1[c], acc[rb][c], 0, 0, 0);
          acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, b0[c], acc[rb][c], 0, 0, 0);
        }
      }
      if constexpr (LDSA) __builtin_amdgcn_s_barrier();
    }
    for (int a = 0; a < 16; ++a)
      for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 4; ++r) s += acc[a][c][r];
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
  out[blockIdx.x * 512 + tid] = s;
  if (tid == 0) stamps[blockIdx.x * 2] = t1 - t0, stamps[blockIdx.x * 2 + 1] = r1 - r0;
}

static int cmp(const void* a, const void* b) {
  const double x = *(const double*)a, y = *(const double*)b;
  return x < y ? -1 : x > y;
}

template <int SHAPE, bool LDSA>
void run(const char* name, float* out, f16x8* in, unsigned long long* st_d) {
  const int iters = 20000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 4; ++rep) {  // ~2 s of back-to-back launches, the last one timed
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<SHAPE, LDSA>), dim3(blocks), dim3(512), 0, 0, out, iters, in, st_d);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  static unsigned long long st[512];
  hipMemcpy(st, st_d, sizeof(st), hipMemcpyDeviceToHost);
  static double clk[256];
  for (int i = 0; i < 256; ++i) clk[i] = (double)st[2 * i] / (double)st[2 * i + 1] / 10.0;
  qsort(clk, 256, sizeof(double), cmp);
  const double cyc = (double)st[0] / iters;
  // fp32-equivalent FLOP: 2 * 256 * 32 * 32 per wave and iteration, 8 waves, 256 CUs (three MFMA products each)
  const double tf = 2.0 * 256 * 32 * 32 * 8 * 256 * iters / (ms * 1e-3) / 1e12;
  printf("%-44s %7.3f us/iter  %6.0f cycles/iter  clock %.2f GHz (median)  %6.1f TF fp32-equivalent (x3 products)\n", name,
         ms * 1e3 / iters, cyc, clk[128], tf);
}

int main() {
  float* out;
  f16x8* in;
  unsigned long long* st;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&in, 4096 * 16);
  hipMalloc(&st, 512 * 8);
  static _Float16 h[4096 * 8];
  srand(1);
  for (int i = 0; i < 4096 * 8; ++i) h[i] = (_Float16)(((rand() & 0xffff) / 32768.0f - 1.0f) * (1.0f + (rand() & 7)));
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int round = 0; round < 2; ++round) {
    run<32, false>("32x32x16 f16, operands in registers", out, in, st);
    run<16, false>("16x16x32 f16, operands in registers", out, in, st);
    run<32, true>("32x32x16 f16, A fragments from LDS + barrier", out, in, st);
    run<16, true>("16x16x32 f16, A fragments from LDS + barrier", out, in, st);
  }
  return 0;
}
