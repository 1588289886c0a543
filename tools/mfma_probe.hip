// Probe: sustained v_mfma_f32_32x32x2_f32 rate under the conditions the conv1x1 kernel creates.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS, bool BARRIER>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters, const float* in) {
  __shared__ float As[2][16][256];
  __shared__ float Bs[2][16][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 16 * 256; i += 256) (&As[0][0][0])[i] = in[i & 1023];
  for (int i = tid; i < 2 * 16 * 128; i += 256) (&Bs[0][0][0])[i] = in[(i * 7) & 1023];
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float av = in[lane], bv = in[lane + 64];
  const int lr = lane & 31, lk = lane >> 5;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float a0 = av, a1 = av, b0 = bv, b1 = bv, b2 = bv, b3 = bv;
      if (LDS) {
        const int k = 2 * kk + lk;
        a0 = As[buf][k][wave * 64 + lr];
        a1 = As[buf][k][wave * 64 + 32 + lr];
        b0 = Bs[buf][k][lr];
        b1 = Bs[buf][k][32 + lr];
        b2 = Bs[buf][k][64 + lr];
        b3 = Bs[buf][k][96 + lr];
      }
      const float aa[2] = {a0, a1};
      const float bb[4] = {b0, b1, b2, b3};
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[j % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa[j >> 2], bb[j & 3], acc[j % NACC], 0, 0, 0);
    }
    if (BARRIER) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int NACC, bool LDS, bool BARRIER>
void run(const char* name, int blocks, float* out, float* in) {
  const int iters = 512;  // 512 * 64 MFMAs per wave
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NACC, LDS, BARRIER>), dim3(blocks), dim3(256), 0, 0, out, iters, in);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4 * iters * 64 * 4096.0;
  printf("%-34s blocks=%4d  %.3f ms  %.1f TF\n", name, blocks, ms, flop / ms / 1e9);
}

int main() {
  float *out, *in;
  hipMalloc(&out, 4096 * 256 * 4);
  hipMalloc(&in, 4096);
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
  for (int blocks : {256, 512, 1024}) {
    run<8, false, false>("8acc regs only", blocks, out, in);
    run<4, false, false>("4acc regs only", blocks, out, in);
    run<8, true, false>("8acc + LDS reads", blocks, out, in);
    run<8, true, true>("8acc + LDS reads + barrier/64", blocks, out, in);
    run<8, false, true>("8acc + barrier/64", blocks, out, in);
  }
  return 0;
}
