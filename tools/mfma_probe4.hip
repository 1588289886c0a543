// Empirical operand / result lane map of v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4x4, K = 1).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(int la, int lb, float* out) {
  const int l = threadIdx.x;
  f32x4 c{0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(l == la ? 1.f : 0.f, l == lb ? 1.f : 0.f, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  float* d;
  hipMalloc(&d, 256 * sizeof(float));
  float h[256];
  int shown = 0;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, la, lb, d);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      for (int i = 0; i < 256; ++i)
        if (h[i] != 0.f && (la < 9 || la % 13 == 0) && shown < 200) {
          printf("A lane %2d x B lane %2d -> D lane %2d reg %d\n", la, lb, i / 4, i % 4);
          ++shown;
        }
    }
  return 0;
}
