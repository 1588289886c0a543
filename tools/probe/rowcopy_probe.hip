// Probe (round 4): what does the ACCESS PATTERN of the depthwise kernel cost?  Copies [N][H][ldt] fp32 rows (N = 32, H = 256,
// T = 3999, ldt = 4224: 131 MB in, 131 MB out -- the benchmark's hidden map) with four work decompositions and reports TB/s:
//   0 linear      flat grid-stride copy, 16 bytes per lane (the ceiling: tools/copy_ceiling.py's device copy)
//   1 dwconv      grid (4, 16, 32), a workgroup = 16 rows x 1024 frames, wave w takes rows w, w+4, w+8, w+12 one after the
//                 other, 4 x 16 bytes per lane per row, eight workgroups per CU (dwconv_wave_kernel's decomposition)
//   2 whole rows  grid (16, 32): wave w takes rows w, w+4, ... of its 16 and streams each row's 4 pieces back to back
//   3 long tiles  grid (64, 32): a workgroup = 4 rows x all frames, one row per wave
//   hipcc -O3 --offload-arch=gfx950 tools/probe/rowcopy_probe.hip -o tools/probe/rowcopy_probe && tools/probe/rowcopy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 32, H = 256, T = 3999, LDT = 4224;

__global__ __launch_bounds__(256) void k_linear(const f32x4* s, f32x4* d, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

template <int MODE>
__global__ __launch_bounds__(256, 8) void k_rows(const float* s, float* d) {
  __shared__ float pad[MODE == 1 ? 5120 : 64];  // (mode 1: the kernel's 20 KiB of strips, i.e. its occupancy)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 1000) pad[0] = 0.f;
  if constexpr (MODE == 1) {
    const int t0 = blockIdx.x * 1024, h0 = blockIdx.y * 16, n = blockIdx.z;
    f32x4 v[4];
    auto ld = [&](int r) {
      const size_t row = ((size_t)n * H + h0 + wave + 4 * r) * LDT;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int f = t0 + (k * 64 + lane) * 4;
        v[k] = f < T ? *reinterpret_cast<const f32x4*>(s + row + f) : f32x4{0, 0, 0, 0};
      }
    };
    auto st = [&](int r) {
      const size_t row = ((size_t)n * H + h0 + wave + 4 * r) * LDT;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int f = t0 + (k * 64 + lane) * 4;
        if (f < T) *reinterpret_cast<f32x4*>(d + row + f) = v[k];
      }
    };
    for (int r = 0; r < 4; ++r) { ld(r); st(r); }
  } else {
    // MODE 2: blockIdx.x = row group of 16, wave takes 4 rows; MODE 3: blockIdx.x = row group of 4, wave takes 1 row
    const int n = blockIdx.y;
    const int rows = MODE == 2 ? 4 : 1;
    for (int r = 0; r < rows; ++r) {
      const int h = MODE == 2 ? blockIdx.x * 16 + wave + 4 * r : blockIdx.x * 4 + wave;
      const size_t row = ((size_t)n * H + h) * LDT;
      f32x4 v[2][4];
      auto ld = [&](int piece, int q) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int f = piece * 1024 + (k * 64 + lane) * 4;
          v[q][k] = f < T ? *reinterpret_cast<const f32x4*>(s + row + f) : f32x4{0, 0, 0, 0};
        }
      };
      auto st = [&](int piece, int q) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int f = piece * 1024 + (k * 64 + lane) * 4;
          if (f < T) *reinterpret_cast<f32x4*>(d + row + f) = v[q][k];
        }
      };
      ld(0, 0);
      ld(1, 1);
      st(0, 0);
      ld(2, 0);
      st(1, 1);
      ld(3, 1);
      st(2, 0);
      st(3, 1);
    }
  }
}

int main() {
  const size_t elems = (size_t)N * H * LDT;
  float *a, *b;
  hipMalloc(&a, elems * 4 * 2);
  hipMalloc(&b, elems * 4 * 2);
  hipMemset(a, 0, elems * 8);
  hipMemset(b, 0, elems * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const double bytes = 2.0 * N * H * T * 4;
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9f, sum = 0.f;
    for (int it = 0; it < 24; ++it) {
      const float* s = a + (it & 1) * elems;
      float* d = b + (it & 1) * elems;
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k_linear, dim3(256 * 8), dim3(256), 0, 0, (const f32x4*)s, (f32x4*)d, elems / 4);
      if (mode == 1) hipLaunchKernelGGL(k_rows<1>, dim3(4, 16, 32), dim3(256), 0, 0, s, d);
      if (mode == 2) hipLaunchKernelGGL(k_rows<2>, dim3(16, 32), dim3(256), 0, 0, s, d);
      if (mode == 3) hipLaunchKernelGGL(k_rows<3>, dim3(64, 32), dim3(256), 0, 0, s, d);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (it >= 4) { sum += ms; best = ms < best ? ms : best; }
    }
    const char* names[4] = {"linear", "dwconv pattern", "whole rows per wave", "4 rows x all frames per workgroup"};
    printf("%-36s avg %.1f us  best %.1f us  %.2f TB/s (of the %s bytes)\n", names[mode], sum / 20 * 1e3, best * 1e3,
           bytes / (sum / 20 * 1e-3) / 1e12, mode == 0 ? "padded-row" : "valid-frame");
  }
  return 0;
}
