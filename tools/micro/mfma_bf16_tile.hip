// Microbenchmark: what v_mfma_f32_16x16x32_bf16 sustains on RANDOM operands in the instruction pattern of the 224x224 value
// gradient (ftv_kernels.hip): per "row block" 12 MFMAs over 3 A fragments x 6 B fragments into 2 accumulators, 8 row blocks =
// 96 MFMAs per "K tile", operands in registers only (no memory in the loop).  Variants: wave count per SIMD (1, 2, 4) and a
// register-only VALU block of V instructions per tile in front of the MFMAs (the split).  Prints ns per MFMA per SIMD, the
// in-kernel clock (s_memtime over s_memrealtime) and cycles per MFMA.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_bf16_tile mfma_bf16_tile.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;
using u32x4 = __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned;
using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;

template <int VALU>
__global__ __launch_bounds__(256) void tile_loop(const u32x4* __restrict__ src, float* out, unsigned long long* clocks, int tiles) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  u32x4 a[8][3], b[3][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int p = 0; p < 3; ++p) a[i][p] = src[(tid * 31 + i * 3 + p) & 4095];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int j = 0; j < 2; ++j) b[p][j] = src[(tid * 17 + p * 2 + j + 99) & 4095];
  f32x4 acc[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float v = (float)tid;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < tiles; ++t) {
#pragma unroll
    for (int k = 0; k < VALU; ++k) v = v * 1.0001f + 0.5f;  // dependent VALU chain in front of the tile's MFMAs
    b[0][0][0] ^= (__float_as_uint(v) & 1u);                // keep it alive without changing magnitudes
    constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i][pa[s]]), __builtin_bit_cast(bf16x8, b[pb[s]][j]),
                                                              acc[i][j], 0, 0, 0);
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = v;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[tid] = s;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int VALU>
void run(const u32x4* src, float* out, unsigned long long* clocks, int wgs_per_cu) {
  const int tiles = 4000, blocks = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(tile_loop<VALU>, dim3(blocks), dim3(256), 0, 0, src, out, clocks, 50);
  hipEventRecord(e0);
  hipLaunchKernelGGL(tile_loop<VALU>, dim3(blocks), dim3(256), 0, 0, src, out, clocks, tiles);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2];
  hipMemcpy(h, clocks, sizeof(h), hipMemcpyDeviceToHost);
  const double mfma_per_simd = (double)tiles * 96 * wgs_per_cu;
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);  // s_memrealtime ticks at 100 MHz
  printf("waves/SIMD=%d VALU/tile=%3d: %.3f ms  %.2f ns/MFMA/SIMD  in-kernel clock %.2f GHz  %.1f cycles/MFMA  %.0f TFLOP/s (useful bf16)\n",
         wgs_per_cu, VALU, ms, ms * 1e6 / mfma_per_simd, ghz, ms * 1e6 / mfma_per_simd * ghz,
         mfma_per_simd * 1024 * 16384 / (ms * 1e-3) / 1e12);
}

int main() {
  u32x4* src; float* out; unsigned long long* clocks;
  hipMalloc(&src, 4096 * sizeof(u32x4)); hipMalloc(&out, 256 * 4 * 256 * sizeof(float)); hipMalloc(&clocks, 2 * 1024 * sizeof(unsigned long long));
  unsigned* h = (unsigned*)malloc(4096 * 16);
  srand(1);
  for (int i = 0; i < 4096 * 4; ++i) {  // two random bf16 of magnitude ~1 per word
    const unsigned lo = 0x3f00u | (rand() & 0xffu) | ((rand() & 1u) << 15), hi = 0x3f00u | (rand() & 0xffu) | ((rand() & 1u) << 15);
    h[i] = lo | (hi << 16);
  }
  hipMemcpy(src, h, 4096 * 16, hipMemcpyHostToDevice);
  for (int w = 1; w <= 4; w *= 2) { run<0>(src, out, clocks, w); run<100>(src, out, clocks, w); run<400>(src, out, clocks, w); }
  return 0;
}
