// Store-pattern probe (development aid): how fast does a streaming kernel write / read-modify-write an [M][N] fp32 tensor when a wave
// instruction covers 16 rows x 64 bytes (the MFMA accumulator layout the fp32 epilogues store from: lane (fl, g) = row fl, channels
// 4g..4g+3 of a 16-column fragment) against 1 KB of consecutive bytes?   hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int PAT, bool RMW, int N>
__global__ __launch_bounds__(256) void k(float* __restrict__ y, long long M) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, fl = lane & 15;
  constexpr int NI = N / 16 > 4 ? 4 : N / 16;       // fragments per wave row (64 columns); N = 256: the workgroup's N-tile is blockIdx.y
  const long long tiles = (M + 127) / 128;
  const int n0 = blockIdx.y * 64;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    const long long r0 = t * 128 + wave * 32;
    if (PAT == 0) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          float4* p = (float4*)(y + (r0 + mi * 16 + fl) * N + n0 + ni * 16 + g * 4);
          float4 v = make_float4(1.f, 2.f, 3.f, (float)t);
          if (RMW) { float4 o = *p; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
          *p = v;
        }
    } else {
      // 32 rows x 64 columns of this wave as 8 instructions of 4 full 256-byte row pieces each (lane = 16-byte chunk of a row piece)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float4* p = (float4*)(y + (r0 + j * 4 + g) * N + n0 + fl * 4);
        float4 v = make_float4(1.f, 2.f, 3.f, (float)t);
        if (RMW) { float4 o = *p; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *p = v;
      }
    }
  }
}
template <int PAT, bool RMW, int N> float run(float* y, long long M, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  dim3 gr(grid, N / 64);
  hipLaunchKernelGGL((k<PAT, RMW, N>), gr, dim3(256), 0, 0, y, M);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<PAT, RMW, N>), gr, dim3(256), 0, 0, y, M);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
  const long long M = 1024ll * 112 * 112;           // the stem's output rows; N = 64 -> 3.29 GB
  float* y; hipMalloc(&y, (size_t)M * 256 * 4 / 4); // N = 256 runs use M / 4 rows (56^2): same bytes
  hipMemset(y, 0, (size_t)M * 64 * 4);
  const double gb = (double)M * 64 * 4 / 1e9;
  for (int grid : {512, 768, 2048}) {
    float t;
    t = run<0, false, 64>(y, M, grid);  printf("grid %4d  N=64  store  16 rows x 64 B : %.3f ms  %.0f GB/s\n", grid, t, gb / t * 1e3);
    t = run<1, false, 64>(y, M, grid);  printf("grid %4d  N=64  store  4 rows x 256 B : %.3f ms  %.0f GB/s\n", grid, t, gb / t * 1e3);
    t = run<0, true, 64>(y, M, grid);   printf("grid %4d  N=64  rmw    16 rows x 64 B : %.3f ms  %.0f GB/s (r+w)\n", grid, t, 2 * gb / t * 1e3);
    t = run<1, true, 64>(y, M, grid);   printf("grid %4d  N=64  rmw    4 rows x 256 B : %.3f ms  %.0f GB/s (r+w)\n", grid, t, 2 * gb / t * 1e3);
    t = run<0, false, 256>(y, M / 4, grid / 4);  printf("grid %4d  N=256 store 16 rows x 64 B : %.3f ms  %.0f GB/s\n", grid, t, gb / t * 1e3);
    t = run<1, false, 256>(y, M / 4, grid / 4);  printf("grid %4d  N=256 store 4 rows x 256 B : %.3f ms  %.0f GB/s\n", grid, t, gb / t * 1e3);
    t = run<0, true, 256>(y, M / 4, grid / 4);   printf("grid %4d  N=256 rmw   16 rows x 64 B : %.3f ms  %.0f GB/s (r+w)\n", grid, t, 2 * gb / t * 1e3);
    t = run<1, true, 256>(y, M / 4, grid / 4);   printf("grid %4d  N=256 rmw   4 rows x 256 B : %.3f ms  %.0f GB/s (r+w)\n", grid, t, 2 * gb / t * 1e3);
  }
  return 0;
}
