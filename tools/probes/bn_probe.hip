// Probe: BatchNorm-apply shaped streaming kernels (bf16, per-channel scale/shift) in different thread->data shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk(float a, float b) { f2 v = {a, b}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2)); }
__device__ __forceinline__ u32x4 bn8(const u32x4 x, const float* sc, const float* sh) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = __uint_as_float(x[i] << 16), b = __uint_as_float(x[i] & 0xffff0000u);
    a = fmaxf(fmaf(a, sc[2 * i], sh[2 * i]), 0.f); b = fmaxf(fmaf(b, sc[2 * i + 1], sh[2 * i + 1]), 0.f);
    o[i] = pk(a, b);
  }
  return o;
}
__device__ __forceinline__ void ldp(const float* p, int c0, float* o) {
  const float4 a = *(const float4*)(p + c0), b = *(const float4*)(p + c0 + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
// current library shape: grid-stride, fixed channel chunk per thread, U rows in flight
template <int U>
__global__ __launch_bounds__(256) void bn_gridstride(const u32x4* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, u32x4* __restrict__ y, long long rows, int cpr) {
  const long long gtid = blockIdx.x * 256ll + threadIdx.x;
  const long long rstride = (gridDim.x * 256ll) / cpr;
  const int cc = (int)(gtid % cpr);
  float sc[8], sh[8]; ldp(scale, cc * 8, sc); ldp(shift, cc * 8, sh);
  for (long long r = gtid / cpr; r < rows; r += U * rstride) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (r + u * rstride < rows) v[u] = x[(r + u * rstride) * cpr + cc];
#pragma unroll
    for (int u = 0; u < U; ++u) if (r + u * rstride < rows) y[(r + u * rstride) * cpr + cc] = bn8(v[u], sc, sh);
  }
}
// flat: one chunk per thread, parameters fetched per thread
__global__ __launch_bounds__(256) void bn_flat(const u32x4* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, u32x4* __restrict__ y, long long n, int cpr) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= n) return;
  const int cc = (int)(i % cpr);
  const u32x4 v = x[i];
  float sc[8], sh[8]; ldp(scale, cc * 8, sc); ldp(shift, cc * 8, sh);
  y[i] = bn8(v, sc, sh);
}
// block-contiguous: each block owns U*256 consecutive chunks; cpr divides 256 => fixed channel chunk per thread
template <int U>
__global__ __launch_bounds__(256) void bn_blockcontig(const u32x4* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift, u32x4* __restrict__ y, long long n, int cpr) {
  const long long base = (long long)blockIdx.x * 256 * U + threadIdx.x;
  const int cc = threadIdx.x % cpr;      // 256 % cpr == 0
  u32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = x[base + u * 256];
  float sc[8], sh[8]; ldp(scale, cc * 8, sc); ldp(shift, cc * 8, sh);
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) y[base + u * 256] = bn8(v[u], sc, sh);
}
// residual form 2r + 1w
template <int U>
__global__ __launch_bounds__(256) void bnres_blockcontig(const u32x4* __restrict__ x, const u32x4* __restrict__ r, const float* __restrict__ scale, const float* __restrict__ shift, u32x4* __restrict__ y, long long n, int cpr) {
  const long long base = (long long)blockIdx.x * 256 * U + threadIdx.x;
  const int cc = threadIdx.x % cpr;
  u32x4 v[U], w[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) { v[u] = x[base + u * 256]; w[u] = r[base + u * 256]; }
  float sc[8], sh[8]; ldp(scale, cc * 8, sc); ldp(shift, cc * 8, sh);
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) {
    u32x4 o = bn8(v[u], sc, sh);
    y[base + u * 256] = o + w[u];
  }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
  const int C = 256, cpr = C / 8;
  const long long rows = 1024ll * 56 * 56, n = rows * cpr, bytes = n * 16;
  u32x4 *a, *b, *c; float *sc, *sh;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&sc, 4096 * 4)); CK(hipMalloc(&sh, 4096 * 4));
  CK(hipMemset(a, 0x3c, bytes)); CK(hipMemset(c, 0x3d, bytes)); CK(hipMemset(sc, 0, 4096 * 4)); CK(hipMemset(sh, 0, 4096 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto bench = [&](const char* name, double traffic, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    float best = 1e9f;
    for (int i = 0; i < 5; ++i) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, best * 1e3, traffic / (best * 1e-3) / 1e12);
  };
  const double rw = 2.0 * bytes, rrw = 3.0 * bytes;
  bench("bn gridstride U4 grid 4096 (library)", rw, [&] { hipLaunchKernelGGL((bn_gridstride<4>), dim3(4096), dim3(256), 0, 0, a, sc, sh, b, rows, cpr); });
  bench("bn flat", rw, [&] { hipLaunchKernelGGL(bn_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, sc, sh, b, n, cpr); });
  bench("bn blockcontig U1", rw, [&] { hipLaunchKernelGGL((bn_blockcontig<1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, sc, sh, b, n, cpr); });
  bench("bn blockcontig U2", rw, [&] { hipLaunchKernelGGL((bn_blockcontig<2>), dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, a, sc, sh, b, n, cpr); });
  bench("bn blockcontig U4", rw, [&] { hipLaunchKernelGGL((bn_blockcontig<4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, a, sc, sh, b, n, cpr); });
  bench("bn+res blockcontig U1 (2r+1w)", rrw, [&] { hipLaunchKernelGGL((bnres_blockcontig<1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, c, sc, sh, b, n, cpr); });
  bench("bn+res blockcontig U2 (2r+1w)", rrw, [&] { hipLaunchKernelGGL((bnres_blockcontig<2>), dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, a, c, sc, sh, b, n, cpr); });
  bench("bn+res blockcontig U4 (2r+1w)", rrw, [&] { hipLaunchKernelGGL((bnres_blockcontig<4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, a, c, sc, sh, b, n, cpr); });
  // smaller tensor (14x14 C1024: 411 MB) to see the dispatch-rate effect
  const long long n2 = 1024ll * 14 * 14 * 128;
  bench("bn blockcontig U1, 411 MB C1024", 2.0 * n2 * 16, [&] { hipLaunchKernelGGL((bn_blockcontig<1>), dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, 0, a, sc, sh, b, n2, 128); });
  bench("bn blockcontig U2, 411 MB C1024", 2.0 * n2 * 16, [&] { hipLaunchKernelGGL((bn_blockcontig<2>), dim3((unsigned)((n2 + 511) / 512)), dim3(256), 0, 0, a, sc, sh, b, n2, 128); });
  bench("bn gridstride U4 grid 4096, 411 MB", 2.0 * n2 * 16, [&] { hipLaunchKernelGGL((bn_gridstride<4>), dim3(4096), dim3(256), 0, 0, a, sc, sh, b, 1024ll * 14 * 14, 128); });
  return 0;
}
