// HBM streaming probe: what read+write bandwidth do different 16-byte/lane copy shapes reach on this GPU?
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/hbm_probe.hip -o gpurun_out/hbm_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_gridstride(const u32x4* __restrict__ a, u32x4* __restrict__ b, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += stride * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < n) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < n) { if (NT) __builtin_nontemporal_store(v[u], b + i + u * stride); else b[i + u * stride] = v[u]; }
  }
}
// every block owns a contiguous run of U*256 chunks
template <int U>
__global__ __launch_bounds__(256) void copy_blockcontig(const u32x4* __restrict__ a, u32x4* __restrict__ b, long long n) {
  const long long base = (long long)blockIdx.x * 256 * U + threadIdx.x;
  u32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = a[base + u * 256];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) b[base + u * 256] = v[u];
}
// persistent: grid of G blocks, each walks contiguous 256*U runs with stride G
template <int U>
__global__ __launch_bounds__(256) void copy_persist(const u32x4* __restrict__ a, u32x4* __restrict__ b, long long n) {
  const long long runs = (n + 256 * U - 1) / (256 * U);
  for (long long r = blockIdx.x; r < runs; r += gridDim.x) {
    const long long base = r * 256 * U + threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = a[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) b[base + u * 256] = v[u];
  }
}
// 2 reads + 1 write (residual-add shape) and read-only sum
template <int U>
__global__ __launch_bounds__(256) void add_gridstride(const u32x4* __restrict__ a, const u32x4* __restrict__ c, u32x4* __restrict__ b, long long n) {
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += stride * U) {
    u32x4 v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < n) { v[u] = a[i + u * stride]; w[u] = c[i + u * stride]; }
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < n) b[i + u * stride] = v[u] + w[u];
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  const long long bytes = 1644167168ll;      // 1024 x 56 x 56 x 256 bf16
  const long long n = bytes / 16;
  u32x4 *a, *b, *c;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes));
  CK(hipMemset(a, 0x3c, bytes)); CK(hipMemset(c, 0x3d, bytes)); CK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto bench = [&](const char* name, double traffic, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    float best = 1e9f;
    for (int i = 0; i < 5; ++i) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-40s %8.1f us  %6.2f TB/s\n", name, best * 1e3, traffic / (best * 1e-3) / 1e12);
    return 0;
  };
  const double rw = 2.0 * bytes, rrw = 3.0 * bytes;
  for (int g : {1024, 2048, 4096, 8192, 16384}) {
    char nm[64];
    snprintf(nm, 64, "gridstride U4 grid %d", g); bench(nm, rw, [&] { hipLaunchKernelGGL((copy_gridstride<4, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
  }
  bench("gridstride U1 grid 4096", rw, [&] { hipLaunchKernelGGL((copy_gridstride<1, false>), dim3(4096), dim3(256), 0, 0, a, b, n); });
  bench("gridstride U2 grid 4096", rw, [&] { hipLaunchKernelGGL((copy_gridstride<2, false>), dim3(4096), dim3(256), 0, 0, a, b, n); });
  bench("gridstride U8 grid 4096", rw, [&] { hipLaunchKernelGGL((copy_gridstride<8, false>), dim3(4096), dim3(256), 0, 0, a, b, n); });
  bench("gridstride U4 NT grid 4096", rw, [&] { hipLaunchKernelGGL((copy_gridstride<4, true>), dim3(4096), dim3(256), 0, 0, a, b, n); });
  bench("flat (1 chunk/thread)", rw, [&] { hipLaunchKernelGGL((copy_blockcontig<1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n); });
  bench("blockcontig U4", rw, [&] { hipLaunchKernelGGL((copy_blockcontig<4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, a, b, n); });
  bench("blockcontig U8", rw, [&] { hipLaunchKernelGGL((copy_blockcontig<8>), dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, 0, a, b, n); });
  for (int g : {512, 1024, 2048, 4096}) {
    char nm[64];
    snprintf(nm, 64, "persist U4 grid %d", g); bench(nm, rw, [&] { hipLaunchKernelGGL((copy_persist<4>), dim3(g), dim3(256), 0, 0, a, b, n); });
    snprintf(nm, 64, "persist U8 grid %d", g); bench(nm, rw, [&] { hipLaunchKernelGGL((copy_persist<8>), dim3(g), dim3(256), 0, 0, a, b, n); });
  }
  bench("add 2r+1w gridstride U4 grid 4096", rrw, [&] { hipLaunchKernelGGL((add_gridstride<4>), dim3(4096), dim3(256), 0, 0, a, c, b, n); });
  bench("add 2r+1w gridstride U2 grid 8192", rrw, [&] { hipLaunchKernelGGL((add_gridstride<2>), dim3(8192), dim3(256), 0, 0, a, c, b, n); });
  { hipEventRecord(e0); for (int i = 0; i < 5; ++i) hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-40s %8.1f us  %6.2f TB/s\n", "hipMemcpyAsync D2D", ms / 5 * 1e3, rw / (ms / 5 * 1e-3) / 1e12); }
  return 0;
}
