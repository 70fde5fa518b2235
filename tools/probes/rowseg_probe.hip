// Row-segment probe: does the SHAPE of a tile's global accesses explain why the big-output short-K convolutions
// (conv3 with the fused BatchNorm apply, the dgrad of a block's first 1x1 convolution) stream at 4.2-4.9 TB/s while the
// layers whose tile covers whole NHWC rows reach 5.4-5.5 (profiles/r03_notes.md)?
// A persistent grid of G workgroups walks (row tile, column segment) pairs of a [M][C] bf16 tensor exactly like
// conv_igemm_persistent does: per tile it READS a residual piece of 128 rows x SEG bytes, and WRITES the same piece of the
// output -- with the segment index either FIXED per workgroup (today's mapping: the sibling segments of a row are written
// by other CUs at other times) or LOOPED inside the workgroup (the whole row leaves one CU back to back).  SEG = row
// bytes is the "tile covers the row" case.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/rowseg_probe.hip -o gpurun_out/rowseg_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// rows per tile 128; SEGC = 16-byte chunks per segment row (8 = 128 B, 16 = 256 B, 32 = 512 B); 256 threads
template <int SEGC, bool NLOOP>
__global__ __launch_bounds__(256) void seg_copy(const u32x4* __restrict__ res, u32x4* __restrict__ out, int M, int rowc /* chunks per row */) {
  const int nseg = rowc / SEGC, m_tiles = (M + 127) / 128;
  const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
  const int cc = threadIdx.x % SEGC, r0 = threadIdx.x / SEGC;
  constexpr int RPP = 256 / SEGC, ER = 128 / RPP;           // rows per pass, rows per thread and tile
  if (NLOOP) {
    const int mslots = gridDim.x;                             // every workgroup is an M slot and loops over the segments
    for (int mt = (l * 8 + xcd); mt < m_tiles; mt += mslots)
      for (int s = 0; s < nseg; ++s) {
        u32x4 v[ER];
#pragma unroll
        for (int i = 0; i < ER; ++i) {
          const long long row = (long long)mt * 128 + r0 + i * RPP;
          v[i] = row < M ? res[row * rowc + s * SEGC + cc] : (u32x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < ER; ++i) {
          const long long row = (long long)mt * 128 + r0 + i * RPP;
          if (row < M) out[row * rowc + s * SEGC + cc] = v[i] + (u32x4){1, 1, 1, 1};
        }
      }
  } else {
    const int s = l % nseg, mslots = gridDim.x / nseg;        // fixed segment per workgroup, siblings on the same XCD
    for (int mt = (l / nseg) * 8 + xcd; mt < m_tiles; mt += mslots) {
      u32x4 v[ER];
#pragma unroll
      for (int i = 0; i < ER; ++i) {
        const long long row = (long long)mt * 128 + r0 + i * RPP;
        v[i] = row < M ? res[row * rowc + s * SEGC + cc] : (u32x4){0, 0, 0, 0};
      }
#pragma unroll
      for (int i = 0; i < ER; ++i) {
        const long long row = (long long)mt * 128 + r0 + i * RPP;
        if (row < M) out[row * rowc + s * SEGC + cc] = v[i] + (u32x4){1, 1, 1, 1};
      }
    }
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int SEGC, bool NLOOP>
static int run(const char* name, const u32x4* a, u32x4* b, int M, int rowc, int grid) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((seg_copy<SEGC, NLOOP>), dim3(grid), dim3(256), 0, 0, a, b, M, rowc);
  CK(hipEventRecord(e0));
  const int it = 5;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((seg_copy<SEGC, NLOOP>), dim3(grid), dim3(256), 0, 0, a, b, M, rowc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = 2.0 * M * rowc * 16.0;
  printf("%-44s grid %5d  %8.1f us  %5.2f TB/s\n", name, grid, ms / it * 1e3, bytes / (ms / it * 1e-3) / 1e12);
  return 0;
}

int main() {
  const int M = 1024 * 56 * 56;
  for (int C : {256, 512}) {                      // 56^2 x 256 channels (512-byte rows), and a 1 KB row
    const int rowc = C * 2 / 16;
    u32x4 *a, *b;
    CK(hipMalloc(&a, (size_t)M * rowc * 16)); CK(hipMalloc(&b, (size_t)M * rowc * 16));
    CK(hipMemset(a, 1, (size_t)M * rowc * 16));
    printf("[M = %d rows x %d bytes]\n", M, C * 2);
    for (int grid : {512, 768, 1024}) {
      run<8, false>("128 B segments, fixed per workgroup", a, b, M, rowc, grid / (8 * (rowc / 8)) * 8 * (rowc / 8));
      run<8, true>("128 B segments, looped in the workgroup", a, b, M, rowc, grid);
      run<16, false>("256 B segments, fixed per workgroup", a, b, M, rowc, grid / (8 * (rowc / 16)) * 8 * (rowc / 16));
      run<16, true>("256 B segments, looped in the workgroup", a, b, M, rowc, grid);
      if (rowc == 32) run<32, true>("512 B = whole rows", a, b, M, rowc, grid);
    }
    CK(hipFree(a)); CK(hipFree(b));
  }
  return 0;
}
