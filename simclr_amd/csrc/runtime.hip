// C-ABI runtime glue: error reporting, version/arch queries and on-device layout probes
// used by the GPU test-suite to pin the MFMA / LDS-transpose lane mappings the kernels
// rely on (see tests/test_gpu_probes.py).
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void simclr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {
typedef __attribute__((ext_vector_type(4))) short s16x4;

// out[lane*4+reg] = D of mfma_16x16x32_bf16 with A[i][k] = a[i*32+k], B[k][j] = b[k*16+j]
__global__ void probe_mfma_bf16(const uint16_t* a, const uint16_t* b, float* out) {
  const int l = threadIdx.x, fl = l & 15, g = l >> 4;
  u32x4 av, bv;
  uint16_t ta[8], tb[8];
  for (int j = 0; j < 8; ++j) { ta[j] = a[fl * 32 + g * 8 + j]; tb[j] = b[(g * 8 + j) * 16 + fl]; }
  for (int j = 0; j < 4; ++j) {
    av[j] = ta[2 * j] | ((uint32_t)ta[2 * j + 1] << 16);
    bv[j] = tb[2 * j] | ((uint32_t)tb[2 * j + 1] << 16);
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
// the same with v_mfma_f32_16x16x32_f16 (fp16 bit patterns): the split-fp16 forward relies on its layout being the bf16 one AND on
// subnormal fp16 inputs taking part in the product (a lo piece below 2^-14 must not be flushed to zero)
__global__ void probe_mfma_f16(const uint16_t* a, const uint16_t* b, float* out) {
  typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
  const int l = threadIdx.x, fl = l & 15, g = l >> 4;
  u32x4 av, bv;
  uint16_t ta[8], tb[8];
  for (int j = 0; j < 8; ++j) { ta[j] = a[fl * 32 + g * 8 + j]; tb[j] = b[(g * 8 + j) * 16 + fl]; }
  for (int j = 0; j < 4; ++j) {
    av[j] = ta[2 * j] | ((uint32_t)ta[2 * j + 1] << 16);
    bv[j] = tb[2 * j] | ((uint32_t)tb[2 * j + 1] << 16);
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
// out[lane*4+reg] = D of mfma_16x16x4f32 with A[i][k] = a[i*4+k], B[k][j] = b[k*16+j]
__global__ void probe_mfma_f32(const float* a, const float* b, float* out) {
  const int l = threadIdx.x, fl = l & 15, g = l >> 4;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[fl * 4 + g], b[g * 16 + fl], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
// ds_read_b64_tr_b16: LDS holds in[0..1023]; lane l supplies byte address 2*addr[l]; out[l*4+j]
__global__ void probe_ds_read_tr16(const uint16_t* in, const int* addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr[l]));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}
}  // namespace

extern "C" {

const char* simclr_last_error(void) { return g_err; }
int simclr_abi_version(void) { return 8; }

// which: 0 = mfma bf16 16x16x32, 1 = mfma f32 16x16x4, 2 = ds_read_b64_tr_b16, 3 = mfma f16 16x16x32 (fp16 bit patterns)
int simclr_probe(int which, const void* a, const void* b, void* out, hipStream_t stream) {
  if (which == 0) hipLaunchKernelGGL(probe_mfma_bf16, dim3(1), dim3(64), 0, stream, (const uint16_t*)a, (const uint16_t*)b, (float*)out);
  else if (which == 1) hipLaunchKernelGGL(probe_mfma_f32, dim3(1), dim3(64), 0, stream, (const float*)a, (const float*)b, (float*)out);
  else if (which == 3) hipLaunchKernelGGL(probe_mfma_f16, dim3(1), dim3(64), 0, stream, (const uint16_t*)a, (const uint16_t*)b, (float*)out);
  else if (which == 2) hipLaunchKernelGGL(probe_ds_read_tr16, dim3(1), dim3(64), 0, stream, (const uint16_t*)a, (const int*)b, (uint16_t*)out);
  else { simclr_set_error("probe: unknown id %d", which); return 1; }
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
