// Shared device helpers for the gfx950 (MI355X, CDNA4) SimCLR hot-path kernels.
// Wave = 64 lanes everywhere.  No portability layer: this code targets gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define SIMCLR_DT_F32 0
#define SIMCLR_DT_BF16 1

// ---- error plumbing (C-ABI: int return codes + simclr_last_error()) ----------
extern "C" const char* simclr_last_error(void);
void simclr_set_error(const char* fmt, ...);

#define SIMCLR_CHECK_ARG(cond, ...)                 \
  do {                                              \
    if (!(cond)) {                                  \
      simclr_set_error(__VA_ARGS__);                \
      return 1;                                     \
    }                                               \
  } while (0)

// SIMCLR_DRY_RUN=1 (tools/instantiation_table.py, a machine without a GPU): launch DECISIONS only -- nothing reaches the device and
// no launch status is checked.  Read once per process and announced loudly on stderr: a variable that leaks into a real run must
// not silently turn every convolution into a no-op.
static inline bool simclr_dry_run() {
  static const int on = [] {
    const char* e = getenv("SIMCLR_DRY_RUN");
    const int v = (e && e[0] == '1') ? 1 : 0;
    if (v) fprintf(stderr, "[simclr] SIMCLR_DRY_RUN=1: kernel launches are SKIPPED and unchecked in this process (decision table mode)\n");
    return v;
  }();
  return on != 0;
}
#define SIMCLR_CHECK_LAUNCH()                                        \
  do {                                                               \
    if (simclr_dry_run()) break;           /* decisions only: no device (conv.hip) */ \
    hipError_t e__ = hipGetLastError();                              \
    if (e__ != hipSuccess) {                                         \
      simclr_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, \
                       hipGetErrorString(e__));                      \
      return 2;                                                      \
    }                                                                \
  } while (0)

// ---- bf16 <-> f32 (round-to-nearest-even, NaN preserved) -----------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
  return __uint_as_float(((uint32_t)b) << 16);
}
// fp32 -> bf16, round-to-nearest-even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, one
// instruction per PAIR); the integer emulation costs ~9 VALU instructions per value and made the
// conv epilogues and the BN kernels instruction-bound.
typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float hw_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  return __builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const hw_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_bf16x2_t));
}

// ---- pre-split block format ("PS") of an fp32 tensor [rows][C], C a multiple of 32 -----------------------------------
// The fast parity mode multiplies fp32 operands as three 16-bit-piece MFMA terms (csrc/conv.hip mma_f32_chunks): x = hi + lo,
// hi = rn16(x), lo = rn16(x - hi), pieces bf16 (8-bit, fp32's range: gradients) or fp16 (11-bit, |x| < 65504: forward operands).
// A tensor whose only heavy consumers are those GEMMs is kept PRE-SPLIT by its (elementwise) producer: same 4 bytes per
// element, same addresses per 128-byte block of 32 channels, but the block holds eight 16-byte chunks of pieces instead of
// eight chunks of floats -- chunk g (g < 4) = the hi pieces of channels {4g..4g+3, 16+4g..16+4g+3} (what lane group g of a
// k-step reads from the fp32 block: chunks g and 4 + g), chunk 4 + g = their lo pieces.  The GEMM's LDS-DMA stream, swizzle
// and fragment reads do not change; its k-loop loses the splitting work (2.5 - 3 VALU instructions per element).
// Elementwise view: the 16-byte fp32 chunk i of the logical tensor (4 consecutive channels) lives as one 8-byte hi quad at
// byte ps_quad_offset(i) and one 8-byte lo quad 64 bytes further.
#define SIMCLR_FMT_PS_IN 0x100    // dtype flag of an entry point: the (first) tensor operand is pre-split
#define SIMCLR_FMT_PS_OUT 0x200   // dtype flag: the output tensor is written pre-split
#define SIMCLR_FMT_PS_F16 0x400   // the pieces are fp16 (default: bf16)
#define SIMCLR_FMT_PS_IN2 0x800   // the second tensor operand (wgrad: x; bn_apply: res) is pre-split
#define SIMCLR_FMT_PS_W 0x100000  // the WEIGHT operand of a forward / data-gradient call is the caller's pre-split copy (simclr_presplit_weights_multi)
typedef _Float16 hw_f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  const hw_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, hw_f16x2_t));      // v_cvt_pk_f16_f32 (round to nearest even)
}
template <bool F16> __device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& h, uint32_t& l) {
  if constexpr (F16) {
    // lo = fp16(x - fp32(hi)) in ONE mixed-precision fma per element (v_fma_mixlo / mixhi_f16: fp16 source half, fp32 addend, the exact
    // sum rounded once to fp16 -- x - hi is exact in fp32, so this IS the rounding of the two-step form): 3 VALU per pair instead of the
    // compiler's 5 (v_cvt_pk_f16_f32, 2 x v_cvt_f32_f16, v_pk_add_f32, v_cvt_pk_f16_f32); the split sits in the forward k-loops.
    // Checked against torch's float16 rounding over 40 binades (tests: presplit_weight_pieces_equal_torch_rounding).
    h = pack_f16x2(x0, x1);
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(x1));
  } else {
    h = pack_bf16x2(x0, x1);
    l = pack_bf16x2(x0 - __uint_as_float(h << 16), x1 - __uint_as_float(h & 0xffff0000u));
  }
}
__device__ __forceinline__ long long ps_quad_offset(long long i) { return (i >> 3) * 128 + (i & 3) * 16 + ((i >> 2) & 1) * 8; }
template <bool F16, bool NT = false> __device__ __forceinline__ void ps_store_quad(void* base, long long i, const float* v) {
  u32x2 h, l;
  uint32_t a, b;
  // The values are pinned in registers first: inlined next to its producer (x = s * t) the residual x - hi would otherwise be
  // contracted into fma(s, t, -hi) -- the residual of the UNROUNDED product, a different (if slightly better) lo than the split of
  // the fp32 value a plain store would have written.  The format is specified as exactly that split (HIP's __fsub_rn is a plain
  // subtraction and contracts too).
  float x0 = v[0], x1 = v[1], x2 = v[2], x3 = v[3];
  asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
  split_pair<F16>(x0, x1, a, b); h[0] = a; l[0] = b;
  split_pair<F16>(x2, x3, a, b); h[1] = a; l[1] = b;
  unsigned char* p = (unsigned char*)base + ps_quad_offset(i);
  if (NT) { __builtin_nontemporal_store(h, (u32x2*)p); __builtin_nontemporal_store(l, (u32x2*)(p + 64)); }
  else { *(u32x2*)p = h; *(u32x2*)(p + 64) = l; }
}
template <bool F16, bool NT = false> __device__ __forceinline__ void ps_load_quad(const void* base, long long i, float* v) {
  const unsigned char* p = (const unsigned char*)base + ps_quad_offset(i);
  const u32x2 h = NT ? __builtin_nontemporal_load((const u32x2*)p) : *(const u32x2*)p;
  const u32x2 l = NT ? __builtin_nontemporal_load((const u32x2*)(p + 64)) : *(const u32x2*)(p + 64);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if constexpr (F16) {
      const hw_f16x2_t hv = __builtin_bit_cast(hw_f16x2_t, h[j]), lv = __builtin_bit_cast(hw_f16x2_t, l[j]);
      v[2 * j] = (float)hv[0] + (float)lv[0];
      v[2 * j + 1] = (float)hv[1] + (float)lv[1];
    } else {
      v[2 * j] = __uint_as_float(h[j] << 16) + __uint_as_float(l[j] << 16);
      v[2 * j + 1] = __uint_as_float(h[j] & 0xffff0000u) + __uint_as_float(l[j] & 0xffff0000u);
    }
  }
}

// Element traits: storage type T is float or uint16_t (bf16 bits).
template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int EPC = 4;  // elements per 16-byte chunk
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<uint16_t> {
  static constexpr int EPC = 8;
  __device__ static __forceinline__ float ld(const uint16_t* p) { return bf16_bits_to_f32(*p); }
  __device__ static __forceinline__ void st(uint16_t* p, float v) { *p = f32_to_bf16_bits(v); }
};

// unpack a 16-byte chunk into floats (4 for f32, 8 for bf16)
template <typename T> __device__ __forceinline__ void chunk_to_f32(const u32x4& c, float* out);
template <> __device__ __forceinline__ void chunk_to_f32<float>(const u32x4& c, float* out) {
  out[0] = __uint_as_float(c[0]); out[1] = __uint_as_float(c[1]);
  out[2] = __uint_as_float(c[2]); out[3] = __uint_as_float(c[3]);
}
template <> __device__ __forceinline__ void chunk_to_f32<uint16_t>(const u32x4& c, float* out) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = __uint_as_float(c[i] << 16);
    out[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u);
  }
}
template <typename T> __device__ __forceinline__ u32x4 f32_to_chunk(const float* in);
template <> __device__ __forceinline__ u32x4 f32_to_chunk<float>(const float* in) {
  u32x4 c;
  c[0] = __float_as_uint(in[0]); c[1] = __float_as_uint(in[1]);
  c[2] = __float_as_uint(in[2]); c[3] = __float_as_uint(in[3]);
  return c;
}
template <> __device__ __forceinline__ u32x4 f32_to_chunk<uint16_t>(const float* in) {
  u32x4 c;
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = pack_bf16x2(in[2 * i], in[2 * i + 1]);
  return c;
}

// ---- wave64 reductions --------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
