// Implicit-GEMM convolution kernels for gfx950 (MI355X): forward, data-gradient and
// weight-gradient of the bias-free NHWC/HWIO convolutions of
// /root/reference/tf2/resnet.py:183-208 (Conv2dFixedPadding: explicit symmetric
// (k-1)//2 padding, VALID for stride>1 / SAME for stride 1) and the Dense layers of
// /root/reference/tf2/model.py:143-154 (a 1x1 conv on a 1x1 image).
//
// GEMM view (forward):  Y[M, N] = A[M, K] * Wt[N, K]^T
//   M = V*OH*OW output pixels, N = Cout, K = KH*KW*Cin (Cin fastest), A gathered on the
//   fly from the NHWC activation (never materialised), Wt = weights re-laid K-contiguous.
// dgrad is the same kernel with a transposed-conv gather (MODE_DGRAD) and weights
// re-laid as [Cin][KH*KW*Cout].  wgrad reduces over pixels: dW[K, N] = A^T * dY.
//
// Matrix cores: v_mfma_f32_16x16x32_bf16 (bf16 storage, fp32 accumulate) or
// v_mfma_f32_16x16x4_f32 (exact fp32 "parity mode").  Both element types share one
// byte geometry: operands move in 16-byte chunks (8 bf16 / 4 f32), an LDS tile row is
// 8 chunks = 128 B, XOR-swizzled (chunk ^= row&7) so ds_read_b128 fragment reads are
// bank-conflict free.  The weight fragment is the MFMA A operand and the activation
// fragment the B operand, so D[n=(lane>>4)*4+reg][m=lane&15]: every lane owns 4
// CONSECUTIVE output channels of one pixel -> 8/16-byte NHWC stores and per-channel
// BatchNorm statistics (tf2/resnet.py:50-60) reduce with 4 xor-shuffles in the epilogue.
#include "common.h"
#include <stdlib.h>

namespace {

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

struct ConvP {
  const void* x;   // gathered tensor [V, IH, IW, *] (pixel pitch = pixpitch elements)
  const void* w;   // [N][K] K-contiguous, K = KH*KW*IC
  void* y;         // output tensor [V, OH, OW, N]
  float* stats;    // nullable: [nslot][2][N] partial (sum, sumsq) per output channel
  int V, IH, IW, IC, OH, OW, N, KH, KW, stride, pad;
  int pixpitch;    // elements between consecutive ix of the gathered tensor (usually IC)
  int M, K;        // M = rows of THIS launch (= V*cls_h*cls_w)
  int nslot, accumulate;
  int m_tiles, n_tiles;
  // Output-pixel class (strided dgrad): rows enumerate pixels (v, a*cs+py, b*cs+px), a<cls_h, b<cls_w.
  // All rows of a class share the same set of contributing taps, so no MFMA work is spent on
  // structurally-zero taps.  Forward / stride-1 dgrad: one class, cs=1, every tap.
  int cs, py, px, cls_h, cls_w;
  int ntaps;
  int taps[9];
  // 4-bit fields, tap i at bits [4i,4i+4): tap id, dy+8, dx+8 (input offset of tap i relative to the per-row
  // base pixel).  Packed words instead of arrays: dynamically indexed kernel-argument arrays go to scratch.
  unsigned long long tap_w, dy_w, dx_w;
  // Fused BatchNorm-backward reduce (dgrad only): the output IS the gradient wrt a BN(+ReLU) output, so
  // the epilogue masks it (ReLU), accumulates sum(dm) and sum(dm * x^) per channel into `stats` and
  // stores dm.  bn_x = that BN's raw input (same shape as y); bn_mask (mode 1) = tensor whose sign
  // gives the ReLU mask (the block output); mode 2 = mask recomputed from bn_x*scale+shift.
  const void* bn_x;
  const void* bn_mask;
  const float *bn_scale, *bn_shift, *bn_mean, *bn_rstd;
  int bn_mode;     // 0 off, 1 mask tensor, 2 recompute, 3 mask bits, 4 mask bits and ONLY sum(dm) (no bn_x: the
                   // sum(dm * x^) of that BatchNorm is derived from the weight-gradient GEMM, csrc/bn.hip bn_fold_s2)
  // K-extension (EXT instantiations, 1x1 stride-1 dgrad only): after the IC channels of x the reduction continues over
  // ic2 channels of a SECOND tensor x2 at the same pixel (weights rows hold [K | ic2] values), and `bias[n]` is added to
  // every output row.  This is how a BatchNorm backward is folded into the consuming convolution by linearity:
  // d(conv input) = dm (a*W)^T + h (W diag(b) W^T) + W d, with dh = a*dm + b*(h W) + d never materialised.
  const void* x2;
  const float* bias;
  int ic2, pixpitch2;
  // Halo-window instantiations (WIN: 3x3, stride 1, pad 1, bf16): the gathered operand of a 128-pixel tile is ONE window
  // of win_j*32 consecutive pixels (the tile plus W+1 pixels either side) per 64-channel chunk, loaded once and read at
  // nine row offsets, instead of nine separately gathered 128-row tiles.  win_bytes = LDS bytes of the window region.
  int win_j, win_bytes;
  int fapply;      // forward with the fused BatchNorm-apply epilogue (FAPPLY instantiations): bn_scale / bn_shift = the
                   // BatchNorm's scale / shift, bn_x = residual (nullable), bn_mask = ReLU bit mask OUT (nullable), bn_mode = relu
  const void* zero;  // 16 zero bytes in device memory (source of padding chunks for direct-to-LDS loads):
                     // a kernel argument stays in SGPRs; &g_zero16 would be re-fetched from the GOT every k-tile
  int diag;        // diagnostic build only (-DSIMCLR_DIAG): bit mask of pipeline parts to skip
  int split;       // fp32 instantiations: 0 = exact fp32 MFMA, 3 / 6 = split-bf16 terms (simclr_set_f32_matmul)
  // fp32 forward with statistics (simclr_conv2d_fwd_pivoted): the epilogue accumulates sum(y - pivot[n]) and sum((y - pivot[n])^2)
  // over the VALID rows instead of the raw moments -- pivot[n] = the convolution output at one interior pixel (conv_pivot_row),
  // so the sums stay at the scale of the channel's spread even when |mean| >> sigma (raw fp32 moments lose (mean/sigma)^2 * 2^-24
  // of the variance); simclr_bn_reduce_slots_pivoted turns them back into raw fp64 moments.  nullptr: raw moments.
  const float* pivot;
  // Split tail (tile-quantisation fix of the persistent grid): every workgroup walks `rem_full` whole M-tiles; the
  // rem_tiles M-tiles left over (fewer than there are workgroups per N-tile) are each shared by rem_parts consecutive
  // workgroups along the REDUCTION (k-steps [j*KT/P, (j+1)*KT/P) for part j).  Parts j > 0 store their fp32 accumulators
  // into part_ws (slot (tile * n_tiles + nt) * (rem_parts - 1) + j - 1, lane-linear) and publish part_flags[slot] = 1
  // (agent-scope release); part 0 acquires, RESETS the flag to 0 (every slot has one producer and one consumer per launch, so the
  // flags are all zero again when the launch ends: no host-side sequence number, a captured hipGraph can replay the launch),
  // adds the parts in the order j = 1, 2, ... (deterministic) and runs the epilogue.  A partner that does not arrive within the
  // bounded wait bumps *part_err (sticky; simclr_conv2d_split_tail_timeouts) -- the tile's result is then wrong AND reported.
  int rem_full, rem_tiles, rem_parts;
  float* part_ws;
  unsigned* part_flags;
  unsigned* part_err;
  int w_ps;          // fp32, three-term launches: p.w is ALREADY the pre-split copy (SIMCLR_FMT_PS_W: simclr_presplit_weights_multi wrote it
                     // once per optimizer step) -> no per-launch presplit_rows
  int x_ps;          // fp32 dgrad, three bf16 terms: the gathered tensor p.x (the upstream gradient) is in the pre-split block format
                     // (common.h; written by simclr_bn_bwd_apply with SIMCLR_FMT_PS_OUT) -> PSX instantiation, no splitting in the k-loop
  unsigned x_bytes;  // conv_igemm_wide: size of the gathered tensor in bytes (range check of its buffer descriptor)
  // conv_igemm_wide: (images, class rows, class columns) that 8, 128 and 136 consecutive GEMM rows advance an output pixel
  int rs_dq[3], rs_drow[3], rs_dcol[3];
};

// Diagnostic build (build.sh diag -> libsimclr_hip_diag.so): parts of a kernel can be switched off at run time
// (env SIMCLR_DIAG) to attribute time to loads / MFMA / epilogue.  Results are WRONG by design; the product
// library compiles every DIAG(..) to `false`.
#ifdef SIMCLR_DIAG
#define DIAG(b) ((p.diag & (b)) != 0)
#else
#define DIAG(b) false
#endif

template <typename T> struct MMA;
template <> struct MMA<uint16_t> {
  __device__ static __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct MMA<float> {
  __device__ static __forceinline__ f32x4 run(const u32x4& a, const u32x4& b, f32x4 c) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
    return c;
  }
};

// ---- split-bf16 arithmetic on fp32 operands (the fast parity mode) --------------------------------------
// gfx950 has no xf32 / TF32 matrix path: fp32-input MFMA runs at 1/16 of the bf16 rate.  A fp32 value is instead split
// after the LDS read into bf16 terms x = hi + lo (+ lo2), hi = bf16(x), lo = bf16(x - hi), lo2 = bf16(x - hi - lo) (each
// subtraction is exact in fp32), and a product a*b becomes 3 (terms of weight >= 2^-9: hi*hi + hi*lo + lo*hi, ~2^-17
// relative) or 6 (weight >= 2^-18: + hi*lo2 + lo2*hi + lo*lo, ~2^-24 = fp32 level) v_mfma_f32_16x16x32_bf16 with fp32
// accumulation -- bf16 products are exact in fp32.  Storage, statistics and every elementwise kernel stay fp32.
// Two consecutive 16-byte chunks (8 floats) of a fragment row become one 8-element bf16 operand.
__device__ __forceinline__ void split_terms2(const u32x4& c0, const u32x4& c1, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = __uint_as_float(j < 2 ? c0[2 * j] : c1[2 * j - 4]), x1 = __uint_as_float(j < 2 ? c0[2 * j + 1] : c1[2 * j - 3]);
    const uint32_t h = pack_bf16x2(x0, x1);
    hi[j] = h;
    lo[j] = pack_bf16x2(x0 - __uint_as_float(h << 16), x1 - __uint_as_float(h & 0xffff0000u));
  }
}
__device__ __forceinline__ void split_terms3(const u32x4& c0, const u32x4& c1, u32x4& hi, u32x4& lo, u32x4& lo2) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = __uint_as_float(j < 2 ? c0[2 * j] : c1[2 * j - 4]), x1 = __uint_as_float(j < 2 ? c0[2 * j + 1] : c1[2 * j - 3]);
    const uint32_t h = pack_bf16x2(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
    const uint32_t l = pack_bf16x2(r0, r1);
    hi[j] = h;
    lo[j] = l;
    lo2[j] = pack_bf16x2(r0 - __uint_as_float(l << 16), r1 - __uint_as_float(l & 0xffff0000u));
  }
}
__device__ __forceinline__ f32x4 mma_bf16(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// ---- split-fp16 arithmetic (terms == 13: "three fp16 terms") ----------------------------------------------------------
// An fp16 piece carries 11 significand bits against bf16's 8: x = hi + lo with hi = fp16(x), lo = fp16(x - hi) represents x to
// ~2^-22 (two bf16 pieces: 2^-16), so the THREE products hi*hi + hi*lo + lo*hi (dropped: lo*lo ~ 2^-22) are as accurate as the
// six bf16 terms at half the MFMA work -- v_mfma_f32_16x16x32_f16 runs at the bf16 rate, fp16 products are exact in fp32.
// The price is fp16's range (|x| < 65504, pieces below 2^-14 are subnormal: absolute resolution 2^-25): fine for what the
// FORWARD multiplies -- BatchNorm outputs, pooled features, images (O(1)) and weights, which are pre-split with a power-of-two
// scale (F16_WSCALE_LOG2, undone on the accumulators: exact) so that their lo pieces stay normal.  Gradients span too many
// binades for fp16 without per-tensor scaling: the backward keeps the bf16 terms.  An operand beyond the fp16 range turns into
// inf / NaN in the output (loud), never into a silently wrong number.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
#define F16_WSCALE_LOG2 8
// (pack_f16x2 / split_pair<F16>: common.h)
__device__ __forceinline__ void split_terms2_f16(const u32x4& c0, const u32x4& c1, u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = __uint_as_float(j < 2 ? c0[2 * j] : c1[2 * j - 4]), x1 = __uint_as_float(j < 2 ? c0[2 * j + 1] : c1[2 * j - 3]);
    uint32_t h, l;
    split_pair<true>(x0, x1, h, l);
    hi[j] = h; lo[j] = l;
  }
}
__device__ __forceinline__ f32x4 mma_f16(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// acc(i, j) += A_i . B_j over the 32 reduction elements of one k-step; lda(i, h) / ldb(j, h) return 16-byte chunk h (0 | 1) of
// fragment row i / j (fp32: 4 values; the two chunks of a row are this lane's 8 reduction elements).
// terms (compile time: the six-term path needs ~60 more registers): 0 = exact fp32 MFMA (8 x v_mfma_f32_16x16x4_f32 per pair), 3 / 6 = split bf16 (small terms accumulated first,
// term-major so that consecutive MFMAs write different accumulators), 13 = three split-fp16 terms.  The B operand is split once, the A operand one
// fragment at a time (register pressure: NB x 12 + 12 split registers instead of (NA + NB) x 12).
// TR: the accumulator array is indexed [j][i] (acc[NB][NA]) instead of [i][j].
// PSA (terms == 3 | 13): the A operand arrives PRE-SPLIT (presplit_rows below): chunk 0 of a fragment row is this lane's
// eight hi values, chunk 1 its eight lo values -- the same two 16-byte reads, no VALU work for that operand.  Bitwise the
// same products as the in-register split (same rounding instruction, same MFMA order).  PSB2: the same for the B operand
// (an activation / gradient tensor kept in the pre-split block format by its producer).
template <int NA, int NB, bool TR, int terms, bool PSA = false, bool PSB2 = false, typename LA, typename LB>
__device__ __forceinline__ void mma_f32_chunks(f32x4* __restrict__ accp, LA lda, LB ldb) {
  static_assert(terms == 0 || terms == 3 || terms == 6 || terms == 13, "0 = exact fp32, 3 / 6 = split-bf16 terms, 13 = three split-fp16 terms");
  static_assert(!(PSA || PSB2) || terms == 3 || terms == 13, "pre-split operand: two planes (hi, lo) = the three-term products only");
#define acc_(i, j) accp[TR ? (j) * NA + (i) : (i) * NB + (j)]
  if constexpr (terms == 0) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 a[NA], b[NB];
#pragma unroll
      for (int i = 0; i < NA; ++i) a[i] = lda(i, ks);
#pragma unroll
      for (int j = 0; j < NB; ++j) b[j] = ldb(j, ks);
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc_(i, j) = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i][e]), __uint_as_float(b[j][e]), acc_(i, j), 0, 0, 0);
    }
  } else if constexpr (terms == 3 || terms == 13) {
    constexpr bool F16 = terms == 13;
    auto mm = [](const u32x4& a, const u32x4& b, f32x4 c) __attribute__((always_inline)) { return F16 ? mma_f16(a, b, c) : mma_bf16(a, b, c); };
    u32x4 bh[NB], bl[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if constexpr (PSB2) { bh[j] = ldb(j, 0); bl[j] = ldb(j, 1); }
      else if constexpr (F16) split_terms2_f16(ldb(j, 0), ldb(j, 1), bh[j], bl[j]);
      else split_terms2(ldb(j, 0), ldb(j, 1), bh[j], bl[j]);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      u32x4 ah, al;
      if constexpr (PSA) { ah = lda(i, 0); al = lda(i, 1); }
      else if constexpr (F16) split_terms2_f16(lda(i, 0), lda(i, 1), ah, al);
      else split_terms2(lda(i, 0), lda(i, 1), ah, al);
#pragma unroll
      for (int j = 0; j < NB; ++j) acc_(i, j) = mm(al, bh[j], acc_(i, j));
#pragma unroll
      for (int j = 0; j < NB; ++j) acc_(i, j) = mm(ah, bl[j], acc_(i, j));
#pragma unroll
      for (int j = 0; j < NB; ++j) acc_(i, j) = mm(ah, bh[j], acc_(i, j));
    }
  } else {
    u32x4 bh[NB], bl[NB], bm[NB];     // h = hi, l = lo, m = lo2
#pragma unroll
    for (int j = 0; j < NB; ++j) split_terms3(ldb(j, 0), ldb(j, 1), bh[j], bl[j], bm[j]);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      u32x4 ah, al, am;
      split_terms3(lda(i, 0), lda(i, 1), ah, al, am);
#pragma unroll
      for (int j = 0; j < NB; ++j) acc_(i, j) = mma_bf16(al, bl[j], acc_(i, j));
#pragma unroll
      for (int j = 0; j < NB; ++j) acc_(i, j) = mma_bf16(am, bh[j], acc_(i, j));
#pragma unroll
      for (int j = 0; j < NB; ++j) acc_(i, j) = mma_bf16(ah, bm[j], acc_(i, j));
#pragma unroll
      for (int j = 0; j < NB; ++j) acc_(i, j) = mma_bf16(al, bh[j], acc_(i, j));
#pragma unroll
      for (int j = 0; j < NB; ++j) acc_(i, j) = mma_bf16(ah, bl[j], acc_(i, j));
#pragma unroll
      for (int j = 0; j < NB; ++j) acc_(i, j) = mma_bf16(ah, bh[j], acc_(i, j));
    }
  }
#undef acc_
}

__device__ __forceinline__ u32x4 ld16(const void* p) { return *(const u32x4*)p; }
__device__ __forceinline__ u32x4 zero16() { return (u32x4){0u, 0u, 0u, 0u}; }

// XCD-aware tile mapping: workgroup b runs on XCD b%8; all N-tiles of one M-tile are
// consecutive on ONE XCD so the gathered A tile is re-read from that XCD's L2.
__device__ __forceinline__ bool tile_of_block(int m_tiles, int n_tiles, int& mt, int& nt) {
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
  nt = idx % n_tiles;
  mt = (idx / n_tiles) * 8 + xcd;
  return mt < m_tiles;
}

// ------------------------------------------------------------------------------------
// forward / dgrad implicit GEMM.  Tile BM=128 x BN (128|64) x 128 bytes of K; 4 waves.
// Epilogue (bf16): the C tile is staged through LDS (row pitch +8 B: conflict-free 8-byte
// writes) and stored as whole 128/256-byte NHWC rows, 8 bytes per lane -> full cache lines.
// ------------------------------------------------------------------------------------
__device__ u32x4 g_zero16;   // source of zero chunks for direct-to-LDS loads (padding / tails)

template <typename T, int MODE, int BN, bool STATS, bool GLDS>
__global__ __launch_bounds__(256) void conv_igemm(const ConvP p) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BM = 128;
  constexpr int BK = 8 * EPC;           // elements per k-tile (128 bytes)
  constexpr int WN = BN / 64;           // waves along N (2 or 1)
  constexpr int WM = 4 / WN;            // waves along M (2 or 4)
  constexpr int MI = BM / WM / 16;      // 16-row m fragments per wave (4 or 2)
  constexpr int NI = 4;                 // 16-col n fragments per wave (64 cols)
  constexpr int AJ = BM / 32;           // A chunks per thread per k-tile (4)
  constexpr int BJ = BN / 32;           // B chunks per thread per k-tile (4 or 2)
  constexpr bool LDS_EPI = sizeof(T) == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* As = (u32x4*)smem;                       // [2][BM*8]
  u32x4* Bs = As + 2 * BM * 8;                    // [2][BN*8]

  int mt, nt;
  if (!tile_of_block(p.m_tiles, p.n_tiles, mt, nt)) return;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR (LDS bases, M0)
  const int g = lane >> 4, fl = lane & 15;
  const int wm = wave / WN, wn = wave % WN;
  const T* __restrict__ X = (const T*)p.x;
  const T* __restrict__ Wt = (const T*)p.w;
  const int cls_hw = p.cls_h * p.cls_w;

  // ---- per-thread gather bookkeeping ----
  // register staging: rows (tid>>3)+32j, chunk tid&7 (written to the swizzled LDS slot).
  // direct-to-LDS (GLDS): wave-instruction j of wave w fills LDS rows 8*(w*AJ+j)..+7 linearly
  // (lane -> row +(lane>>3), slot lane&7), so the XOR swizzle moves to the SOURCE: the lane
  // fetches logical chunk (lane&7)^(lane>>3) (guide rule 21: linear dest + swizzled source).
  const int kc = GLDS ? ((lane & 7) ^ (lane >> 3)) : (tid & 7);
  auto a_row = [&](int j) -> int { return GLDS ? 8 * (wave * AJ + j) + (lane >> 3) : (tid >> 3) + 32 * j; };
  auto b_row = [&](int j) -> int { return GLDS ? 8 * (wave * BJ + j) + (lane >> 3) : (tid >> 3) + 32 * j; };
  int a_by[AJ], a_bx[AJ];
  long long a_img[AJ];
  bool a_ok[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = m0 + a_row(j);
    a_ok[j] = m < p.M;
    const int mm = a_ok[j] ? m : 0;
    const int v = mm / cls_hw;
    const int rem = mm - v * cls_hw;
    const int ca = rem / p.cls_w, cb = rem - ca * p.cls_w;
    const int oy = ca * p.cs + p.py, ox = cb * p.cs + p.px;
    a_img[j] = (long long)v * p.IH * p.IW;
    if (MODE == MODE_FWD) { a_by[j] = oy * p.stride - p.pad; a_bx[j] = ox * p.stride - p.pad; }
    else { a_by[j] = oy + p.pad; a_bx[j] = ox + p.pad; }
  }
  const int kpt = p.IC / BK;            // k-tiles per tap
  const int KT = p.ntaps * kpt;

  u32x4 ra[AJ], rb[BJ];
  auto load_tile = [&](int kt) {
    const int ti = kt / kpt;
    const int ci0 = (kt - ti * kpt) * BK;
    const int tap = p.taps[ti];
    const int ty = tap / p.KW, tx = tap - ty * p.KW;
    const int k0 = tap * p.IC + ci0;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      int iy, ix;
      bool ok = a_ok[j];
      if (MODE == MODE_FWD) {
        iy = a_by[j] + ty; ix = a_bx[j] + tx;
      } else {
        const int tyy = a_by[j] - ty, txx = a_bx[j] - tx;   // divisible by stride for this class's taps
        ok = ok && tyy >= 0 && txx >= 0;
        if (p.stride > 1) { iy = tyy / p.stride; ix = txx / p.stride; }
        else { iy = tyy; ix = txx; }
      }
      ok = ok && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      ra[j] = zero16();
      if (ok) ra[j] = ld16(X + ((a_img[j] + (long long)iy * p.IW + ix) * p.pixpitch + ci0 + kc * EPC));
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int n = n0 + b_row(j);
      rb[j] = zero16();
      if (n < p.N) rb[j] = ld16(Wt + ((long long)n * p.K + k0 + kc * EPC));
    }
  };
  // direct-to-LDS version of load_tile: 16-byte global_load_lds per lane, no VGPR staging
  auto issue_tile = [&](int kt, int buf) {
    const int ti = kt / kpt;
    const int ci0 = (kt - ti * kpt) * BK;
    const int tap = p.taps[ti];
    const int ty = tap / p.KW, tx = tap - ty * p.KW;
    const int k0 = tap * p.IC + ci0;
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    unsigned char* a_dst = (unsigned char*)(As + buf * BM * 8) + wave * AJ * 1024;
    unsigned char* b_dst = (unsigned char*)(Bs + buf * BN * 8) + wave * BJ * 1024;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      int iy, ix;
      bool ok = a_ok[j];
      if (MODE == MODE_FWD) {
        iy = a_by[j] + ty; ix = a_bx[j] + tx;
      } else {
        const int tyy = a_by[j] - ty, txx = a_bx[j] - tx;
        ok = ok && tyy >= 0 && txx >= 0;
        if (p.stride > 1) { iy = tyy / p.stride; ix = txx / p.stride; }
        else { iy = tyy; ix = txx; }
      }
      ok = ok && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
      const void* src = ok ? (const void*)(X + ((a_img[j] + (long long)iy * p.IW + ix) * p.pixpitch + ci0 + kc * EPC))
                           : p.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(a_dst + j * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int n = n0 + b_row(j);
      const void* src = (n < p.N) ? (const void*)(Wt + ((long long)n * p.K + k0 + kc * EPC)) : p.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(b_dst + j * 1024), 16, 0, 0);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int r = (tid >> 3) + 32 * j;
      As[buf * BM * 8 + r * 8 + (kc ^ (r & 7))] = ra[j];
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int r = (tid >> 3) + 32 * j;
      Bs[buf * BN * 8 + r * 8 + (kc ^ (r & 7))] = rb[j];
    }
  };

  f32x4 acc[NI][MI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute_tile = [&](int buf) {
    if constexpr (sizeof(T) == 4) {
      // the weight fragment is the MFMA A operand; the activation side (MI fragments) is the one split once per k-tile
      mma_f32_chunks<NI, MI, false, 0>(&acc[0][0],
          [&](int i, int ks) { const int r = wn * 64 + i * 16 + fl; return Bs[buf * BN * 8 + r * 8 + ((ks * 4 + g) ^ (r & 7))]; },
          [&](int i, int ks) { const int r = wm * (MI * 16) + i * 16 + fl; return As[buf * BM * 8 + r * 8 + ((ks * 4 + g) ^ (r & 7))]; });
    } else
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 af[MI], bf[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = wm * (MI * 16) + i * 16 + fl;
        af[i] = As[buf * BM * 8 + r * 8 + ((ks * 4 + g) ^ (r & 7))];
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int r = wn * 64 + i * 16 + fl;
        bf[i] = Bs[buf * BN * 8 + r * 8 + ((ks * 4 + g) ^ (r & 7))];
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = MMA<T>::run(bf[ni], af[mi], acc[ni][mi]);
    }
  };
  if (GLDS) {
    // tile kt+1 streams into the other LDS buffer (LDS-DMA, no registers) while tile kt is consumed;
    // one barrier per k-tile: it both publishes tile kt and retires the reads of tile kt-1.
    if (KT > 0) issue_tile(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
      const int buf = kt & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < KT) issue_tile(kt + 1, buf ^ 1);
      compute_tile(buf);
    }
    __syncthreads();
  } else {
    if (KT > 0) {
      load_tile(0);
      store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
      const int buf = kt & 1;
      if (kt + 1 < KT) load_tile(kt + 1);
      compute_tile(buf);
      if (kt + 1 < KT) store_tile(buf ^ 1);
      __syncthreads();
    }
  }

  // output address of class-local row m (element offset of channel 0), or -1
  auto row_offset = [&](int m) -> long long {
    if (m >= p.M) return -1;
    const int v = m / cls_hw;
    const int rem = m - v * cls_hw;
    const int ca = rem / p.cls_w, cb = rem - ca * p.cls_w;
    return (((long long)v * p.OH + ca * p.cs + p.py) * p.OW + cb * p.cs + p.px) * p.N;
  };
  T* __restrict__ Y = (T*)p.y;

  if (STATS) {
    // per-channel sum / sum-of-squares over this tile's rows (rows >= M are exact zeros)
    float* red = (float*)smem;  // [WM][BN][2] floats, reuse LDS (all MFMA reads are done)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) { const float v = acc[ni][mi][r]; s += v; ss += v * v; }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
        if (fl == 0) {
          const int nl = wn * 64 + ni * 16 + g * 4 + r;
          red[(wm * BN + nl) * 2] = s;
          red[(wm * BN + nl) * 2 + 1] = ss;
        }
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { s += red[(w * BN + tid) * 2]; ss += red[(w * BN + tid) * 2 + 1]; }
      float* st = p.stats + (long long)(mt % p.nslot) * 2 * p.N;
      atomicAdd(st + n0 + tid, s);
      atomicAdd(st + p.N + n0 + tid, ss);
    }
    __syncthreads();
  }

  if (LDS_EPI) {
    // ---- stage the bf16 C tile in LDS, then store whole rows (8 B per lane, full lines) ----
    constexpr int PITCH = BN * 2 + 8;                       // bytes
    constexpr int LPR = BN * 2 / 8;                         // lanes per row (32 or 16)
    unsigned char* Cs = smem;                               // [BM][PITCH]
    long long* rowoff = (long long*)(smem + BM * PITCH);    // [BM]
    if (tid < BM) rowoff[tid] = row_offset(m0 + tid);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int nl = wn * 64 + ni * 16 + g * 4;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int ml = wm * (MI * 16) + mi * 16 + fl;
        u32x2 pk;
        pk[0] = pack_bf16x2(acc[ni][mi][0], acc[ni][mi][1]);
        pk[1] = pack_bf16x2(acc[ni][mi][2], acc[ni][mi][3]);
        *(u32x2*)(Cs + ml * PITCH + nl * 2) = pk;
      }
    }
    __syncthreads();
    const int c8 = tid % LPR;                               // 8-byte column within the row
    const int ncol = n0 + c8 * 4;
#pragma unroll 4
    for (int r = tid / LPR; r < BM; r += 256 / LPR) {
      const long long off = rowoff[r];
      if (off >= 0 && ncol < p.N) {
        u32x2 v = *(const u32x2*)(Cs + r * PITCH + c8 * 8);
        uint16_t* dst = (uint16_t*)Y + off + ncol;
        if (p.accumulate) {
          const u32x2 o = *(const u32x2*)dst;
          v[0] = pack_bf16x2(__uint_as_float(v[0] << 16) + __uint_as_float(o[0] << 16),
                             __uint_as_float(v[0] & 0xffff0000u) + __uint_as_float(o[0] & 0xffff0000u));
          v[1] = pack_bf16x2(__uint_as_float(v[1] << 16) + __uint_as_float(o[1] << 16),
                             __uint_as_float(v[1] & 0xffff0000u) + __uint_as_float(o[1] & 0xffff0000u));
        }
        *(u32x2*)dst = v;
      }
    }
  } else {
    // ---- fp32 parity mode: direct 16-byte stores (4 consecutive channels per lane) ----
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const long long off = row_offset(m0 + wm * (MI * 16) + mi * 16 + fl);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * 64 + ni * 16 + g * 4;
        if (off >= 0 && n < p.N) {
          T* dst = Y + off + n;
          float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
          if (p.accumulate) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += Elem<T>::ld(dst + r);
          }
          *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Persistent variant of conv_igemm: a fixed grid of workgroups (2 per CU) walks the tile list.
// The (tile, k-tile) steps are flattened into ONE software pipeline, so the direct-to-LDS loads
// of the next tile's first k-tile are in flight while the current tile finishes its MFMAs and
// stores its outputs -- short-K (1x1) layers no longer pay a cold prologue per tile.  Every
// workgroup keeps a fixed N-tile, so the BatchNorm statistics (sum, sum of squares per output
// channel) stay in registers across all its M-tiles and are flushed with ONE set of atomics at
// the end instead of one per tile.  XCD-aware: workgroups that share an M-tile (different
// N-tiles) sit on the same XCD and advance in lockstep, so the gathered A tile is served by
// that XCD's L2.
// ------------------------------------------------------------------------------------
#ifndef SIMCLR_BN64_WPE
#define SIMCLR_BN64_WPE 3   // waves per SIMD of the 64-wide bf16 instantiations: 3 workgroups per CU (a few spilled dwords) beat 2 (profiles/r02_notes.md)
#endif
// (launch bounds: the 64-wide tiles are sized for THREE workgroups per CU -- igemm_persistent_grid gives them 768 workgroups -- in bf16
// and, since round 6, in the fp32 three-term instantiations with pre-split weights (<= 168 VGPRs, no spills): with two resident the third
// third of the persistent grid started when the first finished, 152.9 -> 151.2 ms per parity-mode step, three interleaved pairs)
template <typename T, int MODE, int BM, int BN, int NW, int STAGES, bool STATS, bool BNEPI, bool EXT = false,
          bool WIN = false, bool FAPPLY = false, int SPL = 0, bool TAIL = false, bool PSB = false, int FAS = 0, int EPS = 0, bool PSX = false>
__global__ __launch_bounds__(NW * 64, (NW == 8 || STAGES == 3) ? 1 : (BN == 64 && (sizeof(T) == 2 || SPL == 13 || (SPL == 3 && PSB))) ? SIMCLR_BN64_WPE : 2) void conv_igemm_persistent(const ConvP p) {
  // EPS (bf16 BNEPI only): the mask mode and the accumulate flag of the fused BatchNorm-backward-reduce epilogue as compile-time
  // constants -- EPS - 1 = 2 * (mode == 4) + accumulate for the modes a ResNet step uses (2: mask recomputed from the BatchNorm
  // input, 4: mask bits, sum(dm) only).  0: read from the kernel arguments.
  static_assert(EPS == 0 || (BNEPI && sizeof(T) == 2 && EPS <= 4), "EPS specialises the bf16 BNEPI epilogue");
  const int bnm = EPS ? (((EPS - 1) >> 1) ? 4 : 2) : p.bn_mode;
  const bool accu = EPS ? ((EPS - 1) & 1) != 0 : p.accumulate != 0;
  // FAS (FAPPLY only): the run-time options of the fused BatchNorm-apply epilogue as compile-time constants for the two shapes every
  // fused bottleneck tail of a ResNet has -- 1: residual + ReLU + ReLU bit mask, Cout a multiple of 32; 2: the same with the
  // residual's own BatchNorm (projection blocks).  0: options read from the kernel arguments.  Same arithmetic, same order.
  static_assert(FAS == 0 || FAPPLY, "FAS specialises the FAPPLY epilogue");
  const bool fa_res = FAS ? true : p.bn_x != nullptr;
  const bool fa_rbn = FAS ? FAS == 2 : p.bn_mean != nullptr;
  const bool fa_relu = FAS ? true : p.bn_mode != 0;
  const bool fa_mask = FAS ? true : p.bn_mask != nullptr;
  const bool fa_n32 = FAS ? true : (p.N & 31) == 0;
  // PSB (fp32 storage, three split-bf16 terms): the weight matrix p.w was rewritten by presplit_rows into (hi, lo) bf16
  // planes per 128-byte k-block -- same bytes, same LDS-DMA stream, no splitting work for that operand in the k-loop
  // (SPL == 13, the forward: fp16 planes of the weights times 2^F16_WSCALE_LOG2 -- undone on the accumulators after the k-loop.)
  // PSX: the GATHERED operand p.x is kept in the same pre-split block format by its producer (an elementwise kernel): with PSB
  // and PSX the k-loop is LDS reads + MFMAs only.
  static_assert(!PSB || ((SPL == 3 || SPL == 13) && sizeof(T) == 4), "pre-split weights: fp32 storage, three terms");
  static_assert(!PSX || ((SPL == 3 || SPL == 13) && sizeof(T) == 4), "pre-split gathered operand: fp32 storage, three terms");
  static_assert(SPL != 13 || PSB, "split-fp16 terms need the pre-split (scaled) weight planes");
  // (fp32 storage, round 6: the three-term instantiations with pre-split weights -- a window row is the 128-byte block of 32 channels, the
  // k-step of these kernels; with PSX the block is the pre-split gradient's -- read through mma_f32_chunks like the gathered tile)
  static_assert(!WIN || (STAGES == 2 && !EXT && NW == 4 && BM == 128 && !FAPPLY && (sizeof(T) == 2 || ((SPL == 3 || SPL == 13) && PSB))),
                "halo-window variant: 2 stages; bf16, or fp32 storage with three terms and pre-split weights");
  // FAPPLY (forward, bf16): the row-wise epilogue applies a BatchNorm (+ residual, + ReLU, + ReLU bit mask) to the tile
  // before it is stored -- y = act(bf16(conv) * scale + shift + res) with exactly the arithmetic of bn_apply (csrc/bn.hip),
  // so the convolution output itself never travels to HBM.  The statistics that scale / shift derive from come from a
  // first, store-free pass of the same convolution (STATS instantiation with y == nullptr).
  // (fp32 storage, round 6: the same epilogue on the fp32 accumulators in the register-direct path below -- the parity mode's block
  // tails lose the convolution-output round trip too: 8 of a bottleneck block's 67 tensor passes)
  static_assert(!FAPPLY || (MODE == MODE_FWD && !STATS && !BNEPI && !EXT), "fused BN-apply epilogue: forward, no statistics");
  static_assert(SPL == 0 || sizeof(T) == 4, "split-bf16 terms: fp32 storage only");
  constexpr bool WPS = WIN && sizeof(T) == 4 && !PSX && (SPL == 3 || SPL == 13);   // fp32 window split in place once per chunk (k-loop below)
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BK = 8 * EPC;
  constexpr int WN = BN / 64;           // waves along N
  constexpr int WM = NW / WN;           // waves along M
  constexpr int MI = BM / WM / 16;      // 16-row m fragments per wave
  constexpr int NI = 4;
  constexpr int AJ = BM / (8 * NW);     // direct-to-LDS wave-instructions (8 rows each) per wave, A tile
  constexpr int BJ = BN / (8 * NW);
  static_assert(AJ >= 1 && BJ >= 1 && MI >= 1, "tile / wave configuration");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LDS: STAGES x [A tile (BM rows) | B tile (BN rows)] of 128-byte rows, then the epilogue scratch.
  // A whole stage (>= BM*BN*2 bytes) doubles as the bf16 C staging buffer once it has been consumed.
  constexpr int STG = (BM + BN) * 8;              // u32x4 per stage
  u32x4* As = (u32x4*)smem;                       // stage s: As + s*STG
  u32x4* Bs = As + BM * 8;                        // stage s: Bs + s*STG
  // WIN layout: [window region (win_bytes; doubles as the C staging buffer)] [2 x B tile] [bnp] [rowoff] [one zero row]
  u32x4* Wn = (u32x4*)smem;
  u32x4* Bw = (u32x4*)(smem + (WIN ? p.win_bytes : 0));
  float* bnp = WIN ? (float*)(Bw + 2 * BN * 8)
                   : (float*)(As + STAGES * STG);  // BNEPI: [4][BN] = scale, shift, mean, rstd of this N-tile (+ [BN] bias if EXT)
  long long* rowoff = (long long*)(bnp + (4 + (EXT ? 1 : 0)) * BN); // [BM] output offsets of the tile rows (LDS epilogue)
  const int zrow = WIN ? (int)(((unsigned char*)(rowoff + BM) - smem) >> 7) : 0;   // WIN: index of a 128-byte row of zeros
  // WSUM (256-wide tile): the row-pass sums are folded after every half tile into a per-WAVE LDS slot [NW][BN][2] (plain
  // read-modify-write, the slot belongs to one wave: deterministic) instead of living in 16 registers across the k-loop,
  // where the 128 accumulator registers leave no room for them
  constexpr bool WSUM = BM == 256;
  float2* wred = (float2*)(rowoff + BM);
  constexpr bool LDS_EPI = sizeof(T) == 2;        // bf16: coalesced row-wise epilogue through LDS
  constexpr int NTH = NW * 64;
  constexpr int CPR = BN / 8;                     // 16-byte chunks (8 channels) per C row
  constexpr int RPP = NTH / CPR;                  // rows per pass of the row-wise epilogue
  // A C tile larger than one stage (256 x 256) goes through the row-wise epilogue in EH halves of BMH rows: the waves of
  // M-half h stage their accumulators, every thread takes part in the row pass, then the other half follows.
  // ROWSTATS: the per-channel sums are taken in the row-wise epilogue pass (16 registers per thread, of the bf16-rounded
  // values -- the tensor the reference's moments see, tf2/resnet.py:50-60) instead of from the accumulators (32 registers
  // held across the whole k-loop): the fused BN-backward reduce always, the forward statistics on the 256-wide tile.
  constexpr bool ROWSTATS = STATS && LDS_EPI && (BNEPI || BM == 256);
  constexpr int EH = (LDS_EPI && BM * BN * 2 > STG * 16) ? 2 : 1;
  constexpr int BMH = BM / EH;
  static_assert(!LDS_EPI || (BMH * BN * 2 <= STG * 16), "C (half) tile must fit in one stage");
  static_assert(EH == 1 || (WM % EH == 0 && !WIN), "half-tile epilogue: whole waves per half");
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR (LDS bases, M0)
  const int g = lane >> 4, fl = lane & 15;
  const int wm = wave / WN, wn = wave % WN;
  const T* __restrict__ X = (const T*)p.x;
  const T* __restrict__ Wt = (const T*)p.w;
  T* __restrict__ Y = (T*)p.y;
  const int cls_hw = p.cls_h * p.cls_w;

  // workgroup -> (N-tile, first M-tile, M-tile stride); gridDim.x is a multiple of 8*n_tiles
  const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
  const int nt = l % p.n_tiles;
  const int mslots = gridDim.x / p.n_tiles;
  const int mslot = (l / p.n_tiles) * 8 + xcd;
  const int n0 = nt * BN;
  // split tail: rem_parts >= 2 -> every workgroup owns exactly rem_full whole tiles, then (mslot < rem_tiles * rem_parts)
  // part `pj` of remainder tile `ptile` over the k-steps (WIN: 64-channel chunks) [pka, pkb)
  // (the part's geometry is recomputed from the kernel arguments where it is needed -- tile transitions only -- instead of
  // living in registers across the k-loop: these kernels sit at the SGPR / VGPR limits)
  // TAIL instantiations only (the launcher picks them when ConvP::rem_parts >= 2): on the short-K streaming layers every
  // k-step is a tile transition, and the extra state of the split tail costs them 30-45 % (spills in that path)
  const int count = (TAIL && p.rem_parts >= 2) ? p.rem_full : ((mslot < p.m_tiles) ? (p.m_tiles - mslot + mslots - 1) / mslots : 0);
  const bool has_part = TAIL && p.rem_parts >= 2 && mslot < p.rem_tiles * p.rem_parts;
  auto part_tile = [&]() __attribute__((always_inline)) { return p.rem_full * mslots + mslot / p.rem_parts; };
  auto part_index = [&]() __attribute__((always_inline)) { return mslot % p.rem_parts; };
  const int kpt = p.IC / BK;
  const int kpt2 = EXT ? p.ic2 / BK : 0;          // k-tiles of the second source (appended after the regular taps)
  const int KT = p.ntaps * kpt + kpt2;
  const int kmain = p.KH * p.KW * p.IC;           // weight-row offset of the extension block
  const T* __restrict__ X2 = (const T*)p.x2;

  const int kc = (lane & 7) ^ (lane >> 3);
  // 1x1 stride-1 (and the Dense layers): output row m reads input pixel m -- no decode at all
  const bool flat = WIN || (p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.cs == 1 &&
                            p.IH == p.OH && p.IW == p.OW);   // WIN: same geometry in and out, output row m = pixel m
  // Row state: (a_ry, a_rx) = base input pixel of the row, a_rb = its address (chunk kc, channel 0).
  // Tap i reads pixel (a_ry + tap_dy[i], a_rx + tap_dx[i]) -- offsets are uniform per tap and
  // precomputed on the host, so a tap change costs two adds, two unsigned compares and one 64-bit
  // add per row, and a k-tile inside a tap costs nothing but `+ ci0`.
  int a_ry[AJ], a_rx[AJ];
  int a_m[AJ];                  // EXT: pixel index of the row (flat convolutions: input pixel == output row)
  const T* a_rb[AJ];
  bool a_ok[AJ];
  const T* a_tp[AJ];
  bool a_tok[AJ];
  auto setup_rows = [&](int mt) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int m = mt * BM + 8 * (wave * AJ + j) + (lane >> 3);
      a_ok[j] = m < p.M;
      const int mm = a_ok[j] ? m : 0;
      long long pix;                       // base input pixel index of the row
      int ry = 0, rx = 0;
      if (flat) {
        pix = mm;
      } else {
        const int v = mm / cls_hw;
        const int rem = mm - v * cls_hw;
        const int ca = rem / p.cls_w, cb = rem - ca * p.cls_w;
        if (MODE == MODE_FWD) { ry = ca * p.stride - p.pad; rx = cb * p.stride - p.pad; }
        else { ry = ca; rx = cb; }
        pix = ((long long)v * p.IH + ry) * p.IW + rx;
      }
      a_ry[j] = ry; a_rx[j] = rx;
      if (EXT) a_m[j] = mm;
      a_rb[j] = X + pix * p.pixpitch + kc * EPC;
      a_tp[j] = a_rb[j];                   // flat: final; otherwise overwritten by set_tap
      a_tok[j] = a_ok[j];
    }
  };
  auto set_tap = [&](int ti) __attribute__((always_inline)) {      // only on a tap change (uniform)
    if (flat) return;
    const int dy = (int)((p.dy_w >> (4 * ti)) & 15) - 8, dx = (int)((p.dx_w >> (4 * ti)) & 15) - 8;
    const long long doff = ((long long)dy * p.IW + dx) * p.pixpitch;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      a_tok[j] = a_ok[j] && (unsigned)(a_ry[j] + dy) < (unsigned)p.IH && (unsigned)(a_rx[j] + dx) < (unsigned)p.IW;
      a_tp[j] = a_rb[j] + doff;
    }
  };
  // weight rows of this workgroup never change
  const T* b_src[BJ];
  bool b_ok[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int n = n0 + 8 * (wave * BJ + j) + (lane >> 3);
    b_ok[j] = n < p.N;
    b_src[j] = Wt + (long long)(b_ok[j] ? n : 0) * p.K + kc * EPC;
  }
  // ti = tap index, ci = k-tile index inside the tap (both tracked incrementally: no divisions)
  auto issue_tile = [&](int ti, int ci, int buf) __attribute__((always_inline)) {
    if (DIAG(2)) return;
    const int ci0 = ci * BK;
    const bool ext = EXT && ti == p.ntaps;         // wave-uniform
    const int k0 = ext ? kmain + ci0 : (int)((p.tap_w >> (4 * ti)) & 15) * p.IC + ci0;
    unsigned char* a_dst = (unsigned char*)(As + buf * STG) + wave * AJ * 1024;
    unsigned char* b_dst = (unsigned char*)(Bs + buf * STG) + wave * BJ * 1024;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const void* src;
      if (ext) src = a_ok[j] ? (const void*)(X2 + (long long)a_m[j] * p.pixpitch2 + kc * EPC + ci0) : p.zero;
      else src = a_tok[j] ? (const void*)(a_tp[j] + ci0) : p.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(a_dst + j * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const void* src = b_ok[j] ? (const void*)(b_src[j] + k0) : p.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(b_dst + j * 1024), 16, 0, 0);
    }
  };

  if (BNEPI || EXT || FAPPLY) {
    for (int i = tid; i < BN; i += NW * 64) {
      const int n = n0 + i;
      const bool ok = n < p.N;
      if (FAPPLY) {
        bnp[i] = ok ? p.bn_scale[n] : 0.f;
        bnp[BN + i] = ok ? p.bn_shift[n] : 0.f;
        bnp[2 * BN + i] = (ok && p.bn_mean) ? p.bn_mean[n] : 1.f;      // the residual's own BatchNorm (projection shortcut):
        bnp[3 * BN + i] = (ok && p.bn_rstd) ? p.bn_rstd[n] : 0.f;      // scale / shift travel in bn_mean / bn_rstd
      }
      if (BNEPI) {
        bnp[i] = (ok && bnm == 2) ? p.bn_scale[n] : 0.f;
        bnp[BN + i] = (ok && bnm == 2) ? p.bn_shift[n] : 0.f;
        bnp[2 * BN + i] = (ok && p.bn_mean) ? p.bn_mean[n] : 0.f;
        bnp[3 * BN + i] = (ok && p.bn_rstd) ? p.bn_rstd[n] : 0.f;
      }
      if (EXT) bnp[4 * BN + i] = (ok && p.bias) ? p.bias[n] : 0.f;
    }
    // visibility: the first barrier of the k-loop (or the explicit one before the flush) orders these writes
  }
  if (WSUM && STATS && LDS_EPI) {
    for (int i = lane; i < BN; i += 64) wred[wave * BN + i] = make_float2(0.f, 0.f);     // own slot: ordered by program order
  }
  // row-wise epilogue state: this thread's fixed 8-channel chunk and its partial sums
  const int e_cc = tid % CPR;
  float e_s[8], e_q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { e_s[e] = 0.f; e_q[e] = 0.f; }
  float st_s[NI][4], st_q[NI][4];
  if (STATS) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) { st_s[i][r] = 0.f; st_q[i][r] = 0.f; }
  }

  // issue cursor: runs STAGES-1 pipeline steps ahead of the compute cursor.  With 3 stages two
  // k-tiles are in flight across every barrier (counted vmcnt leaves the newest one outstanding).
  int it = 0, iti = 0, ici = 0, ibuf = 0, buf = 0;
  int issued = 0, consumed = 0;
  const int KU = WIN ? kpt : KT;                          // reduction units of one tile the tail is split in
  auto part_begin = [&]() __attribute__((always_inline)) { return (part_index() * KU) / p.rem_parts; };
  auto part_end = [&]() __attribute__((always_inline)) { return ((part_index() + 1) * KU) / p.rem_parts; };
  const int part_units = has_part ? part_end() - part_begin() : 0;
  const int total = WIN ? (count * kpt + part_units) * 9 : count * KT + part_units;
  auto issue_next = [&]() __attribute__((always_inline)) {
    issue_tile(iti, ici, ibuf);
    ibuf = (ibuf + 1 == STAGES) ? 0 : ibuf + 1;
    ++issued;
    if (++ici == ((EXT && iti == p.ntaps) ? kpt2 : kpt)) {
      ici = 0;
      if (++iti == p.ntaps + ((EXT && kpt2 > 0) ? 1 : 0)) {
        iti = 0;
        ++it;
        if (it < count) setup_rows(mslot + it * mslots);
        else if (it == count && has_part) {           // the partial tile starts at k-step part_begin()
          setup_rows(part_tile());
          const int pka = part_begin();
          iti = pka / kpt; ici = pka - iti * kpt;      // (EXT: steps >= ntaps * kpt belong to the second source)
          if (EXT && iti > p.ntaps) { iti = p.ntaps; ici = pka - p.ntaps * kpt; }
        }
      }
      if (it < count + (has_part ? 1 : 0) && iti < p.ntaps) set_tap(iti);
    }
  };
  // ---- halo-window variant: B tiles run one step ahead through the 2-stage ring, the A window is per (tile, chunk)
  const int Wd = p.IW;
  int wti = 0, wci = (WIN && count == 0 && has_part) ? part_begin() : 0, wit = 0;   // B issue cursor: (tap, chunk) of tile wit, chunk-major steps
  auto issue_b = [&]() __attribute__((always_inline)) {
    const int k0 = (int)((p.tap_w >> (4 * wti)) & 15) * p.IC + wci * BK;
    unsigned char* b_dst = (unsigned char*)(Bw + ibuf * (BN * 8)) + wave * BJ * 1024;
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const void* src = b_ok[j] ? (const void*)(b_src[j] + k0) : p.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(b_dst + j * 1024), 16, 0, 0);
    }
    ibuf ^= 1;
    ++issued;
    if (++wti == 9) {
      wti = 0;
      if (++wci == kpt) { wci = 0; if (++wit == count && has_part) wci = part_begin(); }     // next: the partial tile's first chunk
    }
  };
  auto issue_win = [&](int m0w, int c) __attribute__((always_inline)) {
    for (int j = 0; j < p.win_j; ++j) {
      const int blk = wave * p.win_j + j;                     // wave-uniform 8-row block of the window
      const long long pix = (long long)m0w - (Wd + 1) + 8 * blk + (lane >> 3);
      const void* src = (pix >= 0 && pix < p.M) ? (const void*)(X + pix * p.pixpitch + c * BK + kc * EPC) : p.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + blk * 1024), 16, 0, 0);
    }
  };
  if (WIN) {
    if (tid < 8) Wn[zrow * 8 + tid] = zero16();
    if (total > 0) issue_b();
  } else if (total > 0) {
    if (count > 0) {
      setup_rows(mslot);
      set_tap(0);
    } else {                                   // only the partial tile: start at k-step part_begin()
      setup_rows(part_tile());
      const int pka = part_begin();
      iti = pka / kpt; ici = pka - iti * kpt;
      if (EXT && iti > p.ntaps) { iti = p.ntaps; ici = pka - p.ntaps * kpt; }
      if (iti < p.ntaps) set_tap(iti);
    }
#pragma unroll
    for (int sidx = 0; sidx < STAGES - 1; ++sidx)
      if (issued < total) issue_next();
  }
  for (int ct = 0; ct < count + (has_part ? 1 : 0); ++ct) {
    const bool part = ct == count;                       // the shared remainder tile (split tail)
    const int m0 = (part ? part_tile() : mslot + ct * mslots) * BM;
    const int kt0 = part ? part_begin() : 0, kt1 = part ? part_end() : KU;
    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (WIN) {
      const int m0w = m0;
      // per fragment row: window row of the centre pixel and the 9-bit validity mask of its 3x3 neighbourhood
      int rb[MI];
      unsigned vm[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int ml = wm * (MI * 16) + i * 16 + fl;
        rb[i] = ml + Wd + 1;
        const int m = m0w + ml;
        unsigned msk = 0;
        if (m < p.M) {
          const int q = m / Wd, x = m - q * Wd, y = q % p.IH;
          const unsigned cm = (x > 0 ? 1u : 0u) | 2u | (x < Wd - 1 ? 4u : 0u);
          msk = (y > 0 ? cm : 0u) | (cm << 3) | (y < p.IH - 1 ? cm << 6 : 0u);
        }
        vm[i] = msk;
      }
      // window row (or the zero row) every fragment row reads at step t
      auto win_rows = [&](int t, int zr, int* rr) __attribute__((always_inline)) {
        const int ody = (int)((p.dy_w >> (4 * t)) & 15) - 8 - (MODE == MODE_FWD ? p.pad : 0);
        const int odx = (int)((p.dx_w >> (4 * t)) & 15) - 8 - (MODE == MODE_FWD ? p.pad : 0);
        const int doff = ody * Wd + odx, vbit = (ody + 1) * 3 + (odx + 1);
#pragma unroll
        for (int i = 0; i < MI; ++i) rr[i] = ((vm[i] >> vbit) & 1u) ? rb[i] + doff : zr;
      };
      auto win_load = [&](const u32x4* Wc, const int* rr, const u32x4* Bt, int ks, u32x4* af, u32x4* bf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = Wc[rr[i] * 8 + ((ks * 4 + g) ^ (rr[i] & 7))];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int r = wn * 64 + i * 16 + fl;
          bf[i] = Bt[r * 8 + ((ks * 4 + g) ^ (r & 7))];
        }
      };
      auto win_mma = [&](const u32x4* af, const u32x4* bf) __attribute__((always_inline)) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[ni][mi] = MMA<T>::run(bf[ni], af[mi], acc[ni][mi]);
      };
      auto win_step = [&](const u32x4* Wc, int zr, const u32x4* Bt, int t) __attribute__((always_inline)) {
        int rr[MI];
        win_rows(t, zr, rr);
        if constexpr (sizeof(T) == 4) {
          mma_f32_chunks<NI, MI, false, SPL, PSB, PSX || WPS>(&acc[0][0],
              [&](int i, int ks) { const int r = wn * 64 + i * 16 + fl; return Bt[r * 8 + ((ks * 4 + g) ^ (r & 7))]; },
              [&](int i, int ks) { return Wc[rr[i] * 8 + ((ks * 4 + g) ^ (rr[i] & 7))]; });
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          u32x4 af[MI], bf[NI];
          win_load(Wc, rr, Bt, ks, af, bf);
          win_mma(af, bf);
        }
        }
      };
      for (int c = kt0; c < kt1; ++c) {
        __syncthreads();                     // every wave is done with the previous window / C staging contents
        issue_win(m0w, c);
        for (int t = 0; t < 9; ++t) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (issued < total) issue_b();
          if constexpr (WPS) {
            if (t == 0) {
              // the window of an fp32 tensor, split IN PLACE once per chunk: the two 16-byte chunks (g, 4 + g) of a row -- the eight values
              // lane group g multiplies in one MFMA -- become their (hi, lo) pieces, i.e. the row becomes the pre-split block the nine tap
              // steps of all four waves then read without any VALU work (in registers every element was split 9 taps x 2 waves times).
              // Same split function on the same pairs: bitwise the in-register result.
              const int nq = p.win_j * 32 * 4;
              for (int q = tid; q < nq; q += NW * 64) {
                const int row = q >> 2, gq = q & 3;
                u32x4* w0 = Wn + row * 8 + (gq ^ (row & 7));
                u32x4* w1 = Wn + row * 8 + ((4 + gq) ^ (row & 7));
                u32x4 hi, lo;
                if constexpr (SPL == 13) split_terms2_f16(*w0, *w1, hi, lo); else split_terms2(*w0, *w1, hi, lo);
                *w0 = hi; *w1 = lo;
              }
              __syncthreads();
            }
          }
          win_step(Wn, zrow, Bw + buf * (BN * 8), t);
          buf ^= 1;
          ++consumed;
        }
      }
    } else
    for (int kt = kt0; kt < kt1; ++kt) {
      // the tile about to be consumed must have landed; newer tiles may stay in flight
      if (STAGES == 3 && issued - consumed >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AJ + BJ) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (STAGES == 3) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      } else {
        __syncthreads();
      }
      if (issued < total) issue_next();
      if constexpr (sizeof(T) == 4) {
        if (!DIAG(1))
          mma_f32_chunks<NI, MI, false, SPL, PSB, PSX>(&acc[0][0],
              [&](int i, int ks) { const int r = wn * 64 + i * 16 + fl; return Bs[buf * STG + r * 8 + ((ks * 4 + g) ^ (r & 7))]; },
              [&](int i, int ks) { const int r = wm * (MI * 16) + i * 16 + fl; return As[buf * STG + r * 8 + ((ks * 4 + g) ^ (r & 7))]; });
      } else
      if (!DIAG(1))
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        constexpr int MH = MI > 4 ? 4 : MI;          // m-fragments in flight (8-fragment waves: two groups of four)
        u32x4 bf[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int r = wn * 64 + i * 16 + fl;
          bf[i] = Bs[buf * STG + r * 8 + ((ks * 4 + g) ^ (r & 7))];
        }
#pragma unroll
        for (int mh = 0; mh < MI; mh += MH) {
          u32x4 af[MH];
#pragma unroll
          for (int i = 0; i < MH; ++i) {
            const int r = wm * (MI * 16) + (mh + i) * 16 + fl;
            af[i] = As[buf * STG + r * 8 + ((ks * 4 + g) ^ (r & 7))];
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MH; ++mi) acc[ni][mh + mi] = MMA<T>::run(bf[ni], af[mi], acc[ni][mh + mi]);
        }
      }
      buf = (buf + 1 == STAGES) ? 0 : buf + 1;
      ++consumed;
    }
    if (DIAG(4)) {
      if (acc[0][0][0] == 12345.678f) Y[0] = (T)0;     // keeps the accumulators live
      continue;
    }
    if constexpr (SPL == 13) {                          // the fp16 weight planes carry 2^F16_WSCALE_LOG2 (exact to undo)
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] *= (1.0f / (float)(1 << F16_WSCALE_LOG2));
    }
    if (part) {
      // split tail: fp32 accumulators travel lane-linear ([fragment][thread] float4: fully coalesced both ways).
      // Publish / acquire as cdna_hip_programming.md section 5 prescribes for in-launch split-K partials (write-through form):
      // sc1 stores, every wave drains vmcnt, barrier, ONE lane stores the flag (relaxed, agent scope); the owner polls
      // relaxed, fences (acquire, agent) once, barrier, plain loads.  Correct for any placement of the parts over CUs / XCDs.
      const long long slot0 = ((long long)(mslot / p.rem_parts) * p.n_tiles + nt) * (p.rem_parts - 1);
      const int pj = part_index();
      constexpr int FR = NI * MI;
      if (pj != 0) {
        // write-through (sc1) stores: the data leaves this XCD's L2 at once, so no release fence is needed -- an agent-scope
        // release (buffer_wbl2) would write back EVERY dirty line of the L2, i.e. the output tiles all the other workgroups
        // are streaming (measured: +2.2 ms per training step with the fence form).  Buffer stores through a descriptor of
        // this part's slot: one 32-bit lane offset + a scalar fragment offset (per-fragment 64-bit lane addresses cost two
        // registers each; an inline-asm global_store with a scalar base misses the VALU-writes-SGPR wait states hipcc only
        // inserts for its own instructions -- memory faults).  aux = 16: sc1.
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((f32x4*)p.part_ws + (slot0 + pj - 1) * (long long)(FR * NTH)), 0, FR * NTH * 16, 0x00020000);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[ni][mi]), rp, tid * 16, (ni * MI + mi) * NTH * 16, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
          __hip_atomic_store(p.part_flags + slot0 + pj - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;                                        // the last tile of this workgroup: on to the statistics flush
      }
      if (tid == 0) {
        for (int j = 1; j < p.rem_parts; ++j) {
          int spins = 0;                                 // bounded: a lost partner must not hang the GPU (~0.5 s)
          while (__hip_atomic_load(p.part_flags + slot0 + j - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u &&
                 ++spins < (1 << 22))
            __builtin_amdgcn_s_sleep(8);
          if (spins >= (1 << 22)) atomicAdd(p.part_err, 1u);          // reported, never silent
          else __hip_atomic_store(p.part_flags + slot0 + j - 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      for (int j = 1; j < p.rem_parts; ++j) {
        const f32x4* src = (const f32x4*)p.part_ws + (slot0 + j - 1) * (long long)(FR * NTH);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {              // one row of fragments at a time: the loads must not all be live at once
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) acc[ni][mi] += src[(ni * MI + mi) * NTH + tid];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    auto row_off = [&](int m) __attribute__((always_inline)) -> long long {
      if (m >= p.M) return -1;
      if (flat) return (long long)m * p.N;
      const int v = m / cls_hw;
      const int rem = m - v * cls_hw;
      const int ca = rem / p.cls_w, cb = rem - ca * p.cls_w;
      return (((long long)v * p.OH + ca * p.cs + p.py) * p.OW + cb * p.cs + p.px) * p.N;
    };
    if (LDS_EPI) {
      // ---- bf16 epilogue through LDS: the stage consumed last is free until the next barrier-issue.
      // 1) barrier (every wave is done reading it)  2) accumulators -> bf16 C tile (8-byte granules,
      // XOR-swizzled by row)  3) barrier  4) row-wise pass: 16-byte coalesced loads of the BN input /
      // mask / previous value, ReLU mask, per-channel sums, 16-byte coalesced stores.
      const int cst = (buf == 0) ? STAGES - 1 : buf - 1;
      unsigned char* Cs = WIN ? smem : (unsigned char*)(As + cst * STG);
#pragma unroll
      for (int eh = 0; eh < EH; ++eh) {
      __syncthreads();
      if (eh == 0 && tid < BM) rowoff[tid] = row_off(m0 + tid);
      if (EH == 1 || wm / (WM / EH) == eh)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int ml = wm * (MI * 16) + mi * 16 + fl - eh * BMH;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if (STATS && !ROWSTATS) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float v = acc[ni][mi][r]; st_s[ni][r] += v; st_q[ni][r] += v * v; }
          }
          const int q = wn * 16 + ni * 4 + g;                    // 8-byte granule (4 channels) in the row
          u32x2 pk;
          pk[0] = pack_bf16x2(acc[ni][mi][0], acc[ni][mi][1]);
          pk[1] = pack_bf16x2(acc[ni][mi][2], acc[ni][mi][3]);
          *(u32x2*)(Cs + ml * (BN * 2) + ((q ^ ((ml & 7) << 1)) << 3)) = pk;
        }
      }
      __syncthreads();
      // All global operands of this thread's ER rows (previous value / BN input / mask) are requested
      // back to back with branch-free addresses, so the row pass pays ONE memory round trip per tile
      // instead of one per row pair.
      // (256-wide tile: the waves of the other half still hold 128 accumulator registers, so the rows go in batches of 4)
      constexpr int ERT = BMH / RPP;
      constexpr int ER = (BM == 256 && ERT > 4) ? 4 : ERT;
      const int ncol = n0 + e_cc * 8;
#pragma unroll 1
      for (int eb = 0; eb < ERT; eb += ER) {
      long long eoff[ER];
      bool erok[ER];
#pragma unroll
      for (int i = 0; i < ER; ++i) {
        const long long off = rowoff[eh * BMH + tid / CPR + (eb + i) * RPP];
        erok[i] = off >= 0 && ncol < p.N && Y != nullptr;     // y == NULL: statistics-only pass, nothing is stored
        eoff[i] = erok[i] ? off + ncol : 0;       // masked rows read (and ignore) element 0
      }
      long long eld[ER];
#pragma unroll
      for (int i = 0; i < ER; ++i) eld[i] = DIAG(8) ? 0 : eoff[i];
      u32x4 e_ov[ER], e_xv[ER], e_mv[ER];
      if (accu) {
#pragma unroll
        for (int i = 0; i < ER; ++i) e_ov[i] = *(const u32x4*)((const uint16_t*)Y + eld[i]);
      }
      if (FAPPLY && fa_res) {          // residual operand of the fused BatchNorm apply
#pragma unroll
        for (int i = 0; i < ER; ++i) e_xv[i] = *(const u32x4*)((const uint16_t*)p.bn_x + eld[i]);
      }
      if (BNEPI) {
        if (bnm != 4) {
#pragma unroll
          for (int i = 0; i < ER; ++i) e_xv[i] = *(const u32x4*)((const uint16_t*)p.bn_x + eld[i]);
        } else {
#pragma unroll
          for (int i = 0; i < ER; ++i) e_xv[i] = zero16();
        }
        if (bnm == 1) {
#pragma unroll
          for (int i = 0; i < ER; ++i) e_mv[i] = *(const u32x4*)((const uint16_t*)p.bn_mask + eld[i]);
        } else if (bnm >= 3) {       // one mask byte per 8-channel chunk (written by simclr_bn_apply)
#pragma unroll
          for (int i = 0; i < ER; ++i) e_mv[i][0] = ((const unsigned char*)p.bn_mask)[eld[i] >> 3];
        }
      }
#pragma unroll
      for (int i = 0; i < ER; ++i) {
        if (!FAPPLY && !(ROWSTATS && !BNEPI) && !erok[i]) continue;
        const int r = tid / CPR + (eb + i) * RPP;
        const u32x4 cv = *(const u32x4*)(Cs + r * (BN * 2) + (((e_cc * 2) ^ ((r & 7) << 1)) << 3));
        uint16_t* dst = (uint16_t*)Y + eoff[i];
        float v[8];
        chunk_to_f32<uint16_t>(cv, v);
        if (ROWSTATS && !BNEPI) {          // rows >= M and columns >= N hold exact zeros: no masking needed
#pragma unroll
          for (int e = 0; e < 8; ++e) { e_s[e] += v[e]; e_q[e] = fmaf(v[e], v[e], e_q[e]); }
          if (!erok[i]) continue;
        }
        if (FAPPLY) {
          // the arithmetic of bn_apply<RES = 0 | 1> on the bf16-rounded convolution result: bitwise the same output as
          // conv -> HBM -> bn_apply
          float qv[8];
          if (fa_res) chunk_to_f32<uint16_t>(e_xv[i], qv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float o = fmaf(v[e], bnp[e_cc * 8 + e], bnp[BN + e_cc * 8 + e]);
            if (fa_res) o += fa_rbn ? fmaf(qv[e], bnp[2 * BN + e_cc * 8 + e], bnp[3 * BN + e_cc * 8 + e]) : qv[e];
            v[e] = fa_relu ? fmaxf(o, 0.f) : o;             // bn_mode doubles as the ReLU flag here
          }
          const u32x4 packed = f32_to_chunk<uint16_t>(v);
          const bool ok = erok[i];                          // invalid rows run along (wave-wide shuffles below), store nothing
          if (ok) *(u32x4*)dst = packed;
          if (fa_mask) {                                    // bit e = (stored y[e] > 0), one byte per 16-byte chunk
            float w8[8];
            chunk_to_f32<uint16_t>(packed, w8);
            unsigned bits = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) bits |= (w8[e] > 0.f ? 1u : 0u) << e;
            if (fa_n32) {
              // four neighbouring lanes hold four consecutive mask bytes of one row (validity is uniform over such a
              // group when N is a multiple of 32): one aligned 4-byte store instead of four 1-byte stores
              const unsigned b1 = __shfl_down(bits, 1, 64), b2 = __shfl_down(bits, 2, 64), b3 = __shfl_down(bits, 3, 64);
              if (ok && (e_cc & 3) == 0)
                *(unsigned*)((unsigned char*)p.bn_mask + (eoff[i] >> 3)) = bits | (b1 << 8) | (b2 << 16) | (b3 << 24);
            } else if (ok) {
              ((unsigned char*)p.bn_mask)[eoff[i] >> 3] = (unsigned char)bits;
            }
          }
          continue;
        }
        if (EXT) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bnp[4 * BN + e_cc * 8 + e];
        }
        if (accu) {
          float o[8];
          chunk_to_f32<uint16_t>(e_ov[i], o);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += o[e];
        }
        if (BNEPI) {
          // This pass is VALU-bound (profiles/r02_notes.md), so it accumulates the RAW moment sum(dm * x) -- one fma per
          // element -- and the flush turns it into sum(dm * x^) = rstd * (sum(dm * x) - mean * sum(dm)) once per
          // workgroup; mode 4 (sums only) touches neither x nor the second moment.
          if (bnm == 4) {
            const unsigned mb = e_mv[i][0];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              v[e] = ((mb >> e) & 1u) ? v[e] : 0.f;
              e_s[e] += v[e];
            }
          } else {
            float xf[8];
            chunk_to_f32<uint16_t>(e_xv[i], xf);
            if (bnm == 1) {
              float mk[8];
              chunk_to_f32<uint16_t>(e_mv[i], mk);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
            } else if (bnm == 3) {
              const unsigned mb = e_mv[i][0];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = ((mb >> e) & 1u) ? v[e] : 0.f;
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e)
                v[e] = fmaf(xf[e], bnp[e_cc * 8 + e], bnp[BN + e_cc * 8 + e]) > 0.f ? v[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              e_s[e] += v[e];
              e_q[e] = fmaf(v[e], xf[e], e_q[e]);
            }
          }
        }
        if (DIAG(16)) continue;
        if (accu || BNEPI || EXT) *(u32x4*)dst = f32_to_chunk<uint16_t>(v);
        else *(u32x4*)dst = cv;
      }
      }   // eb
      if (WSUM && ROWSTATS) {
        static_assert(!WSUM || CPR == 32, "lanes l and l + 32 of a wave own the same channel chunk");
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float s1 = e_s[e] + __shfl_xor(e_s[e], 32, 64), s2 = e_q[e] + __shfl_xor(e_q[e], 32, 64);
          if (lane < 32) {
            float2 o = wred[wave * BN + e_cc * 8 + e];
            o.x += s1; o.y += s2;
            wred[wave * BN + e_cc * 8 + e] = o;
          }
          e_s[e] = 0.f; e_q[e] = 0.f;
        }
      }
      }   // eh
    } else if constexpr (FAPPLY) {
    // ---- fp32 fused BatchNorm-apply epilogue: y = act(conv * scale + shift + res [* rscale + rshift]) with bn_apply<float>'s arithmetic
    // (csrc/bn.hip) on the fp32 accumulators -- bitwise what conv -> HBM -> bn_apply produces.  All residual operands of the tile are
    // requested up front (one memory round trip per tile, not one per fragment); ReLU bits: one byte per 16-byte chunk = one lane's 4
    // channels, the 4 lane groups of a fragment row hold 4 consecutive bytes -> one 4-byte store.
    long long offs[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) offs[mi] = row_off(m0 + wm * (MI * 16) + mi * 16 + fl);
    const bool n16 = (p.N & 15) == 0;
    constexpr int MB = 1;                        // fragment rows per batch of residual loads (16 registers in flight; 2 rows spill)
#pragma unroll
    for (int mb = 0; mb < MI; mb += MB) {
    float4 rr[MI][NI];
    if (fa_res) {
#pragma unroll
      for (int mi = mb; mi < mb + MB; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int n = n0 + wn * 64 + ni * 16 + g * 4;
          const bool ok = offs[mi] >= 0 && n < p.N;
          rr[mi][ni] = *(const float4*)((const float*)p.bn_x + (ok ? offs[mi] + n : 0));      // masked lanes read (and ignore) element 0
        }
    }
#pragma unroll
    for (int mi = mb; mi < mb + MB; ++mi) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int nl = wn * 64 + ni * 16 + g * 4, n = n0 + nl;
        const bool ok = offs[mi] >= 0 && n < p.N;
        const float4 sc = *(const float4*)(bnp + nl), sh = *(const float4*)(bnp + BN + nl);
        float o[4] = {fmaf(acc[ni][mi][0], sc.x, sh.x), fmaf(acc[ni][mi][1], sc.y, sh.y), fmaf(acc[ni][mi][2], sc.z, sh.z), fmaf(acc[ni][mi][3], sc.w, sh.w)};
        if (fa_res) {
          const float4 r = rr[mi][ni];
          if (fa_rbn) {
            const float4 rs = *(const float4*)(bnp + 2 * BN + nl), rb = *(const float4*)(bnp + 3 * BN + nl);
            o[0] += fmaf(r.x, rs.x, rb.x); o[1] += fmaf(r.y, rs.y, rb.y); o[2] += fmaf(r.z, rs.z, rb.z); o[3] += fmaf(r.w, rs.w, rb.w);
          } else { o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w; }
        }
        if (fa_relu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = fmaxf(o[r], 0.f);
        }
        if (ok) *(float4*)(Y + offs[mi] + n) = make_float4(o[0], o[1], o[2], o[3]);
        if (fa_mask) {
          const unsigned bits = (o[0] > 0.f ? 1u : 0u) | (o[1] > 0.f ? 2u : 0u) | (o[2] > 0.f ? 4u : 0u) | (o[3] > 0.f ? 8u : 0u);
          if (n16) {       // validity is uniform over the 4 lane groups of a row when N is a multiple of 16
            const unsigned b1 = __shfl(bits, lane + 16, 64), b2 = __shfl(bits, lane + 32, 64), b3 = __shfl(bits, lane + 48, 64);
            if (ok && g == 0) *(unsigned*)((unsigned char*)p.bn_mask + ((offs[mi] + n) >> 2)) = bits | (b1 << 8) | (b2 << 16) | (b3 << 24);
          } else if (ok) {
            ((unsigned char*)p.bn_mask)[(offs[mi] + n) >> 2] = (unsigned char)bits;
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);      // the next batch's 32 load registers are not live before this batch has been stored
    }   // mb
    } else {
    // ---- fp32 parity mode: registers -> NHWC (4 consecutive channels per lane), no LDS, no barrier
    float4 pvt[NI];                                   // pivoted statistics: this lane's 4 x NI pivots, once per tile
    if (sizeof(T) == 4 && MODE == MODE_FWD && STATS && !BNEPI && p.pivot) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * 64 + ni * 16 + g * 4;
        pvt[ni] = n < p.N ? *(const float4*)(p.pivot + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = m0 + wm * (MI * 16) + mi * 16 + fl;
      long long off = -1;
      // accumulate == 2: dx holds earlier data ONLY at the pixels with even row and even column (a stride-2 1x1 data gradient that
      // skipped its empty pixel classes, accumulate == 3 there) -- the other three quarters are neither read nor zero-filled by anyone
      bool prior = p.accumulate != 0;
      if (flat) {
        if (m < p.M) off = (long long)m * p.N;
        if (p.accumulate == 2) {
          const int rem = m % (p.OH * p.OW), iy = rem / p.OW, ix = rem - iy * p.OW;
          prior = ((iy | ix) & 1) == 0;
        }
      } else if (m < p.M) {
        const int v = m / cls_hw;
        const int rem = m - v * cls_hw;
        const int ca = rem / p.cls_w, cb = rem - ca * p.cls_w;
        off = (((long long)v * p.OH + ca * p.cs + p.py) * p.OW + cb * p.cs + p.px) * p.N;
        if (p.accumulate == 2) prior = (((ca * p.cs + p.py) | (cb * p.cs + p.px)) & 1) == 0;
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * 64 + ni * 16 + g * 4;
        if (STATS && !BNEPI) {
          if (sizeof(T) == 4 && MODE == MODE_FWD && p.pivot) {
            if (off >= 0 && n < p.N) {          // padding rows / columns hold zeros, which are NOT zero about the pivot
              const float pa[4] = {pvt[ni].x, pvt[ni].y, pvt[ni].z, pvt[ni].w};
#pragma unroll
              for (int r = 0; r < 4; ++r) { const float v = acc[ni][mi][r] - pa[r]; st_s[ni][r] += v; st_q[ni][r] += v * v; }
            }
          } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float v = acc[ni][mi][r]; st_s[ni][r] += v; st_q[ni][r] += v * v; }
          }
        }
        if (off >= 0 && n < p.N) {
          T* dst = Y + off + n;
          float v[4] = {acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]};
          if (EXT) {
            const float4 bi = *(const float4*)(bnp + 4 * BN + wn * 64 + ni * 16 + g * 4);
            v[0] += bi.x; v[1] += bi.y; v[2] += bi.z; v[3] += bi.w;
          }
          if (prior) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += Elem<T>::ld(dst + r);
          }
          if (BNEPI) {
            // v = gradient wrt the BN(+ReLU) output at (row, channels n..n+3)
            const int nl = wn * 64 + ni * 16 + g * 4;
            float xf[4] = {0.f, 0.f, 0.f, 0.f}, mk[4];
            if (p.bn_mode == 4) {
            } else if (sizeof(T) == 4) {
              const float4 xv = *(const float4*)((const float*)p.bn_x + off + n);
              xf[0] = xv.x; xf[1] = xv.y; xf[2] = xv.z; xf[3] = xv.w;
            } else {
              const u32x2 xv = *(const u32x2*)((const uint16_t*)p.bn_x + off + n);
              xf[0] = __uint_as_float(xv[0] << 16); xf[1] = __uint_as_float(xv[0] & 0xffff0000u);
              xf[2] = __uint_as_float(xv[1] << 16); xf[3] = __uint_as_float(xv[1] & 0xffff0000u);
            }
            if (p.bn_mode >= 3) {
              const unsigned mb = ((const unsigned char*)p.bn_mask)[(off + n) / EPC] >> (n % EPC);
#pragma unroll
              for (int r = 0; r < 4; ++r) mk[r] = ((mb >> r) & 1u) ? 1.f : 0.f;
            } else if (p.bn_mode == 1) {
              if (sizeof(T) == 4) {
                const float4 mv = *(const float4*)((const float*)p.bn_mask + off + n);
                mk[0] = mv.x; mk[1] = mv.y; mk[2] = mv.z; mk[3] = mv.w;
              } else {
                const u32x2 mv = *(const u32x2*)((const uint16_t*)p.bn_mask + off + n);
                mk[0] = __uint_as_float(mv[0] << 16); mk[1] = __uint_as_float(mv[0] & 0xffff0000u);
                mk[2] = __uint_as_float(mv[1] << 16); mk[3] = __uint_as_float(mv[1] & 0xffff0000u);
              }
            } else {
              const float4 sc = *(const float4*)(bnp + nl), sh = *(const float4*)(bnp + BN + nl);
              mk[0] = fmaf(xf[0], sc.x, sh.x); mk[1] = fmaf(xf[1], sc.y, sh.y);
              mk[2] = fmaf(xf[2], sc.z, sh.z); mk[3] = fmaf(xf[3], sc.w, sh.w);
            }
            const float4 mu = *(const float4*)(bnp + 2 * BN + nl), rs = *(const float4*)(bnp + 3 * BN + nl);
            const float mua[4] = {mu.x, mu.y, mu.z, mu.w}, rsa[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = mk[r] > 0.f ? v[r] : 0.f;
              st_s[ni][r] += v[r];
              st_q[ni][r] += v[r] * (xf[r] - mua[r]) * rsa[r];
            }
          }
          if (sizeof(T) == 4) {
            *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            u32x2 pk; pk[0] = pack_bf16x2(v[0], v[1]); pk[1] = pack_bf16x2(v[2], v[3]);
            *(u32x2*)dst = pk;
          }
        }
      }
    }
    }
  }
  if (ROWSTATS) {
    // flush of the row-wise sums: NTH threads = RPP row-lanes x CPR channel chunks -> LDS -> atomics
    __syncthreads();
    float* red = WSUM ? (float*)wred : (float*)smem;  // [RPP | NW][BN][2]
    constexpr int NRED = WSUM ? NW : RPP;
    if (!WSUM) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[((tid / CPR) * BN + e_cc * 8 + e) * 2] = e_s[e];
        red[((tid / CPR) * BN + e_cc * 8 + e) * 2 + 1] = e_q[e];
      }
      __syncthreads();
    }
    // nslot >= mslots (what simclr_conv2d_stats_slots returns): every workgroup owns slot `mslot` of its channels and
    // stores its sums there -- no float atomics, so the statistics are bit-identical from run to run (idle
    // workgroups store zeros).  Fewer slots: atomics into slot mslot % nslot (order-dependent rounding).
    const bool own_slot = p.nslot >= mslots;
    if (tid < BN && n0 + tid < p.N && (count > 0 || own_slot)) {
      float s1 = 0.f, s2 = 0.f;
      for (int w = 0; w < NRED; ++w) { s1 += red[(w * BN + tid) * 2]; s2 += red[(w * BN + tid) * 2 + 1]; }
      if (BNEPI) s2 = (s2 - bnp[2 * BN + tid] * s1) * bnp[3 * BN + tid];     // raw moment -> sum(dm * x^)  (mode 4: s2 = 0, mean = rstd = 0)
      float* st = p.stats + (long long)(own_slot ? mslot : mslot % p.nslot) * 2 * p.N;
      if (own_slot) { st[n0 + tid] = s1; st[p.N + n0 + tid] = s2; }
      else { atomicAdd(st + n0 + tid, s1); atomicAdd(st + p.N + n0 + tid, s2); }
    }
    return;
  }
  if (STATS) {
    // one flush per workgroup: lanes -> 16-lane groups -> waves (LDS) -> atomics into a slot
    __syncthreads();
    float* red = (float*)smem;  // [WM][BN][2]
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = st_s[ni][r], ss = st_q[ni][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
        if (fl == 0) {
          const int nl = wn * 64 + ni * 16 + g * 4 + r;
          red[(wm * BN + nl) * 2] = s;
          red[(wm * BN + nl) * 2 + 1] = ss;
        }
      }
    }
    __syncthreads();
    const bool own_slot = p.nslot >= mslots;       // see the BNEPI flush above
    if (tid < BN && n0 + tid < p.N && (count > 0 || own_slot)) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { s += red[(w * BN + tid) * 2]; ss += red[(w * BN + tid) * 2 + 1]; }
      float* st = p.stats + (long long)(own_slot ? mslot : mslot % p.nslot) * 2 * p.N;
      if (own_slot) { st[n0 + tid] = s; st[p.N + n0 + tid] = ss; }
      else { atomicAdd(st + n0 + tid, s); atomicAdd(st + p.N + n0 + tid, ss); }
    }
  }
}

#include "igemm_wide.h"

// ------------------------------------------------------------------------------------
// wgrad: dW[K, N] (fp32 split-slabs) = sum over pixels A(m, k) * dY(m, n).
// Tile BKW (k rows: one tap x BKW input channels) x BNW, reduction chunks of BR pixels.
// LDS tiles keep the natural [pixel][channel] layout (coalesced 16-byte staging); the
// 32-byte channel blocks are XOR-permuted by the pixel group so the per-element
// fragment gathers of the 4 lane groups hit different banks.
// ------------------------------------------------------------------------------------
struct WgradP {
  const void* x;    // activation [V, IH, IW, pixpitch...]
  const void* dy;   // [M, N]
  float* dw;        // [splits][K][N] fp32 slabs
  int V, IH, IW, IC, OH, OW, N, KH, KW, stride, pad, pixpitch;
  int M, K, splits, chunks_per_split;
  int k_tiles, n_tiles;
  int xcd_map;   // 1: all tiles of a pixel range on one XCD (big tensors); 0: plain interleaving
  const void* zero;   // 16 zero bytes (padding source for the direct-to-LDS loads)
  int diag;
  int split;     // fp32: 0 = exact fp32 MFMA, 3 / 6 = split-bf16 terms (simclr_set_f32_matmul)
  int dy_ps;     // fp32, three terms: dy is in the pre-split block format (PSD instantiations)
};

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

// XOR key that spreads the 32-byte channel blocks of a pixel row over the LDS banks.
//  bf16: the 8 pixel rows one 32-lane half touches in a ds_read_b64_tr_b16 (pixels 8g+q, q<4)
//        get 8 distinct keys;
//  f32:  the four lane groups of a fragment read read 16 floats (64 bytes = TWO blocks) of four CONSECUTIVE pixels, so those pixels need
//        keys that differ above bit 0: (px & 3) << 1 | bit 2 of px -- eight consecutive pixels still get eight distinct keys.  (Round 6:
//        the old key px & 7 put pixels 4j and 4j + 1 into the same 64-byte bank range -- SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50
//        on every fp32 weight-gradient launch that splits in registers and on the exact Gram launches, r06_call29.)
template <typename T> __device__ __forceinline__ int px_key(int px) {
  return sizeof(T) == 2 ? ((px & 3) | (((px >> 3) & 1) << 2)) : (((px & 3) << 1) | ((px >> 2) & 1));
}

// MT (multi-tap k-tile): the BKW rows of the tile span SEVERAL taps of IC < BKW channels each (the stem: 7 kernel rows x 32
// packed elements = 224 rows in one 256-row tile), so the gradient tensor is read once instead of once per tap.
template <typename T, int BKW, int BNW, bool MT = false>
__global__ __launch_bounds__(256) void conv_wgrad(const WgradP p) {
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BR = 8 * EPC;                 // pixels per reduction chunk (64 bf16 / 32 f32)
  constexpr int WK = 2;                       // waves along k rows
  constexpr int WNN = 2;                      // waves along n
  constexpr int KI = BKW / WK / 16;           // 16-row fragments per wave along k
  constexpr int NI = BNW / WNN / 16;
  constexpr int A_CPR = BKW / EPC;            // 16B chunks per pixel row of the A tile
  constexpr int B_CPR = BNW / EPC;
  constexpr int A_CH = BR * A_CPR / 256;      // chunks per thread
  constexpr int B_CH = BR * B_CPR / 256;
  constexpr int A_RB = BKW * sizeof(T);       // row bytes
  constexpr int B_RB = BNW * sizeof(T);
  constexpr int A_BLK = A_RB / 32;            // 32-byte blocks per pixel row
  constexpr int B_BLK = B_RB / 32;
  constexpr int BUF = BR * (A_RB + B_RB);     // bytes per LDS buffer
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR (LDS bases, M0)
  const int g = lane >> 4, fl = lane & 15;
  const int wk = wave / WNN, wn = wave % WNN;
  // XCD-aware mapping (workgroup b runs on XCD b%8): every (tap/k-tile, n-tile) workgroup of one
  // pixel range ("split") sits on the SAME XCD, so the 9 taps x N-tiles that re-read the same
  // activation / gradient rows are served by that XCD's L2 instead of 8 separate fabric fetches.
  // Measured per layer (profiles/r01_notes.md): a win for the 56x56 layers and the 28x28 1x1 layers
  // (tensors far larger than L2/MALL), a loss for the small-spatial ones -> chosen by the host.
  const int tiles = p.k_tiles * p.n_tiles;
  int tile, split;
  if (p.xcd_map) {
    const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
    tile = bidx % tiles;
    split = (bidx / tiles) * 8 + xcd;
  } else {
    tile = blockIdx.x % tiles;
    split = blockIdx.x / tiles;
  }
  if (split >= p.splits) return;
  const int ktile = tile % p.k_tiles;
  const int ntile = tile / p.k_tiles;
  const int kk0 = ktile * BKW;
  const int tap = kk0 / p.IC, ci0 = kk0 - tap * p.IC;
  const int ty = tap / p.KW, tx = tap - ty * p.KW;
  const int n0 = ntile * BNW;
  const T* __restrict__ X = (const T*)p.x;
  const T* __restrict__ DY = (const T*)p.dy;

  f32x4 acc[KI][NI];
#pragma unroll
  for (int i = 0; i < KI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = (p.M + BR - 1) / BR;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(nchunks, c_begin + p.chunks_per_split);

  u32x4 ra[A_CH], rb[B_CH];
#ifdef SIMCLR_DIAG
#pragma unroll
  for (int j = 0; j < A_CH; ++j) ra[j] = zero16();
#pragma unroll
  for (int j = 0; j < B_CH; ++j) rb[j] = zero16();
#endif
  // 1x1 stride-1 (and Dense): reduction pixel m reads input pixel m -- no decode
  const bool flat = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.IH == p.OH && p.IW == p.OW;
  const int ohow = p.OH * p.OW;
  const float inv_ow = 1.0f / (float)p.OW;
  auto load_chunk = [&](int c) {
    if (DIAG(2)) return;
    const int mbase = c * BR;
    // (image, pixel-in-image) of the chunk's first pixel: ONE division per chunk (uniform), the
    // per-lane pixel then needs at most two wrap steps and a reciprocal multiply (exact: rem < 2^14)
    const int vbase = flat ? 0 : mbase / ohow;
    const int rbase = flat ? 0 : mbase - vbase * ohow;
#pragma unroll
    for (int j = 0; j < A_CH; ++j) {
      const int id = tid + 256 * j;
      const int px = id / A_CPR, cc = id % A_CPR;
      const int m = mbase + px;
      ra[j] = zero16();
      if (MT) {
        const int kidx = kk0 + cc * EPC;                 // this chunk's own tap / channel offset
        if (m < p.M && kidx < p.K) {
          const int tapc = kidx / p.IC, cic = kidx - tapc * p.IC;
          const int tyc = tapc / p.KW, txc = tapc - tyc * p.KW;
          int v = vbase, rem = rbase + px;
          if (rem >= ohow) { rem -= ohow; ++v; }
          if (rem >= ohow) { rem -= ohow; ++v; }
          if (rem >= ohow) { v = m / ohow; rem = m - v * ohow; }
          const int oy = (int)(((float)rem + 0.5f) * inv_ow), ox = rem - oy * p.OW;
          const int iy = oy * p.stride - p.pad + tyc, ix = ox * p.stride - p.pad + txc;
          if ((unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW)
            ra[j] = ld16(X + (((long long)v * p.IH + iy) * p.IW + ix) * p.pixpitch + cic);
        }
      } else if (m < p.M) {
        if (flat) {
          ra[j] = ld16(X + (long long)m * p.pixpitch + ci0 + cc * EPC);
        } else {
          int v = vbase, rem = rbase + px;
          if (rem >= ohow) { rem -= ohow; ++v; }
          if (rem >= ohow) { rem -= ohow; ++v; }
          if (rem >= ohow) { v = m / ohow; rem = m - v * ohow; }     // tiny images only
          const int oy = (int)(((float)rem + 0.5f) * inv_ow), ox = rem - oy * p.OW;
          const int iy = oy * p.stride - p.pad + ty, ix = ox * p.stride - p.pad + tx;
          if ((unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW)
            ra[j] = ld16(X + (((long long)v * p.IH + iy) * p.IW + ix) * p.pixpitch + ci0 + cc * EPC);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < B_CH; ++j) {
      const int id = tid + 256 * j;
      const int px = id / B_CPR, cc = id % B_CPR;
      const int m = mbase + px;
      rb[j] = zero16();
      if (m < p.M && n0 + cc * EPC < p.N) rb[j] = ld16(DY + (long long)m * p.N + n0 + cc * EPC);
    }
  };
  auto store_chunk = [&](int buf) {
    if (DIAG(4)) return;
    unsigned char* As = smem + buf * BUF;
    unsigned char* Bs = As + BR * A_RB;
#pragma unroll
    for (int j = 0; j < A_CH; ++j) {
      const int id = tid + 256 * j;
      const int px = id / A_CPR, cc = id % A_CPR;
      const int blk = (cc >> 1) ^ (px_key<T>(px) & (A_BLK - 1));
      *(u32x4*)(As + px * A_RB + blk * 32 + (cc & 1) * 16) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < B_CH; ++j) {
      const int id = tid + 256 * j;
      const int px = id / B_CPR, cc = id % B_CPR;
      const int blk = (cc >> 1) ^ (px_key<T>(px) & (B_BLK - 1));
      *(u32x4*)(Bs + px * B_RB + blk * 32 + (cc & 1) * 16) = rb[j];
    }
  };
  // byte offset of element (pixel px, channel ch) inside a tile with RB row bytes / NBLK blocks
  auto elem_off = [&](int RB, int NBLK, int px, int ch) -> int {
    const int byte = ch * (int)sizeof(T);
    return px * RB + (((byte >> 5) ^ (px_key<T>(px) & (NBLK - 1))) << 5) + (byte & 31);
  };
  auto compute = [&](int buf) {
    if (DIAG(1)) return;
    const unsigned char* As = smem + buf * BUF;
    const unsigned char* Bs = As + BR * A_RB;
    if (sizeof(T) == 2) {
      // bf16: hardware transpose reads.  Lane (fl,g) supplies the address of pixel row
      // 8g+(fl>>2) (+4 for the second read), channel sub-chunk (fl&3)*4 of the 16-channel group
      // and receives, for channel `fl`, pixels 8g..8g+3 (+4..7) = the MFMA k layout.
#pragma unroll
      for (int ks = 0; ks < BR / 32; ++ks) {
        u32x4 af[KI], bf[NI];
        const int px0 = ks * 32 + g * 8 + (fl >> 2);
#pragma unroll
        for (int i = 0; i < KI; ++i) {
          const int ch = wk * (KI * 16) + i * 16 + (fl & 3) * 4;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(As + elem_off(A_RB, A_BLK, px0, ch)));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(As + elem_off(A_RB, A_BLK, px0 + 4, ch)));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          af[i] = (u32x4){l2[0], l2[1], h2[0], h2[1]};
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int ch = wn * (NI * 16) + i * 16 + (fl & 3) * 4;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Bs + elem_off(B_RB, B_BLK, px0, ch)));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Bs + elem_off(B_RB, B_BLK, px0 + 4, ch)));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          bf[i] = (u32x4){l2[0], l2[1], h2[0], h2[1]};
        }
#pragma unroll
        for (int ki = 0; ki < KI; ++ki)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, bf[ni]), __builtin_bit_cast(bf16x8, af[ki]), acc[ki][ni], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int ks = 0; ks < BR / 4; ++ks) {
        float af[KI], bf[NI];
        const int px = ks * 4 + g;
#pragma unroll
        for (int i = 0; i < KI; ++i)
          af[i] = *(const float*)(As + elem_off(A_RB, A_BLK, px, wk * (KI * 16) + i * 16 + fl));
#pragma unroll
        for (int i = 0; i < NI; ++i)
          bf[i] = *(const float*)(Bs + elem_off(B_RB, B_BLK, px, wn * (NI * 16) + i * 16 + fl));
#pragma unroll
        for (int ki = 0; ki < KI; ++ki)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[ni], af[ki], acc[ki][ni], 0, 0, 0);
      }
    }
  };

  if (c_begin < c_end) {
    load_chunk(c_begin);
    store_chunk(0);
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; ++c) {
    const int buf = (c - c_begin) & 1;
    if (c + 1 < c_end) load_chunk(c + 1);
    compute(buf);
    if (c + 1 < c_end) store_chunk(buf ^ 1);
    __syncthreads();
  }
  // D[n = g*4+reg][k row = fl]  ->  slab[split][kk][n .. n+3]
  float* slab = p.dw + (long long)split * p.K * p.N;
#pragma unroll
  for (int ki = 0; ki < KI; ++ki) {
    const int kk = kk0 + wk * (KI * 16) + ki * 16 + fl;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * (NI * 16) + ni * 16 + g * 4;
      if (kk < p.K && n < p.N)
        *(float4*)(slab + (long long)kk * p.N + n) =
            make_float4(acc[ki][ni][0], acc[ki][ni][1], acc[ki][ni][2], acc[ki][ni][3]);
    }
  }
}

// ------------------------------------------------------------------------------------
// conv_wgrad_dma: same tiling / LDS layout / MFMA schedule as conv_wgrad, but the operands travel
// global -> LDS by LDS-DMA (global_load_lds, 16 B per lane, the XOR block permutation applied to the
// SOURCE address) through a STAGES-deep ring, and every per-lane source address is advanced
// INCREMENTALLY from chunk to chunk (one division per lane at kernel start, none in the loop).
// The register-staged version spent ~5x more VALU cycles on address generation per chunk than the
// MFMA pipe needed for the chunk's math (profiles/r01_notes.md, tools/diag_conv.py).
//   BRM: reduction chunk = BRM * 4 * EPC pixels (bf16: 32 / 64, f32: 16 / 32).
// ------------------------------------------------------------------------------------
//   WK x WNN waves: 2 x 2 (128 x 128 and smaller tiles, 2-3 workgroups per CU) or 2 x 4 (256 x 256 tile, one
//   8-wave workgroup per CU: half the L2->LDS bytes per FLOP for the layers whose reduction is short).
// GRAM: the "gradient" operand IS the activation tile (dy == x, BNW == BKW, one tile spans all channels): h^T h with a single
// DMA stream, plus the column sums of h from one extra MFMA per k-fragment against an all-ones fragment (slab layout
// [K*K | K]).  Used by the folded BatchNorm backward (csrc/bn.hip bn_fold_*).
// MT (multi-tap k-tile, the stem: KW = 1, IC = 32 packed elements per kernel row, all 7 kernel rows in ONE 256-row k-tile so
// that the gradient tensor is read once): every 16-byte chunk of a tile row has its OWN tap, so the tap offset is per lane
// instead of per workgroup.  Needs 16-byte aligned sources (stride 2 on an even-width packed image: every pixel index even).
// PSD (fp32 storage, three bf16 terms): bit 0 = the gradient operand dy is in the pre-split block format (common.h; written by
// simclr_bn_bwd_apply with SIMCLR_FMT_PS_OUT).  Its LDS tile then holds, per 128-byte block of 32 channels, two 32-byte runs of hi
// pieces and two of lo pieces -- each run is a [pixel][16 channel] bf16 tile row, so the fragments come from ds_read_b64_tr_b16
// exactly like the bf16 kernel's (2 + 2 transposing reads per fragment, no VALU), and lane group g holds pixels 8g..8g+7 of a
// 32-pixel step; the activation operand (fp32, split in registers) is read with the same pixel assignment.  The 16 positions of
// a run are channels {0-3, 16-19, 4-7, 20-23} / {8-11, 24-27, 12-15, 28-31} of the block: the slab store maps them back.
template <typename T, int BKW, int BNW, int BRM, int STAGES, int WK = 2, int WNN = 2, bool GRAM = false, bool MT = false, int SPL = 0, int PSD = 0>
__global__ __launch_bounds__(WK * WNN * 64,
                             WK * WNN == 8 ? 1 : ((SPL == 0 && STAGES * BRM * 4 * Elem<T>::EPC * (BKW + BNW) * (int)sizeof(T) <= 53 * 1024) ? 3 : 2))
void conv_wgrad_dma(const WgradP p) {
  // PSD == 2: dy arrives as plain fp32 (a gradient nobody pre-split: conv3's h^T dm under the folded tail BatchNorm) and its LDS tile is
  // rewritten IN PLACE into the pre-split block format once per chunk -- each value split once per workgroup instead of once per wave
  // that reads it -- after which the PSD == 1 k-loop applies unchanged (transposing reads, no VALU for that operand).
  static_assert(PSD == 0 || ((PSD == 1 || PSD == 2) && SPL == 3 && sizeof(T) == 4 && !GRAM && !MT && BNW % 32 == 0), "pre-split dy: fp32 storage, three terms");
  // LDS bank keys of the two tiles (XOR on the 32-byte blocks of a pixel row).  PSD: the dy tile takes the bf16 kernel's key (the 8
  // pixel rows {8g+q, q<4} of a half-wave transposing read get 8 distinct keys); the fp32 activation tile is read 16 lanes x 4 bytes
  // per lane group at pixels 8g+j, so the four groups need distinct 64-byte bank ranges: key = g << 1.
  auto key_a = [](int px) __attribute__((always_inline)) { return PSD ? (((px >> 3) & 3) << 1) : px_key<T>(px); };
  auto key_b = [](int px) __attribute__((always_inline)) { return PSD ? ((px & 3) | (((px >> 3) & 1) << 2)) : px_key<T>(px); };
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BR = BRM * 4 * EPC;           // pixels per reduction chunk
  constexpr int NW = WK * WNN;
  constexpr int KI = BKW / WK / 16;           // 16-row fragments per wave along k
  constexpr int NI = BNW / WNN / 16;
  constexpr int A_RB = BKW * sizeof(T);       // row bytes
  constexpr int B_RB = BNW * sizeof(T);
  constexpr int A_CPR = A_RB / 16;            // 16-byte chunks per pixel row
  constexpr int B_CPR = B_RB / 16;
  constexpr int A_BLK = A_RB / 32;            // 32-byte blocks per pixel row
  constexpr int B_BLK = B_RB / 32;
  constexpr int A_RPI = 64 / A_CPR;           // pixel rows covered by one wave-wide 1024-byte DMA instruction
  constexpr int B_RPI = 64 / B_CPR;
  constexpr int AJ = BR / A_RPI / NW;         // DMA instructions per wave per chunk
  constexpr int BJ = BR / B_RPI / NW;
  static_assert(AJ >= 1 && BJ >= 1 && A_CPR <= 64 && B_CPR <= 64, "tile / chunk configuration");
  constexpr int BUF = GRAM ? BR * A_RB : BR * (A_RB + B_RB);     // bytes per stage: [A tile | B tile] (GRAM: A only)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, fl = lane & 15;
  const int wk = wave / WNN, wn = wave % WNN;
  const int tiles = p.k_tiles * p.n_tiles;
  int tile, split;
  if (p.xcd_map) {
    const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
    tile = bidx % tiles;
    split = (bidx / tiles) * 8 + xcd;
  } else {
    tile = blockIdx.x % tiles;
    split = blockIdx.x / tiles;
  }
  if (split >= p.splits) return;
  const int ktile = tile % p.k_tiles;
  const int ntile = tile / p.k_tiles;
  const int kk0 = ktile * BKW;
  const int tap = kk0 / p.IC, ci0 = kk0 - tap * p.IC;
  const int ty = tap / p.KW, tx = tap - ty * p.KW;
  const int n0 = ntile * BNW;
  const T* __restrict__ X = (const T*)p.x;
  const T* __restrict__ DY = (const T*)p.dy;

  f32x4 acc[KI][NI];
#pragma unroll
  for (int i = 0; i < KI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 acs[KI];          // GRAM: column sums (every n row of the fragment holds the same value)
#pragma unroll
  for (int i = 0; i < KI; ++i) acs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int LPC = GRAM ? AJ : AJ + BJ;       // LDS-DMA instructions per wave per chunk

  const int nchunks = (p.M + BR - 1) / BR;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(nchunks, c_begin + p.chunks_per_split);
  const bool flat = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.IH == p.OH && p.IW == p.OW;
  const bool shift = !flat && p.stride == 1 && p.IH == p.OH && p.IW == p.OW;
  const int ohow = p.OH * p.OW;
  // decomposition of one chunk step (BR pixels) into whole images / rows / columns
  const int dq = BR / ohow, drem = BR - dq * ohow, drow = drem / p.OW, dcol = drem - drow * p.OW;

  // ---- per-lane source state (fixed LDS slot per lane and instruction; only the pixel advances) ----
  int a_m[AJ], a_v[AJ], a_oy[AJ], a_ox[AJ];
  int a_ty[AJ];              // MT: kernel row of this lane's chunk, -1 = beyond K (zero padding of the 256-row tile)
  const T* a_ptr[AJ];        // flat: current source; otherwise X + channel offset (pixel part recomputed)
  int b_m[BJ];
  const T* b_ptr[BJ];
  bool b_cok[BJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int px = (wave * AJ + j) * A_RPI + lane / A_CPR, pc = lane % A_CPR;
    const int lc = (((pc >> 1) ^ (key_a(px) & (A_BLK - 1))) << 1) | (pc & 1);   // logical chunk held by this slot
    const int m = c_begin * BR + px;
    a_m[j] = m;
    if (flat) {
      a_ptr[j] = X + (long long)m * p.pixpitch + ci0 + lc * EPC;
      a_v[j] = a_oy[j] = a_ox[j] = 0;
    } else {
      const int v = m / ohow, rem = m - v * ohow;
      a_v[j] = v; a_oy[j] = rem / p.OW; a_ox[j] = rem - a_oy[j] * p.OW;
      a_ptr[j] = X + ci0 + lc * EPC;
      if (MT) {
        const int kidx = kk0 + lc * EPC;
        a_ty[j] = kidx < p.K ? kidx / p.IC : -1;
        a_ptr[j] = X + (kidx - max(a_ty[j], 0) * p.IC);
      }
      if (shift) a_ptr[j] += ((long long)m + (ty - p.pad) * p.IW + (tx - p.pad)) * p.pixpitch;
    }
  }
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int px = (wave * BJ + j) * B_RPI + lane / B_CPR, pc = lane % B_CPR;
    const int lc = (((pc >> 1) ^ (key_b(px) & (B_BLK - 1))) << 1) | (pc & 1);
    const int m = c_begin * BR + px;
    b_m[j] = m;
    b_cok[j] = n0 + lc * EPC < p.N;
    b_ptr[j] = DY + (long long)m * p.N + n0 + lc * EPC;
  }
  const long long a_step = (long long)BR * p.pixpitch, b_step = (long long)BR * p.N;

  auto issue = [&](int stage) __attribute__((always_inline)) {
    if (DIAG(2)) return;
    unsigned char* a_dst = smem + stage * BUF + wave * (AJ * 1024);
    unsigned char* b_dst = smem + stage * BUF + BR * A_RB + wave * (BJ * 1024);
    if (flat) {
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const void* src = a_m[j] < p.M ? (const void*)a_ptr[j] : p.zero;
        a_ptr[j] += a_step;
        a_m[j] += BR;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(a_dst + j * 1024), 16, 0, 0);
      }
    } else if (shift) {
      // stride-1 "same" convolution: the tap's input pixel is output pixel m + constant, so the source
      // pointer advances linearly; only the border test needs the (row, column) of the pixel
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int iy = a_oy[j] - p.pad + ty, ix = a_ox[j] - p.pad + tx;
        const bool ok = a_m[j] < p.M && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        const void* src = ok ? (const void*)a_ptr[j] : p.zero;
        a_ptr[j] += a_step;
        a_ox[j] += dcol;
        if (a_ox[j] >= p.OW) { a_ox[j] -= p.OW; ++a_oy[j]; }
        a_oy[j] += drow;
        if (a_oy[j] >= p.OH) a_oy[j] -= p.OH;
        a_m[j] += BR;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(a_dst + j * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int iy = a_oy[j] * p.stride - p.pad + (MT ? a_ty[j] : ty), ix = a_ox[j] * p.stride - p.pad + tx;
        const bool ok = a_m[j] < p.M && (!MT || a_ty[j] >= 0) && (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
        const int pix = (a_v[j] * p.IH + iy) * p.IW + ix;
        const void* src = ok ? (const void*)(a_ptr[j] + (long long)pix * p.pixpitch) : p.zero;
        a_ox[j] += dcol;
        if (a_ox[j] >= p.OW) { a_ox[j] -= p.OW; ++a_oy[j]; }
        a_oy[j] += drow;
        if (a_oy[j] >= p.OH) { a_oy[j] -= p.OH; ++a_v[j]; }
        a_v[j] += dq;
        a_m[j] += BR;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(a_dst + j * 1024), 16, 0, 0);
      }
    }
    if (!GRAM) {
#pragma unroll
      for (int j = 0; j < BJ; ++j) {
        const void* src = (b_m[j] < p.M && b_cok[j]) ? (const void*)b_ptr[j] : p.zero;
        b_ptr[j] += b_step;
        b_m[j] += BR;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(b_dst + j * 1024), 16, 0, 0);
      }
    }
  };
  auto elem_off = [&](int RB, int NBLK, int px, int ch) -> int {
    const int byte = ch * (int)sizeof(T);
    return px * RB + (((byte >> 5) ^ (px_key<T>(px) & (NBLK - 1))) << 5) + (byte & 31);
  };
  auto a_off = [&](int px, int byte) -> int { return px * A_RB + (((byte >> 5) ^ (key_a(px) & (A_BLK - 1))) << 5) + (byte & 31); };
  auto b_off = [&](int px, int byte) -> int { return px * B_RB + (((byte >> 5) ^ (key_b(px) & (B_BLK - 1))) << 5) + (byte & 31); };
  auto compute = [&](int stage) __attribute__((always_inline)) {
    if (DIAG(1)) return;
    const unsigned char* As = smem + stage * BUF;
    const unsigned char* Bs = GRAM ? As : As + BR * A_RB;
    if (sizeof(T) == 2) {
#pragma unroll
      for (int ks = 0; ks < BR / 32; ++ks) {
        u32x4 af[KI], bf[NI];
        const int px0 = ks * 32 + g * 8 + (fl >> 2);
#pragma unroll
        for (int i = 0; i < KI; ++i) {
          const int ch = wk * (KI * 16) + i * 16 + (fl & 3) * 4;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(As + elem_off(A_RB, A_BLK, px0, ch)));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(As + elem_off(A_RB, A_BLK, px0 + 4, ch)));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          af[i] = (u32x4){l2[0], l2[1], h2[0], h2[1]};
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          const int ch = wn * (NI * 16) + i * 16 + (fl & 3) * 4;
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Bs + elem_off(B_RB, B_BLK, px0, ch)));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Bs + elem_off(B_RB, B_BLK, px0 + 4, ch)));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          bf[i] = (u32x4){l2[0], l2[1], h2[0], h2[1]};
        }
#pragma unroll
        for (int ki = 0; ki < KI; ++ki)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(bf16x8, bf[ni]), __builtin_bit_cast(bf16x8, af[ki]), acc[ki][ni], 0, 0, 0);
        if (GRAM && wn == 0) {
          const u32x4 ones = (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};   // eight bf16 1.0
#pragma unroll
          for (int ki = 0; ki < KI; ++ki)
            acs[ki] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ones),
                                                              __builtin_bit_cast(bf16x8, af[ki]), acs[ki], 0, 0, 0);
        }
      }
    } else if constexpr (PSD != 0) {
      __builtin_amdgcn_s_setprio(1);
      // pre-split dy: transposing reads of the hi / lo runs (bf16 pieces), activation split in registers; lane group g = pixels 8g..8g+7
#pragma unroll
      for (int ks = 0; ks < BR / 32; ++ks) {
        const int px0 = ks * 32 + g * 8 + (fl >> 2);
        mma_f32_chunks<NI, KI, true, 3, true, false>(&acc[0][0],
            [&](int i, int h) {
              const int nf = wn * NI + i;                              // 16-position run: block nf >> 1, run nf & 1; h = 0 hi, 1 lo
              const int byte = (nf >> 1) * 128 + h * 64 + (nf & 1) * 32 + (fl & 3) * 8;
              const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Bs + b_off(px0, byte)));
              const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Bs + b_off(px0 + 4, byte)));
              const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
              return (u32x4){l2[0], l2[1], h2[0], h2[1]};
            },
            [&](int i, int h) {
              u32x4 c;
#pragma unroll
              for (int e = 0; e < 4; ++e)
                c[e] = *(const uint32_t*)(As + a_off(ks * 32 + g * 8 + h * 4 + e, (wk * (KI * 16) + i * 16 + fl) * 4));
              return c;
            });
      }
      __builtin_amdgcn_s_setprio(0);
    } else if constexpr (SPL != 0) {
      // split-bf16 terms: lane group g holds pixels {4j + g : j = 0..7} of a 32-pixel step for both operands
      // GRAM (six terms only: the statistics the Gram matrix feeds need fp32-level products): both operands come from the one tile; the
      // column sums ride on the activation loader below as exact fp32 MFMAs against ones (one per 4-pixel group, as in the exact kernel)
      static_assert((!GRAM || SPL == 6) && BR % 32 == 0, "split-bf16 weight gradient: 32-pixel reduction steps; Gram variant: six terms");
#pragma unroll
      for (int ks = 0; ks < BR / 32; ++ks) {
        mma_f32_chunks<NI, KI, true, SPL>(&acc[0][0],
            [&](int i, int h) {
              u32x4 c;
#pragma unroll
              for (int e = 0; e < 4; ++e)
                c[e] = *(const uint32_t*)(Bs + elem_off(B_RB, B_BLK, ks * 32 + (h * 4 + e) * 4 + g, wn * (NI * 16) + i * 16 + fl));
              return c;
            },
            [&](int i, int h) {
              u32x4 c;
#pragma unroll
              for (int e = 0; e < 4; ++e)
                c[e] = *(const uint32_t*)(As + elem_off(A_RB, A_BLK, ks * 32 + (h * 4 + e) * 4 + g, wk * (KI * 16) + i * 16 + fl));
              if (GRAM && wn == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acs[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, __uint_as_float(c[e]), acs[i], 0, 0, 0);
              }
              return c;
            });
      }
    } else {
#pragma unroll 4
      for (int ks = 0; ks < BR / 4; ++ks) {
        float af[KI], bf[NI];
        const int px = ks * 4 + g;
#pragma unroll
        for (int i = 0; i < KI; ++i)
          af[i] = *(const float*)(As + elem_off(A_RB, A_BLK, px, wk * (KI * 16) + i * 16 + fl));
#pragma unroll
        for (int i = 0; i < NI; ++i)
          bf[i] = *(const float*)(Bs + elem_off(B_RB, B_BLK, px, wn * (NI * 16) + i * 16 + fl));
#pragma unroll
        for (int ki = 0; ki < KI; ++ki)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[ki][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[ni], af[ki], acc[ki][ni], 0, 0, 0);
        if (GRAM && wn == 0) {
#pragma unroll
          for (int ki = 0; ki < KI; ++ki) acs[ki] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, af[ki], acs[ki], 0, 0, 0);
        }
      }
    }
  };

  // ring: chunk c lives in stage (c - c_begin) % STAGES; STAGES-1 chunks are in flight ahead of the math
  int issued = c_begin;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (issued < c_end) { issue(s); ++issued; }
  int cs = 0, is = STAGES - 1;          // compute stage, next issue stage
  for (int c = c_begin; c < c_end; ++c) {
    // chunk c must have landed; newer chunks may stay in flight (each wave issued AJ+BJ loads per chunk)
    const int ahead = issued - c - 1;
    if (STAGES >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPC) : "memory");
    else if (STAGES >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (issued < c_end) { issue(is); ++issued; is = (is + 1 == STAGES) ? 0 : is + 1; }
    if constexpr (PSD == 2) {
      // (pixel row, 128-byte block, lane-group chunk gq): chunks gq and 4 + gq -- the eight channels one lane group multiplies -- become
      // their (hi, lo) bf16 pieces in the same two slots: the block is then what simclr_bn_bwd_apply(SIMCLR_FMT_PS_OUT) would have stored
      unsigned char* Bt = smem + cs * BUF + BR * A_RB;
      for (int q = tid; q < BR * (BNW / 32) * 4; q += NW * 64) {
        const int gq = q & 3, blk = (q >> 2) % (BNW / 32), px = q / (BNW / 8);
        u32x4* w0 = (u32x4*)(Bt + b_off(px, blk * 128 + gq * 16));
        u32x4* w1 = (u32x4*)(Bt + b_off(px, blk * 128 + (4 + gq) * 16));
        u32x4 hi, lo;
        split_terms2(*w0, *w1, hi, lo);
        *w0 = hi; *w1 = lo;
      }
      __syncthreads();
    }
    compute(cs);
    cs = (cs + 1 == STAGES) ? 0 : cs + 1;
  }
  // D[n = g*4+reg][k row = fl]  ->  slab[split][kk][n .. n+3]   (GRAM: every slab carries K extra floats = column sums)
  float* slab = p.dw + (long long)split * ((long long)p.K * p.N + (GRAM ? p.K : 0));
#pragma unroll
  for (int ki = 0; ki < KI; ++ki) {
    const int kk = kk0 + wk * (KI * 16) + ki * 16 + fl;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      int n = n0 + wn * (NI * 16) + ni * 16 + g * 4;
      if constexpr (PSD != 0) {            // run position 4 g + reg of run nf -> channel (common.h block layout)
        const int nf = wn * NI + ni, Q = 4 * (nf & 1) + g;
        n = n0 + (nf >> 1) * 32 + (Q & 1) * 16 + (Q >> 1) * 4;
      }
      if (kk < p.K && n < p.N)
        *(float4*)(slab + (long long)kk * p.N + n) =
            make_float4(acc[ki][ni][0], acc[ki][ni][1], acc[ki][ni][2], acc[ki][ni][3]);
    }
    if (GRAM && wn == 0 && g == 0 && kk < p.K) slab[(long long)p.K * p.N + kk] = acs[ki][0];
  }
}

// ------------------------------------------------------------------------------------
// Nine-tap weight gradient of the stride-1 3x3 "same" convolutions in bf16 (tf2/resnet.py:183-208 under tape.gradient).
// The per-tap kernels above re-read the activation and the gradient once per tap (9x through L2 and LDS); here a
// workgroup owns a (64 input channels x 64 output channels) tile of ALL nine taps: per 64-pixel chunk it loads the
// gradient rows once and ONE activation window with a halo of W+1 pixels on either side ([64 + 2W + 2] pixel rows), and
// forms tap (dy,dx) from the window shifted by dy*W + dx.  Wave w owns input channels [16w, 16w+16) x all 64 output
// channels x nine taps (144 accumulator registers): per 32-pixel k-step it reads 4 gradient fragments (shared by the nine
// taps) and 9 activation fragments for 36 MFMAs -- 0.36 KB of LDS reads per MFMA against 0.5 of the per-tap 64x64 wave
// tile.  Pixel pairs that the shift carries across an image border are removed by masking the ACTIVATION fragment
// (one fragment per tap: 4 ANDs) with per-lane 8-pixel edge masks.  The 32-byte channel blocks of an LDS row are
// XOR-permuted by key(row) = bit1 | bit3 << 1 of the row: the 8 rows {R..R+3, R+8..R+11} one half-wave transposing read
// touches then fall into 8 different bank groups for EVERY window shift R.
// ------------------------------------------------------------------------------------
struct Wgrad3P {
  const void* x;      // [V, H, W, pixpitch...] bf16
  const void* dy;     // [M, N] bf16
  float* dw;          // [splits][9*IC][N] fp32 slabs
  const void* zero;
  int V, H, W, IC, N, pixpitch, M;
  int splits, chunks_per_split, ci_tiles, co_tiles;
  int hpp;            // halo window rows, rounded up to 8
};

__device__ __forceinline__ int w3_key(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

// ST = LDS ring depth: 3 for the W <= 28 layers (two workgroups of 3 x <= 26 KB fit a CU), where the 2-deep ring left
// 42 % of the wave cycles waiting (PMC SQ_WAIT_ANY); hpp is a multiple of 32 rows, so every wave issues hpp / 32 window
// loads + 2 gradient loads per chunk and the counted vmcnt leaves exactly the newest chunk in flight.
template <int ST>
__global__ __launch_bounds__(256, 2) void conv_wgrad3x3_bf16(const Wgrad3P p) {
  constexpr int BR = 64, RB = 128;                 // pixels per chunk, bytes per LDS row (64 bf16 channels)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, fl = lane & 15;
  const int tiles = p.ci_tiles * p.co_tiles;
  const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
  const int tile = bidx % tiles, split = (bidx / tiles) * 8 + xcd;
  if (split >= p.splits) return;
  const int ci0 = (tile % p.ci_tiles) * 64, n0 = (tile / p.ci_tiles) * 64;
  const uint16_t* __restrict__ X = (const uint16_t*)p.x;
  const uint16_t* __restrict__ DY = (const uint16_t*)p.dy;
  const int W1 = p.W + 1;
  const int stage_bytes = (p.hpp + BR) * RB;

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = (p.M + BR - 1) / BR;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(nchunks, c_begin + p.chunks_per_split);

  // ---- direct-to-LDS loads of chunk c into stage s: window rows [mc - W1, mc - W1 + hpp), gradient rows [mc, mc+64)
  const int lrow = lane >> 3, lpc = lane & 7;
  auto issue = [&](int c, int s) __attribute__((always_inline)) {
    unsigned char* xs = smem + s * stage_bytes;
    unsigned char* ds = xs + p.hpp * RB;
    const int mc = c * BR;
    const int nxi = p.hpp >> 3;
    for (int q = wave; q < nxi; q += 4) {
      const int r = q * 8 + lrow;
      const int lc = (((lpc >> 1) ^ w3_key(r)) << 1) | (lpc & 1);        // logical chunk held by this LDS slot
      const int gp = mc + r - W1;
      const void* src = (gp >= 0 && gp < p.M) ? (const void*)(X + (long long)gp * p.pixpitch + ci0 + lc * 8) : p.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xs + q * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = wave * 2 + j;
      const int r = q * 8 + lrow;
      const int lc = (((lpc >> 1) ^ w3_key(r)) << 1) | (lpc & 1);
      const int m = mc + r;
      const void* src = (m < p.M) ? (const void*)(DY + (long long)m * p.N + n0 + lc * 8) : p.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ds + q * 1024), 16, 0, 0);
    }
  };

  // ---- per-lane constants of the fragment reads: byte offset of (row, channel byte) in a [rows][128 B] tile
  auto off_of = [&](int row, int byte) -> int { return row * RB + ((((byte >> 5) ^ w3_key(row)) & 3) << 5) + (byte & 31); };
  const int px_lane = g * 8 + (fl >> 2);                     // this lane's pixel row inside a 32-pixel k-step
  // window read offsets of k-step 0, rows r and r + 4 (k-step 1: + 32 rows = + 4096 bytes, key(row + 32) == key(row))
  int a_lo[9], a_hi[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int r = px_lane + W1 + (t / 3 - 1) * p.W + (t % 3 - 1);
    a_lo[t] = off_of(r, (wave * 16 + (fl & 3) * 4) * 2);
    a_hi[t] = off_of(r + 4, (wave * 16 + (fl & 3) * 4) * 2);
  }
  int b_lo[4], b_hi[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    b_lo[ni] = off_of(px_lane, (ni * 16 + (fl & 3) * 4) * 2);
    b_hi[ni] = off_of(px_lane + 4, (ni * 16 + (fl & 3) * 4) * 2);
  }

  // ---- (row, column) of the first pixel of this lane's two 8-pixel groups (k-steps 0 and 1), advanced per chunk
  const int hw = p.H * p.W;
  const int dq = BR / hw, drem = BR - dq * hw, drow = drem / p.W, dcol = drem - drow * p.W;
  int gy[2], gx[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const int m = c_begin * BR + ks * 32 + g * 8;
    const int rem = m % hw;
    gy[ks] = rem / p.W; gx[ks] = rem - gy[ks] * p.W;
  }

  auto compute = [&](int s) __attribute__((always_inline)) {
    const unsigned char* xs = smem + s * stage_bytes;
    const unsigned char* ds = xs + p.hpp * RB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // gradient fragments (shared by the nine taps)
      u32x4 bf[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(ds + ks * 32 * RB + b_lo[ni]));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(ds + ks * 32 * RB + b_hi[ni]));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        bf[ni] = (u32x4){l2[0], l2[1], h2[0], h2[1]};
      }
      // edge masks of this lane's 8 pixels (bit j set = the neighbour in that direction exists).  W >= 7, so the
      // 8 consecutive pixels cross at most one row end: at position jw = W - gx (>= 8: no crossing).
      const int jw = p.W - gx[ks];
      const unsigned lowm = jw >= 8 ? 0xffu : ((1u << jw) - 1u);          // pixels still in row gy
      const int y1 = (gy[ks] + 1 == p.H) ? 0 : gy[ks] + 1;                  // row after the crossing (next image: 0)
      const unsigned up = (gy[ks] > 0 ? lowm : 0u) | (y1 > 0 ? (0xffu & ~lowm) : 0u);
      const unsigned dn = (gy[ks] < p.H - 1 ? lowm : 0u) | (y1 < p.H - 1 ? (0xffu & ~lowm) : 0u);
      const unsigned lf = 0xffu & ~((gx[ks] == 0 ? 1u : 0u) | (jw < 8 ? (1u << jw) : 0u));          // x == 0 at j = 0 / jw
      const unsigned rt = 0xffu & ~((jw - 1 < 8 ? (1u << (jw - 1)) : 0u) |                            // x == W-1 at j = jw-1
                                    (jw - 1 + p.W < 8 ? (1u << (jw - 1 + p.W)) : 0u));                // and once more when W = 7
      // 8-bit lane masks -> dword masks (dword d holds pixels 2d, 2d+1), once per direction
      u32x4 wup, wdn, wlf, wrt;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        wup[d] = (((up >> (2 * d)) & 1u) ? 0x0000ffffu : 0u) | (((up >> (2 * d + 1)) & 1u) ? 0xffff0000u : 0u);
        wdn[d] = (((dn >> (2 * d)) & 1u) ? 0x0000ffffu : 0u) | (((dn >> (2 * d + 1)) & 1u) ? 0xffff0000u : 0u);
        wlf[d] = (((lf >> (2 * d)) & 1u) ? 0x0000ffffu : 0u) | (((lf >> (2 * d + 1)) & 1u) ? 0xffff0000u : 0u);
        wrt[d] = (((rt >> (2 * d)) & 1u) ? 0x0000ffffu : 0u) | (((rt >> (2 * d + 1)) & 1u) ? 0xffff0000u : 0u);
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ty = t / 3, tx = t % 3;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(xs + ks * 32 * RB + a_lo[t]));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(xs + ks * 32 * RB + a_hi[t]));
        const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        u32x4 af = (u32x4){l2[0], l2[1], h2[0], h2[1]};
        if (t != 4) {
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            unsigned w = 0xffffffffu;
            if (ty == 0) w &= wup[d]; else if (ty == 2) w &= wdn[d];
            if (tx == 0) w &= wlf[d]; else if (tx == 2) w &= wrt[d];
            af[d] &= w;
          }
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[t][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8, bf[ni]), __builtin_bit_cast(bf16x8, af), acc[t][ni], 0, 0, 0);
      }
      // advance this group's pixel coordinates by one chunk
      gx[ks] += dcol;
      if (gx[ks] >= p.W) { gx[ks] -= p.W; ++gy[ks]; }
      gy[ks] += drow;
      if (gy[ks] >= p.H) gy[ks] -= p.H;
    }
  };

  int issued = c_begin, cs = 0, is = ST - 1;
#pragma unroll
  for (int s = 0; s < ST - 1; ++s)
    if (issued < c_end) { issue(issued, s); ++issued; }
  const int nq = p.hpp >> 5;
  for (int c = c_begin; c < c_end; ++c) {
    // chunk c must have landed; with three stages chunk c + 1 (nq + 2 loads per wave) may stay in flight
    if (ST == 3 && issued - c - 1 >= 1 && nq == 3) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if (ST == 3 && issued - c - 1 >= 1 && nq == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (issued < c_end) { issue(issued, is); ++issued; is = (is + 1 == ST) ? 0 : is + 1; }
    compute(cs);
    cs = (cs + 1 == ST) ? 0 : cs + 1;
  }
  // D[n = g*4+reg][ci = fl]  ->  slab[split][tap*IC + ci][n .. n+3]
  float* slab = p.dw + (long long)split * 9 * p.IC * p.N;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int kk = t * p.IC + ci0 + wave * 16 + fl;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + ni * 16 + g * 4;
      *(float4*)(slab + (long long)kk * p.N + n) = make_float4(acc[t][ni][0], acc[t][ni][1], acc[t][ni][2], acc[t][ni][3]);
    }
  }
}

// ------------------------------------------------------------------------------------
// Nine-tap weight gradient of the stride-1 3x3 convolutions in fp32 storage, three bf16-piece terms, gradient pre-split (round 6).
// The structure of conv_wgrad3x3_bf16 -- a workgroup owns 64 input x 64 output channels x ALL nine taps over a pixel range, one
// activation window with a halo of W + 1 pixels either side per chunk, tap (dy, dx) = the window shifted by dy * W + dx -- with the
// operands of the parity mode: dy arrives in the pre-split block format (written by simclr_bn_bwd_apply, SIMCLR_FMT_PS_OUT) and the
// fp32 activation window is rewritten IN PLACE into that format once per chunk (chunks gq and 4 + gq of a 128-byte block -> their hi
// and lo bf16 pieces), so every fragment of either operand is two ds_read_b64_tr_b16 per piece and the k-loop holds no VALU work
// besides the edge masks.  Against the per-tap kernel (conv_wgrad_dma<float, ..., PSD = 1>): the gradient rows are loaded and read
// once per nine taps, the activation once per window instead of once per tap -- 8 LDS-DMA instructions per wave and 32 pixels as
// before, but for 108 MFMAs instead of 48 -- and nothing is split in registers.  32-pixel chunks (rows are 256 bytes: 64 fp32
// channels = two blocks), two stages, two workgroups per CU: W <= 47 (28^2, 14^2, 7^2); 56^2 keeps the per-tap kernel.
// Wave w owns the 16 positions of run w of the 64 input channels (block w >> 1, run w & 1: channels {0-3, 16-19, 4-7, 20-23} or
// {8-11, 24-27, 12-15, 28-31} of the block) x all 64 output channels x nine taps; the slab store maps positions back to channels.
// LDS bank key (XOR on the eight 32-byte blocks of a row): (row & 3) | bit 3 of the row << 2 -- the 8 rows {R..R+3, R+8..R+11} of a
// half-wave transposing read get 8 distinct keys for EVERY window shift R.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int w3f_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

__global__ __launch_bounds__(256, 2) void conv_wgrad3x3_f32ps(const Wgrad3P p) {
  constexpr int BR = 32, RB = 256;                 // pixels per chunk, bytes per LDS row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, fl = lane & 15;
  const int tiles = p.ci_tiles * p.co_tiles;
  const int xcd = blockIdx.x & 7, bidx = blockIdx.x >> 3;
  const int tile = bidx % tiles, split = (bidx / tiles) * 8 + xcd;
  if (split >= p.splits) return;
  const int ci0 = (tile % p.ci_tiles) * 64, n0 = (tile / p.ci_tiles) * 64;
  const float* __restrict__ X = (const float*)p.x;
  const float* __restrict__ DY = (const float*)p.dy;       // pre-split blocks: same bytes, same addresses as the fp32 tensor
  const int W1 = p.W + 1;
  const int stage_bytes = (p.hpp + BR) * RB;

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunks = (p.M + BR - 1) / BR;
  const int c_begin = split * p.chunks_per_split;
  const int c_end = min(nchunks, c_begin + p.chunks_per_split);

  // ---- direct-to-LDS loads of chunk c into stage s: window rows [mc - W1, mc - W1 + hpp), gradient rows [mc, mc + 32); one
  // instruction = 4 rows of 16 chunks; the XOR permutation of the 32-byte blocks is applied to the SOURCE chunk
  const int lrow = lane >> 4, lpc = lane & 15;
  auto issue = [&](int c, int s) __attribute__((always_inline)) {
    unsigned char* xs = smem + s * stage_bytes;
    unsigned char* ds = xs + p.hpp * RB;
    const int mc = c * BR;
    const int nxi = p.hpp >> 2;
    for (int q = wave; q < nxi; q += 4) {
      const int r = q * 4 + lrow;
      const int lc = (((lpc >> 1) ^ w3f_key(r)) << 1) | (lpc & 1);        // logical chunk held by this LDS slot
      const int gp = mc + r - W1;
      const void* src = (gp >= 0 && gp < p.M) ? (const void*)(X + (long long)gp * p.pixpitch + ci0 + lc * 4) : p.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xs + q * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q = wave * 2 + j;
      const int r = q * 4 + lrow;
      const int lc = (((lpc >> 1) ^ w3f_key(r)) << 1) | (lpc & 1);
      const int m = mc + r;
      const void* src = (m < p.M) ? (const void*)(DY + (long long)m * p.N + n0 + lc * 4) : p.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ds + q * 1024), 16, 0, 0);
    }
  };

  // byte offset of (row, byte in the 256-byte row); the lo pieces of a run sit 64 bytes after its hi pieces: block index + 2 = the
  // offset XOR 64 (the run's block index has bit 1 clear, every tile base is a multiple of 256)
  auto off_of = [&](int row, int byte) -> int { return row * RB + (((byte >> 5) ^ w3f_key(row)) << 5) + (byte & 31); };
  const int px_lane = g * 8 + (fl >> 2);                     // this lane's pixel row inside the 32-pixel chunk
  const int xbyte = (wave >> 1) * 128 + (wave & 1) * 32 + (fl & 3) * 8;
  // (register budget: 144 accumulators + 32 gradient fragment registers leave ~70 for everything else.  The offset of row + 4 is derived
  //  from the offset of the row: + 1024 bytes, and key bit 2 flips when bit 2 of the row is set -- that bit travels in bit 0 of the stored
  //  offset, which is a multiple of 8; the four edge masks stay 8-bit values and are widened per tap)
  auto hi_of = [](int o) __attribute__((always_inline)) -> int { return ((o & ~7) + 4 * RB) ^ ((o & 1) << 7); };
  int a_lo[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int r = px_lane + W1 + (t / 3 - 1) * p.W + (t % 3 - 1);
    a_lo[t] = off_of(r, xbyte) | ((r >> 2) & 1);
  }
  int b_lo[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int nbyte = (ni >> 1) * 128 + (ni & 1) * 32 + (fl & 3) * 8;
    b_lo[ni] = off_of(px_lane, nbyte) | ((px_lane >> 2) & 1);
  }
  auto frag = [&](const unsigned char* base, int o_lo, int o_hi) __attribute__((always_inline)) -> u32x4 {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base + o_lo));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(base + o_hi));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    return (u32x4){l2[0], l2[1], h2[0], h2[1]};
  };

  // ---- (row, column) of the first pixel of this lane's 8-pixel group, advanced per chunk
  const int hw = p.H * p.W;
  const int dq = BR / hw, drem = BR - dq * hw, drow = drem / p.W, dcol = drem - drow * p.W;
  int gy, gx;
  {
    const int m = c_begin * BR + g * 8;
    const int rem = m % hw;
    gy = rem / p.W; gx = rem - gy * p.W;
  }

  auto convert = [&](int s) __attribute__((always_inline)) {
    unsigned char* xs = smem + s * stage_bytes;
    for (int q = tid; q < p.hpp * 8; q += 256) {
      const int row = q >> 3, j = (q >> 2) & 1, gq = q & 3;
      u32x4* w0 = (u32x4*)(xs + off_of(row, (j * 8 + gq) * 16));
      u32x4* w1 = (u32x4*)(xs + off_of(row, (j * 8 + 4 + gq) * 16));
      u32x4 hi, lo;
      split_terms2(*w0, *w1, hi, lo);
      *w0 = hi; *w1 = lo;
    }
  };

  auto compute = [&](int s) __attribute__((always_inline)) {
    const unsigned char* xs = smem + s * stage_bytes;
    const unsigned char* ds = xs + p.hpp * RB;
    u32x4 dh[4], dl[4];                    // gradient fragments (shared by the nine taps)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int o0 = b_lo[ni] & ~7, o1 = hi_of(b_lo[ni]);
      dh[ni] = frag(ds, o0, o1);
      dl[ni] = frag(ds, o0 ^ 64, o1 ^ 64);
    }
    // edge masks of this lane's 8 pixels, as in conv_wgrad3x3_bf16 (W >= 7: the 8 pixels cross at most one row end)
    const int jw = p.W - gx;
    const unsigned lowm = jw >= 8 ? 0xffu : ((1u << jw) - 1u);
    const int y1 = (gy + 1 == p.H) ? 0 : gy + 1;
    const unsigned up = (gy > 0 ? lowm : 0u) | (y1 > 0 ? (0xffu & ~lowm) : 0u);
    const unsigned dn = (gy < p.H - 1 ? lowm : 0u) | (y1 < p.H - 1 ? (0xffu & ~lowm) : 0u);
    const unsigned lf = 0xffu & ~((gx == 0 ? 1u : 0u) | (jw < 8 ? (1u << jw) : 0u));
    const unsigned rt = 0xffu & ~((jw - 1 < 8 ? (1u << (jw - 1)) : 0u) | (jw - 1 + p.W < 8 ? (1u << (jw - 1 + p.W)) : 0u));
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int ty = t / 3, tx = t % 3;
      const int o0 = a_lo[t] & ~7, o1 = hi_of(a_lo[t]);
      u32x4 xh = frag(xs, o0, o1);
      u32x4 xl = frag(xs, o0 ^ 64, o1 ^ 64);
      if (t != 4) {
        const unsigned m8 = (ty == 0 ? up : ty == 2 ? dn : 0xffu) & (tx == 0 ? lf : tx == 2 ? rt : 0xffu);
#pragma unroll
        for (int d = 0; d < 4; ++d) {      // dword d holds pixels 2d, 2d + 1: two sign-extended mask bits -> 0x0000ffff | 0xffff0000
          const unsigned w = ((unsigned)__builtin_amdgcn_sbfe((int)m8, 2 * d, 1) & 0x0000ffffu) |
                             ((unsigned)__builtin_amdgcn_sbfe((int)m8, 2 * d + 1, 1) & 0xffff0000u);
          xh[d] &= w; xl[d] &= w;
        }
      }
      // three terms, small ones first, term-major (consecutive MFMAs write different accumulators) -- the order of mma_f32_chunks
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[t][ni] = mma_bf16(dl[ni], xh, acc[t][ni]);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[t][ni] = mma_bf16(dh[ni], xl, acc[t][ni]);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[t][ni] = mma_bf16(dh[ni], xh, acc[t][ni]);
      __builtin_amdgcn_sched_barrier(0);          // one tap's fragments at a time (left alone the scheduler prefetches several taps and spills)
    }
    __builtin_amdgcn_s_setprio(0);
    gx += dcol;
    if (gx >= p.W) { gx -= p.W; ++gy; }
    gy += drow;
    if (gy >= p.H) gy -= p.H;
  };

  int issued = c_begin, cs = 0, is = 1;
  if (issued < c_end) { issue(issued, 0); ++issued; }
  for (int c = c_begin; c < c_end; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // chunk c has landed; every wave is done with the other stage
    if (issued < c_end) { issue(issued, is); ++issued; is ^= 1; }
    convert(cs);
    __syncthreads();
    compute(cs);
    cs ^= 1;
  }
  // D[n position 4 g + reg of run ni][ci position fl of run `wave`]  ->  slab[split][tap * IC + ci][n .. n + 3]
  float* slab = p.dw + (long long)split * 9 * p.IC * p.N;
  const int Qx = 4 * (wave & 1) + (fl >> 2);
  const int ci = ci0 + (wave >> 1) * 32 + (Qx & 1) * 16 + (Qx >> 1) * 4 + (fl & 3);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int kk = t * p.IC + ci;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int Q = 4 * (ni & 1) + g;
      const int n = n0 + (ni >> 1) * 32 + (Q & 1) * 16 + (Q >> 1) * 4;
      *(float4*)(slab + (long long)kk * p.N + n) = make_float4(acc[t][ni][0], acc[t][ni][1], acc[t][ni][2], acc[t][ni][3]);
    }
  }
}

// C[m][n] = sum_k A[m][k] * B[n][k]   (fp32, row-major, K contiguous in both; m, n multiples of 16, k of 32).
// Small helper GEMM of the folded BatchNorm backward ((W*b) W^T and (h^T h) W: at most 512 x 2048 x 2048), exact f32
// MFMA.  A workgroup owns a TT x TT tile (64: waves 2 x 2 of 32 x 32; 32: waves 2 x 2 of 16 x 16 -- picked so that the
// launch has >= 1024 waves, one per SIMD, whenever the problem has that many fragments); the operands go global ->
// registers -> LDS in 32-wide k-chunks with full 128-byte rows per request (the former one-wave-per-strip version read
// 64-byte row pieces straight from L2 and spent 150 us on the 512 x 512 x 2048 product against an MFMA floor of 27).
template <int TT>
__global__ __launch_bounds__(256) void small_gemm_nt_f32(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, int M, int N, int K) {
  constexpr int LDR = 36;                          // padded row (floats): 16-byte aligned, rows 4 banks apart
  constexpr int FR = TT / 32;                      // 16 x 16 fragments per wave along each dimension
  constexpr int LPT = TT * 8 / 256;                // float4 loads per thread per operand per chunk
  __shared__ __attribute__((aligned(16))) float As[TT * LDR], Bs[TT * LDR];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fl = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * TT, n0 = blockIdx.x * TT;
  f32x4 acc[FR][FR];
#pragma unroll
  for (int i = 0; i < FR; ++i)
#pragma unroll
    for (int j = 0; j < FR; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 ra[LPT], rb[LPT];
  auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      const int idx = tid + l * 256, r = idx >> 3, c = (idx & 7) * 4;
      ra[l] = (m0 + r < M) ? *(const float4*)(A + (long long)(m0 + r) * K + k0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[l] = (n0 + r < N) ? *(const float4*)(B + (long long)(n0 + r) * K + k0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += 32) {
    __syncthreads();                               // the previous chunk's fragment reads are done
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      const int idx = tid + l * 256, r = idx >> 3, c = (idx & 7) * 4;
      *(float4*)(As + r * LDR + c) = ra[l];
      *(float4*)(Bs + r * LDR + c) = rb[l];
    }
    __syncthreads();
    if (k0 + 32 < K) gload(k0 + 32);               // in flight during the MFMAs of this chunk
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float4 af[FR], bf[FR];
#pragma unroll
      for (int i = 0; i < FR; ++i) {
        af[i] = *(const float4*)(As + (wm * (FR * 16) + i * 16 + fl) * LDR + h * 16 + 4 * q);
        bf[i] = *(const float4*)(Bs + (wn * (FR * 16) + i * 16 + fl) * LDR + h * 16 + 4 * q);
      }
      // D[n-row = g*4+reg][m-col = fl]: B fragment as the first operand, like the conv kernels; the four k values of a
      // lane's float4 go to four MFMAs (any k order: both operands use the same one)
#pragma unroll
      for (int i = 0; i < FR; ++i)
#pragma unroll
        for (int j = 0; j < FR; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].x, af[i].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].y, af[i].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].z, af[i].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j].w, af[i].w, acc[i][j], 0, 0, 0);
        }
    }
  }
#pragma unroll
  for (int i = 0; i < FR; ++i) {
    const int m = m0 + wm * (FR * 16) + i * 16 + fl;
#pragma unroll
    for (int j = 0; j < FR; ++j) {
      const int n = n0 + wn * (FR * 16) + j * 16 + q * 4;
      if (m < M && n < N) *(float4*)(C + (long long)m * N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  }
}

// dw[i] = sum_s slabs[s][i]  (+ dw[i] if accumulate).  A workgroup owns 16 float4 columns; its 256 threads are 16 columns
// x 16 slab lanes: lane l adds slabs l, l+16, l+32, ... (four independent loads in flight), the 16 partial sums meet in
// LDS and are added in lane order.  The summation tree depends only on `splits`, never on the launch geometry, so the
// gradient is bit-identical from run to run; spreading the slab loop over 16 lanes matters because the layers with few
// weights (64 x 64 ... 64 x 256) have up to 256 slabs and only a few thousand columns -- a one-thread-per-column loop
// is a chain of 32 dependent memory round trips (22 us per launch, 73 launches per step).
__global__ __launch_bounds__(256) void slab_reduce(const float* __restrict__ slabs, int splits, long long numel,
                                                   float* __restrict__ out, int accumulate) {
  __shared__ float4 red[16][16];
  const long long n4 = numel / 4;
  const int tx = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long long i = blockIdx.x * 16ll + tx;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float* src = slabs + i * 4;
    int s = sl;
    for (; s + 48 < splits; s += 64) {
      float4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = *(const float4*)(src + (long long)(s + 16 * j) * numel);
#pragma unroll
      for (int j = 0; j < 4; ++j) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
    }
    for (; s < splits; s += 16) {
      const float4 v = *(const float4*)(src + (long long)s * numel);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[sl][tx] = a;
  __syncthreads();
  if (sl == 0 && i < n4) {
#pragma unroll
    for (int l = 1; l < 16; ++l) { const float4 v = red[l][tx]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (accumulate) {
      const float4 v = *(const float4*)(out + i * 4);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *(float4*)(out + i * 4) = a;
  }
}

// pivot[n] = the convolution output at output pixel (image 0, OH / 2, OW / 2) -- an interior pixel, so that every tap of a
// padded 3x3 convolution contributes -- for the fp32 forward's pivoted statistics (ConvP::pivot).  One workgroup per output
// channel, its 256 threads stride the K = KH * KW * Cin reduction (w_t rows are K-contiguous: coalesced; 18 dependent iterations
// for the 3x3 x 512 layers instead of the 72 of a single wave), fixed-order combine.
__global__ __launch_bounds__(256) void conv_pivot_row(const float* __restrict__ x, const float* __restrict__ w_t, float* __restrict__ pivot,
                                                      int IH, int IW, int Cin, int pixpitch, int OH, int OW, int Cout, int KH, int KW,
                                                      int stride, int pad, int w_ps) {
  // w_ps: w_t is the pre-split forward copy (fp16 pieces of 2^8 * w per 128-byte k-block, see presplit_rows): element k of a row =
  // (hi + lo) * 2^-8 -- the pivot only has to be NEAR the channel mean, any finite value gives the same statistics
  __shared__ float red[4];
  const int n = blockIdx.x, lane = threadIdx.x & 63;
  const int oy = OH / 2, ox = OW / 2, K = KH * KW * Cin;
  float a = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const int tap = k / Cin, ci = k - tap * Cin;
    const int iy = oy * stride - pad + tap / KW, ix = ox * stride - pad + tap % KW;
    if ((unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW) {
      float wv;
      if (w_ps) {
        const int within = k & 31, c16 = within >> 2, e = within & 3;
        const _Float16* blk = (const _Float16*)(w_t + (long long)n * K + (k & ~31)) + (c16 & 3) * 8 + (c16 >> 2) * 4 + e;
        wv = ((float)blk[0] + (float)blk[32]) * (1.0f / (float)(1 << F16_WSCALE_LOG2));
      } else {
        wv = w_t[(long long)n * K + k];
      }
      a = fmaf(x[((long long)iy * IW + ix) * pixpitch + ci], wv, a);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) pivot[n] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Pre-split copy of an fp32 matrix [rows][K] (K a multiple of 32) for the three-term split-bf16 product (PSB / PSA
// instantiations): every 128-byte k-block (32 floats = eight 16-byte chunks c0..c7) becomes eight 16-byte chunks of bf16:
// chunk g (g < 4) = hi of the eight values lane group g reads in a k-step (c_g, then c_{4+g}); chunk 4 + g = their lo
// = bf16(x - hi).  Same bytes, same addresses per block, so the LDS-DMA stream and the swizzle of the consumer do not change;
// its two fragment reads (chunks g and 4 + g) return the operands split_terms2 would have produced -- bit for bit.
// F16: fp16 planes of scale * x (scale a power of two: the split-fp16 forward's weights, F16_WSCALE_LOG2).
template <bool F16>
__global__ __launch_bounds__(256) void presplit_rows(const float* __restrict__ src, uint32_t* __restrict__ dst, long long nblocks, float scale) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;       // one output chunk per thread
  if (i >= nblocks * 8) return;
  const long long blk = i >> 3;
  const int c = (int)(i & 7), gq = c & 3;
  u32x4 c0 = *(const u32x4*)(src + blk * 32 + gq * 4), c1 = *(const u32x4*)(src + blk * 32 + 16 + gq * 4);
  u32x4 hi, lo;
  if constexpr (F16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { c0[e] = __float_as_uint(__uint_as_float(c0[e]) * scale); c1[e] = __float_as_uint(__uint_as_float(c1[e]) * scale); }
    split_terms2_f16(c0, c1, hi, lo);
  } else {
    split_terms2(c0, c1, hi, lo);
  }
  *(u32x4*)(dst + i * 4) = (c < 4) ? hi : lo;
}

// ------------------------------------------------------------------------------------
// Stem convolution (Cin=3): tf2/resnet.py:593-599 (7x7 s2) and :551-556 (CIFAR 3x3 s1).
// The input is pre-packed (simclr_pack_views) as [V][HP][WP][4] with physical zero
// borders and a zero 4th channel, so one kernel row of KWP taps x 4 channels is one
// contiguous, bounds-check-free run and every MFMA A fragment is a single 16-byte
// global load (no LDS for the activation); the padded weights [N][KHP][KWP][4] sit
// in LDS for the whole (persistent) workgroup.
// ------------------------------------------------------------------------------------
struct StemP {
  const void* xp;  // [V][HP][WP][4]
  const void* w;   // [N][KP] padded weights, KP = KHP*KWP*4
  void* y;         // [M][N]
  float* stats;
  int V, HP, WP, OH, OW, N, KHP, KWP, stride, M, KP, nslot, m_tiles;
  int wpad;          // bytes added to a weight row in LDS so that the row pitch is 32 (mod 256): stem_lds_pad
  int diag;          // diagnostic build only (SIMCLR_DIAG): 1 = no MFMA, 2 = no activation loads in the tile loop, 4 = no stores
};

// LDS pitch of the stem's weight rows.  A ds_read_b128 is served in four lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...
// (MI355X_MICROARCH.md, LDS): eight rows fl of k-chunk g together with the OTHER eight rows of chunk g + 1 (16 bytes further).  With a
// pitch of 32 (mod 256) the first eight land on the even 16-byte slots of the 256-byte bank row and the second eight on the odd ones --
// conflict-free; the old pitch (row bytes + 16 = 144 mod 256 for the 7 x 7 stem in fp32) spread the 16 rows of ONE chunk, which is not
// what the hardware groups: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.75 (r06_call29), the LDS pipe busier than the MFMA pipe.
static inline int stem_lds_pad(int row_bytes) { return ((32 - row_bytes) % 256 + 256) % 256; }

// KS > 0 (the 7x7 stem in bf16: 7 k-steps, one padded kernel row of 8 taps x 4 channels each): the k-loop is unrolled and
// all 14 activation fragments of a tile are requested before the first MFMA -- one L1/L2 round trip per tile instead of
// one per kernel row (the runtime loop exposed the load latency seven times per tile).
// SPL (fp32 storage, simclr_set_f32_matmul): 3 / 6 split-bf16 terms per product instead of the exact fp32 MFMA -- two
// consecutive 16-element k-steps of a lane become the eight reduction elements of one v_mfma_f32_16x16x32_bf16 (the same pairing
// for the activation and the weight operand, so every product is formed exactly once; an odd trailing k-step is paired with zeros).
// SPL > 0 with KS > 0 (fp32 storage, KS = the padded K in 16-element k-steps, even: 14 for the 7x7 stem): the k-loop is unrolled, the
// weights sit PRE-SPLIT in LDS (hi pieces in the even k-step's chunk, lo pieces in the odd one's: split once per workgroup instead of
// once per tile) and the activation fragments ROLL: as soon as the two k-steps of a pair have been split into their pieces, the same
// registers are reloaded with the NEXT tile's fragments, so the L2 round trip of a tile's 28 loads hides behind the previous tile's
// 168 MFMAs (the runtime loop waited for it seven times per tile: 2.5 ms for 0.45 ms of matrix work, profiles/r06_notes.md).
template <typename T, bool STATS, int KS = 0, int SPL = 0>
__global__ __launch_bounds__(256, (SPL > 0 && KS > 0) ? 2 : 1) void stem_conv_fwd(const StemP p) {
  static_assert(SPL == 0 || sizeof(T) == 4, "split terms: fp32 storage");
  static_assert(SPL == 0 || KS % 2 == 0, "split terms, unrolled: whole k-step pairs");
  constexpr bool ROLL = SPL > 0 && KS > 0;
  constexpr bool PSW = ROLL && (SPL == 3 || SPL == 13);     // weights pre-split in LDS (two planes: the three-term products only)
  constexpr int EPC = Elem<T>::EPC;
  constexpr int KSTEP = 4 * EPC;     // elements per MFMA k-step
  constexpr int BN = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPR (LDS bases, M0)
  const int g = lane >> 4, fl = lane & 15;
  const int n0 = blockIdx.y * BN;
  const int pitch = p.KP * (int)sizeof(T) + p.wpad;  // bytes per weight row in LDS (32 mod 256: stem_lds_pad)
  const T* __restrict__ Wt = (const T*)p.w;
  const int cpr = p.KP / EPC;  // chunks per weight row
  if constexpr (PSW) {
    // (row, k-step pair, lane group): the eight values a lane multiplies in one MFMA = chunks (2 kp) * 4 + g and (2 kp + 1) * 4 + g
    for (int id = tid; id < BN * (KS / 2) * 4; id += 256) {
      const int r = id / (KS * 2), q = id % (KS * 2), kp = q >> 2, gq = q & 3;
      u32x4 c0 = zero16(), c1 = zero16(), hi, lo;
      if (n0 + r < p.N) {
        c0 = ld16(Wt + (long long)(n0 + r) * p.KP + ((2 * kp) * 4 + gq) * EPC);
        c1 = ld16(Wt + (long long)(n0 + r) * p.KP + ((2 * kp + 1) * 4 + gq) * EPC);
      }
      if constexpr (SPL == 13) split_terms2_f16(c0, c1, hi, lo); else split_terms2(c0, c1, hi, lo);
      *(u32x4*)(smem + r * pitch + ((2 * kp) * 4 + gq) * 16) = hi;
      *(u32x4*)(smem + r * pitch + ((2 * kp + 1) * 4 + gq) * 16) = lo;
    }
  } else
  for (int id = tid; id < BN * cpr; id += 256) {
    const int r = id / cpr, c = id % cpr;
    u32x4 v = zero16();
    if (n0 + r < p.N) v = ld16(Wt + (long long)(n0 + r) * p.KP + c * EPC);
    *(u32x4*)(smem + r * pitch + c * 16) = v;
  }
  __syncthreads();
  float* red = (float*)(smem + BN * pitch);  // [4 waves][64][2]
  const T* __restrict__ X = (const T*)p.xp;
  T* __restrict__ Y = (T*)p.y;
  const int row_elems = p.KWP * 4;             // elements per kernel row
  const int ksteps = p.KP / KSTEP;
  // The workgroup keeps its channel sums in registers over all its tiles (ascending order) and stores them once into its own
  // slot (nslot >= gridDim.x: plain stores, deterministic statistics; fewer slots: one set of atomics per workgroup).
  float st_s[4][4], st_q[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) { st_s[i][r] = 0.f; st_q[i][r] = 0.f; }
  float* cst = red + 4 * BN * 2;               // bf16: [128][64] C staging tile (16 KB) behind the reduction scratch
  // (ROLL: 32-bit BYTE offsets from the image base -- the launcher checks that the packed image is smaller than 4 GB; four registers less,
  //  and the loads can take the scalar base + 32-bit lane offset form)
  typedef typename std::conditional<ROLL, unsigned, long long>::type off_t_;
  auto tile_rows = [&](int mt_, off_t_* base_, bool* ok_) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mt_ * 128 + wave * 32 + i * 16 + fl;
      ok_[i] = m < p.M;
      const int mm = ok_[i] ? m : 0;
      const int v = mm / (p.OH * p.OW);
      const int rem = mm - v * (p.OH * p.OW);
      const int oy = rem / p.OW, ox = rem - oy * p.OW;
      base_[i] = (off_t_)((((long long)v * p.HP + oy * p.stride) * p.WP + ox * p.stride) * 4 * (ROLL ? (int)sizeof(T) : 1));
    }
  };
  // element offset of k-step ks of this lane's group inside the packed image (kernel row, position in the row)
  // ROLL (the launcher guarantees one padded kernel row = 32 elements = two k-steps): k-step ks is half ks & 1 of kernel row ks >> 1, so
  // the offset is a wave-uniform multiple of the packed row pitch plus ONE per-lane term -- a table of KS per-lane offsets cost 13
  // registers, which this kernel (256 VGPRs at two workgroups per CU) spilled; the spill reloads at the top of every tile made the
  // compiler wait for vmcnt(0) there, i.e. for the previous tile's STORES (r06: ISA of stem_conv_fwd<float, true, 14, 13>)
  const unsigned wp4 = p.WP * 4 * (unsigned)sizeof(T), goff = g * 16;                              // bytes
  auto k_off = [&](int ks) -> unsigned { return (ks >> 1) * wp4 + (ks & 1) * (KSTEP * (unsigned)sizeof(T)) + goff; };
  u32x4 roll[ROLL ? KS : 1][2];      // ROLL: the activation fragments of the tile about to be computed
  if constexpr (ROLL) {
    if ((int)blockIdx.x < p.m_tiles) {
      off_t_ b0[2];
      bool k0[2];
      tile_rows(blockIdx.x, b0, k0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) roll[ks][i] = ld16((const unsigned char*)X + (b0[i] + k_off(ks)));      // (rows beyond M read pixel 0: zeroed below)
    }
  }
  for (int mt = blockIdx.x; mt < p.m_tiles; mt += gridDim.x) {
    const int mw = mt * 128 + wave * 32;
    off_t_ base[2];
    bool ok[2];
    tile_rows(mt, base, ok);
    f32x4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (ROLL) {
      const int mtn = mt + gridDim.x;
      off_t_ nbase[2];
      bool nok[2];
      tile_rows(mtn < p.m_tiles ? mtn : mt, nbase, nok);
      if (mt * 128 + 128 > p.M) {                            // the partial last tile: rows beyond M multiply zeros
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            if (!ok[i]) roll[ks][i] = zero16();
      }
#pragma unroll
      for (int kp = 0; kp < KS; kp += 2) {
        if (!DIAG(1))
        mma_f32_chunks<4, 2, false, SPL, PSW>(&acc[0][0],
            [&](int ni, int h) { return *(const u32x4*)(smem + (ni * 16 + fl) * pitch + ((kp + h) * 4 + g) * 16); },
            [&](int mi, int h) { return roll[kp + h][mi]; });
        // (unconditional: a workgroup's last tile re-reads its own fragments -- a branch here would make the compiler count the loads
        // of this tile as possibly absent and wait for ALL outstanding ones at the later pairs)
        // The scheduling fences keep the reloads where they are written: left alone, the scheduler gathers all 28 at the end of the tile.
        __builtin_amdgcn_sched_barrier(0);
        if (!DIAG(2)) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < 2; ++i) roll[kp + h][i] = ld16((const unsigned char*)X + (nbase[i] + k_off(kp + h)));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (KS > 0) {
      u32x4 afk[KS > 0 ? KS : 1][2];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          afk[ks][i] = ok[i] ? ld16(X + base[i] + (long long)ks * p.WP * 4 + g * EPC) : zero16();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const u32x4 bf = *(const u32x4*)(smem + (ni * 16 + fl) * pitch + (ks * 4 + g) * 16);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = MMA<T>::run(bf, afk[ks][mi], acc[ni][mi]);
        }
    } else if constexpr (SPL > 0) {
      for (int kp = 0; kp < ksteps; kp += 2) {
        mma_f32_chunks<4, 2, false, SPL>(&acc[0][0],
            [&](int ni, int h) {
              const int ks = kp + h;
              return ks < ksteps ? *(const u32x4*)(smem + (ni * 16 + fl) * pitch + (ks * 4 + g) * 16) : zero16();
            },
            [&](int mi, int h) {
              const int ks = kp + h;
              const int e = ks * KSTEP + g * EPC;
              const int kh = e / row_elems, within = e - kh * row_elems;
              return (ok[mi] && ks < ksteps) ? ld16(X + base[mi] + (long long)kh * p.WP * 4 + within) : zero16();
            });
      }
    } else
    for (int ks = 0; ks < ksteps; ++ks) {
      const int e = ks * KSTEP + g * EPC;          // element offset along padded K
      const int kh = e / row_elems, within = e - kh * row_elems;
      u32x4 af[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = ok[i] ? ld16(X + base[i] + (long long)kh * p.WP * 4 + within) : zero16();
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const u32x4 bf = *(const u32x4*)(smem + (ni * 16 + fl) * pitch + (ks * 4 + g) * 16);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = MMA<T>::run(bf, af[mi], acc[ni][mi]);
      }
    }
    if (STATS) {       // per-lane partial sums over all tiles of this workgroup (tile order = ascending: deterministic)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) { const float v = acc[ni][mi][r]; st_s[ni][r] += v; st_q[ni][r] += v * v; }
    }
    if (sizeof(T) == 2) {
      // bf16: stage the 128 x 64 tile in LDS (8-byte granules, XOR-swizzled by row) and store whole 128-byte rows,
      // 16 bytes per lane -- the direct register stores were 8 bytes per lane in 32-byte row pieces
      unsigned char* Cs = (unsigned char*)cst;
      __syncthreads();                                  // the previous tile's row pass is done with the staging buffer
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int ml = wave * 32 + mi * 16 + fl;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int q = ni * 4 + g;                     // 8-byte granule (4 channels) in the row
          u32x2 pk;
          pk[0] = pack_bf16x2(acc[ni][mi][0], acc[ni][mi][1]);
          pk[1] = pack_bf16x2(acc[ni][mi][2], acc[ni][mi][3]);
          *(u32x2*)(Cs + ml * (BN * 2) + ((q ^ ((ml & 7) << 1)) << 3)) = pk;
        }
      }
      __syncthreads();
      const int cc = tid & 7;                           // 16-byte chunk (8 channels) of the row
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 3) + i * 32;
        const int m = mt * 128 + r;
        if (m < p.M && n0 + cc * 8 < p.N)
          *(u32x4*)((uint16_t*)Y + (long long)m * p.N + n0 + cc * 8) =
              *(const u32x4*)(Cs + r * (BN * 2) + (((cc * 2) ^ ((r & 7) << 1)) << 3));
      }
    } else {
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + ni * 16 + g * 4;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int m = mw + mi * 16 + fl;
          if (m < p.M && n < p.N && !DIAG(4))
            *(float4*)(Y + (long long)m * p.N + n) = make_float4(acc[ni][mi][0], acc[ni][mi][1], acc[ni][mi][2], acc[ni][mi][3]);
        }
      }
    }
  }
  if (STATS) {
    // one flush per workgroup: lanes -> 16-lane groups -> waves (LDS) -> this workgroup's slot (or atomics into slot b % nslot)
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s_ = st_s[ni][r], ss = st_q[ni][r];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s_ += __shfl_xor(s_, o, 64); ss += __shfl_xor(ss, o, 64); }
        if (fl == 0) {
          const int nl = ni * 16 + g * 4 + r;
          red[(wave * BN + nl) * 2] = s_;
          red[(wave * BN + nl) * 2 + 1] = ss;
        }
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float s_ = 0.f, ss = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s_ += red[(w * BN + tid) * 2]; ss += red[(w * BN + tid) * 2 + 1]; }
      const bool own_slot = p.nslot >= (int)gridDim.x;
      float* st = p.stats + (long long)(own_slot ? blockIdx.x : blockIdx.x % p.nslot) * 2 * p.N;
      if (own_slot) { st[n0 + tid] = s_; st[p.N + n0 + tid] = ss; }
      else { atomicAdd(st + n0 + tid, s_); atomicAdd(st + p.N + n0 + tid, ss); }
    }
  }
}

// ---- weight re-layout (fp32 HWIO master -> compute copies) ---------------------------
// mode 0: fwd   dst[co][(kh*KW+kw)*CIP+ci]     (CIP >= CI, COP >= CO: zero-padded channel dims)
// mode 1: dgrad dst[ci][(kh*KW+kw)*COP+co]
// mode 2: stem  dst[co][(kh*KWP+kw)*4+ci]   zero padded to KHP x KWP x 4 (COP rows)
template <typename T>
__global__ void prep_weights(const float* __restrict__ w, T* __restrict__ dst, int KH, int KW, int CI,
                             int CO, int mode, int KHP, int KWP, int CIP, int COP) {
  long long total;
  if (mode == 2) total = (long long)COP * KHP * KWP * 4;
  else total = (long long)KH * KW * CIP * COP;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (mode == 0) {
      const int co = (int)(i / ((long long)KH * KW * CIP));
      const int k = (int)(i % ((long long)KH * KW * CIP));
      const int tap = k / CIP, ci = k % CIP;
      if (ci < CI && co < CO) v = w[((long long)tap * CI + ci) * CO + co];
    } else if (mode == 1) {
      const int ci = (int)(i / ((long long)KH * KW * COP));
      const int k = (int)(i % ((long long)KH * KW * COP));
      const int tap = k / COP, co = k % COP;
      if (ci < CI && co < CO) v = w[((long long)tap * CI + ci) * CO + co];
    } else {
      const int co = (int)(i / (KHP * KWP * 4));
      const int k = (int)(i % (KHP * KWP * 4));
      const int kh = k / (KWP * 4), kw = (k / 4) % KWP, ci = k % 4;
      if (kh < KH && kw < KW && ci < CI && co < CO) v = w[(((long long)kh * KW + kw) * CI + ci) * CO + co];
    }
    Elem<T>::st(dst + i, v);
  }
}

// both compute copies of one conv weight in one launch: dst_t[co][(tap)*CIP+ci] (fwd) and dst_d[ci][(tap)*COP+co]
// (dgrad); threads walk the padded (tap, ci, co) index space, co fastest (coalesced source reads and dst_d writes)
template <typename T>
__global__ void prep_weights_pair(const float* __restrict__ w, T* __restrict__ dst_t, T* __restrict__ dst_d,
                                  int taps, int CI, int CO, int CIP, int COP) {
  const long long total = (long long)taps * CIP * COP;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int co = (int)(i % COP);
    const long long r = i / COP;
    const int ci = (int)(r % CIP), tap = (int)(r / CIP);
    const float v = (ci < CI && co < CO) ? w[((long long)tap * CI + ci) * CO + co] : 0.f;
    Elem<T>::st(dst_t + (long long)co * taps * CIP + (long long)tap * CIP + ci, v);
    Elem<T>::st(dst_d + (long long)ci * taps * COP + (long long)tap * COP + co, v);
  }
}

// prep_weights_pair for MANY convolutions in one launch (the per-step refresh of every bf16 compute copy after the
// optimizer step: 52 launches of ~8 us each otherwise, all of them dominated by 2-byte scattered stores).
// table[t] = {src, dst_t, dst_d, taps, CI, CO, CIP, COP}; workgroup b handles the 64 x 64 (ci, co) tile chunks[b] =
// {t, tile} of one tap: the tile goes through LDS so that BOTH copies are written in 128-byte runs (dst_d along co as
// read, dst_t along ci after the transpose).
constexpr int kPrepTile = 64;
template <typename T>
__global__ __launch_bounds__(256) void prep_weights_pair_multi(const long long* __restrict__ table,
                                                               const long long* __restrict__ chunks) {
  __shared__ float tile[kPrepTile][kPrepTile + 1];
  const long long t = chunks[2 * blockIdx.x];
  int idx = (int)chunks[2 * blockIdx.x + 1];
  const long long* e = table + 8 * t;
  const float* __restrict__ w = (const float*)e[0];
  T* __restrict__ dst_t = (T*)e[1];
  T* __restrict__ dst_d = (T*)e[2];
  const int taps = (int)e[3], CI = (int)e[4], CO = (int)e[5], CIP = (int)e[6], COP = (int)e[7];
  const int nci = (CIP + kPrepTile - 1) / kPrepTile, nco = (COP + kPrepTile - 1) / kPrepTile;
  const int tap = idx / (nci * nco);
  idx -= tap * nci * nco;
  const int ci0 = (idx / nco) * kPrepTile, co0 = (idx % nco) * kPrepTile;
  const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
#pragma unroll 4
  for (int r = r0; r < kPrepTile; r += 4) {
    const int ci = ci0 + r, co = co0 + c;
    const float v = (ci < CI && co < CO) ? w[((long long)tap * CI + ci) * CO + co] : 0.f;
    tile[r][c] = v;
    if (ci < CIP && co < COP) Elem<T>::st(dst_d + ((long long)ci * taps + tap) * COP + co, v);
  }
  __syncthreads();
#pragma unroll 4
  for (int r = r0; r < kPrepTile; r += 4) {
    const int co = co0 + r, ci = ci0 + c;
    if (ci < CIP && co < COP) Elem<T>::st(dst_t + ((long long)co * taps + tap) * CIP + ci, tile[c][r]);
  }
}

// ------------------------------------------------------------------------------------
// Stem weight gradient of the three-term parity mode from PRE-SPLIT operands (tf2/resnet.py:593-599 under tape.gradient, fp32 storage).
// The multi-tap tile of conv_wgrad_dma above splits both operands in registers (2.75 VALU per element, redone by every wave that
// shares a row), reads them with 4-byte LDS loads and gathers every packed pixel SEVEN times (once per kernel row): 3.3 ms per step
// for 0.3 ms of matrix work (profiles/r06_notes.md).  Here both operands arrive as bf16 pieces:
//   xq  [V][HP][WP] packed pixels of 16 bytes: the four hi pieces (c0..c3; c3 = 0), then the four lo pieces (simclr_presplit_packed);
//   dy  [M][64] in the pre-split block format of common.h (simclr_bn_bwd_apply with SIMCLR_FMT_PS_OUT).
// A workgroup (two waves) walks 32-pixel segments of output rows.  Per segment ONE window of the image -- KH rows x (31 * STRIDE + 8)
// packed pixels, 7.7 KB for the 7x7 / stride-2 stem instead of 28 KB of gathered rows -- and the 8 KB gradient tile arrive by LDS-DMA
// (ring of STAGES segments).  Every MFMA fragment is two transposing LDS reads (ds_read_b64_tr_b16), no VALU: for the image operand
// lane (fl, g) addresses packed pixel STRIDE * (8 g + (fl >> 2)) + tap (tap = 4 * half + (fl & 3)) of kernel row r, i.e. the k-rows
// r * 32 + half * 16 .. + 15 = (four taps x four channels) of fragment 2 r + half; the gradient operand is read as in the PSD = 1
// instantiation above.  Wave w owns fragments KH * w .. KH * w + KH - 1 (7 x 4 accumulator tiles for the 7x7 stem) over all 64
// output channels: per segment 8 + 4 * KH transposing reads feed 12 * KH MFMAs.
// Output: fp32 slabs [split][KH * 32][64] (k = kh * 32 + kw * 4 + ci, the layout unpack_stem_dw reads) reduced by slab_reduce.
// ------------------------------------------------------------------------------------
struct StemWgP {
  const void* xq;
  const void* dy;
  float* dw;
  const void* zero;
  int HP, WP, OH, OW, segs;
  int chunks, splits, chunks_per_split;
};

template <int KH, int STRIDE, int STAGES>
__global__ __launch_bounds__(128, 2) void stem_wgrad_ps(const StemWgP p) {
  constexpr int WCOLS = 31 * STRIDE + 8;                 // packed pixels per window row
  constexpr int WSLOTS = KH * WCOLS;
  constexpr int WJ = (WSLOTS + 63) / 64;                 // LDS-DMA instructions per window
  constexpr int WBYTES = WJ * 1024;
  constexpr int DJ = 8;                                  // ... per gradient tile: 32 pixels x 256 bytes
  constexpr int STAGE = WBYTES + DJ * 1024;
  constexpr int WJW = (WJ + 1) / 2, DJW = DJ / 2;        // per wave
  constexpr int LPC = WJW + DJW;
  constexpr int NI = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, fl = lane & 15;
  const int split = blockIdx.x;
  if (split >= p.splits) return;
  const int c_begin = split * p.chunks_per_split, c_end = min(p.chunks, c_begin + p.chunks_per_split);
  const unsigned char* __restrict__ XQ = (const unsigned char*)p.xq;
  const unsigned char* __restrict__ DY = (const unsigned char*)p.dy;
  auto key_b = [](int px) __attribute__((always_inline)) { return (px & 3) | (((px >> 3) & 1) << 2); };

  // ---- per-lane source state: fixed for the whole kernel, only the segment's base moves ----
  int w_off[WJW], w_col[WJW];       // byte offset of this lane's packed pixel inside the window's image rows; its column (-1: no slot)
#pragma unroll
  for (int j = 0; j < WJW; ++j) {
    const int slot = (wave * WJW + j) * 64 + lane;
    const int r = slot / WCOLS, col = slot - r * WCOLS;
    const bool ok = slot < WSLOTS && wave * WJW + j < WJ;
    w_col[j] = ok ? col : -1;
    w_off[j] = (r * p.WP + col) * 16;
  }
  int d_off[DJW], d_px[DJW];
#pragma unroll
  for (int j = 0; j < DJW; ++j) {
    const int px = (wave * DJW + j) * 4 + (lane >> 4), pc = lane & 15;
    const int lc = (((pc >> 1) ^ key_b(px)) << 1) | (pc & 1);          // logical 16-byte chunk held by this LDS slot
    d_px[j] = px;
    d_off[j] = px * 256 + lc * 16;
  }
  auto issue = [&](int stage, int c) __attribute__((always_inline)) {
    const int row = c / p.segs, seg = c - row * p.segs;                 // (workgroup-uniform)
    const int v = row / p.OH, oy = row - v * p.OH, ox0 = seg * 32;
    const long long xbase = (((long long)v * p.HP + oy * STRIDE) * p.WP + ox0 * STRIDE) * 16;
    const long long dbase = ((long long)row * p.OW + ox0) * 256;
    unsigned char* w_dst = smem + stage * STAGE + wave * (WJW * 1024);
    unsigned char* d_dst = smem + stage * STAGE + WBYTES + wave * (DJW * 1024);
#pragma unroll
    for (int j = 0; j < WJW; ++j) {
      if (wave * WJW + j < WJ) {
        const bool ok = w_col[j] >= 0 && ox0 * STRIDE + w_col[j] < p.WP;
        const void* src = ok ? (const void*)(XQ + xbase + w_off[j]) : p.zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(w_dst + j * 1024), 16, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < DJW; ++j) {
      const void* src = ox0 + d_px[j] < p.OW ? (const void*)(DY + dbase + d_off[j]) : p.zero;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(d_dst + j * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[KH][NI];
#pragma unroll
  for (int i = 0; i < KH; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int stage) __attribute__((always_inline)) {
    const unsigned char* Ws = smem + stage * STAGE;
    const unsigned char* Ds = Ws + WBYTES;
    const int px0 = g * 8 + (fl >> 2);
    mma_f32_chunks<NI, KH, true, 3, true, true>(&acc[0][0],
        [&](int nf, int h) {                                     // gradient: 16-position run nf of the block format, h = 0 hi | 1 lo
          const int byte = (nf >> 1) * 128 + h * 64 + (nf & 1) * 32 + (fl & 3) * 8;
          auto off = [&](int px) { return px * 256 + (((byte >> 5) ^ key_b(px)) << 5) + (byte & 31); };
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Ds + off(px0)));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Ds + off(px0 + 4)));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          return (u32x4){l2[0], l2[1], h2[0], h2[1]};
        },
        [&](int i, int h) {                                      // image: fragment f = 2 r + half of this wave
          const int f = wave * KH + i, r = f >> 1, half = f & 1;
          const int slot = r * WCOLS + STRIDE * px0 + 4 * half + (fl & 3);
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Ws + slot * 16 + h * 8));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Ws + (slot + 4 * STRIDE) * 16 + h * 8));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          return (u32x4){l2[0], l2[1], h2[0], h2[1]};
        });
  };

  int issued = c_begin;
#pragma unroll
  for (int s_ = 0; s_ < STAGES - 1; ++s_)
    if (issued < c_end) { issue(s_, issued); ++issued; }
  int cs = 0, is = STAGES - 1;
  for (int c = c_begin; c < c_end; ++c) {
    const int ahead = issued - c - 1;
    if (STAGES >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPC) : "memory");
    else if (STAGES >= 3 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (issued < c_end) { issue(is, issued); ++issued; is = (is + 1 == STAGES) ? 0 : is + 1; }
    compute(cs);
    cs = (cs + 1 == STAGES) ? 0 : cs + 1;
  }
  // D[n position g * 4 + reg of run nf][k row fl of fragment f]  ->  slab[split][f * 16 + fl][channel .. + 3]
  float* slab = p.dw + (long long)split * (KH * 32 * 64);
#pragma unroll
  for (int i = 0; i < KH; ++i) {
    const int kk = (wave * KH + i) * 16 + fl;
#pragma unroll
    for (int nf = 0; nf < NI; ++nf) {
      const int Q = 4 * (nf & 1) + g;
      const int n = (nf >> 1) * 32 + (Q & 1) * 16 + (Q >> 1) * 4;
      *(float4*)(slab + kk * 64 + n) = make_float4(acc[i][nf][0], acc[i][nf][1], acc[i][nf][2], acc[i][nf][3]);
    }
  }
}

// presplit_rows for many matrices in one launch (blockIdx.y = matrix): table [n][4] = (src, dst, 128-byte blocks, fp16 pieces?)
__global__ __launch_bounds__(256) void presplit_rows_multi(const long long* __restrict__ table, float f16_scale) {
  const long long* e = table + 4ll * blockIdx.y;
  const float* __restrict__ src = (const float*)e[0];
  uint32_t* __restrict__ dst = (uint32_t*)e[1];
  const long long nchunks = e[2] * 8;
  const bool f16 = e[3] != 0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nchunks; i += (long long)gridDim.x * 256ll) {
    const long long blk = i >> 3;
    const int c = (int)(i & 7), gq = c & 3;
    u32x4 c0 = *(const u32x4*)(src + blk * 32 + gq * 4), c1 = *(const u32x4*)(src + blk * 32 + 16 + gq * 4);
    u32x4 hi, lo;
    if (f16) {
#pragma unroll
      for (int q = 0; q < 4; ++q) { c0[q] = __float_as_uint(__uint_as_float(c0[q]) * f16_scale); c1[q] = __float_as_uint(__uint_as_float(c1[q]) * f16_scale); }
      split_terms2_f16(c0, c1, hi, lo);
    } else {
      split_terms2(c0, c1, hi, lo);
    }
    *(u32x4*)(dst + i * 4) = (c < 4) ? hi : lo;
  }
}

// packed image [npix][4] fp32 -> [npix] x (four bf16 hi pieces, four bf16 lo pieces): the image operand of stem_wgrad_ps
__global__ __launch_bounds__(256) void presplit_packed(const float4* __restrict__ src, u32x4* __restrict__ dst, long long npix) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < npix; i += (long long)gridDim.x * 256ll) {
    const float4 v = src[i];
    uint32_t h0, l0, h1, l1;
    split_pair<false>(v.x, v.y, h0, l0);
    split_pair<false>(v.z, v.w, h1, l1);
    dst[i] = (u32x4){h0, h1, l0, l1};
  }
}

// stem wgrad result [KHP*KWP*4][CO] -> HWIO [KH][KW][CI][CO]  (+= if accumulate)
__global__ void unpack_stem_dw(const float* __restrict__ src, float* __restrict__ dst, int KH, int KW,
                               int CI, int CO, int KWP, int accumulate) {
  const int total = KH * KW * CI * CO;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int co = i % CO;
    const int ci = (i / CO) % CI;
    const int kw = (i / (CO * CI)) % KW;
    const int kh = i / (CO * CI * KW);
    const float v = src[(long long)((kh * KWP + kw) * 4 + ci) * CO + co];
    dst[i] = accumulate ? dst[i] + v : v;
  }
}

// device address of g_zero16 (looked up once); nullptr if the lookup fails -- entry points refuse to launch then
// fp32 matrix arithmetic of the forward and of the two backward GEMMs (process-wide, simclr_set_f32_matmul)
static int g_f32_terms_fwd = 0, g_f32_terms_bwd = 0;
// Per-call override (re-entrant form): an entry point's `dtype` argument may carry the matrix arithmetic of THIS
// call in bits 12..19 -- SIMCLR_FMT_TERMS(t) = (t + 1) << 12, t in {0, 3, 6, 13} -- in which case the process-wide default set by
// simclr_set_f32_matmul is not consulted at all.  terms_of() returns the terms a call runs with and strips the field.
static bool valid_terms(int t, bool fwd) { return t == 0 || t == 3 || t == 6 || (fwd && t == 13); }
static int terms_of(int* dtype, bool fwd) {
  const int f = (*dtype >> 12) & 0xff;
  *dtype &= ~(0xff << 12);
  if (f == 0) return fwd ? g_f32_terms_fwd : g_f32_terms_bwd;
  return valid_terms(f - 1, fwd) ? f - 1 : -1;
}

// dtype | SIMCLR_FMT_PS_W (fp32 storage): the weight operand of this call is the pre-split copy simclr_presplit_weights(_multi) made of it --
// fp16 pieces of 2^8 * w for a forward call (terms 13), bf16 pieces for a data-gradient call (terms 3).  Strips the flag.
static bool take_ps_w(int* dtype) {
  const bool f = (*dtype & SIMCLR_FMT_PS_W) != 0;
  *dtype &= ~SIMCLR_FMT_PS_W;
  return f;
}

// ---- which instantiation ran: ONE place that records every forward / dgrad launch decision ----------
// SIMCLR_LAUNCH replaces hipLaunchKernelGGL inside launch_igemm_one: it writes the kernel's template-argument list (as spelled
// at the launch site, with the element size and the GEMM role appended) plus grid / block / LDS into g_last_inst and launches --
// unless SIMCLR_DRY_RUN=1, in which case NOTHING touches the device: tools/instantiation_table.py walks the layer classes of
// the benchmark models on a machine without a GPU, dumps (layer class -> instantiation) to profiles/, and a CPU test asserts
// that the table has not changed behind anybody's back.  simclr_conv2d_last_instantiation() returns the record.
static char g_last_inst[512] = "";
static bool dry_run() { return simclr_dry_run(); }
template <typename T, int MODE> static void record_inst(const char* kern, unsigned grid, unsigned block, size_t lds, const ConvP& p) {
  snprintf(g_last_inst, sizeof(g_last_inst), "%s elt=%d role=%s grid=%u block=%u lds=%zu m_tiles=%d n_tiles=%d tail_parts=%d", kern,
           (int)sizeof(T), MODE == MODE_FWD ? "fwd" : "dgrad", grid, block, lds, p.m_tiles, p.n_tiles, p.rem_parts);
}
#define SIMCLR_LAUNCH(kern, grid, block, lds, stream, ...)                                   \
  do {                                                                                       \
    record_inst<T, MODE>(#kern, (grid).x, (block).x, (size_t)(lds), p);                      \
    if (!dry_run()) hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);         \
  } while (0)

static const void* zero_page() {
  static void* zp = nullptr;
  if (dry_run()) return (const void*)&g_last_inst;          // never dereferenced: nothing is launched in a dry run
  if (!zp && hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero16)) != hipSuccess) zp = nullptr;
  return zp;
}

// Persistent-grid geometry of conv_igemm_persistent for an [M, N] output: tile width, N-tiles, grid size.
// 2-3 workgroups per CU, rounded to a multiple of 8*n_tiles, at most one workgroup per tile.
static void igemm_persistent_grid(long long M, int N, int* bn, int* n_tiles, int* grid, bool narrow = false) {
  const int BN = (N <= 64 || narrow) ? 64 : 128;
  const int m_tiles = (int)((M + 127) / 128);
  const int nt = ceil_div(N, BN);
  const int unit = 8 * nt;
  const int resident = (BN == 64 ? 3 : 2) * 256;      // workgroups that fit: LDS 48 KB / 64 KB each
  int pg = (resident / unit) * unit;
  if (pg < unit) pg = unit;
  const int need = ceil_div(m_tiles, 8) * unit;       // enough workgroups to give every M-tile a slot
  if (pg > need) pg = need;
  *bn = BN; *n_tiles = nt; *grid = pg;
}

// Tile choice of the bf16 forward / dgrad launches: true = 256 x 256 (8 waves), false = 128 x (64 | 128) (4 waves).
// Measured per ResNet-50 layer class at 1024 views (profiles/r03_notes.md, tools/microbench.py --what tile): the wide tile
// wins on the 1x1 convolutions with >= 256 output channels and >= 150 k rows (-5...-18 %: forward with statistics, forward
// with the fused BatchNorm-apply epilogue, plain and K-extended dgrad), ties at 7^2 (196 M-tiles leave the last round of
// 256 workgroups 3/4 empty) and LOSES wherever the row-wise epilogue carries the fused BN-backward reduce (one workgroup
// per CU cannot hide that pass behind another workgroup's MFMAs: +5...+18 %) and on the 3x3 layers (the 128-wide
// halo-window path moves fewer L2->LDS bytes than the 256-wide gather).
static bool igemm_use_256(const ConvP& p, bool fwd) {
  if (p.N % 256 != 0 || p.ntaps <= 0 || p.N / 256 > 32) return false;
  const char* e = getenv("SIMCLR_IGEMM_TILE");
  const int mode = e ? atoi(e) : 0;
  if (mode == 256) return true;
  if (mode == 128) return false;
  if (p.KH != 1 || p.KW != 1 || p.M < 150000) return false;
  if (p.bn_mode && !p.fapply) return false;          // fused BN-backward reduce epilogue
  if (fwd && !p.fapply && p.stats && p.K < 128) return false;
  // classes (SIMCLR_IGEMM_256_CLASSES, bit mask): 1 = forward with the fused BatchNorm-apply epilogue, 2 = other forward
  // launches, 4 = plain / K-extended dgrad.  Inside the training step class 1 LOSES 2.5 ms of 67 on ResNet-50 1x although it
  // wins stand-alone (a 151 KB workgroup needs a whole CU's LDS: it cannot start until the previous kernel has drained
  // from that CU, nothing can start beside it, and the fused epilogue has no second workgroup to hide behind); classes
  // 2 and 4 are neutral there (66.96 / 67.01 vs 66.9 ms) and gain 2.5 % on ResNet-50 2x + SK (298.9 vs 305.7 ms,
  // profiles/r03_notes.md) -> default 6.
  const char* c = getenv("SIMCLR_IGEMM_256_CLASSES");
  const int classes = c ? atoi(c) : 6;
  const int cls = p.fapply ? 1 : (fwd ? 2 : 4);
  return (classes & cls) != 0;
}

// Eight-phase 256 x 256 tile (csrc/igemm_wide.h).  SIMCLR_IGEMM_WIDE: 0 = off, 1 = on for the layers the rule below picks,
// 2 = on wherever the kernel is applicable (tests).  Applicable: bf16, Cout a multiple of 256, whole 64-channel k-tiles, no
// K-extension / fused BatchNorm-apply epilogue (those stay on conv_igemm_persistent).
static bool igemm_use_wide(const ConvP& p, bool fwd) {
  const char* e = getenv("SIMCLR_IGEMM_WIDE");
  const int mode = e ? atoi(e) : 1;
  if (mode <= 0) return false;
  if (p.N % 256 != 0 || p.N / 256 > 32 || p.ntaps <= 0 || p.IC % 64 != 0 || p.x2 || p.fapply) return false;
  if (p.bn_mode && p.bn_mode != 4 && !p.bn_x) return false;
  // 32-bit byte offsets into the operands, below the out-of-range marker of the kernel (0xF0000000)
  if ((long long)p.V * p.IH * p.IW * p.pixpitch * 2 >= 0xE0000000ll || (long long)p.N * p.K * 2 >= 0xE0000000ll) return false;
  if (mode == 2) return true;
  // Rule from the per-layer A/B at 1024 views (tools/microbench.py --what wide, profiles/r04_notes.md; us, 128-wide -> wide):
  //   forward (statistics, plain or statistics-only): wins wherever it applies -- 56^2 64->256 544 -> 443, 14^2 1024->512
  //   254 -> 223, 14^2 1024->2048 s2 288 -> 239, 7^2 2048->512 132 -> 113 -- except the 3x3 stride-1 layers, where the
  //   halo-window path of the 128-wide tile moves fewer bytes (14^2 260 vs 267);
  //   plain dgrad (no fused BatchNorm reduce): 1x1 layers with >= 8 k-tiles (14^2 1024->512 267 -> 243, 1024->2048 s2
  //   360 -> 313, 7^2 512->2048 115 -> 105); shorter reductions and the strided 3x3 lose;
  //   dgrad + BatchNorm-backward reduce: loses everywhere but at 7^2 (+4 ... +29 %: one workgroup per CU cannot hide the
  //   row pass's operand loads) -> stays on conv_igemm_persistent.
  if (p.M < 32768) return false;
  // modes 3 / 4 (A/B runs): the rule for the forward launches only / for the data-gradient launches only.
  // Forward, round 5: only launches of >= 300 GFLOP.  Stand-alone the wide tile wins on every layer it applies to, but a 150 KB workgroup
  // needs a whole CU's LDS -- it cannot start until the previous kernel has drained from that CU and nothing starts beside it -- and
  // INSIDE the training step that fixed cost outweighs the gain on ResNet-50 1x (105 ... 237 GFLOP per launch: 63.19 -> 62.65 ms per
  // step with the forward rule off, three interleaved runs) while ResNet-50 2x + SK (420 ... 950 GFLOP per launch) loses 2.4 ms of 286
  // without it (profiles/r05_notes.md section 8).
  // mode 5: the rule of round 4 (no work threshold on the forward launches), kept for A/B runs against the current default
  if (fwd) return mode != 4 && !(p.KH == 3 && p.KW == 3 && p.stride == 1) && (mode == 3 || mode == 5 || 2.0 * (double)p.M * p.K * p.N >= 3.0e11);
  return mode != 3 && !p.bn_mode && !p.accumulate && p.KH == 1 && p.KW == 1 && p.IC >= 512;
}

// Short-K layers (1x1 convolutions from <= SIMCLR_IGEMM_BN64_K channels, default 128: one or two k-tiles per output tile)
// are pure streaming: with the 64-wide tile three workgroups fit a CU instead of two, so the load phase of one tile, the
// MFMA / staging of another and the store phase of a third overlap (the gathered operand is then re-read from L2 by N/64
// workgroups instead of N/128 -- irrelevant for an HBM-bound layer).  Measured in the step: 66.49 / 66.33 vs 66.72 / 66.66 ms
// (-0.3 ms, two interleaved pairs, profiles/r03_notes.md); K <= 64 alone and K <= 256 give 66.60 / 66.50.
static bool igemm_narrow(const ConvP& p, size_t esz, bool dgrad = false) {
  // Round 5 (three interleaved runs each on one box, profiles/r05_notes.md section 8): K <= 64 only -- 63.19 -> 62.56 ms per step; the K = 128
  // layers (28^2 128 -> 512) are better off on the 128-wide tile now that its epilogues carry their options as compile-time constants.
  static const int kmax = getenv("SIMCLR_IGEMM_BN64_K") ? atoi(getenv("SIMCLR_IGEMM_BN64_K")) : 64;
  // fp32 storage, three-term DATA GRADIENT launches (their 64-wide tiles run three per CU): K <= 128 (SIMCLR_IGEMM_BN64_K32).  Measured in
  // the step, three interleaved triples on one box (r06_call23): data-gradient family 44.40 -> 43.63 ms (K <= 64: 43.93); the forward
  // launches LOSE on the narrow tile (38.69 -> 39.77 ms) and keep 128 columns.
  static const int kmax32 = getenv("SIMCLR_IGEMM_BN64_K32") ? atoi(getenv("SIMCLR_IGEMM_BN64_K32")) : 128;
  if (esz == 4)
    return dgrad && kmax32 > 0 && p.split == 3 && p.N > 64 && p.KH == 1 && p.KW == 1 && p.stride == 1 && !p.x2 && p.K <= kmax32 && p.N / 64 <= 64;
  return esz == 2 && p.N > 64 && p.KH == 1 && p.KW == 1 && p.stride == 1 && !p.x2 && p.K <= kmax && p.N / 64 <= 64;
}

// ---- split tail of the persistent forward / dgrad grid (ConvP::rem_*) -------------------------------------------------
// The persistent grid gives every workgroup ceil(m_tiles / mslots) tiles although the last round is mostly empty: 3.06
// rounds of work cost 4 (7^2 layers at 1024 views: +31 %), 6.1 cost 7 (14^2: +14 %).  With the split tail every workgroup
// runs floor(m_tiles / mslots) whole tiles and the R left-over tiles are shared along the reduction by P = mslots / R
// workgroups each (at most 8, at least two k-steps per part), combined through an fp32 scratch owned by the library:
// one buffer per stream (launches of one stream are ordered, so a slot is free again when the next launch starts),
// allocated on first use, grown when a launch needs more.  SIMCLR_IGEMM_SPLIT=0 switches it off (A/B runs).
struct SplitScratch { hipStream_t stream; float* ws; size_t floats; unsigned* flags; size_t nflags; };   // flags[nflags] = time-out counter
static SplitScratch g_split_scratch[8];
static int g_split_scratch_n = 0;

// units: pieces one tile's reduction can be cut in (k-steps; 64-channel chunks of nine k-steps on the halo-window path)
static int g_last_split_parts = 0;      // simclr_conv2d_last_split_parts (tests)
// `wide`: the caller is the eight-phase 256 x 256 launch.  Default policy (SIMCLR_IGEMM_SPLIT unset): split the tail of the
// wide launches only -- measured per ResNet-50 layer at 1024 views (profiles/r04_notes.md) the 128-wide tiles gain 4-6 % on the
// 3x3 layers at 14^2 / 7^2 and LOSE up to 14 % on the 1x1 layers at 7^2 (a 17-30 us tile is not long enough to pay for the
// exchange), -0.65 ms per training step in sum; the 256-wide tiles gain 1-4 % (their tiles are four times longer).
// SIMCLR_IGEMM_SPLIT=0: never, =1: every eligible launch, >= 2: every eligible launch with at most that many parts.
static bool igemm_split_tail(ConvP& p, int grid, int units, int unit_steps, size_t tile_floats, hipStream_t stream, bool wide = false) {
  g_last_split_parts = 0;
  p.rem_parts = 0; p.rem_full = 0; p.rem_tiles = 0; p.part_ws = nullptr; p.part_flags = nullptr; p.part_err = nullptr;
  const char* e = getenv("SIMCLR_IGEMM_SPLIT");
  const int mode = e ? atoi(e) : (wide ? 1 : 0);
  if (mode == 0 || p.n_tiles <= 0 || grid % p.n_tiles != 0) return true;
  const int mslots = grid / p.n_tiles;
  const int full = p.m_tiles / mslots, R = p.m_tiles - full * mslots;
  const int max_parts = mode > 1 ? mode : 8;
  int P = R > 0 ? mslots / R : 0;
  if (P > max_parts) P = max_parts;
  // at least `min_steps` k-steps per part: a part must be worth its exchange (64 - 256 KB written through + re-read, one
  // acquire on the owner: 3 - 6 us, i.e. several k-steps)
  constexpr int min_steps = 8;
  int max_parts_u = (units * unit_steps) / (min_steps > 0 ? min_steps : 1);      // parts are whole units
  if (max_parts_u > units) max_parts_u = units;
  if (P > max_parts_u) P = max_parts_u;
  // nothing to gain: no remainder, too few workgroups per remainder tile, short reductions (streaming layers: the
  // partial exchange would cost more than the idle round), or so many rounds that one more is noise
  if (R == 0 || full == 0 || P < 2 || full > 24) return true;
  const size_t slots = (size_t)R * p.n_tiles * (P - 1);
  if (dry_run()) {                                  // decision only: nothing is allocated, nothing will be launched
    p.rem_full = full; p.rem_tiles = R; p.rem_parts = P; g_last_split_parts = P;
    return true;
  }
  SplitScratch* sc = nullptr;
  for (int i = 0; i < g_split_scratch_n; ++i)
    if (g_split_scratch[i].stream == stream) sc = &g_split_scratch[i];
  if (!sc) {
    if (g_split_scratch_n == 8) return true;        // more streams than scratch buffers: run without the split
    sc = &g_split_scratch[g_split_scratch_n++];
    *sc = SplitScratch{stream, nullptr, 0, nullptr, 0};
  }
  if (sc->floats < slots * tile_floats || sc->nflags < slots) {
    // grow.  Not while the stream is being captured into a hipGraph (allocation is not a capturable operation): the un-split
    // schedule is still correct.  hipFree synchronises with the device: earlier launches that use the old buffers have finished.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (stream && (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)) {
      (void)hipGetLastError();
      return true;
    }
    unsigned lost = 0;                                 // the time-out count travels to the new allocation (sticky)
    if (sc->flags) (void)hipMemcpy(&lost, sc->flags + sc->nflags, sizeof(unsigned), hipMemcpyDeviceToHost);
    if (sc->ws) (void)hipFree(sc->ws);
    if (sc->flags) (void)hipFree(sc->flags);
    sc->ws = nullptr; sc->flags = nullptr; sc->floats = 0; sc->nflags = 0;
    const size_t want_f = slots * tile_floats, want_n = slots < 1024 ? 1024 : slots;
    // the flags are zeroed ON THE LAUNCH STREAM (ordered before the launch that follows, whatever kind of stream it is)
    if (hipMalloc((void**)&sc->ws, want_f * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&sc->flags, (want_n + 1) * sizeof(unsigned)) != hipSuccess ||
        hipMemsetAsync(sc->flags, 0, (want_n + 1) * sizeof(unsigned), stream) != hipSuccess ||
        (lost && hipMemcpyAsync(sc->flags + want_n, &lost, sizeof(unsigned), hipMemcpyHostToDevice, stream) != hipSuccess) ||
        (lost && hipStreamSynchronize(stream) != hipSuccess)) {
      (void)hipGetLastError();
      if (sc->ws) (void)hipFree(sc->ws);
      if (sc->flags) (void)hipFree(sc->flags);
      sc->ws = nullptr; sc->flags = nullptr;
      return true;                                   // no scratch: the un-split schedule is still correct
    }
    sc->floats = want_f; sc->nflags = want_n;
  }
  p.rem_full = full; p.rem_tiles = R; p.rem_parts = P;
  g_last_split_parts = P;
  p.part_ws = sc->ws; p.part_flags = sc->flags; p.part_err = sc->flags + sc->nflags;
  return true;
}

// Number of split-tail partners that never arrived (summed over every stream's scratch) since the library was loaded: 0 on a
// healthy device.  Synchronous (copies the sticky counters back): call it where the host synchronises anyway.
static int split_tail_timeouts(unsigned* host_total) {
  SIMCLR_CHECK_ARG(host_total != nullptr, "conv2d_split_tail_timeouts: null argument");
  unsigned total = 0;
  for (int i = 0; i < g_split_scratch_n; ++i) {
    if (!g_split_scratch[i].flags) continue;
    unsigned v = 0;
    if (hipMemcpy(&v, g_split_scratch[i].flags + g_split_scratch[i].nflags, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) {
      simclr_set_error("conv2d_split_tail_timeouts: %s", hipGetErrorString(hipGetLastError()));
      return 2;
    }
    total += v;
    // After a time-out the late producer may still publish its flag: it would be taken for the partner of a LATER launch (or hipGraph
    // replay) that uses the slot.  The caller has synchronised (see above), so nothing is in flight: clear the flags, keep the counter.
    if (v != 0 && hipMemset(g_split_scratch[i].flags, 0, g_split_scratch[i].nflags * sizeof(unsigned)) != hipSuccess) {
      simclr_set_error("conv2d_split_tail_timeouts: %s", hipGetErrorString(hipGetLastError()));
      return 2;
    }
  }
  *host_total = total;
  return 0;
}

// Pre-split weights of the fp32 data gradient under three split-bf16 terms (simclr_set_f32_matmul(*, 3)): the weight
// operand of a launch is rewritten by presplit_rows into a library-owned per-stream buffer (launches of one stream are
// ordered, so the buffer is free again when the next launch's copy starts) and the PSB instantiation reads it without any
// splitting work -- the k-loop of the three-term kernels carried ~4 VALU instructions per MFMA, half of them for the
// weights.  SIMCLR_F32_PRESPLIT=0 keeps the in-register split (read per launch: A/B runs and the bitwise test).
struct PresplitScratch { hipStream_t stream; void* buf; size_t bytes; };
static PresplitScratch g_presplit[8];
static int g_presplit_n = 0;
static int g_last_presplit = 0;          // simclr_conv2d_last_presplit (tests): 1 if the most recent dgrad launch ran PSB
static int g_igemm_fail = 0;             // set by launch_igemm_one when a launch cannot honour its operand format (pre-split input without PSB)
static const void* presplit_weights(const void* w, long long rows, int K, hipStream_t stream, bool f16 = false) {
  const char* e = getenv("SIMCLR_F32_PRESPLIT");
  if ((e && atoi(e) == 0) || K % 32 != 0 || rows <= 0) return nullptr;
  if (dry_run()) return w;                           // decision only
  PresplitScratch* sc = nullptr;
  for (int i = 0; i < g_presplit_n; ++i)
    if (g_presplit[i].stream == stream) sc = &g_presplit[i];
  if (!sc) {
    if (g_presplit_n == 8) return nullptr;
    sc = &g_presplit[g_presplit_n++];
    *sc = PresplitScratch{stream, nullptr, 0};
  }
  const size_t need = (size_t)rows * K * sizeof(float);
  if (sc->bytes < need) {
    if (sc->buf) (void)hipFree(sc->buf);          // synchronises with the device: earlier readers have finished
    sc->buf = nullptr; sc->bytes = 0;
    const size_t want = need < (16u << 20) ? (16u << 20) : need;
    if (hipMalloc(&sc->buf, want) != hipSuccess) { (void)hipGetLastError(); sc->buf = nullptr; return nullptr; }
    sc->bytes = want;
  }
  const long long nblocks = rows * (K / 32);
  if (f16) hipLaunchKernelGGL(presplit_rows<true>, dim3((unsigned)ceil_div(nblocks * 8, 256)), dim3(256), 0, stream,
                              (const float*)w, (uint32_t*)sc->buf, nblocks, (float)(1 << F16_WSCALE_LOG2));
  else hipLaunchKernelGGL(presplit_rows<false>, dim3((unsigned)ceil_div(nblocks * 8, 256)), dim3(256), 0, stream,
                          (const float*)w, (uint32_t*)sc->buf, nblocks, 1.0f);
  return sc->buf;
}

template <typename T, int MODE>
void launch_igemm_one(ConvP p, hipStream_t stream) {
  const bool narrow = igemm_narrow(p, sizeof(T), MODE == MODE_DGRAD);
  const int BN = (p.N <= 64 || narrow) ? 64 : 128;
  p.m_tiles = ceil_div(p.M, 128);
  p.n_tiles = ceil_div(p.N, BN);
  if (p.M <= 0) return;
  // p.split: set by the entry point (terms_of: the call's own terms or the process-wide default)
#ifdef SIMCLR_DIAG
  { const char* e = getenv("SIMCLR_DIAG"); p.diag = e ? atoi(e) : 0; }
#endif
  const int grid = ceil_div(p.m_tiles, 8) * 8 * p.n_tiles;
  size_t lds = 2 * (128 + BN) * 128;
  const size_t epi = 128 * (BN * 2 + 8) + 128 * sizeof(long long);
  if (sizeof(T) == 2 && epi > lds) lds = epi;
  const bool st = p.stats != nullptr;
  static const bool no_glds = getenv("SIMCLR_NO_GLDS") != nullptr;   // A/B switches for benchmarking
  static const bool no_persist = getenv("SIMCLR_NO_PERSISTENT") != nullptr;
  if ((p.bn_mode || p.fapply || !p.y || (!no_glds && !no_persist)) && p.ntaps > 0 && p.n_tiles <= 64) {
    // Tile choice: 128x128 / 128x64, 4 waves, 2-stage LDS ring, 2-3 workgroups per CU.  Measured
    // alternatives that were slower on every ResNet-50 layer (profiles/r01_notes.md): 256x128 with 8
    // waves (one workgroup per CU), and 3-stage rings with counted vmcnt for either tile.
    // persistent grid: 2-3 workgroups per CU, rounded to a multiple of 8*n_tiles, at most one per tile
    int bn_, nt_, pg;
    igemm_persistent_grid(p.M, p.N, &bn_, &nt_, &pg, narrow);
    const size_t plds = 2 * (128 + BN) * 128 + 5 * BN * sizeof(float) + 128 * sizeof(long long);
    // 256 x 256 tile, 8 waves of 128 x 64 (one workgroup per CU): half the L2->LDS bytes per FLOP of the 128 x 128 tile
    // (which needs 64 B/clk/CU from L2 at the MFMA peak -- more than an XCD's L2 delivers) and 0.375 instead of 0.5 KB
    // of LDS fragment reads per MFMA.  SIMCLR_IGEMM_TILE: 128 / 256 force a tile, unset = igemm_use_256 (measured per
    // layer class, profiles/r03_notes.md).  The statistics slots of the 128-wide geometry (simclr_conv2d_stats_slots)
    // are never fewer than this grid's M-slots.
    if constexpr (sizeof(T) == 2) {
      if (igemm_use_wide(p, MODE == MODE_FWD)) {
        // eight-phase 256 x 256 tile (csrc/igemm_wide.h): one 8-wave workgroup per CU, 150 KB of LDS
        p.x_bytes = (unsigned)((long long)p.V * p.IH * p.IW * p.pixpitch * 2);
        { const int dist[3] = {8, 128, 136}, chw = p.cls_h * p.cls_w;
          for (int i = 0; i < 3; ++i) {
            p.rs_dq[i] = dist[i] / chw;
            const int r = dist[i] - p.rs_dq[i] * chw;
            p.rs_drow[i] = r / p.cls_w; p.rs_dcol[i] = r - p.rs_drow[i] * p.cls_w;
          } }
        p.m_tiles = ceil_div(p.M, 256);
        p.n_tiles = p.N / 256;
        const int unit = 8 * p.n_tiles;
        int pgw = max(unit, (256 / unit) * unit);
        pgw = min(pgw, ceil_div(p.m_tiles, 8) * unit);
        // ring 128 KB + BN parameters 4 KB + row offsets 2 KB + per-wave statistics 16 KB + border coordinates 8 KB
        const size_t ldsw = 131072 + 4 * 256 * sizeof(float) + 256 * sizeof(long long) + 8 * 256 * 2 * sizeof(float) + 4 * 512 * sizeof(int);
        igemm_split_tail(p, pgw, p.ntaps * (p.IC / 64), 1, (size_t)256 * 256, stream, true);
        const bool flatw = p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0 && p.cs == 1 && p.IH == p.OH && p.IW == p.OW;
#define LWD(STv, BEv)                                                                                                          \
        do {                                                                                                                   \
          if (flatw) SIMCLR_LAUNCH((conv_igemm_wide<MODE, STv, BEv, true>), dim3(pgw), dim3(512), ldsw, stream, p);       \
          else SIMCLR_LAUNCH((conv_igemm_wide<MODE, STv, BEv, false>), dim3(pgw), dim3(512), ldsw, stream, p);            \
        } while (0)
        if (p.bn_mode) LWD(true, true);
        else if (st) LWD(true, false);
        else LWD(false, false);
#undef LWD
        return;
      }
      if (igemm_use_256(p, MODE == MODE_FWD)) {
        p.m_tiles = ceil_div(p.M, 256);
        p.n_tiles = p.N / 256;
        const int unit = 8 * p.n_tiles;
        int pg2 = max(unit, (256 / unit) * unit);
        pg2 = min(pg2, ceil_div(p.m_tiles, 8) * unit);
        const size_t lds2 = 2 * (256 + 256) * 128 + 5 * 256 * sizeof(float) + 256 * sizeof(long long) + 8 * 256 * 2 * sizeof(float);
        igemm_split_tail(p, pg2, p.ntaps * (p.IC / 64) + (p.x2 ? p.ic2 / 64 : 0), 1, (size_t)256 * 256, stream);
        if (p.fapply) { p.rem_parts = 0; g_last_split_parts = 0; }
#define L2(STv, BEv, EXv, FAv)                                                                                                \
        do {                                                                                                                   \
          if (p.rem_parts >= 2) { if constexpr (!(FAv)) SIMCLR_LAUNCH((conv_igemm_persistent<uint16_t, MODE, 256, 256, 8, 2, STv, BEv, EXv, false, FAv, 0, true>), dim3(pg2), dim3(512), lds2, stream, p); } \
          else SIMCLR_LAUNCH((conv_igemm_persistent<uint16_t, MODE, 256, 256, 8, 2, STv, BEv, EXv, false, FAv>), dim3(pg2), dim3(512), lds2, stream, p); \
        } while (0)
        if (p.fapply) { if constexpr (MODE == MODE_FWD) L2(false, false, false, true); }
        else if (p.x2 && p.bn_mode) L2(true, true, true, false);
        else if (p.x2) L2(false, false, true, false);
        else if (p.bn_mode) L2(true, true, false, false);
        else if (st) L2(true, false, false, false);
        else L2(false, false, false, false);
#undef L2
        return;
      }
    }
    const int bk_elems = 128 / (int)sizeof(T);
    const bool win3 = [&] {
      static const bool no_win = []{ const char* e = getenv("SIMCLR_CONV3_WIN"); return e && e[0] == '0'; }();
      // fp32 storage (round 6): the three-term launches with pre-split weights (checked below); SIMCLR_CONV3_WIN=0 gathers in both storages
      // (A/B of the fp32 path: 143.94 / 144.15 / 144.00 -> 141.75 / 141.58 / 141.22 ms per step, r06_call35)
      if (sizeof(T) == 4 && !(p.split == 3 || p.split == 13)) return false;
      return !no_win && !p.fapply && !p.x2 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.cs == 1 &&
             p.ntaps == 9 && p.IH == p.OH && p.IW == p.OW && p.IW <= 62 && p.accumulate < 2;
    }();
    // reduction units the tail can be split in: k-steps, or 64-channel chunks on the halo-window path
    // (bf16 128 x 128 tiles only: the 64-wide tiles are the short-K streaming layers, the fp32 kernels keep whole tiles)
    if (sizeof(T) == 2 && BN == 128 && !p.fapply)
      igemm_split_tail(p, pg, win3 ? p.IC / bk_elems : p.ntaps * (p.IC / bk_elems) + (p.x2 ? p.ic2 / bk_elems : 0),
                       win3 ? 9 : 1, (size_t)128 * BN, stream);
    else { p.rem_parts = 0; g_last_split_parts = 0; }
    // fp32 storage: one instantiation per matrix arithmetic (exact fp32 MFMA, 3 or 6 split-bf16 terms)
    // three-term data gradient: weights pre-split once per launch (PSB)
    bool psb = false;
    if constexpr (sizeof(T) == 4 && MODE == MODE_DGRAD) {
      g_last_presplit = 0;
      if (p.split == 3) {
        if (p.w_ps) { psb = true; g_last_presplit = 1; }
        else {
          const void* ws = presplit_weights(p.w, p.N, p.K, stream);
          if (ws) { p.w = ws; psb = true; g_last_presplit = 1; }
        }
      }
    }
    if (p.x_ps && !psb) { g_igemm_fail = 1; return; }      // a pre-split operand cannot be read as floats: refuse (the entry point reports it)
    // split-fp16 forward (13): fp16 weight planes times 2^F16_WSCALE_LOG2, once per launch; no scratch -> six bf16 terms
    if constexpr (sizeof(T) == 4 && MODE == MODE_FWD) {
      if (p.split == 13) {
        if (p.w_ps) psb = true;
        else {
          const void* ws = presplit_weights(p.w, p.N, p.K, stream, true);
          if (ws) { p.w = ws; psb = true; } else p.split = 6;
        }
      }
    }
    if (p.w_ps && !psb) { g_igemm_fail = 1; return; }      // pre-split weights on a launch that would read them as floats (other terms): refuse
    // fused BatchNorm-backward-reduce epilogue with its mask mode / accumulate flag compiled in (EPS instantiations, bf16):
    // 1 = (mode 2, store), 3 = (mode 4, store), 4 = (mode 4, accumulate); SIMCLR_BNEPI_SPECIAL=0 (read per launch) = generic
    int eps = 0;
    if (sizeof(T) == 2 && MODE == MODE_DGRAD && p.bn_mode) {
      if (p.bn_mode == 2 && !p.accumulate) eps = 1;
      else if (p.bn_mode == 4) eps = p.accumulate ? 4 : 3;
      const char* e = getenv("SIMCLR_BNEPI_SPECIAL");
      if (e && atoi(e) == 0) eps = 0;
    }
#define LPE(BNv, EXv, EPSv) SIMCLR_LAUNCH((conv_igemm_persistent<T, MODE, 128, BNv, 4, 2, true, true, EXv, false, false, 0, false, false, 0, EPSv>), dim3(pg), dim3(256), plds, stream, p)
#define LPX(BNv, STv, BEv, EXv)                                                                                              \
    do {                                                                                                                     \
      if constexpr (sizeof(T) == 4) {                                                                                        \
        if (psb && p.x_ps) { if constexpr (MODE == MODE_DGRAD && !(EXv)) SIMCLR_LAUNCH((conv_igemm_persistent<T, MODE, 128, BNv, 4, 2, STv, BEv, false, false, false, 3, false, true, 0, 0, true>), dim3(pg), dim3(256), plds, stream, p); \
                             else g_igemm_fail = 1; } \
        else if (psb) { if constexpr (MODE == MODE_DGRAD) SIMCLR_LAUNCH((conv_igemm_persistent<T, MODE, 128, BNv, 4, 2, STv, BEv, EXv, false, false, 3, false, true>), dim3(pg), dim3(256), plds, stream, p); \
                   else if constexpr (!(BEv) && !(EXv)) SIMCLR_LAUNCH((conv_igemm_persistent<T, MODE, 128, BNv, 4, 2, STv, false, false, false, false, 13, false, true>), dim3(pg), dim3(256), plds, stream, p); } \
        else if (p.split == 3) SIMCLR_LAUNCH((conv_igemm_persistent<T, MODE, 128, BNv, 4, 2, STv, BEv, EXv, false, false, 3>), dim3(pg), dim3(256), plds, stream, p); \
        else if (p.split == 6) SIMCLR_LAUNCH((conv_igemm_persistent<T, MODE, 128, BNv, 4, 2, STv, BEv, EXv, false, false, 6>), dim3(pg), dim3(256), plds, stream, p); \
        else SIMCLR_LAUNCH((conv_igemm_persistent<T, MODE, 128, BNv, 4, 2, STv, BEv, EXv>), dim3(pg), dim3(256), plds, stream, p); \
      } else if ((BNv) == 128 && p.rem_parts >= 2) {                                                                         \
        SIMCLR_LAUNCH((conv_igemm_persistent<T, MODE, 128, 128, 4, 2, STv, BEv, EXv, false, false, 0, true>), dim3(pg), dim3(256), plds, stream, p); \
      } else {                                                                                                               \
        SIMCLR_LAUNCH((conv_igemm_persistent<T, MODE, 128, BNv, 4, 2, STv, BEv, EXv>), dim3(pg), dim3(256), plds, stream, p); \
      }                                                                                                                      \
    } while (0)
#define LP(BNv, STv, BEv) LPX(BNv, STv, BEv, false)
    if (p.fapply) {
      // the two option sets every fused bottleneck tail uses get instantiations with the options compiled in (FAS);
      // SIMCLR_FAPPLY_SPECIAL=0 (read per launch: A/B runs, bitwise test) keeps the generic epilogue
      int fas = 0;
      if (p.bn_x && p.bn_mode && p.bn_mask && (p.N & 31) == 0) fas = p.bn_mean ? 2 : 1;
      { const char* e = getenv("SIMCLR_FAPPLY_SPECIAL"); if (e && atoi(e) == 0) fas = 0; }
      if constexpr (sizeof(T) == 4 && MODE == MODE_FWD) {
        // fp32 storage: generic options (FAS = 0); three fp16-piece terms with the pre-split weights, six bf16 terms, or exact fp32
#define LF32(BNv)                                                                                                              \
        do {                                                                                                                   \
          if (psb && p.split == 13) SIMCLR_LAUNCH((conv_igemm_persistent<float, MODE_FWD, 128, BNv, 4, 2, false, false, false, false, true, 13, false, true>), dim3(pg), dim3(256), plds, stream, p); \
          else if (p.split == 6) SIMCLR_LAUNCH((conv_igemm_persistent<float, MODE_FWD, 128, BNv, 4, 2, false, false, false, false, true, 6>), dim3(pg), dim3(256), plds, stream, p); \
          else if (p.split == 3) SIMCLR_LAUNCH((conv_igemm_persistent<float, MODE_FWD, 128, BNv, 4, 2, false, false, false, false, true, 3>), dim3(pg), dim3(256), plds, stream, p); \
          else SIMCLR_LAUNCH((conv_igemm_persistent<float, MODE_FWD, 128, BNv, 4, 2, false, false, false, false, true>), dim3(pg), dim3(256), plds, stream, p); \
        } while (0)
        if (BN == 64) LF32(64); else LF32(128);
#undef LF32
        return;
      }
#define LF(BNv)                                                                                                                \
      do {                                                                                                                     \
        if (fas == 1) SIMCLR_LAUNCH((conv_igemm_persistent<uint16_t, MODE_FWD, 128, BNv, 4, 2, false, false, false, false, true, 0, false, false, 1>), dim3(pg), dim3(256), plds, stream, p); \
        else if (fas == 2) SIMCLR_LAUNCH((conv_igemm_persistent<uint16_t, MODE_FWD, 128, BNv, 4, 2, false, false, false, false, true, 0, false, false, 2>), dim3(pg), dim3(256), plds, stream, p); \
        else SIMCLR_LAUNCH((conv_igemm_persistent<uint16_t, MODE_FWD, 128, BNv, 4, 2, false, false, false, false, true>), dim3(pg), dim3(256), plds, stream, p); \
      } while (0)
      if (BN == 64) LF(64); else LF(128);
#undef LF
      return;
    }
    // 3x3 stride-1: halo-window operand path (one window load per 128-byte channel chunk instead of nine gathers)
    bool win_now = win3;
    if constexpr (sizeof(T) == 4) {
      // fp32 storage: forward = three fp16-piece terms with the pre-split weights; data gradient = three bf16-piece terms with both
      // operands pre-split, plain or with the fused BatchNorm-backward reduce (the launches a ResNet step makes); everything else gathers
      if (MODE == MODE_FWD) win_now = win3 && psb && p.split == 13 && !p.bn_mode;
      else win_now = win3 && psb && p.split == 3 && p.x_ps && (p.bn_mode != 0) == st;
    }
    if (win_now) {
      const int rows = ((128 + 2 * (p.IW + 1)) + 31) / 32 * 32;
      p.win_j = rows / 32;
      p.win_bytes = (sizeof(T) == 4 || rows * 128 > 128 * BN * 2) ? rows * 128 : 128 * BN * 2;     // (bf16: doubles as the C staging tile)
      const size_t wlds = (size_t)p.win_bytes + 2 * BN * 128 + 5 * BN * sizeof(float) + 128 * sizeof(long long) + 128;
      if constexpr (sizeof(T) == 4) {
#define LW32(BNv, STv, BEv)                                                                                                   \
        do {                                                                                                                   \
          if constexpr (MODE == MODE_FWD) { if constexpr (!(BEv)) SIMCLR_LAUNCH((conv_igemm_persistent<float, MODE_FWD, 128, BNv, 4, 2, STv, false, false, true, false, 13, false, true>), dim3(pg), dim3(256), wlds, stream, p); } \
          else SIMCLR_LAUNCH((conv_igemm_persistent<float, MODE_DGRAD, 128, BNv, 4, 2, STv, BEv, false, true, false, 3, false, true, 0, 0, true>), dim3(pg), dim3(256), wlds, stream, p); \
        } while (0)
        if (p.bn_mode) { if (BN == 64) LW32(64, true, true); else LW32(128, true, true); }
        else if (BN == 64) { if (st) LW32(64, true, false); else LW32(64, false, false); }
        else { if (st) LW32(128, true, false); else LW32(128, false, false); }
#undef LW32
        return;
      } else {
#define LW(BNv, STv, BEv)                                                                                                     \
      do {                                                                                                                     \
        if ((BNv) == 128 && p.rem_parts >= 2) SIMCLR_LAUNCH((conv_igemm_persistent<uint16_t, MODE, 128, 128, 4, 2, STv, BEv, false, true, false, 0, true>), dim3(pg), dim3(256), wlds, stream, p); \
        else SIMCLR_LAUNCH((conv_igemm_persistent<uint16_t, MODE, 128, BNv, 4, 2, STv, BEv, false, true>), dim3(pg), dim3(256), wlds, stream, p); \
      } while (0)
      if (p.bn_mode && eps == 1 && p.rem_parts < 2) {
        if (BN == 64) SIMCLR_LAUNCH((conv_igemm_persistent<uint16_t, MODE, 128, 64, 4, 2, true, true, false, true, false, 0, false, false, 0, 1>), dim3(pg), dim3(256), wlds, stream, p);
        else SIMCLR_LAUNCH((conv_igemm_persistent<uint16_t, MODE, 128, 128, 4, 2, true, true, false, true, false, 0, false, false, 0, 1>), dim3(pg), dim3(256), wlds, stream, p);
      }
      else if (p.bn_mode) { if (BN == 64) LW(64, true, true); else LW(128, true, true); }
      else if (BN == 64) { if (st) LW(64, true, false); else LW(64, false, false); }
      else { if (st) LW(128, true, false); else LW(128, false, false); }
#undef LW
      return;
      }
    }
    if (!p.bn_mode && p.x2) {  // K-extended dgrad, plain epilogue (the conv input is not a BatchNorm output: block entry)
      if (BN == 64) LPX(64, false, false, true); else LPX(128, false, false, true);
    } else if (p.bn_mode && p.x2) {   // K-extended dgrad (folded BatchNorm backward of the consumer's output) + fused BN reduce
      if (eps == 1 && p.rem_parts < 2) { if constexpr (sizeof(T) == 2) { if (BN == 64) LPE(64, true, 1); else LPE(128, true, 1); } }
      else if (BN == 64) LPX(64, true, true, true); else LPX(128, true, true, true);
    } else if (p.bn_mode) {    // dgrad with fused BN-backward reduce (statistics = sum dm, sum dm*x^)
      if ((eps == 1 || eps >= 3) && p.rem_parts < 2) {
        if constexpr (sizeof(T) == 2) {
          if (eps == 1) { if (BN == 64) LPE(64, false, 1); else LPE(128, false, 1); }
          else if (eps == 3) { if (BN == 64) LPE(64, false, 3); else LPE(128, false, 3); }
          else { if (BN == 64) LPE(64, false, 4); else LPE(128, false, 4); }
        }
      }
      else if (BN == 64) LP(64, true, true); else LP(128, true, true);
    } else if (BN == 64) { if (st) LP(64, true, false); else LP(64, false, false); }
    else { if (st) LP(128, true, false); else LP(128, false, false); }
#undef LP
#undef LPX
#undef LPE
    return;
  }
  if (p.x_ps || (p.w_ps && p.ntaps > 0) || p.accumulate == 2) { g_igemm_fail = 1; return; }
#define L(BNv, STv)                                                                                    \
  do {                                                                                                 \
    if (no_glds) SIMCLR_LAUNCH((conv_igemm<T, MODE, BNv, STv, false>), dim3(grid), dim3(256), lds, stream, p); \
    else SIMCLR_LAUNCH((conv_igemm<T, MODE, BNv, STv, true>), dim3(grid), dim3(256), lds, stream, p);          \
  } while (0)
  if (BN == 64) { if (st) L(64, true); else L(64, false); }
  else { if (st) L(128, true); else L(128, false); }
#undef L
}

// Forward and stride-1 dgrad: one class with every tap.  Strided dgrad: one launch per output
// parity class with only the taps that can contribute (1x1 s2: a single class has work, the
// other three are zero-filled unless accumulating).
template <typename T, int MODE>
int launch_igemm(const ConvP& p0, hipStream_t stream) {
  ConvP p = p0;
  if (MODE == MODE_FWD || p.stride == 1) {
    p.cs = 1; p.py = 0; p.px = 0; p.cls_h = p.OH; p.cls_w = p.OW;
    p.ntaps = p.KH * p.KW;
    p.tap_w = p.dy_w = p.dx_w = 0;
    for (int t = 0; t < p.ntaps; ++t) {
      p.taps[t] = t;
      const int ty = t / p.KW, tx = t % p.KW;
      // fwd: row base = (oy*s-pad, ox*s-pad), tap adds (+ty,+tx); dgrad s1: base = (oy, ox), tap adds (pad-ty, pad-tx)
      const int dy = (MODE == MODE_FWD) ? ty : p.pad - ty, dx = (MODE == MODE_FWD) ? tx : p.pad - tx;
      p.tap_w |= (unsigned long long)t << (4 * t);
      p.dy_w |= (unsigned long long)(dy + 8) << (4 * t);
      p.dx_w |= (unsigned long long)(dx + 8) << (4 * t);
    }
    p.M = p.V * p.OH * p.OW;
    launch_igemm_one<T, MODE>(p, stream);
    return 0;
  }
  const int S = p.stride;
  for (int py = 0; py < S; ++py)
    for (int px = 0; px < S; ++px) {
      ConvP q = p;
      q.cs = S; q.py = py; q.px = px;
      q.cls_h = (p.OH - py + S - 1) / S;
      q.cls_w = (p.OW - px + S - 1) / S;
      q.ntaps = 0;
      q.tap_w = q.dy_w = q.dx_w = 0;
      for (int ty = 0; ty < p.KH; ++ty)
        for (int tx = 0; tx < p.KW; ++tx)
          if ((py + p.pad - ty) % S == 0 && (px + p.pad - tx) % S == 0 && q.ntaps < 9) {
            // class rows have base pixel (a, b); this tap reads dy pixel (a + (py+pad-ty)/S, b + (px+pad-tx)/S)
            q.tap_w |= (unsigned long long)(ty * p.KW + tx) << (4 * q.ntaps);
            q.dy_w |= (unsigned long long)((py + p.pad - ty) / S + 8) << (4 * q.ntaps);
            q.dx_w |= (unsigned long long)((px + p.pad - tx) / S + 8) << (4 * q.ntaps);
            q.taps[q.ntaps++] = ty * p.KW + tx;
          }
      q.M = p.V * q.cls_h * q.cls_w;
      // nothing to add (accumulate == 3: nothing to store, the class stays untouched; accumulate == 2: only a class of even pixels
      // holds earlier data -- an empty class of other pixels has to be zero-filled like a plain store)
      if (q.ntaps == 0 && p.accumulate == 2 && ((py | px) & 1)) q.accumulate = 0;
      else if (q.ntaps == 0 && p.accumulate) continue;
      if (p.accumulate == 3) q.accumulate = 0;      // sparse store: the classes with taps are plain stores
      if (q.ntaps == 0) q.x_ps = 0;                 // zero fill: the gathered tensor is not read at all
      launch_igemm_one<T, MODE>(q, stream);
    }
  return 0;
}

}  // namespace

extern "C" {

// Matrix arithmetic of the fp32 (parity) convolution / dense kernels: number of bf16 terms per product, separately for
// the forward GEMM and for the two backward GEMMs (dgrad, wgrad).  0 = exact fp32 MFMA (v_mfma_f32_16x16x4_f32, the
// default), 3 = hi*hi + hi*lo + lo*hi (~2^-17 relative per product), 6 = all terms of weight >= 2^-18 (fp32 level).
// Storage stays fp32; bf16 launches are unaffected.  Process-wide; returns 1 on a bad argument.
int simclr_set_f32_matmul(int fwd_terms, int bwd_terms) {
  SIMCLR_CHECK_ARG((fwd_terms == 0 || fwd_terms == 3 || fwd_terms == 6 || fwd_terms == 13) && (bwd_terms == 0 || bwd_terms == 3 || bwd_terms == 6),
                   "set_f32_matmul: terms must be 0, 3 or 6 (forward also 13 = three split-fp16 terms) (got %d, %d)", fwd_terms, bwd_terms);
  g_f32_terms_fwd = fwd_terms;
  g_f32_terms_bwd = bwd_terms;
  return 0;
}
int simclr_get_f32_matmul(int which) { return which == 0 ? g_f32_terms_fwd : g_f32_terms_bwd; }
// Test hook: into how many parts the most recent forward / dgrad launch of this process split each left-over tile
// (0 = the launch ran whole tiles only).  See "split tail" above.
int simclr_conv2d_last_split_parts(void) { return g_last_split_parts; }
const char* simclr_conv2d_last_instantiation(void) { return g_last_inst; }
int simclr_conv2d_split_tail_timeouts(unsigned* host_total) { return split_tail_timeouts(host_total); }
// Test hook: 1 if the most recent fp32 data-gradient launch read pre-split weights (three split-bf16 terms, see presplit_weights).
int simclr_conv2d_last_presplit(void) { return g_last_presplit; }

// Number of partial-statistics slots that makes the statistics of simclr_conv2d_fwd / simclr_conv2d_dgrad_bn
// deterministic for an output of M rows x C channels (one slot per persistent workgroup of an N-tile).
int simclr_conv2d_stats_slots(long long M, int C) {
  int bn, nt, pg;
  igemm_persistent_grid(M, C, &bn, &nt, &pg);
  return pg / nt;
}
// same for simclr_stem_conv_fwd (M = V*OH*OW output pixels)
int simclr_stem_stats_slots(long long M) { return (int)min((M + 127) / 128, 2048ll); }

// Forward conv / dense.  x [V,IH,IW,Cin] (T), w_t [Cout][KH*KW*Cin] (T), y [V,OH,OW,Cout] (T).
// stats (nullable): float [nslot][2][Cout], must be zeroed by the caller; receives
// per-channel partial (sum, sum of squares) of the fp32 results (BatchNorm statistics).  With
// nslot >= simclr_conv2d_stats_slots(V*OH*OW, Cout) every workgroup stores into its own slot (no float
// atomics): the statistics are bit-identical from run to run, like the reference's (tf2/resnet.py:54-60).
int simclr_conv2d_fwd(const void* x, const void* w_t, void* y, float* stats, int nslot, int V,
                      int IH, int IW, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride,
                      int pad, int dtype, hipStream_t stream) {
  const int terms = terms_of(&dtype, true);
  SIMCLR_CHECK_ARG(terms >= 0, "conv2d_fwd: bad matrix-arithmetic field in dtype (SIMCLR_FMT_TERMS)");
  const bool w_ps = take_ps_w(&dtype);
  SIMCLR_CHECK_ARG(!w_ps || ((dtype & 0xff) == SIMCLR_DT_F32 && terms == 13), "conv2d_fwd: pre-split weights (SIMCLR_FMT_PS_W) need fp32 storage and terms 13");
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "conv2d_fwd: bad dtype %d", dtype);
  SIMCLR_CHECK_ARG(Cin % (8 * epc) == 0, "conv2d_fwd: Cin=%d must be a multiple of %d", Cin, 8 * epc);
  SIMCLR_CHECK_ARG(Cout % 4 == 0, "conv2d_fwd: Cout=%d must be a multiple of 4", Cout);
  SIMCLR_CHECK_ARG(V > 0 && OH > 0 && OW > 0 && stride >= 1, "conv2d_fwd: bad geometry");
  SIMCLR_CHECK_ARG((long long)V * OH * OW < (1ll << 31), "conv2d_fwd: M overflows int32");
  SIMCLR_CHECK_ARG(!stats || nslot > 0, "conv2d_fwd: nslot must be > 0 with stats");
  SIMCLR_CHECK_ARG(KH * KW <= 9, "conv2d_fwd: at most 9 taps (got %dx%d)", KH, KW);
  SIMCLR_CHECK_ARG(y || (stats && dtype == SIMCLR_DT_BF16), "conv2d_fwd: y == NULL (statistics-only pass) needs stats and bf16");
  ConvP p = {};
  p.w_ps = w_ps ? 1 : 0;
  p.split = terms;
  p.zero = zero_page();
  SIMCLR_CHECK_ARG(p.zero != nullptr, "conv2d: zero page symbol not found");
  p.x = x; p.w = w_t; p.y = y; p.stats = stats; p.nslot = nslot;
  p.V = V; p.IH = IH; p.IW = IW; p.IC = Cin; p.OH = OH; p.OW = OW; p.N = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.pixpitch = Cin;
  p.M = V * OH * OW; p.K = KH * KW * Cin;
  g_igemm_fail = 0;
  if (dtype == SIMCLR_DT_BF16) launch_igemm<uint16_t, MODE_FWD>(p, stream);
  else launch_igemm<float, MODE_FWD>(p, stream);
  SIMCLR_CHECK_ARG(!g_igemm_fail, "conv2d_fwd: no kernel reads pre-split weights on this launch path");
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// simclr_conv2d_fwd for SIMCLR_DT_F32 with PIVOTED statistics: pivot [Cout] (out) receives the convolution output at one
// interior pixel and the slots receive sum(y - pivot), sum((y - pivot)^2) per channel; simclr_bn_reduce_slots_pivoted
// converts them into the raw fp64 moments the BatchNorm finalize expects.  Why: BatchNorm variance from raw fp32 moments loses
// (mean / sigma)^2 * 2^-24 (tf2/resnet.py:50-60 computes the moments in one fp32 pass too, but on TPU/GPU reductions in a
// tree; a channel with |mean| >> sigma is where the two drift apart).  A launch that does not take the persistent fp32
// kernel writes zeros into pivot and raw moments into the slots (the conversion is then the identity).
int simclr_conv2d_fwd_pivoted(const void* x, const void* w_t, void* y, float* stats, int nslot, float* pivot, int V,
                              int IH, int IW, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride,
                              int pad, int dtype, hipStream_t stream) {
  const int terms = terms_of(&dtype, true);
  SIMCLR_CHECK_ARG(terms >= 0, "conv2d_fwd_pivoted: bad matrix-arithmetic field in dtype (SIMCLR_FMT_TERMS)");
  const bool w_ps = take_ps_w(&dtype);
  SIMCLR_CHECK_ARG(!w_ps || ((dtype & 0xff) == SIMCLR_DT_F32 && terms == 13), "conv2d_fwd_pivoted: pre-split weights (SIMCLR_FMT_PS_W) need fp32 storage and terms 13");
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_F32, "conv2d_fwd_pivoted: fp32 only (dtype %d)", dtype);
  SIMCLR_CHECK_ARG(Cin % 32 == 0, "conv2d_fwd_pivoted: Cin=%d must be a multiple of 32", Cin);
  SIMCLR_CHECK_ARG(Cout % 4 == 0, "conv2d_fwd_pivoted: Cout=%d must be a multiple of 4", Cout);
  SIMCLR_CHECK_ARG(V > 0 && OH > 0 && OW > 0 && stride >= 1, "conv2d_fwd_pivoted: bad geometry");
  SIMCLR_CHECK_ARG((long long)V * OH * OW < (1ll << 31), "conv2d_fwd_pivoted: M overflows int32");
  SIMCLR_CHECK_ARG(x && w_t && y && stats && pivot && nslot > 0, "conv2d_fwd_pivoted: null argument");
  SIMCLR_CHECK_ARG(KH * KW <= 9, "conv2d_fwd_pivoted: at most 9 taps (got %dx%d)", KH, KW);
  ConvP p = {};
  p.w_ps = w_ps ? 1 : 0;
  p.split = terms;
  p.zero = zero_page();
  SIMCLR_CHECK_ARG(p.zero != nullptr, "conv2d: zero page symbol not found");
  p.x = x; p.w = w_t; p.y = y; p.stats = stats; p.nslot = nslot;
  p.V = V; p.IH = IH; p.IW = IW; p.IC = Cin; p.OH = OH; p.OW = OW; p.N = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.pixpitch = Cin;
  p.M = V * OH * OW; p.K = KH * KW * Cin;
  // the persistent fp32 kernel is the one that knows about the pivot (launch_igemm_one: n_tiles <= 64, no A/B switches)
  const bool persistent = Cout <= 64 * 64 && !getenv("SIMCLR_NO_GLDS") && !getenv("SIMCLR_NO_PERSISTENT");
  if (persistent) {
    if (!dry_run())
      hipLaunchKernelGGL(conv_pivot_row, dim3(Cout), dim3(256), 0, stream, (const float*)x, (const float*)w_t, pivot,
                         IH, IW, Cin, Cin, OH, OW, Cout, KH, KW, stride, pad, p.w_ps);
    p.pivot = pivot;
  } else if (!dry_run()) {
    if (hipMemsetAsync(pivot, 0, (size_t)Cout * sizeof(float), stream) != hipSuccess) { simclr_set_error("conv2d_fwd_pivoted: memset failed"); return 2; }
  }
  g_igemm_fail = 0;
  launch_igemm<float, MODE_FWD>(p, stream);
  SIMCLR_CHECK_ARG(!g_igemm_fail, "conv2d_fwd_pivoted: no kernel reads pre-split weights on this launch path");
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// Forward conv with the BatchNorm apply of its consumer fused into the epilogue (bf16, and since round 6 fp32 storage):
//   y = act(T(conv(x)) * scale + shift + r),  r = res or res * rscale + rshift (a projection shortcut's own BatchNorm),
//   relu_bits[i] bit e = (y[EPC i + e] > 0), one byte per 16-byte chunk of y (EPC = 8 bf16 / 4 fp32 elements)
// -- bitwise what simclr_conv2d_fwd followed by simclr_bn_apply produces, without the convolution output ever reaching
// memory.  scale / shift [Cout] come from the statistics of a first pass (simclr_conv2d_fwd with y == NULL: statistics
// only) through simclr_bn_finalize.  res (nullable): residual [V,OH,OW,Cout]; relu_bits (nullable): uint8 [V*OH*OW*Cout/8].
// tf2/resnet.py:470-487 (conv3 -> bn3 -> + shortcut -> relu of the bottleneck block).
int simclr_conv2d_fwd_bn_apply(const void* x, const void* w_t, void* y, const float* scale, const float* shift,
                               const void* res, const float* rscale, const float* rshift, int relu,
                               unsigned char* relu_bits, int V, int IH, int IW, int Cin,
                               int OH, int OW, int Cout, int KH, int KW, int stride, int pad, int dtype,
                               hipStream_t stream) {
  const int terms = terms_of(&dtype, true);
  SIMCLR_CHECK_ARG(terms >= 0, "conv2d_fwd_bn_apply: bad matrix-arithmetic field in dtype (SIMCLR_FMT_TERMS)");
  const bool w_ps = take_ps_w(&dtype);
  SIMCLR_CHECK_ARG(!w_ps || ((dtype & 0xff) == SIMCLR_DT_F32 && terms == 13), "conv2d_fwd_bn_apply: pre-split weights (SIMCLR_FMT_PS_W) need fp32 storage and terms 13");
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "conv2d_fwd_bn_apply: bad dtype %d", dtype);
  SIMCLR_CHECK_ARG(Cin % (dtype == SIMCLR_DT_BF16 ? 64 : 32) == 0, "conv2d_fwd_bn_apply: Cin=%d must be a multiple of %d", Cin, dtype == SIMCLR_DT_BF16 ? 64 : 32);
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || (Cout <= 64 * 64 && !getenv("SIMCLR_NO_GLDS") && !getenv("SIMCLR_NO_PERSISTENT")),
                   "conv2d_fwd_bn_apply: the fp32 epilogue lives in the persistent kernel only");
  SIMCLR_CHECK_ARG(Cout % 8 == 0, "conv2d_fwd_bn_apply: Cout=%d must be a multiple of 8", Cout);
  SIMCLR_CHECK_ARG(x && w_t && y && scale && shift, "conv2d_fwd_bn_apply: null argument");
  SIMCLR_CHECK_ARG((rscale == nullptr) == (rshift == nullptr) && (!rscale || res), "conv2d_fwd_bn_apply: rscale / rshift come together and need res");
  SIMCLR_CHECK_ARG(V > 0 && OH > 0 && OW > 0 && stride >= 1, "conv2d_fwd_bn_apply: bad geometry");
  SIMCLR_CHECK_ARG((long long)V * OH * OW < (1ll << 31), "conv2d_fwd_bn_apply: M overflows int32");
  SIMCLR_CHECK_ARG(KH * KW <= 9, "conv2d_fwd_bn_apply: at most 9 taps (got %dx%d)", KH, KW);
  ConvP p = {};
  p.w_ps = w_ps ? 1 : 0;
  (void)terms;
  p.zero = zero_page();
  SIMCLR_CHECK_ARG(p.zero != nullptr, "conv2d: zero page symbol not found");
  p.x = x; p.w = w_t; p.y = y;
  p.V = V; p.IH = IH; p.IW = IW; p.IC = Cin; p.OH = OH; p.OW = OW; p.N = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.pixpitch = Cin;
  p.M = V * OH * OW; p.K = KH * KW * Cin;
  p.fapply = 1; p.bn_scale = scale; p.bn_shift = shift; p.bn_x = res; p.bn_mask = relu_bits; p.bn_mode = relu ? 1 : 0;
  p.bn_mean = rscale; p.bn_rstd = rshift;
  (void)terms;
  g_igemm_fail = 0;
  if (dtype == SIMCLR_DT_F32) { p.split = terms; launch_igemm<float, MODE_FWD>(p, stream); }
  else launch_igemm<uint16_t, MODE_FWD>(p, stream);
  SIMCLR_CHECK_ARG(!g_igemm_fail, "conv2d_fwd_bn_apply: no kernel reads pre-split weights on this launch path");
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// Data gradient.  dy [V,OH,OW,Cout] (T), w_d [Cin][KH*KW*Cout] (T), dx [V,IH,IW,Cin] (T).
// accumulate != 0: dx += result.
int simclr_conv2d_dgrad(const void* dy, const void* w_d, void* dx, int accumulate, int V, int IH,
                        int IW, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad,
                        int dtype, hipStream_t stream) {
  const int terms = terms_of(&dtype, false);
  SIMCLR_CHECK_ARG(terms >= 0, "conv2d_dgrad: bad matrix-arithmetic field in dtype (SIMCLR_FMT_TERMS)");
  const bool w_ps = take_ps_w(&dtype);
  SIMCLR_CHECK_ARG(!w_ps || ((dtype & 0xff) == SIMCLR_DT_F32 && terms == 3), "conv2d_dgrad: pre-split weights (SIMCLR_FMT_PS_W) need fp32 storage and terms 3");
  // dtype | SIMCLR_FMT_PS_IN (fp32, three bf16 backward terms, Cout a multiple of 32): dy is in the pre-split block format (common.h)
  const bool dy_ps = (dtype & SIMCLR_FMT_PS_IN) != 0;
  dtype &= 0xff;
  SIMCLR_CHECK_ARG(!dy_ps || (dtype == SIMCLR_DT_F32 && terms == 3 && Cout % 32 == 0),
                   "conv2d_dgrad: a pre-split dy needs fp32 storage, three backward terms (simclr_set_f32_matmul) and Cout %% 32 == 0");
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "conv2d_dgrad: bad dtype %d", dtype);
  SIMCLR_CHECK_ARG(Cout % (8 * epc) == 0, "conv2d_dgrad: Cout=%d must be a multiple of %d", Cout, 8 * epc);
  SIMCLR_CHECK_ARG(Cin % 4 == 0, "conv2d_dgrad: Cin=%d must be a multiple of 4", Cin);
  SIMCLR_CHECK_ARG((long long)V * IH * IW < (1ll << 31), "conv2d_dgrad: M overflows int32");
  SIMCLR_CHECK_ARG(KH * KW <= 9 && stride <= 2, "conv2d_dgrad: at most 9 taps and stride <= 2");
  // accumulate: 0 store | 1 dx += result | 2 (fp32) dx += result where dx holds earlier data only at even (row, column) pixels -- the
  // output of a call with 3 | 3 (stride 2) store the pixel classes that receive a tap and leave the others UNTOUCHED (no zero fill)
  SIMCLR_CHECK_ARG(accumulate >= 0 && accumulate <= 3, "conv2d_dgrad: accumulate must be 0 ... 3 (got %d)", accumulate);
  SIMCLR_CHECK_ARG(accumulate != 2 || dtype == SIMCLR_DT_F32, "conv2d_dgrad: accumulate = 2 (sparse earlier data) is an fp32 path");
  SIMCLR_CHECK_ARG(accumulate != 3 || stride == 2, "conv2d_dgrad: accumulate = 3 (sparse store) needs stride 2");
  ConvP p = {};
  p.w_ps = w_ps ? 1 : 0;
  p.split = terms;
  p.zero = zero_page();
  SIMCLR_CHECK_ARG(p.zero != nullptr, "conv2d: zero page symbol not found");
  p.x = dy; p.w = w_d; p.y = dx; p.stats = nullptr; p.nslot = 1; p.accumulate = accumulate;
  p.V = V; p.IH = OH; p.IW = OW; p.IC = Cout; p.OH = IH; p.OW = IW; p.N = Cin;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.pixpitch = Cout;
  p.M = V * IH * IW; p.K = KH * KW * Cout;
  p.x_ps = dy_ps ? 1 : 0;
  g_igemm_fail = 0;
  if (dtype == SIMCLR_DT_BF16) launch_igemm<uint16_t, MODE_DGRAD>(p, stream);
  else launch_igemm<float, MODE_DGRAD>(p, stream);
  SIMCLR_CHECK_ARG(!g_igemm_fail, "conv2d_dgrad: no kernel for a pre-split dy on this launch path (pre-split weights unavailable?)");
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// Data gradient with the BatchNorm-backward reduce of the PRODUCER layer fused into the epilogue.
// dx here is the gradient wrt the output of a BatchNormRelu whose raw input is bn_x ([V,IH,IW,Cin]):
// the kernel stores dm = dx * relu_mask and adds per-channel (sum dm, sum dm*x^) into
// stats[nslot][2][Cin] (zeroed by the caller).  mask_mode 1: mask = bn_mask > 0 (bn_mask = the tensor
// after the ReLU, e.g. the residual block output); 2: mask = bn_x*scale+shift > 0.
int simclr_conv2d_dgrad_bn(const void* dy, const void* w_d, void* dx, int accumulate, const void* bn_x,
                           const void* bn_mask, const float* bn_scale, const float* bn_shift,
                           const float* bn_mean, const float* bn_rstd, int mask_mode, float* stats, int nslot,
                           int V, int IH, int IW, int Cin, int OH, int OW, int Cout, int KH, int KW,
                           int stride, int pad, int dtype, hipStream_t stream) {
  const int terms = terms_of(&dtype, false);
  SIMCLR_CHECK_ARG(terms >= 0, "conv2d_dgrad_bn: bad matrix-arithmetic field in dtype (SIMCLR_FMT_TERMS)");
  const bool w_ps = take_ps_w(&dtype);
  SIMCLR_CHECK_ARG(!w_ps || ((dtype & 0xff) == SIMCLR_DT_F32 && terms == 3), "conv2d_dgrad_bn: pre-split weights (SIMCLR_FMT_PS_W) need fp32 storage and terms 3");
  const bool dy_ps = (dtype & SIMCLR_FMT_PS_IN) != 0;          // see simclr_conv2d_dgrad
  dtype &= 0xff;
  SIMCLR_CHECK_ARG(!dy_ps || (dtype == SIMCLR_DT_F32 && terms == 3 && Cout % 32 == 0),
                   "conv2d_dgrad_bn: a pre-split dy needs fp32 storage, three backward terms (simclr_set_f32_matmul) and Cout %% 32 == 0");
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "conv2d_dgrad_bn: bad dtype %d", dtype);
  SIMCLR_CHECK_ARG(Cout % (8 * epc) == 0, "conv2d_dgrad_bn: Cout=%d must be a multiple of %d", Cout, 8 * epc);
  SIMCLR_CHECK_ARG(Cin % 4 == 0, "conv2d_dgrad_bn: Cin=%d must be a multiple of 4", Cin);
  SIMCLR_CHECK_ARG((long long)V * IH * IW < (1ll << 31), "conv2d_dgrad_bn: M overflows int32");
  SIMCLR_CHECK_ARG(KH * KW <= 9 && stride == 1, "conv2d_dgrad_bn: stride-1 convolutions with <= 9 taps only");
  SIMCLR_CHECK_ARG(accumulate >= 0 && accumulate <= 2 && (accumulate != 2 || dtype == SIMCLR_DT_F32),
                   "conv2d_dgrad_bn: accumulate must be 0, 1 or (fp32) 2 = earlier data at even pixels only, see simclr_conv2d_dgrad (got %d)", accumulate);
  SIMCLR_CHECK_ARG(mask_mode >= 1 && mask_mode <= 4, "conv2d_dgrad_bn: mask_mode must be 1, 2, 3 or 4");
  SIMCLR_CHECK_ARG(stats && nslot > 0 && (mask_mode == 4 || (bn_x && bn_mean && bn_rstd)), "conv2d_dgrad_bn: null BN argument");
  SIMCLR_CHECK_ARG(mask_mode == 2 || bn_mask, "conv2d_dgrad_bn: mask_mode 1 / 3 / 4 need bn_mask");
  SIMCLR_CHECK_ARG(mask_mode != 2 || (bn_scale && bn_shift), "conv2d_dgrad_bn: mask_mode 2 needs scale/shift");
  ConvP p = {};
  p.w_ps = w_ps ? 1 : 0;
  p.split = terms;
  p.zero = zero_page();
  SIMCLR_CHECK_ARG(p.zero != nullptr, "conv2d: zero page symbol not found");
  p.x = dy; p.w = w_d; p.y = dx; p.stats = stats; p.nslot = nslot; p.accumulate = accumulate;
  p.V = V; p.IH = OH; p.IW = OW; p.IC = Cout; p.OH = IH; p.OW = IW; p.N = Cin;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.pixpitch = Cout;
  p.M = V * IH * IW; p.K = KH * KW * Cout;
  p.bn_x = bn_x; p.bn_mask = bn_mask; p.bn_scale = bn_scale; p.bn_shift = bn_shift;
  p.bn_mean = bn_mean; p.bn_rstd = bn_rstd; p.bn_mode = mask_mode;
  p.x_ps = dy_ps ? 1 : 0;
  g_igemm_fail = 0;
  if (dtype == SIMCLR_DT_BF16) launch_igemm<uint16_t, MODE_DGRAD>(p, stream);
  else launch_igemm<float, MODE_DGRAD>(p, stream);
  SIMCLR_CHECK_ARG(!g_igemm_fail, "conv2d_dgrad_bn: no kernel for a pre-split dy on this launch path (pre-split weights unavailable?)");
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// 1x1 stride-1 data gradient whose upstream gradient passes through a BatchNorm backward that is FOLDED into this
// convolution by linearity (the BN of this conv's own output, e.g. the residual tail BN3 after the expand conv3):
//   dh = a*dm + b*c + d  per output channel (BN backward: a = scale, b = -scale*k2*rstd, d = scale*(k2*mean*rstd - k1)),
//   c = h W  (this conv's forward)  =>  d(h) = dm (a*W)^T + h (W diag(b) W^T) + W d.
// dm [M, Cout] (masked gradient wrt the BN output), h [M, Cin] (this conv's forward input), w_ext [Cin][Cout + Cin] =
// [a*W | Q^T] rows (T), bias [Cin] fp32 = W d.  The rest as simclr_conv2d_dgrad_bn (fused reduce of the PRODUCER BN of h).
int simclr_conv2d_dgrad_bn_ext(const void* dm, const void* h, const void* w_ext, const float* bias, void* dx,
                               int accumulate, const void* bn_x, const void* bn_mask, const float* bn_scale,
                               const float* bn_shift, const float* bn_mean, const float* bn_rstd, int mask_mode,
                               float* stats, int nslot, int V, int H, int W, int Cin, int Cout, int dtype,
                               hipStream_t stream) {
  const int terms = terms_of(&dtype, false);
  SIMCLR_CHECK_ARG(terms >= 0, "conv2d_dgrad_bn_ext: bad matrix-arithmetic field in dtype (SIMCLR_FMT_TERMS)");
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "conv2d_dgrad_bn_ext: bad dtype %d", dtype);
  SIMCLR_CHECK_ARG(Cout % (8 * epc) == 0 && Cin % (8 * epc) == 0, "conv2d_dgrad_bn_ext: Cin=%d / Cout=%d must be multiples of %d", Cin, Cout, 8 * epc);
  SIMCLR_CHECK_ARG((long long)V * H * W < (1ll << 31), "conv2d_dgrad_bn_ext: M overflows int32");
  SIMCLR_CHECK_ARG(mask_mode >= 1 && mask_mode <= 4, "conv2d_dgrad_bn_ext: mask_mode must be 1, 2, 3 or 4");
  SIMCLR_CHECK_ARG(dm && h && w_ext && stats && nslot > 0 && (mask_mode == 4 || (bn_x && bn_mean && bn_rstd)), "conv2d_dgrad_bn_ext: null argument");
  SIMCLR_CHECK_ARG(mask_mode == 2 || bn_mask, "conv2d_dgrad_bn_ext: mask_mode 1 / 3 / 4 need bn_mask");
  SIMCLR_CHECK_ARG(mask_mode != 2 || (bn_scale && bn_shift), "conv2d_dgrad_bn_ext: mask_mode 2 needs scale/shift");
  ConvP p = {};
  p.split = terms;
  p.zero = zero_page();
  SIMCLR_CHECK_ARG(p.zero != nullptr, "conv2d: zero page symbol not found");
  p.x = dm; p.w = w_ext; p.y = dx; p.stats = stats; p.nslot = nslot; p.accumulate = accumulate;
  p.V = V; p.IH = H; p.IW = W; p.IC = Cout; p.OH = H; p.OW = W; p.N = Cin;
  p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.pixpitch = Cout;
  p.M = V * H * W; p.K = Cout + Cin;
  p.x2 = h; p.ic2 = Cin; p.pixpitch2 = Cin; p.bias = bias;
  p.bn_x = bn_x; p.bn_mask = bn_mask; p.bn_scale = bn_scale; p.bn_shift = bn_shift;
  p.bn_mean = bn_mean; p.bn_rstd = bn_rstd; p.bn_mode = mask_mode;
  if (dtype == SIMCLR_DT_BF16) launch_igemm<uint16_t, MODE_DGRAD>(p, stream);
  else launch_igemm<float, MODE_DGRAD>(p, stream);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// The same K-extended 1x1 data gradient with a plain epilogue (dx [+]= dm (a*W)^T + h (W diag(b) W^T) + W d): for a folded
// BatchNorm whose conv input is not itself a BatchNorm output (the projection shortcut at a block entry).
int simclr_conv2d_dgrad_ext(const void* dm, const void* h, const void* w_ext, const float* bias, void* dx, int accumulate,
                            int V, int H, int W, int Cin, int Cout, int dtype, hipStream_t stream) {
  const int terms = terms_of(&dtype, false);
  SIMCLR_CHECK_ARG(terms >= 0, "conv2d_dgrad_ext: bad matrix-arithmetic field in dtype (SIMCLR_FMT_TERMS)");
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "conv2d_dgrad_ext: bad dtype %d", dtype);
  SIMCLR_CHECK_ARG(Cout % (8 * epc) == 0 && Cin % (8 * epc) == 0, "conv2d_dgrad_ext: Cin=%d / Cout=%d must be multiples of %d", Cin, Cout, 8 * epc);
  SIMCLR_CHECK_ARG((long long)V * H * W < (1ll << 31), "conv2d_dgrad_ext: M overflows int32");
  SIMCLR_CHECK_ARG(dm && h && w_ext && dx, "conv2d_dgrad_ext: null argument");
  ConvP p = {};
  p.split = terms;
  p.zero = zero_page();
  SIMCLR_CHECK_ARG(p.zero != nullptr, "conv2d: zero page symbol not found");
  p.x = dm; p.w = w_ext; p.y = dx; p.stats = nullptr; p.nslot = 1; p.accumulate = accumulate;
  p.V = V; p.IH = H; p.IW = W; p.IC = Cout; p.OH = H; p.OW = W; p.N = Cin;
  p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.pixpitch = Cout;
  p.M = V * H * W; p.K = Cout + Cin;
  p.x2 = h; p.ic2 = Cin; p.pixpitch2 = Cin; p.bias = bias;
  if (dtype == SIMCLR_DT_BF16) launch_igemm<uint16_t, MODE_DGRAD>(p, stream);
  else launch_igemm<float, MODE_DGRAD>(p, stream);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

size_t simclr_conv2d_wgrad_workspace_bytes(int V, int OH, int OW, int Cin, int Cout, int KH, int KW,
                                           int dtype);

static int wgrad_splits(long long M, int K, int N, int bkw, int bnw, int br, int* chunks_per_split, int cap = 256,
                        int want = 0) {
  const int tiles = ceil_div(K, bkw) * ceil_div(N, bnw);
  const int nchunks = ceil_div(M, br);
  // ~1024-2048 workgroups in total; a multiple of 8 pixel ranges so that the XCD-aware mapping (one
  // pixel range per XCD at a time) keeps all 8 XCDs equally loaded
  static const int target_env = getenv("SIMCLR_WGRAD_BLOCKS") ? atoi(getenv("SIMCLR_WGRAD_BLOCKS")) : 1536;
  // 256 x 256 tiles run one 8-wave workgroup per CU: two rounds of 256 instead of three rounds of 512
  const int target = want > 0 ? want : (bkw == 256 ? 512 : target_env);
  int splits = max(1, min(nchunks, target / max(1, tiles)));
  if (nchunks >= 8) splits = min(nchunks / 8 * 8, max(8, (splits + 7) / 8 * 8));
  splits = min(splits, cap);
  *chunks_per_split = ceil_div(nchunks, splits);
  int eff = ceil_div(nchunks, *chunks_per_split);
  // keep the effective split count a multiple of 8 when possible
  while (nchunks >= 8 && eff % 8 != 0 && *chunks_per_split > 1) {
    --*chunks_per_split;
    eff = ceil_div(nchunks, *chunks_per_split);
    if (eff > cap) { ++*chunks_per_split; eff = ceil_div(nchunks, *chunks_per_split); break; }
  }
  return eff;
}
// nine-tap 3x3 kernel: default for every eligible layer (bf16, stride 1, 7 <= W, whole 64-channel blocks); measured
// against the per-tap kernels at 1024 views (profiles/r03_notes.md): 56^2 645 -> 516 us, 28^2 463 -> 381, 14^2 387 -> 367,
// 7^2 360 -> 349; SIMCLR_WGRAD_3X3=0 switches back (read per call so that a test can compare both in-process)
static bool wgrad_use_3x3(int dtype, long long M, int Cin, int Cout, int KH, int KW, int stride, int pad, int IH, int IW,
                          int OH, int OW, int pixpitch) {
  const char* e = getenv("SIMCLR_WGRAD_3X3");
  if (e && atoi(e) <= 0) return false;
  return dtype == SIMCLR_DT_BF16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && IH == OH && IW == OW &&
         Cin % 64 == 0 && Cout % 64 == 0 && (pixpitch * 2) % 16 == 0 && IW >= 7 && ((64 + 2 * IW + 2 + 31) / 32 * 32 + 64) * 128 * 2 <= 160 * 1024;
}
static bool wgrad_use_256() {
#ifdef SIMCLR_DIAG
  const char* e = getenv("SIMCLR_WGRAD_256");       // per launch: sweeps
  return !e || atoi(e) != 0;
#else
  static const bool on = !getenv("SIMCLR_WGRAD_256") || atoi(getenv("SIMCLR_WGRAD_256")) != 0;
  return on;
#endif
}
static void wgrad_tile(int Cin, int Cout, int dtype, long long M, int taps, int* bkw, int* bnw, bool f32ps256 = false) {
  *bkw = (Cin % 128 == 0) ? 128 : (Cin % 64 == 0 ? 64 : 32);
  *bnw = (Cout % 128 == 0 || Cout > 128) ? 128 : 64;
  if (*bkw == 32) *bnw = 64;
  // 256 x 256 tile (one 8-wave workgroup per CU, half the L2->LDS bytes per FLOP): measured per layer
  // (tools/diag_conv.py --wgrad) a win for the long reductions and the 256-channel 3x3 layers, a loss
  // for the short ones (7x7, 14x14 1x1) where it leaves too few workgroups.
  if (dtype == SIMCLR_DT_BF16 && Cin % 256 == 0 && Cout % 256 == 0 && wgrad_use_256() &&
      (M >= 500000 || (taps == 9 && M >= 150000 && Cin == 256))) { *bkw = 256; *bnw = 256; }
  // fp32 storage, pre-split gradient, 1x1 layers of 256-channel multiples: 256 x 256 tile, eight waves along k (32 k-rows x all 256 columns
  // each: the activation operand split once per workgroup, 8 LDS-DMA instructions per wave for 96 MFMAs instead of 48), one workgroup per
  // CU.  Per layer at 1024 views (r06_call48): 28^2 512->256 713 -> 673 us, 14^2 1024->512 659 -> 620, 1024->256 361 -> 358, 256->1024
  // 366 -> 360, 7^2 340 -> 331 / 350 -> 340; no layer loses
  if (f32ps256 && taps == 1 && Cin % 256 == 0 && Cout % 256 == 0) { *bkw = 256; *bnw = 256; }
}

size_t simclr_conv2d_wgrad_workspace_bytes(int V, int OH, int OW, int Cin, int Cout, int KH, int KW,
                                           int dtype) {
  int bkw, bnw, cps;
  wgrad_tile(Cin, Cout, dtype, (long long)V * OH * OW, KH * KW, &bkw, &bnw);
  // the bf16 kernel variants reduce in chunks of 64 or 32 pixels: size for whichever needs more slabs
  int splits = wgrad_splits((long long)V * OH * OW, KH * KW * Cin, Cout, bkw, bnw, dtype == SIMCLR_DT_BF16 ? 64 : 32, &cps);
  splits = max(splits, wgrad_splits((long long)V * OH * OW, KH * KW * Cin, Cout, bkw, bnw, 32, &cps));
  if (KH == 3 && KW == 3 && Cin % 64 == 0 && Cout % 64 == 0)      // nine-tap kernel: 64x64 tiles of all taps, up to 2048 ranges
    splits = max(splits, max(wgrad_splits((long long)V * OH * OW, Cin, Cout, 64, 64, 64, &cps, 2048, 4096),
                             wgrad_splits((long long)V * OH * OW, Cin, Cout, 64, 64, 32, &cps, 2048, 4096)));    // (fp32 nine-tap kernel: 32-pixel chunks)
  if (dtype != SIMCLR_DT_BF16 && KH * KW == 1 && Cin % 256 == 0 && Cout % 256 == 0)   // fp32 256 x 256 tile on a pre-split gradient: fewer tiles, more ranges
    splits = max(splits, wgrad_splits((long long)V * OH * OW, Cin, Cout, 256, 256, 32, &cps));
  if (Cin == 32 && KH * KW * Cin <= 256 && Cout <= 64)            // stem: one 256-row k-tile, up to 1024 pixel ranges
    splits = max(splits, wgrad_splits((long long)V * OH * OW, KH * KW * Cin, Cout, 256, 64, dtype == SIMCLR_DT_BF16 ? 64 : 32, &cps, 1024, 1024));
  return (size_t)splits * KH * KW * Cin * Cout * sizeof(float);
}

// Weight gradient.  x [V,IH,IW,*] (T, pixel pitch `pixpitch` elements, Cin channels used),
// dy [V,OH,OW,Cout] (T), dw fp32 [KH*KW*Cin][Cout] (= HWIO).  accumulate != 0: dw += result.
int simclr_conv2d_wgrad(const void* x, const void* dy, float* dw, int accumulate, void* workspace,
                        int V, int IH, int IW, int Cin, int pixpitch, int OH, int OW, int Cout, int KH,
                        int KW, int stride, int pad, int dtype, hipStream_t stream) {
  const int terms = terms_of(&dtype, false);
  SIMCLR_CHECK_ARG(terms >= 0, "conv2d_wgrad: bad matrix-arithmetic field in dtype (SIMCLR_FMT_TERMS)");
  // dtype | SIMCLR_FMT_PS_IN (fp32, three bf16 backward terms, Cin a multiple of 64, Cout of 32, 16-byte aligned pixels): dy is in
  // the pre-split block format (common.h)
  const bool dy_ps = (dtype & SIMCLR_FMT_PS_IN) != 0;
  dtype &= 0xff;
  SIMCLR_CHECK_ARG(!dy_ps || (dtype == SIMCLR_DT_F32 && terms == 3 && Cin % 64 == 0 && Cout % 32 == 0 && (pixpitch * 4) % 16 == 0),
                   "conv2d_wgrad: a pre-split dy needs fp32 storage, three backward terms, Cin %% 64 == 0 and Cout %% 32 == 0 (Cin=%d Cout=%d)", Cin, Cout);
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "conv2d_wgrad: bad dtype %d", dtype);
  SIMCLR_CHECK_ARG(Cin % 32 == 0, "conv2d_wgrad: Cin=%d must be a multiple of 32", Cin);
  SIMCLR_CHECK_ARG(Cout % 8 == 0, "conv2d_wgrad: Cout=%d must be a multiple of 8", Cout);
  SIMCLR_CHECK_ARG((long long)V * OH * OW < (1ll << 31), "conv2d_wgrad: M overflows int32");
  WgradP p = {};
  p.x = x; p.dy = dy; p.dw = (float*)workspace;
  p.V = V; p.IH = IH; p.IW = IW; p.IC = Cin; p.OH = OH; p.OW = OW; p.N = Cout;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.pixpitch = pixpitch;
  p.M = V * OH * OW; p.K = KH * KW * Cin;
  if (wgrad_use_3x3(dtype, p.M, Cin, Cout, KH, KW, stride, pad, IH, IW, OH, OW, pixpitch)) {
    Wgrad3P q = {};
    q.x = x; q.dy = dy; q.dw = (float*)workspace; q.zero = zero_page();
    SIMCLR_CHECK_ARG(q.zero != nullptr, "conv2d_wgrad: zero page symbol not found");
    q.V = V; q.H = IH; q.W = IW; q.IC = Cin; q.N = Cout; q.pixpitch = pixpitch; q.M = p.M;
    q.ci_tiles = Cin / 64; q.co_tiles = Cout / 64;
    // pixel ranges (SIMCLR_WGRAD3_BLOCKS overrides the total workgroup target), at most 2048 ranges
    // measured (us at 1024 views, targets 512 / 768 / 1024 / 1536 / 2048): 56^2 437 517 434 425 418 | 28^2 362 440 370 450
    // 507 | 14^2 323 379 344 371 515 | 7^2 295 311 312 340 378 -> one full round of 512 (2 per CU), more ranges only
    // for the single-tile 64 -> 64 layer; partial rounds (768, 1536) are the worst choice
    const int tiles3 = (Cin / 64) * (Cout / 64);
    // (never more than the 4096 simclr_conv2d_wgrad_workspace_bytes sizes the slabs for)
    const int want3 = min(4096, max(8, getenv("SIMCLR_WGRAD3_BLOCKS") ? atoi(getenv("SIMCLR_WGRAD3_BLOCKS")) : (tiles3 == 1 ? 2048 : 512)));
    q.splits = wgrad_splits(p.M, Cin, Cout, 64, 64, 64, &q.chunks_per_split, 2048, want3);
    q.hpp = (64 + 2 * IW + 2 + 31) / 32 * 32;
    const int tiles = q.ci_tiles * q.co_tiles;
    const int grid3 = tiles * ceil_div(q.splits, 8) * 8;
    const size_t stage3 = (size_t)(q.hpp + 64) * 128;
    constexpr bool ring3 = true;          // three-deep ring wherever two workgroups of it fit a CU (round 3: -0.1 ms per step)
    if (ring3 && 2 * 3 * stage3 <= 160 * 1024)
      hipLaunchKernelGGL(conv_wgrad3x3_bf16<3>, dim3(grid3), dim3(256), 3 * stage3, stream, q);
    else
      hipLaunchKernelGGL(conv_wgrad3x3_bf16<2>, dim3(grid3), dim3(256), 2 * stage3, stream, q);
    SIMCLR_CHECK_LAUNCH();
    const long long numel3 = (long long)p.K * p.N;
    hipLaunchKernelGGL(slab_reduce, dim3(max(1, (int)ceil_div(numel3 / 4, 16))), dim3(256), 0, stream,
                       (const float*)workspace, q.splits, numel3, dw, accumulate);
    SIMCLR_CHECK_LAUNCH();
    return 0;
  }
  // fp32 storage, three backward terms, pre-split gradient: the nine-tap kernel for the 3x3 stride-1 layers whose window fits two
  // workgroups per CU (W <= 47: 28^2, 14^2, 7^2); SIMCLR_WGRAD_3X3=0 keeps the per-tap kernel (read per call, as for bf16)
  {
    const char* e3 = getenv("SIMCLR_WGRAD_3X3");
    const int hpp = (32 + 2 * IW + 2 + 31) / 32 * 32;
    if (!(e3 && atoi(e3) <= 0) && dtype == SIMCLR_DT_F32 && terms == 3 && dy_ps && KH == 3 && KW == 3 && stride == 1 && pad == 1 &&
        IH == OH && IW == OW && Cin % 64 == 0 && Cout % 64 == 0 && IW >= 7 && 2 * 2 * (hpp + 32) * 256 <= 160 * 1024) {
      Wgrad3P q = {};
      q.x = x; q.dy = dy; q.dw = (float*)workspace; q.zero = zero_page();
      SIMCLR_CHECK_ARG(q.zero != nullptr, "conv2d_wgrad: zero page symbol not found");
      q.V = V; q.H = IH; q.W = IW; q.IC = Cin; q.N = Cout; q.pixpitch = pixpitch; q.M = p.M;
      q.ci_tiles = Cin / 64; q.co_tiles = Cout / 64;
      const int tiles3 = q.ci_tiles * q.co_tiles;
      const int want3 = min(2048, max(8, getenv("SIMCLR_WGRAD3_BLOCKS") ? atoi(getenv("SIMCLR_WGRAD3_BLOCKS")) : (tiles3 == 1 ? 2048 : 512)));
      q.splits = wgrad_splits(p.M, Cin, Cout, 64, 64, 32, &q.chunks_per_split, 2048, want3);
      q.hpp = hpp;
      const int grid3 = tiles3 * ceil_div(q.splits, 8) * 8;
      hipLaunchKernelGGL(conv_wgrad3x3_f32ps, dim3(grid3), dim3(256), (size_t)2 * (hpp + 32) * 256, stream, q);
      SIMCLR_CHECK_LAUNCH();
      const long long numel3 = (long long)p.K * p.N;
      hipLaunchKernelGGL(slab_reduce, dim3(max(1, (int)ceil_div(numel3 / 4, 16))), dim3(256), 0, stream,
                         (const float*)workspace, q.splits, numel3, dw, accumulate);
      SIMCLR_CHECK_LAUNCH();
      return 0;
    }
  }
  int bkw, bnw;
  constexpr bool f32_256 = true;
  wgrad_tile(Cin, Cout, dtype, p.M, KH * KW, &bkw, &bnw, f32_256 && dtype == SIMCLR_DT_F32 && terms == 3 && dy_ps && stride == 1);
  if ((pixpitch * (dtype == SIMCLR_DT_BF16 ? 2 : 4)) % 16 != 0 && bkw == 256) { bkw = 128; bnw = 128; }
  // stem (packed input, 32 elements per kernel row): all kernel rows in ONE 256-row k-tile, so dY is read once
  constexpr bool stem_mt_on = true;
  const bool stem_mt = stem_mt_on && bkw == 32 && Cin == 32 && p.K <= 256 && Cout <= 64;
  if (stem_mt) { bkw = 256; bnw = 64; }
  // kernel variant: 1 = LDS-DMA ring, 64-pixel chunks x 2 stages; 2 = 32-pixel chunks x 3 stages for the
  // 128x128 tile (3 workgroups/CU), 3 stages elsewhere; 3 = 32-pixel chunks x 4 stages; 0 = register-staged
  // kernel (kept for A/B runs: SIMCLR_WGRAD_CFG).
#ifdef SIMCLR_DIAG
  const int cfg_env = getenv("SIMCLR_WGRAD_CFG") ? atoi(getenv("SIMCLR_WGRAD_CFG")) : -1;   // per launch: sweeps
#else
  static const int cfg_env = getenv("SIMCLR_WGRAD_CFG") ? atoi(getenv("SIMCLR_WGRAD_CFG")) : -1;
#endif
  int cfg = cfg_env >= 0 ? cfg_env : 1;
  if (dtype != SIMCLR_DT_BF16 && cfg > 1) cfg = 1;
  // the stem's packed input (pixel pitch 4 elements) gives 8-byte-aligned sources: keep the register-staged kernel
  if (bkw == 32 || stem_mt || (pixpitch * (dtype == SIMCLR_DT_BF16 ? 2 : 4)) % 16 != 0) cfg = 0;
  const bool big = bkw == 128 && bnw == 128;
  const bool big256 = bkw == 256 && !stem_mt;
  if (big256 && cfg == 0) cfg = 1;
  const bool bf = dtype == SIMCLR_DT_BF16;
  const int brm = ((cfg >= 2 && big) || (big256 && bf)) ? 1 : 2;
  const int stages = big256 ? (bf ? 4 : 2) : (cfg <= 1 ? 2 : (big ? (cfg == 2 ? 3 : 4) : 3));
  const int br = (dtype == SIMCLR_DT_BF16 ? 32 : 16) * brm;
  // workgroup target of the bf16 launches: 1024 (round 5 sweep, interleaved pairs on one box, ms per step at 768 / 1024 / 1536 / 2048 /
  // 3072: 64.1 / 63.7 / 63.9 / 64.6 / 65.1 -- fewer, longer pixel ranges write fewer fp32 slabs); the fp32 launches keep the 1536 of
  // SIMCLR_WGRAD_BLOCKS' default (parity mode 190.5 / 187.4 / 187.2 at 1024 / 1536 / 2048).  The workspace is sized for 1536.
  static const bool blocks_env = getenv("SIMCLR_WGRAD_BLOCKS") != nullptr;
  const int want_wg = (dtype == SIMCLR_DT_BF16 && !blocks_env && bkw != 256) ? 1024 : 0;
  p.splits = stem_mt ? wgrad_splits(p.M, p.K, p.N, bkw, bnw, br, &p.chunks_per_split, 1024, 1024)
                     : wgrad_splits(p.M, p.K, p.N, bkw, bnw, br, &p.chunks_per_split, 256, want_wg);
  p.k_tiles = ceil_div(p.K, bkw);
  p.n_tiles = ceil_div(p.N, bnw);
  // XCD-aware mapping: with the LDS-DMA kernel a win (or neutral) on every ResNet-50 layer; the register-staged
  // kernel (cfg 0: stem) keeps the old size rule
  p.xcd_map = cfg != 0 || (p.M >= 1500000) || (p.M >= 500000 && KH * KW == 1 && stride == 1);
#ifdef SIMCLR_DIAG
  { const char* e = getenv("SIMCLR_DIAG"); p.diag = e ? atoi(e) : 0; }
  { const char* e = getenv("SIMCLR_WGRAD_XCD"); if (e && atoi(e) >= 0) p.xcd_map = atoi(e); }
#endif
  p.zero = zero_page();
  p.split = terms;
  p.dy_ps = dy_ps ? 1 : 0;
  SIMCLR_CHECK_ARG(p.zero != nullptr, "conv2d_wgrad: zero page symbol not found");
  SIMCLR_CHECK_ARG(!dy_ps || (cfg != 0 && !stem_mt), "conv2d_wgrad: no pre-split kernel on the register-staged path");
  const int grid = p.k_tiles * p.n_tiles * (p.xcd_map ? ceil_div(p.splits, 8) * 8 : p.splits);
  const size_t esz = dtype == SIMCLR_DT_BF16 ? 2 : 4;
  const size_t lds = (size_t)(cfg == 0 ? 2 : stages) * br * (bkw + bnw) * esz;
#define LW(TT, A, B) hipLaunchKernelGGL((conv_wgrad<TT, A, B>), dim3(grid), dim3(256), lds, stream, p)
#define LD(TT, A, B, M_, S_) hipLaunchKernelGGL((conv_wgrad_dma<TT, A, B, M_, S_>), dim3(grid), dim3(256), lds, stream, p)
  // stem over LDS-DMA: stride 2 on an even-width packed image makes every source 16-byte aligned
  constexpr bool stem_dma_on = true;
  const bool stem_dma = stem_mt && stem_dma_on && dtype == SIMCLR_DT_BF16 && KW == 1 && pad == 0 && stride % 2 == 0 && IW % 2 == 0 &&
                        pixpitch == 4 && Cin == 32;
  // fp32 stem under split-bf16 terms: a packed pixel is 4 floats = 16 bytes, so every source of the multi-tap k-tile is 16-byte
  // aligned at ANY stride / width; 32-pixel chunks x 2 stages.  The exact fp32 arithmetic keeps the register-staged kernel.
  static const bool stem_split_on = !getenv("SIMCLR_STEM_SPLIT") || atoi(getenv("SIMCLR_STEM_SPLIT")) != 0;
  const bool stem_dma_f32 = stem_mt && stem_dma_on && stem_split_on && dtype == SIMCLR_DT_F32 && p.split != 0 && KW == 1 && pad == 0 &&
                            pixpitch == 4 && Cin == 32;
  if (big256 && !bf) {
    hipLaunchKernelGGL((conv_wgrad_dma<float, 256, 256, 2, 2, 8, 1, false, false, 3, 1>), dim3(grid), dim3(512), lds, stream, p);
  } else if (big256) {
    hipLaunchKernelGGL((conv_wgrad_dma<uint16_t, 256, 256, 1, 4, 2, 4>), dim3(grid), dim3(512), lds, stream, p);
  } else if (stem_dma_f32) {
    const size_t lds_s = (size_t)2 * 32 * (256 + 64) * 4;      // 2 stages x 32 pixels x (256 + 64) fp32
    p.xcd_map = 1;
    const int grid_s = p.k_tiles * p.n_tiles * ceil_div(p.splits, 8) * 8;
    // 8 waves (4 along the 256 k-rows x 2 along n): 4 k-fragments and 4 LDS-DMA source states per wave -- the 4-wave shape keeps 8 + 8
    // of them next to the split fragments and spills ~100 registers
    // SIMCLR_STEM_WGRAD_STAGES=3: a three-deep ring (120 KB) -- two 32-pixel chunks in flight behind the one being multiplied
    static const int stem_stages = getenv("SIMCLR_STEM_WGRAD_STAGES") ? atoi(getenv("SIMCLR_STEM_WGRAD_STAGES")) : 2;
    if (p.split == 3 && stem_stages == 3)
      hipLaunchKernelGGL((conv_wgrad_dma<float, 256, 64, 2, 3, 4, 2, false, true, 3>), dim3(grid_s), dim3(512), lds_s / 2 * 3, stream, p);
    else if (p.split == 3) hipLaunchKernelGGL((conv_wgrad_dma<float, 256, 64, 2, 2, 4, 2, false, true, 3>), dim3(grid_s), dim3(512), lds_s, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_dma<float, 256, 64, 2, 2, 4, 2, false, true, 6>), dim3(grid_s), dim3(512), lds_s, stream, p);
  } else if (stem_dma) {
    const size_t lds_s = (size_t)2 * 64 * (256 + 64) * 2;      // 2 stages x 64 pixels x (256 + 64) bf16
    p.xcd_map = 1;
    const int grid_s = p.k_tiles * p.n_tiles * ceil_div(p.splits, 8) * 8;
    hipLaunchKernelGGL((conv_wgrad_dma<uint16_t, 256, 64, 2, 2, 2, 2, false, true>), dim3(grid_s), dim3(256), lds_s, stream, p);
  } else if (stem_mt) {
    if (dtype == SIMCLR_DT_BF16) hipLaunchKernelGGL((conv_wgrad<uint16_t, 256, 64, true>), dim3(grid), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((conv_wgrad<float, 256, 64, true>), dim3(grid), dim3(256), lds, stream, p);
  } else if (cfg == 0) {
    if (dtype == SIMCLR_DT_BF16) {
      if (bkw == 128 && bnw == 128) LW(uint16_t, 128, 128);
      else if (bkw == 128) LW(uint16_t, 128, 64);
      else if (bkw == 64 && bnw == 128) LW(uint16_t, 64, 128);
      else if (bkw == 64) LW(uint16_t, 64, 64);
      else LW(uint16_t, 32, 64);
    } else {
      if (bkw == 128 && bnw == 128) LW(float, 128, 128);
      else if (bkw == 128) LW(float, 128, 64);
      else if (bkw == 64 && bnw == 128) LW(float, 64, 128);
      else if (bkw == 64) LW(float, 64, 64);
      else LW(float, 32, 64);
    }
  } else if (dtype != SIMCLR_DT_BF16) {
    // settled A/B switches of round 6 (profiles/r06_notes.md sections 16, 17): four waves along k on a pre-split gradient (family 39.93 ->
    // 38.54 ms), plain fp32 gradients split in LDS (37.40 -> 37.21 ms)
    constexpr bool wk4 = true, ldsps = true;
#define LDS_(A, B)                                                                                                          \
    do {                                                                                                                     \
      if (p.split == 3 && p.dy_ps) {                                                                                         \
        /* pre-split gradient: four waves along k (each 32 | 16 k-rows x all columns) -- the activation operand, the one still split in   \
           registers, is then split ONCE per workgroup instead of once per column half (96 -> 48 VALU per 48 MFMAs) */                      \
        if constexpr ((A) >= 64) { if (wk4) hipLaunchKernelGGL((conv_wgrad_dma<float, A, B, 2, 2, 4, 1, false, false, 3, 1>), dim3(grid), dim3(256), lds, stream, p); \
                                   else hipLaunchKernelGGL((conv_wgrad_dma<float, A, B, 2, 2, 2, 2, false, false, 3, 1>), dim3(grid), dim3(256), lds, stream, p); } \
        else hipLaunchKernelGGL((conv_wgrad_dma<float, A, B, 2, 2, 2, 2, false, false, 3, 1>), dim3(grid), dim3(256), lds, stream, p); \
      }                                                                                                                      \
      else if (p.split == 3) {                                                                                               \
        /* plain fp32 gradient: split in LDS once per chunk (PSD = 2), then the pre-split kernel's loop with four waves along k */         \
        if constexpr ((A) >= 64 && (B) % 32 == 0) { if (ldsps) hipLaunchKernelGGL((conv_wgrad_dma<float, A, B, 2, 2, 4, 1, false, false, 3, 2>), dim3(grid), dim3(256), lds, stream, p); \
                                   else hipLaunchKernelGGL((conv_wgrad_dma<float, A, B, 2, 2, 2, 2, false, false, 3>), dim3(grid), dim3(256), lds, stream, p); } \
        else hipLaunchKernelGGL((conv_wgrad_dma<float, A, B, 2, 2, 2, 2, false, false, 3>), dim3(grid), dim3(256), lds, stream, p); \
      }                                                                                                                      \
      else if (p.split == 6) hipLaunchKernelGGL((conv_wgrad_dma<float, A, B, 2, 2, 2, 2, false, false, 6>), dim3(grid), dim3(256), lds, stream, p); \
      else LD(float, A, B, 2, 2);                                                                                            \
    } while (0)
    if (bkw == 128 && bnw == 128) LDS_(128, 128);
    else if (bkw == 128) LDS_(128, 64);
    else if (bkw == 64 && bnw == 128) LDS_(64, 128);
    else if (bkw == 64) LDS_(64, 64);
    else LDS_(32, 64);
#undef LDS_
  } else if (stages == 2) {
    if (bkw == 128 && bnw == 128) LD(uint16_t, 128, 128, 2, 2);
    else if (bkw == 128) LD(uint16_t, 128, 64, 2, 2);
    else if (bkw == 64 && bnw == 128) LD(uint16_t, 64, 128, 2, 2);
    else if (bkw == 64) LD(uint16_t, 64, 64, 2, 2);
    else LD(uint16_t, 32, 64, 2, 2);
  } else {
    if (big && stages == 3) LD(uint16_t, 128, 128, 1, 3);
    else if (big) LD(uint16_t, 128, 128, 1, 4);
    else if (bkw == 128) LD(uint16_t, 128, 64, 2, 3);
    else if (bkw == 64 && bnw == 128) LD(uint16_t, 64, 128, 2, 3);
    else if (bkw == 64) LD(uint16_t, 64, 64, 2, 3);
    else LD(uint16_t, 32, 64, 2, 3);
  }
#undef LD
#undef LW
  SIMCLR_CHECK_LAUNCH();
  const long long numel = (long long)p.K * p.N;
  hipLaunchKernelGGL(slab_reduce, dim3(max(1, (int)ceil_div(numel / 4, 16))), dim3(256), 0, stream,
                     (const float*)workspace, p.splits, numel, dw, accumulate);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// h^T h and the column sums of h for an activation h [M = V*H*W rows][K channels] (T): out [K*K + K] fp32
// (Gram matrix row-major, then the K column sums).  K in {64, 128, 256}: one tile spans all channels, the activation
// is streamed ONCE.  workspace: simclr_conv2d_gram_workspace_bytes.
size_t simclr_conv2d_gram_workspace_bytes(long long M, int K, int dtype) {
  int cps;
  const int br = (dtype == SIMCLR_DT_BF16 ? 64 : 32);
  const int splits = wgrad_splits(M, K, K, K, K, K == 256 ? br / 2 : br, &cps, K == 256 ? 512 : 1024, K == 256 ? 512 : 1024);
  return (size_t)splits * ((size_t)K * K + K) * sizeof(float);
}
int simclr_conv2d_gram(const void* h, float* out, void* workspace, long long M, int K, int dtype, hipStream_t stream) {
  // fp32 storage: dtype may carry the matrix-arithmetic field (SIMCLR_FMT_TERMS) of the forward pass this Gram matrix serves.  Exact
  // arithmetic -> exact fp32 MFMA; any split mode -> SIX bf16-piece terms (fp32-level products at 2.7x the fp32 MFMA rate: the Gram
  // launches were the last exact-fp32 matrix work of the fast parity step)
  const int terms = terms_of(&dtype, true);
  SIMCLR_CHECK_ARG(terms >= 0, "conv2d_gram: bad matrix-arithmetic field in dtype (SIMCLR_FMT_TERMS)");
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "conv2d_gram: bad dtype %d", dtype);
  SIMCLR_CHECK_ARG(K == 64 || K == 128 || (K == 256 && dtype == SIMCLR_DT_BF16), "conv2d_gram: K=%d not supported (64, 128, bf16 256)", K);
  SIMCLR_CHECK_ARG(M > 0 && M < (1ll << 31), "conv2d_gram: bad M");
  WgradP p = {};
  p.x = h; p.dy = h; p.dw = (float*)workspace;
  p.V = 1; p.IH = 1; p.IW = (int)M; p.IC = K; p.OH = 1; p.OW = (int)M; p.N = K;
  p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.pixpitch = K;
  p.M = (int)M; p.K = K;
  const int br = (dtype == SIMCLR_DT_BF16 ? 64 : 32) / (K == 256 ? 2 : 1);
  // one tile per pixel range: up to 1024 workgroups (4 per CU) share the streaming
  p.splits = wgrad_splits(M, K, K, K, K, br, &p.chunks_per_split, K == 256 ? 512 : 1024, K == 256 ? 512 : 1024);
  p.k_tiles = 1; p.n_tiles = 1; p.xcd_map = 1;
  p.zero = zero_page();
  SIMCLR_CHECK_ARG(p.zero != nullptr, "conv2d_gram: zero page symbol not found");
  const int grid = ceil_div(p.splits, 8) * 8;
  const size_t esz = dtype == SIMCLR_DT_BF16 ? 2 : 4;
  if (K == 256) {
    const size_t lds = (size_t)4 * br * K * esz;
    hipLaunchKernelGGL((conv_wgrad_dma<uint16_t, 256, 256, 1, 4, 2, 4, true>), dim3(grid), dim3(512), lds, stream, p);
  } else {
    // 4-stage ring of 64-pixel (bf16) chunks holding the activation tile only: three chunks in flight per workgroup
    const size_t lds = (size_t)4 * br * K * esz;
    if (dtype == SIMCLR_DT_BF16) {
      if (K == 64) hipLaunchKernelGGL((conv_wgrad_dma<uint16_t, 64, 64, 2, 4, 2, 2, true>), dim3(grid), dim3(256), lds, stream, p);
      else hipLaunchKernelGGL((conv_wgrad_dma<uint16_t, 128, 128, 2, 4, 2, 2, true>), dim3(grid), dim3(256), lds, stream, p);
    } else {
      // (A/B on one box, r06_call52: step 140.16 -> 139.71 ms in three pairs; every gate of the fused-tail / fold / step tests unchanged)
      if (terms != 0) {
        if (K == 64) hipLaunchKernelGGL((conv_wgrad_dma<float, 64, 64, 2, 4, 2, 2, true, false, 6>), dim3(grid), dim3(256), lds, stream, p);
        else hipLaunchKernelGGL((conv_wgrad_dma<float, 128, 128, 2, 4, 2, 2, true, false, 6>), dim3(grid), dim3(256), lds, stream, p);
      }
      else if (K == 64) hipLaunchKernelGGL((conv_wgrad_dma<float, 64, 64, 2, 4, 2, 2, true>), dim3(grid), dim3(256), lds, stream, p);
      else hipLaunchKernelGGL((conv_wgrad_dma<float, 128, 128, 2, 4, 2, 2, true>), dim3(grid), dim3(256), lds, stream, p);
    }
  }
  SIMCLR_CHECK_LAUNCH();
  const long long numel = (long long)K * K + K;
  hipLaunchKernelGGL(slab_reduce, dim3(max(1, (int)ceil_div(numel / 4, 16))), dim3(256), 0, stream,
                     (const float*)workspace, p.splits, numel, out, 0);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// C [M][N] = A [M][K] B[N][K]^T in fp32 on the matrix cores; M, N multiples of 16, K of 16 (small helper GEMMs)
int simclr_small_gemm_nt_f32(const float* A, const float* B, float* C, int M, int N, int K, hipStream_t stream) {
  SIMCLR_CHECK_ARG(M > 0 && N > 0 && K > 0 && M % 16 == 0 && N % 16 == 0 && K % 32 == 0, "small_gemm_nt_f32: M=%d N=%d must be multiples of 16, K=%d of 32", M, N, K);
  if ((long long)ceil_div(M, 64) * ceil_div(N, 64) >= 256)
    hipLaunchKernelGGL((small_gemm_nt_f32<64>), dim3(ceil_div(N, 64), ceil_div(M, 64)), dim3(256), 0, stream, A, B, C, M, N, K);
  else
    hipLaunchKernelGGL((small_gemm_nt_f32<32>), dim3(ceil_div(N, 32), ceil_div(M, 32)), dim3(256), 0, stream, A, B, C, M, N, K);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// Stem forward on the packed input (see simclr_pack_views).  w_s [Cout][KHP*KWP*4] (T).
int simclr_stem_conv_fwd(const void* xp, const void* w_s, void* y, float* stats, int nslot, int V,
                         int HP, int WP, int OH, int OW, int Cout, int KHP, int KWP, int stride,
                         int dtype, hipStream_t stream) {
  const int terms = terms_of(&dtype, true);
  SIMCLR_CHECK_ARG(terms >= 0, "stem_conv_fwd: bad matrix-arithmetic field in dtype (SIMCLR_FMT_TERMS)");
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "stem_conv_fwd: bad dtype %d", dtype);
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  StemP p = {};
  p.xp = xp; p.w = w_s; p.y = y; p.stats = stats; p.nslot = nslot > 0 ? nslot : 1;
  p.V = V; p.HP = HP; p.WP = WP; p.OH = OH; p.OW = OW; p.N = Cout; p.KHP = KHP; p.KWP = KWP;
  p.stride = stride; p.M = V * OH * OW; p.KP = KHP * KWP * 4;
  SIMCLR_CHECK_ARG(p.KP % (4 * epc) == 0, "stem_conv_fwd: padded K=%d not a multiple of %d", p.KP, 4 * epc);
  SIMCLR_CHECK_ARG((KWP * 4) % epc == 0, "stem_conv_fwd: KWP*4 must be a multiple of %d", epc);
  SIMCLR_CHECK_ARG(Cout % (dtype == SIMCLR_DT_BF16 ? 8 : 4) == 0, "stem_conv_fwd: Cout=%d must be a multiple of %d", Cout, dtype == SIMCLR_DT_BF16 ? 8 : 4);
  p.m_tiles = ceil_div(p.M, 128);
#ifdef SIMCLR_DIAG
  { const char* e = getenv("SIMCLR_DIAG"); p.diag = e ? atoi(e) : 0; }
#endif
  const size_t esz = dtype == SIMCLR_DT_BF16 ? 2 : 4;
  p.wpad = stem_lds_pad((int)(p.KP * esz));
  const size_t lds = 64 * (p.KP * esz + p.wpad) + 4 * 64 * 2 * sizeof(float) + (esz == 2 ? 128 * 64 * 2 : 0);
  dim3 grid(min(p.m_tiles, 2048), ceil_div(Cout, 64));
  if (dtype == SIMCLR_DT_BF16) {
    constexpr bool unroll_on = true;
    if (unroll_on && KWP * 4 == 32 && p.KP == 7 * 32) {      // one k-step = one padded kernel row, 7 rows
      if (stats) hipLaunchKernelGGL((stem_conv_fwd<uint16_t, true, 7>), grid, dim3(256), lds, stream, p);
      else hipLaunchKernelGGL((stem_conv_fwd<uint16_t, false, 7>), grid, dim3(256), lds, stream, p);
    } else if (stats) hipLaunchKernelGGL((stem_conv_fwd<uint16_t, true>), grid, dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((stem_conv_fwd<uint16_t, false>), grid, dim3(256), lds, stream, p);
  } else {
    // fp32 storage: the forward terms of simclr_set_f32_matmul (0 = exact fp32 MFMA; SIMCLR_STEM_SPLIT=0 keeps the exact kernel)
    static const bool stem_split_on = !getenv("SIMCLR_STEM_SPLIT") || atoi(getenv("SIMCLR_STEM_SPLIT")) != 0;
    // (13 = three fp16-piece terms, both operands split in registers: images lie in [0, 1], and the stem's weights -- fan-in 147, |w| ~ 0.1 --
    // keep ~2^-21 relative in their unscaled lo pieces; -0.8 ms per step against six bf16 terms, profiles/r06_notes.md)
    constexpr bool stem_f16_on = true;
    const int spl = stem_split_on ? ((terms == 13 && !stem_f16_on) ? 6 : terms) : 0;     // split-fp16 forward: the stem keeps six bf16 terms
    // the 7x7 stem (7 padded kernel rows of 8 taps x 4 channels = 14 k-steps of 16): unrolled, weights pre-split in LDS, rolling fragments
    static const bool stem_roll_on = !getenv("SIMCLR_STEM_ROLL") || atoi(getenv("SIMCLR_STEM_ROLL")) != 0;
    const bool roll = stem_roll_on && p.KP == 14 * 16 && KWP * 4 == 32 &&     // (7 padded kernel rows of 8 taps x 4 channels)
                      (long long)V * HP * WP * 16 < (1ll << 32);              // (32-bit byte offsets in the rolling kernel)
#define LSF(STv)                                                                                              \
    do {                                                                                                       \
      if (spl == 13 && roll) hipLaunchKernelGGL((stem_conv_fwd<float, STv, 14, 13>), grid, dim3(256), lds, stream, p);  \
      else if (spl == 3 && roll) hipLaunchKernelGGL((stem_conv_fwd<float, STv, 14, 3>), grid, dim3(256), lds, stream, p);    \
      else if (spl == 13) hipLaunchKernelGGL((stem_conv_fwd<float, STv, 0, 13>), grid, dim3(256), lds, stream, p);  \
      else if (spl == 3) hipLaunchKernelGGL((stem_conv_fwd<float, STv, 0, 3>), grid, dim3(256), lds, stream, p);    \
      else if (spl == 6) hipLaunchKernelGGL((stem_conv_fwd<float, STv, 0, 6>), grid, dim3(256), lds, stream, p); \
      else hipLaunchKernelGGL((stem_conv_fwd<float, STv>), grid, dim3(256), lds, stream, p);                   \
    } while (0)
    if (stats) LSF(true); else LSF(false);
#undef LSF
  }
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// fp32 HWIO master weights -> compute-dtype copies.  mode 0 fwd, 1 dgrad, 2 stem (padded).
// CinP/CoutP (>= Cin/Cout, 0 = no padding): channel dims of the destination, zero-filled beyond the
// real channels (used where a layer's channel count is not a multiple of the 64-element k-tile).
int simclr_prep_weights(const float* w_hwio, void* dst, int KH, int KW, int Cin, int Cout, int mode,
                        int KHP, int KWP, int CinP, int CoutP, int dtype, hipStream_t stream) {
  SIMCLR_CHECK_ARG(mode >= 0 && mode <= 2, "prep_weights: bad mode %d", mode);
  if (CinP <= 0) CinP = Cin;
  if (CoutP <= 0) CoutP = Cout;
  SIMCLR_CHECK_ARG(CinP >= Cin && CoutP >= Cout, "prep_weights: padded dims smaller than real dims");
  const long long total = mode == 2 ? (long long)CoutP * KHP * KWP * 4 : (long long)KH * KW * CinP * CoutP;
  const int grid = min(4096, ceil_div(total, 256));
  if (dtype == SIMCLR_DT_BF16)
    hipLaunchKernelGGL((prep_weights<uint16_t>), dim3(grid), dim3(256), 0, stream, w_hwio, (uint16_t*)dst,
                       KH, KW, Cin, Cout, mode, KHP, KWP, CinP, CoutP);
  else
    hipLaunchKernelGGL((prep_weights<float>), dim3(grid), dim3(256), 0, stream, w_hwio, (float*)dst, KH,
                       KW, Cin, Cout, mode, KHP, KWP, CinP, CoutP);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// modes 0 and 1 of simclr_prep_weights for the same weight in ONE launch (one per conv layer and step).
int simclr_prep_weights_pair(const float* w_hwio, void* dst_t, void* dst_d, int KH, int KW, int Cin, int Cout,
                             int CinP, int CoutP, int dtype, hipStream_t stream) {
  if (CinP <= 0) CinP = Cin;
  if (CoutP <= 0) CoutP = Cout;
  SIMCLR_CHECK_ARG(CinP >= Cin && CoutP >= Cout, "prep_weights_pair: padded dims smaller than real dims");
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "prep_weights_pair: bad dtype %d", dtype);
  const long long total = (long long)KH * KW * CinP * CoutP;
  const int grid = (int)max(1ll, min(1ll << 20, (total + 255) / 256));
  if (dtype == SIMCLR_DT_BF16)
    hipLaunchKernelGGL((prep_weights_pair<uint16_t>), dim3(grid), dim3(256), 0, stream, w_hwio, (uint16_t*)dst_t,
                       (uint16_t*)dst_d, KH * KW, Cin, Cout, CinP, CoutP);
  else
    hipLaunchKernelGGL((prep_weights_pair<float>), dim3(grid), dim3(256), 0, stream, w_hwio, (float*)dst_t,
                       (float*)dst_d, KH * KW, Cin, Cout, CinP, CoutP);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// Every (w_t, w_d) pair of a model in one launch.  table: device int64 [T][8] = {w_hwio, dst_t, dst_d, KH*KW, Cin, Cout,
// CinP, CoutP}; chunks: device int64 [nchunks][2] = {tensor, tile}, tile = (tap * nci + ci_tile) * nco + co_tile with
// nci = ceil(CinP / E), nco = ceil(CoutP / E), E = simclr_prep_chunk_elems() (the tile edge).
int simclr_prep_chunk_elems(void) { return kPrepTile; }
int simclr_prep_weights_pair_multi(const long long* table, const long long* chunks, int nchunks, int dtype,
                                   hipStream_t stream) {
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "prep_weights_pair_multi: bad dtype %d", dtype);
  SIMCLR_CHECK_ARG(table && chunks && nchunks > 0, "prep_weights_pair_multi: empty table");
  if (dtype == SIMCLR_DT_BF16)
    hipLaunchKernelGGL((prep_weights_pair_multi<uint16_t>), dim3(nchunks), dim3(256), 0, stream, table, chunks);
  else
    hipLaunchKernelGGL((prep_weights_pair_multi<float>), dim3(nchunks), dim3(256), 0, stream, table, chunks);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// ---- pre-split weight copies, made once per optimizer step by the caller (SIMCLR_FMT_PS_W) ----
// table [n][4] (device, int64): source fp32 matrix, destination, 128-byte k-blocks (rows * K / 32), pieces (0 = bf16: the data
// gradient's copy of w_d; 1 = fp16 of 2^8 * w: the split-fp16 forward's copy of w_t).  One launch for all n matrices.
int simclr_presplit_weights_multi(const long long* table, int n, long long max_blocks, hipStream_t stream) {
  SIMCLR_CHECK_ARG(table && n > 0 && max_blocks > 0, "presplit_weights_multi: empty table");
  const unsigned gx = (unsigned)min((max_blocks * 8 + 255) / 256, 4096ll);
  hipLaunchKernelGGL(presplit_rows_multi, dim3(gx, n), dim3(256), 0, stream, table, (float)(1 << F16_WSCALE_LOG2));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// ---- stem weight gradient from pre-split operands (three bf16 backward terms; see stem_wgrad_ps) ----
int simclr_presplit_packed(const void* xp, void* xq, long long npix, hipStream_t stream) {
  SIMCLR_CHECK_ARG(xp && xq && npix > 0, "presplit_packed: empty input");
  hipLaunchKernelGGL(presplit_packed, dim3((unsigned)min((npix + 255) / 256, 65536ll)), dim3(256), 0, stream, (const float4*)xp,
                     (u32x4*)xq, npix);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}
// 1 if simclr_stem_wgrad_ps has a kernel for this stem (the 7x7 / stride-2 ImageNet stem with 64 output channels)
int simclr_stem_wgrad_ps_supported(int KH, int KWP, int stride, int Cout) { return KH == 7 && KWP == 8 && stride == 2 && Cout == 64; }
static int stem_wgrad_ps_splits(long long chunks, int* cps) {
  // four two-wave workgroups per CU (two per SIMD pair), every one of them resident for the whole launch
  static const int want_env = getenv("SIMCLR_STEM_WGRAD_BLOCKS") ? atoi(getenv("SIMCLR_STEM_WGRAD_BLOCKS")) : 1024;
  const int want = (int)max(1ll, min((long long)max(8, min(want_env, 1024)), chunks));
  *cps = (int)((chunks + want - 1) / want);
  return (int)((chunks + *cps - 1) / *cps);
}
size_t simclr_stem_wgrad_ps_workspace_bytes(int V, int OH, int OW, int KH) {
  int cps;
  const long long chunks = (long long)V * OH * ((OW + 31) / 32);
  return (size_t)stem_wgrad_ps_splits(chunks, &cps) * KH * 32 * 64 * sizeof(float);
}
// xq: simclr_presplit_packed of the packed views [V][HP][WP][4]; dy_ps [V*OH*OW][64] in the pre-split block format;
// dw_kn: fp32 [KH*KWP*4][64] (what simclr_unpack_stem_dw reads); accumulate != 0: dw_kn += result.
int simclr_stem_wgrad_ps(const void* xq, const void* dy_ps, float* dw_kn, int accumulate, void* workspace, int V, int HP, int WP,
                         int OH, int OW, int Cout, int KH, int KWP, int stride, hipStream_t stream) {
  SIMCLR_CHECK_ARG(simclr_stem_wgrad_ps_supported(KH, KWP, stride, Cout), "stem_wgrad_ps: only the 7x7 / stride-2 stem with 64 output channels "
                   "(KH=%d KWP=%d stride=%d Cout=%d)", KH, KWP, stride, Cout);
  SIMCLR_CHECK_ARG((long long)V * OH * OW < (1ll << 31) && (long long)V * HP * WP * 16 < (1ll << 40), "stem_wgrad_ps: sizes overflow");
  SIMCLR_CHECK_ARG(HP >= (OH - 1) * stride + KH && WP >= (OW - 1) * stride + KWP, "stem_wgrad_ps: packed image %d x %d too small", HP, WP);
  StemWgP p = {};
  p.xq = xq; p.dy = dy_ps; p.dw = (float*)workspace; p.zero = zero_page();
  SIMCLR_CHECK_ARG(p.zero != nullptr, "stem_wgrad_ps: zero page symbol not found");
  p.HP = HP; p.WP = WP; p.OH = OH; p.OW = OW; p.segs = (OW + 31) / 32;
  const long long chunks = (long long)V * OH * p.segs;
  p.chunks = (int)chunks;
  p.splits = stem_wgrad_ps_splits(chunks, &p.chunks_per_split);
  static const int stages_env = getenv("SIMCLR_STEM_WGRAD_PS_STAGES") ? atoi(getenv("SIMCLR_STEM_WGRAD_PS_STAGES")) : 2;
  constexpr int stage_bytes = ((7 * 70 + 63) / 64 + 8) * 1024;
  if (stages_env == 3) hipLaunchKernelGGL((stem_wgrad_ps<7, 2, 3>), dim3(p.splits), dim3(128), 3 * stage_bytes, stream, p);
  else hipLaunchKernelGGL((stem_wgrad_ps<7, 2, 2>), dim3(p.splits), dim3(128), 2 * stage_bytes, stream, p);
  SIMCLR_CHECK_LAUNCH();
  const long long numel = (long long)KH * 32 * 64;
  hipLaunchKernelGGL(slab_reduce, dim3(max(1, (int)ceil_div(numel / 4, 16))), dim3(256), 0, stream, (const float*)workspace, p.splits,
                     numel, dw_kn, accumulate);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

int simclr_unpack_stem_dw(const float* src, float* dst, int KH, int KW, int Cin, int Cout, int KWP,
                          int accumulate, hipStream_t stream) {
  hipLaunchKernelGGL(unpack_stem_dw, dim3(ceil_div(KH * KW * Cin * Cout, 256)), dim3(256), 0, stream, src,
                     dst, KH, KW, Cin, Cout, KWP, accumulate);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
