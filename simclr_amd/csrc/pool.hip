// View packing, stem BN+ReLU+max-pool, global average pool, supervised-head loss and
// small elementwise helpers for gfx950.  HBM-bound kernels: 16-byte accesses along the
// contiguous channel axis, grid-stride loops.
//
// References: /root/reference/tf2/model.py:250-259 (split the 3k-channel input into k
// views and concatenate on the batch axis), /root/reference/tf2/resnet.py:602-611
// (BN+ReLU then MaxPooling2D(3, 2, 'SAME')), :693-696 (global mean over H,W),
// /root/reference/tf2/objective.py:27-32 + /root/reference/tf2/metrics.py:49-55
// (supervised softmax cross-entropy and top-1 accuracy).
#include "common.h"

namespace {

// images f32 [b][H][W][3k] -> xp T [k*b][HP][WP][4], interior at (pad, pad), zero elsewhere
template <typename T>
__global__ void pack_views(const float* __restrict__ img, T* __restrict__ xp, int b, int H, int W,
                           int k, int HP, int WP, int pad, u32x4* __restrict__ xq = nullptr) {
  const long long total = (long long)k * b * HP * WP;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int px = (int)(i % WP);
    const int py = (int)((i / WP) % HP);
    const int v = (int)(i / ((long long)WP * HP));
    const int view = v / b, n = v % b;
    const int y = py - pad, x = px - pad;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
      const float* s = img + (((long long)n * H + y) * W + x) * (3 * k) + 3 * view;
      c[0] = s[0]; c[1] = s[1]; c[2] = s[2];
    }
    if (sizeof(T) == 4) {
      *(float4*)((float*)xp + i * 4) = make_float4(c[0], c[1], c[2], c[3]);
      if (xq) {            // the same pixel as (four bf16 hi pieces, four bf16 lo pieces): simclr_presplit_packed in the same pass
        uint32_t h0, l0, h1, l1;
        split_pair<false>(c[0], c[1], h0, l0);
        split_pair<false>(c[2], c[3], h1, l1);
        xq[i] = (u32x4){h0, h1, l0, l1};
      }
    } else {
      u32x2 pk; pk[0] = pack_bf16x2(c[0], c[1]); pk[1] = pack_bf16x2(c[2], c[3]);
      *(u32x2*)((uint16_t*)xp + i * 4) = pk;
    }
  }
}

// y[v,oy,ox,c] = max over the k x k window of relu(x*scale+shift); arg = first max tap.
template <typename T>
__global__ void bnrelu_maxpool_fwd(const T* __restrict__ x, const float* __restrict__ scale,
                                   const float* __restrict__ shift, T* __restrict__ y,
                                   uint8_t* __restrict__ arg, int V, int H, int W, int C, int OH, int OW,
                                   int ksz, int stride, int pad_t, int pad_l) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = C / EPC;
  const long long total = (long long)V * OH * OW * cpr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const unsigned pix = (unsigned)(i / cpr);            // V*OH*OW < 2^31 (checked by the caller)
    const int cc = (int)(i - (long long)pix * cpr);
    const unsigned row = pix / (unsigned)OW;
    const int ox = (int)(pix - row * OW), v = (int)(row / (unsigned)OH), oy = (int)(row - (unsigned)v * OH);
    const int c0 = cc * EPC;
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e]; }
    float best[EPC];
    int bi[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    for (int ky = 0; ky < ksz; ++ky) {
      const int iy = oy * stride - pad_t + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kx = 0; kx < ksz; ++kx) {
        const int ix = ox * stride - pad_l + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        float xv[EPC];
        chunk_to_f32<T>(*(const u32x4*)(x + (((long long)v * H + iy) * W + ix) * C + c0), xv);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float a = fmaxf(fmaf(xv[e], sc[e], sh[e]), 0.f);
          if (a > best[e]) { best[e] = a; bi[e] = ky * ksz + kx; }
        }
      }
    }
    *(u32x4*)(y + (long long)pix * C + c0) = f32_to_chunk<T>(best);
    // tap ids of the EPC channels as ONE 4- or 8-byte store
    uint32_t w[EPC / 4];
#pragma unroll
    for (int q = 0; q < EPC / 4; ++q)
      w[q] = (uint32_t)bi[4 * q] | ((uint32_t)bi[4 * q + 1] << 8) | ((uint32_t)bi[4 * q + 2] << 16) | ((uint32_t)bi[4 * q + 3] << 24);
    uint32_t* ap = (uint32_t*)(arg + (long long)pix * C + c0);
#pragma unroll
    for (int q = 0; q < EPC / 4; ++q) ap[q] = w[q];
  }
}

// dx[v,iy,ix,c] = sum over windows containing (iy,ix) whose argmax is this tap of dy
template <typename T>
__global__ void maxpool_bwd(const T* __restrict__ dy, const uint8_t* __restrict__ arg,
                            T* __restrict__ dx, int V, int H, int W, int C, int OH, int OW, int ksz,
                            int stride, int pad_t, int pad_l) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = C / EPC;
  const long long total = (long long)V * H * W * cpr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const unsigned pix = (unsigned)(i / cpr);            // V*H*W < 2^31 (checked by the caller)
    const int cc = (int)(i - (long long)pix * cpr);
    const unsigned row = pix / (unsigned)W;
    const int ix = (int)(pix - row * W), v = (int)(row / (unsigned)H), iy = (int)(row - (unsigned)v * H);
    const int c0 = cc * EPC;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    // windows that contain (iy, ix): oy*stride - pad_t <= iy <= oy*stride - pad_t + ksz - 1  (at most
    // ceil(ksz/stride)^2 of them -- 4 for the 3x3 stride-2 stem pool), enumerated directly
    const int ty = iy + pad_t, tx = ix + pad_l;
    if (ksz == 3 && stride == 2) {
      // the stem pool: at most 2 x 2 windows contain a pixel.  All four candidates are requested back to back from
      // clamped addresses (8 loads in flight per thread) and the invalid ones are masked afterwards -- the loop below
      // issues its loads one dependent round trip at a time.  Same order of additions: (hi,hi) (hi,lo) (lo,hi) (lo,lo).
      const int oy1 = ty >> 1, ox1 = tx >> 1;
      int oys[2] = {oy1, oy1 - 1}, oxs[2] = {ox1, ox1 - 1};
      // (the second candidate carries the same upper bound as the generic loop's clamp: a caller-supplied OH / OW smaller than
      // the pooled geometry excludes it here exactly as it does there)
      bool vy[2] = {oy1 <= OH - 1, oy1 >= 1 && oy1 - 1 <= OH - 1 && (ty & 1) == 0}, vx[2] = {ox1 <= OW - 1, ox1 >= 1 && ox1 - 1 <= OW - 1 && (tx & 1) == 0};
      u32x4 dv[4];
      uint32_t av[4][EPC / 4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const int oy = min(max(oys[a], 0), OH - 1), ox = min(max(oxs[b], 0), OW - 1);
          const long long op = (((long long)v * OH + oy) * OW + ox) * C + c0;
          dv[a * 2 + b] = *(const u32x4*)(dy + op);
          const uint32_t* ap = (const uint32_t*)(arg + op);
#pragma unroll
          for (int q = 0; q < EPC / 4; ++q) av[a * 2 + b][q] = ap[q];
        }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (!(vy[a] && vx[b])) continue;
          const uint32_t tapid = (uint32_t)((ty - oys[a] * 2) * 3 + (tx - oxs[b] * 2));
          float d[EPC];
          chunk_to_f32<T>(dv[a * 2 + b], d);
#pragma unroll
          for (int q = 0; q < EPC / 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (((av[a * 2 + b][q] >> (8 * e)) & 0xffu) == tapid) acc[4 * q + e] += d[4 * q + e];
        }
      *(u32x4*)(dx + (long long)pix * C + c0) = f32_to_chunk<T>(acc);
      continue;
    }
    const int oy_hi = min(OH - 1, ty / stride), ox_hi = min(OW - 1, tx / stride);
    const int oy_lo = max(0, (ty - ksz + stride) / stride), ox_lo = max(0, (tx - ksz + stride) / stride);
    for (int oy = oy_hi; oy >= oy_lo; --oy) {          // ky ascending, kx ascending: fixed summation order
      const int ky = ty - oy * stride;
      for (int ox = ox_hi; ox >= ox_lo; --ox) {
        const int kx = tx - ox * stride;
        const long long op = (((long long)v * OH + oy) * OW + ox) * C + c0;
        float d[EPC];
        chunk_to_f32<T>(*(const u32x4*)(dy + op), d);
        const uint32_t tapid = (uint32_t)(ky * ksz + kx);
        const uint32_t* ap = (const uint32_t*)(arg + op);       // EPC tap ids: one 4-byte load per 4 channels
#pragma unroll
        for (int q = 0; q < EPC / 4; ++q) {
          const uint32_t w = ap[q];
#pragma unroll
          for (int b = 0; b < 4; ++b)
            if (((w >> (8 * b)) & 0xffu) == tapid) acc[4 * q + b] += d[4 * q + b];
        }
      }
    }
    *(u32x4*)(dx + (long long)pix * C + c0) = f32_to_chunk<T>(acc);
  }
}

// Gradient wrt the ReLU output at input pixel (v, iy, ix), channels c0..c0+EPC-1, gathered from the pooled gradient and
// the arg-max taps (the body of maxpool_bwd): the (at most ceil(k/s)^2) windows containing the pixel, fixed order.
template <typename T>
__device__ __forceinline__ void maxpool_gather(const T* __restrict__ dy, const uint8_t* __restrict__ arg, int v, int iy, int ix,
                                               int c0, int C, int OH, int OW, int ksz, int stride, int pad_t, int pad_l,
                                               float* acc) {
  constexpr int EPC = Elem<T>::EPC;
#pragma unroll
  for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
  const int ty = iy + pad_t, tx = ix + pad_l;
  if (ksz == 3 && stride == 2) {
    // the stem pool, as in maxpool_bwd: the (at most) 2 x 2 windows are requested back to back from clamped addresses and the
    // invalid ones masked afterwards; same windows, same order of additions as the loop below
    const int oy1 = ty >> 1, ox1 = tx >> 1;
    const int oys[2] = {oy1, oy1 - 1}, oxs[2] = {ox1, ox1 - 1};
    const bool vy[2] = {oy1 <= OH - 1, oy1 >= 1 && oy1 - 1 <= OH - 1 && (ty & 1) == 0};
    const bool vx[2] = {ox1 <= OW - 1, ox1 >= 1 && ox1 - 1 <= OW - 1 && (tx & 1) == 0};
    u32x4 dv[4];
    uint32_t av[4][EPC / 4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int oy = min(max(oys[a], 0), OH - 1), ox = min(max(oxs[b], 0), OW - 1);
        const long long op = (((long long)v * OH + oy) * OW + ox) * C + c0;
        dv[a * 2 + b] = *(const u32x4*)(dy + op);
        const uint32_t* ap = (const uint32_t*)(arg + op);
#pragma unroll
        for (int q = 0; q < EPC / 4; ++q) av[a * 2 + b][q] = ap[q];
      }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const bool ok = vy[a] && vx[b];
        const uint32_t tapid = (uint32_t)((ty - oys[a] * 2) * 3 + (tx - oxs[b] * 2));
        float d[EPC];
        chunk_to_f32<T>(dv[a * 2 + b], d);
#pragma unroll
        for (int q = 0; q < EPC / 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (ok && ((av[a * 2 + b][q] >> (8 * e)) & 0xffu) == tapid) acc[4 * q + e] += d[4 * q + e];
      }
    return;
  }
  const int oy_hi = min(OH - 1, ty / stride), ox_hi = min(OW - 1, tx / stride);
  const int oy_lo = max(0, (ty - ksz + stride) / stride), ox_lo = max(0, (tx - ksz + stride) / stride);
  for (int oy = oy_hi; oy >= oy_lo; --oy) {
    const int ky = ty - oy * stride;
    for (int ox = ox_hi; ox >= ox_lo; --ox) {
      const int kx = tx - ox * stride;
      const long long op = (((long long)v * OH + oy) * OW + ox) * C + c0;
      float d[EPC];
      chunk_to_f32<T>(*(const u32x4*)(dy + op), d);
      const uint32_t tapid = (uint32_t)(ky * ksz + kx);
      const uint32_t* ap = (const uint32_t*)(arg + op);
#pragma unroll
      for (int q = 0; q < EPC / 4; ++q) {
        const uint32_t w = ap[q];
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (((w >> (8 * b)) & 0xffu) == tapid) acc[4 * q + b] += d[4 * q + b];
      }
    }
  }
}

// Stem backward without the un-pooled gradient tensor: BatchNorm-backward REDUCE straight from the pooled gradient.
// partial[slot][2][C] = (sum dm, sum dm * x^) with dm = maxpool_bwd(dy)[pixel] * (x*scale+shift > 0); one slot per
// workgroup (plain stores: deterministic).  Same thread layout as bn_bwd_reduce: C/EPC channel chunks x row lanes.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool(
    const T* __restrict__ dy, const uint8_t* __restrict__ arg, const T* __restrict__ x, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd, int V, int H, int W,
    int C, int OH, int OW, int ksz, int stride, int pad_t, int pad_l, int rows_per_block, float* __restrict__ partial) {
  constexpr int EPC = Elem<T>::EPC;
  __shared__ float red[256 * 2 * EPC];
  const int cpr = C / EPC;                     // <= 256 (checked by the host)
  const int rl = 256 / cpr;
  const int tcol = threadIdx.x % cpr, trow = threadIdx.x / cpr;
  const long long rows = (long long)V * H * W;
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  const int c0 = tcol * EPC;
  float s1[EPC], s2[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (trow < rl) {
    float mu[EPC], rs[EPC], sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { mu[e] = mean[c0 + e]; rs[e] = rstd[c0 + e]; sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e]; }
    for (long long r = r0 + trow; r < r1; r += rl) {
      const unsigned ru = (unsigned)r;                       // V*H*W < 2^31 (checked by the host): 32-bit divisions
      const unsigned row = ru / (unsigned)W;
      const int ix = (int)(ru - row * (unsigned)W), v = (int)(row / (unsigned)H), iy = (int)(row - (unsigned)v * H);
      float d[EPC], xf[EPC];
      maxpool_gather<T>(dy, arg, v, iy, ix, c0, C, OH, OW, ksz, stride, pad_t, pad_l, d);
      chunk_to_f32<T>(*(const u32x4*)(x + r * C + c0), xf);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float dm = fmaf(xf[e], sc[e], sh[e]) > 0.f ? d[e] : 0.f;
        s1[e] += dm;
        s2[e] += dm * (xf[e] - mu[e]) * rs[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    red[(threadIdx.x * EPC + e) * 2] = s1[e];
    red[(threadIdx.x * EPC + e) * 2 + 1] = s2[e];
  }
  __syncthreads();
  if (trow == 0) {
    float* slot = partial + (long long)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float a = 0.f, b = 0.f;
      for (int q = 0; q < rl; ++q) {
        a += red[((q * cpr + tcol) * EPC + e) * 2];
        b += red[((q * cpr + tcol) * EPC + e) * 2 + 1];
      }
      slot[c0 + e] = a;
      slot[C + c0 + e] = b;
    }
  }
}

// The same sums walked over the POOLED pixels (round 6): sum_p dm[p] = sum_o dy[o] * [bn(x[arg(o)]) > 0] -- every pooled element names the
// one input pixel its gradient goes to, so instead of gathering (at most) four windows per input element (nine loads per 16 bytes of x,
// 3.2 TB/s) a thread reads one pooled chunk, its tap ids, and the EPC input values the taps point at (scattered 4-byte loads inside a
// 3 x 3 neighbourhood: the cache lines of x are still touched about once).  Same slots, same finalize; the order of the additions differs.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_out(
    const T* __restrict__ dy, const uint8_t* __restrict__ arg, const T* __restrict__ x, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd, int V, int H, int W,
    int C, int OH, int OW, int ksz, int stride, int pad_t, int pad_l, int rows_per_block, float* __restrict__ partial) {
  constexpr int EPC = Elem<T>::EPC;
  __shared__ float red[256 * 2 * EPC];
  const int cpr = C / EPC;
  const int rl = 256 / cpr;
  const int tcol = threadIdx.x % cpr, trow = threadIdx.x / cpr;
  const long long rows = (long long)V * OH * OW;
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  const int c0 = tcol * EPC;
  float s1[EPC], s2[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (trow < rl) {
    float mu[EPC], rs[EPC], sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { mu[e] = mean[c0 + e]; rs[e] = rstd[c0 + e]; sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e]; }
    for (long long r = r0 + trow; r < r1; r += rl) {
      const unsigned ru = (unsigned)r;
      const unsigned orow = ru / (unsigned)OW;
      const int ox = (int)(ru - orow * (unsigned)OW), v = (int)(orow / (unsigned)OH), oy = (int)(orow - (unsigned)v * OH);
      float d[EPC];
      chunk_to_f32<T>(*(const u32x4*)(dy + r * C + c0), d);
      uint32_t av[EPC / 4];
#pragma unroll
      for (int q = 0; q < EPC / 4; ++q) av[q] = ((const uint32_t*)(arg + r * C + c0))[q];
      const int by = oy * stride - pad_t, bx = ox * stride - pad_l;
      float xf[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int tap = (int)((av[e >> 2] >> (8 * (e & 3))) & 0xffu);
        const int ky = tap / ksz, kx = tap - ky * ksz;
        const int iy = min(max(by + ky, 0), H - 1), ix = min(max(bx + kx, 0), W - 1);      // (a tap id always names a real pixel; clamped all the same)
        xf[e] = Elem<T>::ld(x + (((long long)v * H + iy) * W + ix) * C + c0 + e);
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float dm = fmaf(xf[e], sc[e], sh[e]) > 0.f ? d[e] : 0.f;
        s1[e] += dm;
        s2[e] += dm * (xf[e] - mu[e]) * rs[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    red[(threadIdx.x * EPC + e) * 2] = s1[e];
    red[(threadIdx.x * EPC + e) * 2 + 1] = s2[e];
  }
  __syncthreads();
  if (trow == 0) {
    float* slot = partial + (long long)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float a = 0.f, b = 0.f;
      for (int q = 0; q < rl; ++q) {
        a += red[((q * cpr + tcol) * EPC + e) * 2];
        b += red[((q * cpr + tcol) * EPC + e) * 2 + 1];
      }
      slot[c0 + e] = a;
      slot[C + c0 + e] = b;
    }
  }
}

// ... and the APPLY: dx = scale * (dm - c1 - x^ * c2), dm gathered / masked as above
// PSO (fp32, C % 32 == 0): dx in the pre-split block format with bf16 pieces (common.h) -- its only consumer is the stem's weight gradient
template <typename T, bool PSO = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_pool(
    const T* __restrict__ dy, const uint8_t* __restrict__ arg, const T* __restrict__ x, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ c1, const float* __restrict__ c2, T* __restrict__ dx, int V, int H, int W, int C, int OH,
    int OW, int ksz, int stride, int pad_t, int pad_l) {
  constexpr int EPC = Elem<T>::EPC;
  static_assert(!PSO || sizeof(T) == 4, "pre-split output: fp32 storage");
  const int cpr = C / EPC;
  const bool pow2 = (cpr & (cpr - 1)) == 0;
  const int cshift = __ffs(cpr) - 1;
  const long long total = (long long)V * H * W * cpr;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256ll) {
    const unsigned pix = pow2 ? (unsigned)(i >> cshift) : (unsigned)(i / cpr);
    const int c0 = (int)(i - (long long)pix * cpr) * EPC;
    const unsigned row = pix / (unsigned)W;
    const int ix = (int)(pix - row * W), v = (int)(row / (unsigned)H), iy = (int)(row - (unsigned)v * H);
    float d[EPC], xf[EPC], o[EPC];
    maxpool_gather<T>(dy, arg, v, iy, ix, c0, C, OH, OW, ksz, stride, pad_t, pad_l, d);
    chunk_to_f32<T>(*(const u32x4*)(x + (long long)pix * C + c0), xf);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float sc = scale[c0 + e];
      const float dm = fmaf(xf[e], sc, shift[c0 + e]) > 0.f ? d[e] : 0.f;
      const float xh = (xf[e] - mean[c0 + e]) * rstd[c0 + e];
      o[e] = sc * (dm - c1[c0 + e] - xh * c2[c0 + e]);
    }
    if constexpr (PSO) ps_store_quad<false>(dx, i, o);           // quad i = channels c0 .. c0 + 3 of pixel pix
    else *(u32x4*)(dx + (long long)pix * C + c0) = f32_to_chunk<T>(o);
  }
}

// The APPLY with a thread owning a 2 x 2 block of input pixels (3 x 3 stride-2 pooling only; round 6): in padded coordinates t = i + pad
// the pixels (2 by + {0, 1}, 2 bx + {0, 1}) are covered by the SAME (at most) four windows {by - 1, by} x {bx - 1, bx} -- the (even, even)
// pixel by all four, the (even, odd) / (odd, even) ones by two, the (odd, odd) one by window (by, bx) alone -- so the four (gradient,
// tap id) pairs are loaded once for four input pixels instead of once per pixel (2 + 1 loads per 16 bytes of x instead of 8 + 1).
// Per pixel the windows are visited in maxpool_gather's order, so every dx value is bit for bit the gathered kernel's.
template <typename T, bool PSO>
__global__ __launch_bounds__(256) void bn_bwd_apply_pool2(
    const T* __restrict__ dy, const uint8_t* __restrict__ arg, const T* __restrict__ x, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ c1, const float* __restrict__ c2, T* __restrict__ dx, int V, int H, int W, int C, int OH,
    int OW, int pad_t, int pad_l, int TBY, int TBX) {
  constexpr int EPC = Elem<T>::EPC;
  static_assert(!PSO || sizeof(T) == 4, "pre-split output: fp32 storage");
  const int cpr = C / EPC;
  const long long total = (long long)V * TBY * TBX * cpr;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256ll) {
    const int cc = (int)(i % cpr);
    const unsigned blk = (unsigned)(i / cpr);
    const unsigned brow = blk / (unsigned)TBX;
    const int bx = (int)(blk - brow * TBX), v = (int)(brow / (unsigned)TBY), by = (int)(brow - (unsigned)v * TBY);
    const int c0 = cc * EPC;
    // the four windows, requested back to back from clamped addresses (invalid ones masked below): index a * 2 + b = (by - a, bx - b)
    u32x4 dv[4];
    uint32_t av[4][EPC / 4];
    bool wok[4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int oy = by - a, ox = bx - b;
        wok[a * 2 + b] = oy >= 0 && oy < OH && ox >= 0 && ox < OW;
        const int oyc = min(max(oy, 0), OH - 1), oxc = min(max(ox, 0), OW - 1);
        const long long op = (((long long)v * OH + oyc) * OW + oxc) * C + c0;
        dv[a * 2 + b] = *(const u32x4*)(dy + op);
        const uint32_t* ap = (const uint32_t*)(arg + op);
#pragma unroll
        for (int q = 0; q < EPC / 4; ++q) av[a * 2 + b][q] = ap[q];
      }
    float dwin[4][EPC];
#pragma unroll
    for (int w = 0; w < 4; ++w) chunk_to_f32<T>(dv[w], dwin[w]);
    float sc[EPC], sh[EPC], mu[EPC], rs[EPC], k1[EPC], k2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e]; mu[e] = mean[c0 + e]; rs[e] = rstd[c0 + e]; k1[e] = c1[c0 + e]; k2[e] = c2[c0 + e]; }
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int iy = 2 * by + py - pad_t, ix = 2 * bx + px - pad_l;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        float d[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) d[e] = 0.f;
        // maxpool_gather's order: (oy1, ox1), (oy1, ox1 - 1), (oy1 - 1, ox1), (oy1 - 1, ox1 - 1) with oy1 = by, ox1 = bx; the windows
        // one row / column back only reach the even pixel of the pair
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            if ((a == 1 && py == 1) || (b == 1 && px == 1)) continue;
            const bool ok = wok[a * 2 + b];
            const uint32_t tapid = (uint32_t)((py + 2 * a) * 3 + (px + 2 * b));
#pragma unroll
            for (int q = 0; q < EPC / 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (ok && ((av[a * 2 + b][q] >> (8 * e)) & 0xffu) == tapid) d[4 * q + e] += dwin[a * 2 + b][4 * q + e];
          }
        const long long pix = ((long long)v * H + iy) * W + ix;
        float xf[EPC], o[EPC];
        chunk_to_f32<T>(*(const u32x4*)(x + pix * C + c0), xf);
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float dm = fmaf(xf[e], sc[e], sh[e]) > 0.f ? d[e] : 0.f;
          const float xh = (xf[e] - mu[e]) * rs[e];
          o[e] = sc[e] * (dm - k1[e] - xh * k2[e]);
        }
        if constexpr (PSO) ps_store_quad<false>(dx, pix * cpr + cc, o);
        else *(u32x4*)(dx + pix * C + c0) = f32_to_chunk<T>(o);
      }
  }
}

// y[v][c] = mean over HW of x[v][hw][c]
// y32 (nullable): also / instead write the fp32 means (the heads may run in fp32 on top of a bf16 encoder: rounding the
// mean of HW bf16 values back to bf16 would throw away the sqrt(HW) averaging gain right before the head's BatchNorm)
template <typename T>
__global__ void global_avgpool_fwd(const T* __restrict__ x, T* __restrict__ y, float* __restrict__ y32, int V, int HW, int C) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = C / EPC;
  const long long total = (long long)V * cpr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr);
    const int v = (int)(i / cpr);
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int p = 0; p < HW; ++p) {
      float xv[EPC];
      chunk_to_f32<T>(*(const u32x4*)(x + ((long long)v * HW + p) * C + cc * EPC), xv);
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += xv[e];
    }
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] *= inv;
    if (y) *(u32x4*)(y + (long long)v * C + cc * EPC) = f32_to_chunk<T>(acc);
    if (y32) {
#pragma unroll
      for (int e = 0; e < EPC; e += 4)
        *(float4*)(y32 + (long long)v * C + cc * EPC + e) = make_float4(acc[e], acc[e + 1], acc[e + 2], acc[e + 3]);
    }
  }
}
// dx[v][hw][c] = (dy[v][c] [+ dy2[v][c]]) / HW, optionally masked by mask_src > 0
template <typename T>
__global__ void global_avgpool_bwd(const T* __restrict__ dy, const T* __restrict__ mask_src,
                                   T* __restrict__ dx, int V, int HW, int C) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = C / EPC;
  const long long total = (long long)V * HW * cpr;
  const float inv = 1.f / (float)HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr);
    const int v = (int)(i / ((long long)cpr * HW));
    float d[EPC];
    chunk_to_f32<T>(*(const u32x4*)(dy + (long long)v * C + cc * EPC), d);
    if (mask_src) {
      float mk[EPC];
      chunk_to_f32<T>(*(const u32x4*)(mask_src + i * EPC), mk);
#pragma unroll
      for (int e = 0; e < EPC; ++e) d[e] = mk[e] > 0.f ? d[e] * inv : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < EPC; ++e) d[e] *= inv;
    }
    *(u32x4*)(dx + i * EPC) = f32_to_chunk<T>(d);
  }
}

// ---- batch_random_blur (tf2/data_util.py:323-361, 413-440; called on device from tf2/model.py:255-258).
// One 1-D pass of the separable depthwise Gaussian (SAME zero padding) over a float32 [b,H,W,C] batch;
// channel c belongs to view c/3, which has its own filter (one sigma per view per batch) and a per-image
// 0/1 selector.  Unselected images are copied.  CLIP: clip_by_value(., 0, 1) on the last pass.
// One thread per (image, pixel, view): 3 channels, taps along x (HORIZONTAL pass, only for selected
// images, writes tmp) or along y (VERTICAL pass: reads tmp for selected images, the input itself for
// unselected ones, clips and writes the result) -- unselected images are touched once.
template <bool VERT>
__global__ void blur1d(const float* __restrict__ in, const float* __restrict__ tmp, float* __restrict__ out,
                       const float* __restrict__ filt, const float* __restrict__ selector, int b, int H, int W,
                       int nviews, int K) {
  const int C = 3 * nviews;
  const long long total = (long long)b * H * W * nviews;
  const int r = K / 2;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int view = (int)(i % nviews);
    const long long pix = i / nviews;
    const int x = (int)(pix % W);
    const int y = (int)((pix / W) % H);
    const int n = (int)(pix / ((long long)W * H));
    const bool sel = selector[view * b + n] != 0.f;
    const long long e = pix * C + 3 * view;
    float a0, a1, a2;
    if (!sel) {
      if (!VERT) continue;                       // horizontal pass skips unselected images
      a0 = in[e]; a1 = in[e + 1]; a2 = in[e + 2];
    } else {
      const float* src = VERT ? tmp : in;
      const float* f = filt + view * K;
      a0 = a1 = a2 = 0.f;
      const long long step = VERT ? (long long)W * C : C;
      const int pos = VERT ? y : x, lim = VERT ? H : W;
      const int t0 = max(0, r - pos), t1 = min(K, lim + r - pos);
      const float* q = src + e + (long long)(t0 - r) * step;
      for (int t = t0; t < t1; ++t, q += step) {
        const float w = f[t];
        a0 = fmaf(w, q[0], a0); a1 = fmaf(w, q[1], a1); a2 = fmaf(w, q[2], a2);
      }
    }
    if (VERT) {                                  // clip_by_value(., 0, 1), tf2/data_util.py:437
      a0 = fminf(fmaxf(a0, 0.f), 1.f); a1 = fminf(fmaxf(a1, 0.f), 1.f); a2 = fminf(fmaxf(a2, 0.f), 1.f);
      out[e] = a0; out[e + 1] = a1; out[e + 2] = a2;
    } else {
      out[e] = a0; out[e + 1] = a1; out[e + 2] = a2;
    }
  }
}

// ---- ResNet-D shortcut: AveragePooling2D(2, strides, SAME if strides==1 else VALID after FixedPadding(2))
// (tf2/resnet.py:330-338, 400-408).  stride 2: windows rows 2oy..2oy+1 (the (0,1) zero pad is part of
// the tensor -> divisor always 4); stride 1 SAME: pad (0,1), TF divides by the number of VALID cells.
template <typename T>
__global__ void avgpool2_fwd(const T* __restrict__ x, T* __restrict__ y, int V, int H, int W, int C, int OH,
                             int OW, int stride) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = C / EPC;
  const long long total = (long long)V * OH * OW * cpr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr);
    const long long pix = i / cpr;
    const int ox = (int)(pix % OW), oy = (int)((pix / OW) % OH), v = (int)(pix / ((long long)OW * OH));
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    int cnt = 0;
    for (int ky = 0; ky < 2; ++ky)
      for (int kx = 0; kx < 2; ++kx) {
        const int iy = oy * stride + ky, ix = ox * stride + kx;
        if (iy < H && ix < W) {
          float xv[EPC];
          chunk_to_f32<T>(*(const u32x4*)(x + (((long long)v * H + iy) * W + ix) * C + cc * EPC), xv);
#pragma unroll
          for (int e = 0; e < EPC; ++e) acc[e] += xv[e];
          ++cnt;
        }
      }
    const float inv = 1.f / (float)(stride == 1 ? cnt : 4);
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] *= inv;
    *(u32x4*)(y + pix * C + cc * EPC) = f32_to_chunk<T>(acc);
  }
}
template <typename T>
__global__ void avgpool2_bwd(const T* __restrict__ dy, T* __restrict__ dx, int V, int H, int W, int C, int OH,
                             int OW, int stride) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = C / EPC;
  const long long total = (long long)V * H * W * cpr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr);
    const long long pix = i / cpr;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), v = (int)(pix / ((long long)W * H));
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int ky = 0; ky < 2; ++ky)
      for (int kx = 0; kx < 2; ++kx) {
        const int ty = iy - ky, tx = ix - kx;
        if (ty < 0 || tx < 0 || ty % stride || tx % stride) continue;
        const int oy = ty / stride, ox = tx / stride;
        if (oy >= OH || ox >= OW) continue;
        float w;
        if (stride == 1) {
          const int cnt = ((oy + 1 < H) ? 2 : 1) * ((ox + 1 < W) ? 2 : 1);
          w = 1.f / (float)cnt;
        } else {
          w = 0.25f;
        }
        float d[EPC];
        chunk_to_f32<T>(*(const u32x4*)(dy + (((long long)v * OH + oy) * OW + ox) * C + cc * EPC), d);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += d[e] * w;
      }
    *(u32x4*)(dx + pix * C + cc * EPC) = f32_to_chunk<T>(acc);
  }
}

// ---- Selective-kernel unit (tf2/resnet.py:266-277).  a = [V, HW, 2f]: stream k occupies channels
// [k*f, (k+1)*f).  g[v,c] = mean_hw(a0+a1);  mix = softmax over the two streams of l[v, k*f+c];
// out = a0*m0 + a1*m1.
template <typename T>
__global__ void sk_pool_fwd(const T* __restrict__ a, T* __restrict__ gout, int V, int HW, int f, int gpitch) {
  // one thread per (v, 4-channel group); gout [V, gpitch] (pad channels zero-filled by the caller)
  const long long total = (long long)V * (f / 4);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % (f / 4)) * 4;
    const int v = (int)(i / (f / 4));
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < HW; ++p) {
      const T* r = a + ((long long)v * HW + p) * 2 * f;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += Elem<T>::ld(r + c0 + e) + Elem<T>::ld(r + f + c0 + e);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) Elem<T>::st(gout + (long long)v * gpitch + c0 + e, acc[e] / (float)HW);
  }
}
template <typename T>
__global__ void sk_mix_fwd(const T* __restrict__ a, const T* __restrict__ l, T* __restrict__ out, int V, int HW,
                           int f, int lpitch) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = f / EPC;
  const long long total = (long long)V * HW * cpr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr);
    const long long pix = i / cpr;
    const int v = (int)(pix / HW);
    float a0[EPC], a1[EPC], o[EPC];
    chunk_to_f32<T>(*(const u32x4*)(a + pix * 2 * f + cc * EPC), a0);
    chunk_to_f32<T>(*(const u32x4*)(a + pix * 2 * f + f + cc * EPC), a1);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float l0 = Elem<T>::ld(l + (long long)v * lpitch + cc * EPC + e);
      const float l1 = Elem<T>::ld(l + (long long)v * lpitch + f + cc * EPC + e);
      const float m0 = 1.f / (1.f + __expf(l1 - l0));
      o[e] = a0[e] * m0 + a1[e] * (1.f - m0);
    }
    *(u32x4*)(out + pix * f + cc * EPC) = f32_to_chunk<T>(o);
  }
}
// dl[v, k*f+c] = softmax-backward of (sum_hw dout*a0, sum_hw dout*a1); one thread per (v, c)
template <typename T>
__global__ void sk_mix_bwd_logits(const T* __restrict__ a, const T* __restrict__ l, const T* __restrict__ dout,
                                  T* __restrict__ dl, int V, int HW, int f, int lpitch) {
  const long long total = (long long)V * f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % f);
    const int v = (int)(i / f);
    float d0 = 0.f, d1 = 0.f;
    for (int p = 0; p < HW; ++p) {
      const long long pix = (long long)v * HW + p;
      const float d = Elem<T>::ld(dout + pix * f + c);
      d0 += d * Elem<T>::ld(a + pix * 2 * f + c);
      d1 += d * Elem<T>::ld(a + pix * 2 * f + f + c);
    }
    const float l0 = Elem<T>::ld(l + (long long)v * lpitch + c), l1 = Elem<T>::ld(l + (long long)v * lpitch + f + c);
    const float m0 = 1.f / (1.f + __expf(l1 - l0)), m1 = 1.f - m0;
    const float dot = m0 * d0 + m1 * d1;
    Elem<T>::st(dl + (long long)v * lpitch + c, m0 * (d0 - dot));
    Elem<T>::st(dl + (long long)v * lpitch + f + c, m1 * (d1 - dot));
  }
}
// da[v,hw,k*f+c] = dout*m_k + dg[v,c]/HW   (dg = gradient wrt the pooled feature g)
template <typename T>
__global__ void sk_mix_bwd_streams(const T* __restrict__ l, const T* __restrict__ dout, const T* __restrict__ dg,
                                   T* __restrict__ da, int V, int HW, int f, int lpitch, int gpitch) {
  constexpr int EPC = Elem<T>::EPC;
  const int cpr = f / EPC;
  const long long total = (long long)V * HW * cpr;
  const float inv = 1.f / (float)HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cpr);
    const long long pix = i / cpr;
    const int v = (int)(pix / HW);
    float d[EPC], o0[EPC], o1[EPC];
    chunk_to_f32<T>(*(const u32x4*)(dout + pix * f + cc * EPC), d);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float l0 = Elem<T>::ld(l + (long long)v * lpitch + cc * EPC + e);
      const float l1 = Elem<T>::ld(l + (long long)v * lpitch + f + cc * EPC + e);
      const float m0 = 1.f / (1.f + __expf(l1 - l0));
      const float gq = Elem<T>::ld(dg + (long long)v * gpitch + cc * EPC + e) * inv;
      o0[e] = d[e] * m0 + gq;
      o1[e] = d[e] * (1.f - m0) + gq;
    }
    *(u32x4*)(da + pix * 2 * f + cc * EPC) = f32_to_chunk<T>(o0);
    *(u32x4*)(da + pix * 2 * f + f + cc * EPC) = f32_to_chunk<T>(o1);
  }
}

// Supervised head tail: logits = z + bias; softmax CE vs int labels (mean over rows);
// dlogits = (softmax - onehot) * gscale / rows ; padded classes (>= nclass) get 0.
// One wave per row.  out[0] += loss contribution, out[1] += top-1 hits  (atomics; caller zeroes)
template <typename T>
__global__ void bias_softmax_xent(const T* __restrict__ z, const float* __restrict__ bias,
                                  const int* __restrict__ labels, int rows, int label_rows, int nclass,
                                  int cpad, float gscale, T* __restrict__ dlogits, float* __restrict__ out) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int label = labels[row % label_rows];   // labels duplicated per view (tf2/run.py:599-600)
  const T* zr = z + (long long)row * cpad;
  float mx = -INFINITY; int am = 0x7fffffff;
  for (int c = lane; c < nclass; c += 64) {
    const float v = Elem<T>::ld(zr + c) + bias[c];
    if (v > mx) { mx = v; am = c; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(mx, o, 64); const int oa = __shfl_xor(am, o, 64);
    if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
  }
  float se = 0.f, zl = 0.f;
  for (int c = lane; c < nclass; c += 64) {
    const float v = Elem<T>::ld(zr + c) + bias[c];
    se += __expf(v - mx);
    if (c == label) zl = v;
  }
  se = wave_sum(se);
  zl = wave_sum(zl);
  const float lse = mx + __logf(se);
  const float g = gscale / (float)rows;
  T* dr = dlogits + (long long)row * cpad;
  for (int c = lane; c < cpad; c += 64) {
    float d = 0.f;
    if (c < nclass) {
      const float v = Elem<T>::ld(zr + c) + bias[c];
      d = (__expf(v - lse) - (c == label ? 1.f : 0.f)) * g;
    }
    Elem<T>::st(dr + c, d);
  }
  if (lane == 0) {
    atomicAdd(out, (lse - zl) / (float)rows);
    atomicAdd(out + 1, (am == label ? 1.f : 0.f) / (float)rows);
  }
}

// out[c] (+)= sum over rows of x[row][c]   (bias gradient)
template <typename T>
__global__ __launch_bounds__(256) void colsum(const T* __restrict__ x, int rows, int C, int cvalid, float* __restrict__ out,
                                              int accumulate) {
  // 16 columns x 16 row-lanes per workgroup: lane rl adds rows rl, rl+16, ... (ascending); fixed-order join -> deterministic
  __shared__ float sh[256];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float a = 0.f;
  if (c < cvalid) {
#pragma unroll 4
    for (int r = rl; r < rows; r += 16) a += Elem<T>::ld(x + (long long)r * C + c);
  }
  sh[threadIdx.x] = a;
  __syncthreads();
  if (rl == 0 && c < cvalid) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q * 16 + cl];
    out[c] = accumulate ? out[c] + t : t;
  }
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    Elem<TO>::st(y + i, Elem<TI>::ld(x + i));
}

// y = a*x + y  (fp32) ; also returns nothing.  Used for the sup-head L2 term (tf2/model.py:49-60)
// dst[i] += scale[i] * *src[i] for up to 16 device scalars in one launch, and optionally total[0] = the sum of the scaled
// terms whose bit is set in total_mask: the running metric sums and the total loss of a training step
// (tf2/run.py:587-613, tf2/metrics.py) without one tiny elementwise launch per scalar.
struct ScalarsP {
  const float* src[16];
  float scale[16];
};
__global__ void accumulate_scalars(const ScalarsP p, int n, float* __restrict__ dst, float* __restrict__ total,
                                   unsigned total_mask, float* __restrict__ copy) {
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < n; ++i) {
      const float v = p.scale[i] * *p.src[i];
      if (dst) dst[i] += v;
      if (copy) copy[i] = v;
      if ((total_mask >> i) & 1u) t += v;
    }
    if (total) total[0] = t;
  }
}

__global__ void axpy_f32(float a, const float* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    y[i] = fmaf(a, x[i], y[i]);
}
// out[0] += 0.5 * sum x^2  (tf.nn.l2_loss); out pre-zeroed; fp64 per-block partials
__global__ __launch_bounds__(256) void l2_loss_f32(const float* __restrict__ x, long long n,
                                                   float* __restrict__ out) {
  __shared__ double sh[256];
  double a = 0.0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) a += (double)x[i] * x[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(out, (float)(0.5 * sh[0]));
}

// one element group per thread whenever possible: a linear sweep in dispatch order streams ~40 % faster than
// a capped grid striding through the tensor (tools/probes/hbm_probe.hip); the kernels keep their grid-stride
// loops for tensors beyond 2^30 threads
int grid_for(long long n) {
  constexpr long long cap = 1ll << 22;
  return (int)max(1ll, min(cap, (n + 255) / 256));
}

}  // namespace

#define DISPATCH_T(dtype, EXPR_BF16, EXPR_F32) \
  do { if ((dtype) == SIMCLR_DT_BF16) { EXPR_BF16; } else { EXPR_F32; } } while (0)

extern "C" {

int simclr_pack_views(const float* images, void* xp, int b, int H, int W, int k, int HP, int WP, int pad,
                      int dtype, hipStream_t stream) {
  SIMCLR_CHECK_ARG(b > 0 && k > 0 && HP >= H + pad && WP >= W + pad, "pack_views: bad geometry");
  const long long total = (long long)k * b * HP * WP;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((pack_views<uint16_t>), dim3(grid_for(total)), dim3(256), 0, stream, images,
                                (uint16_t*)xp, b, H, W, k, HP, WP, pad),
             hipLaunchKernelGGL((pack_views<float>), dim3(grid_for(total)), dim3(256), 0, stream, images,
                                (float*)xp, b, H, W, k, HP, WP, pad));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// fp32 packed views AND their pre-split copy (simclr_presplit_packed) in one pass over the images
int simclr_pack_views_ps(const float* images, void* xp, void* xq, int b, int H, int W, int k, int HP, int WP, int pad,
                         hipStream_t stream) {
  SIMCLR_CHECK_ARG(b > 0 && k > 0 && HP >= H + pad && WP >= W + pad && xp && xq, "pack_views_ps: bad geometry");
  const long long total = (long long)k * b * HP * WP;
  hipLaunchKernelGGL((pack_views<float>), dim3(grid_for(total)), dim3(256), 0, stream, images, (float*)xp, b, H, W, k, HP, WP, pad,
                     (u32x4*)xq);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

int simclr_bnrelu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y,
                              unsigned char* arg, int V, int H, int W, int C, int OH, int OW, int ksz,
                              int stride, int pad_t, int pad_l, int dtype, hipStream_t stream) {
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(C % epc == 0, "bnrelu_maxpool_fwd: C %% %d != 0", epc);
  SIMCLR_CHECK_ARG(ksz * ksz <= 255, "bnrelu_maxpool_fwd: window too large");
  SIMCLR_CHECK_ARG((long long)V * H * W < (1ll << 31) && (long long)V * OH * OW < (1ll << 31), "bnrelu_maxpool_fwd: pixel count overflows int32");
  const long long total = (long long)V * OH * OW * (C / epc);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((bnrelu_maxpool_fwd<uint16_t>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const uint16_t*)x, scale, shift, (uint16_t*)y, arg, V, H, W, C, OH, OW, ksz,
                                stride, pad_t, pad_l),
             hipLaunchKernelGGL((bnrelu_maxpool_fwd<float>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const float*)x, scale, shift, (float*)y, arg, V, H, W, C, OH, OW, ksz, stride,
                                pad_t, pad_l));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

int simclr_maxpool_bwd(const void* dy, const unsigned char* arg, void* dx, int V, int H, int W, int C,
                       int OH, int OW, int ksz, int stride, int pad_t, int pad_l, int dtype,
                       hipStream_t stream) {
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(C % epc == 0, "maxpool_bwd: C %% %d != 0", epc);
  SIMCLR_CHECK_ARG((long long)V * H * W < (1ll << 31), "maxpool_bwd: pixel count overflows int32");
  const long long total = (long long)V * H * W * (C / epc);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((maxpool_bwd<uint16_t>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const uint16_t*)dy, arg, (uint16_t*)dx, V, H, W, C, OH, OW, ksz, stride, pad_t,
                                pad_l),
             hipLaunchKernelGGL((maxpool_bwd<float>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const float*)dy, arg, (float*)dx, V, H, W, C, OH, OW, ksz, stride, pad_t,
                                pad_l));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

static int avgpool_fwd_impl(const void* x, void* y, float* y32, int V, int HW, int C, int dtype, hipStream_t stream) {
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(C % epc == 0, "global_avgpool_fwd: C %% %d != 0", epc);
  SIMCLR_CHECK_ARG(y || y32, "global_avgpool_fwd: no output");
  const long long total = (long long)V * (C / epc);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((global_avgpool_fwd<uint16_t>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const uint16_t*)x, (uint16_t*)y, y32, V, HW, C),
             hipLaunchKernelGGL((global_avgpool_fwd<float>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const float*)x, (float*)y, y32, V, HW, C));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}
int simclr_global_avgpool_fwd(const void* x, void* y, int V, int HW, int C, int dtype, hipStream_t stream) {
  return avgpool_fwd_impl(x, y, nullptr, V, HW, C, dtype, stream);
}
// the same mean, written as float32 whatever the activation dtype (input of fp32 heads)
int simclr_global_avgpool_fwd_f32(const void* x, float* y32, int V, int HW, int C, int dtype, hipStream_t stream) {
  return avgpool_fwd_impl(x, nullptr, y32, V, HW, C, dtype, stream);
}

// dx = dy/HW broadcast; mask_src (nullable, same shape as dx): zero where mask_src <= 0
// Stem backward fused with the max-pool backward (tf2/resnet.py:602-611 under tape.gradient): the gradient wrt the stem
// BN+ReLU output is never materialised.  dy [V,OH,OW,C] pooled gradient, arg uint8 tap ids (simclr_bnrelu_maxpool_fwd),
// x [V,H,W,C] raw stem-conv output.  reduce: partial [slots][2][C] with slots = simclr_bn_bwd_pool_slots (one per workgroup).
static int bwd_pool_grid(long long rows, int C, int epc, int* rows_per_block) {
  const int rl = 256 / max(1, C / epc);
  *rows_per_block = (int)max((long long)rl * 8, (rows + 2047) / 2048);
  return (int)((rows + *rows_per_block - 1) / *rows_per_block);
}
int simclr_bn_bwd_pool_slots(long long rows, int C, int dtype) {
  int rpb;
  return bwd_pool_grid(rows, C, dtype == SIMCLR_DT_BF16 ? 8 : 4, &rpb);
}
int simclr_bn_bwd_reduce_pool(const void* dy, const unsigned char* arg, const void* x, const float* scale,
                              const float* shift, const float* mean, const float* rstd, int V, int H, int W, int C,
                              int OH, int OW, int ksz, int stride, int pad_t, int pad_l, float* partial, int nslot,
                              int dtype, hipStream_t stream) {
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(C % epc == 0 && C / epc <= 256 && 256 % (C / epc) == 0, "bn_bwd_reduce_pool: C=%d not supported", C);
  SIMCLR_CHECK_ARG((long long)V * H * W < (1ll << 31), "bn_bwd_reduce_pool: pixel count overflows int32");
  int rpb;
  const int grid = bwd_pool_grid((long long)V * H * W, C, epc, &rpb);
  SIMCLR_CHECK_ARG(nslot >= grid, "bn_bwd_reduce_pool: need %d slots (simclr_bn_bwd_pool_slots), got %d", grid, nslot);
  // fp32 storage: the walk over the pooled pixels (same grid = same slots, every slot written): 1.30 -> 0.85 ms at 1024 views of 112^2 x 64,
  // step 139.81 -> 139.50 ms in three interleaved pairs (r06_call55); bf16 keeps the gather (its fused pair is not the default)
  if (dtype == SIMCLR_DT_F32 && (long long)V * OH * OW >= grid) {
    const int rpb_o = (int)(((long long)V * OH * OW + grid - 1) / grid);
    hipLaunchKernelGGL((bn_bwd_reduce_pool_out<float>), dim3(grid), dim3(256), 0, stream, (const float*)dy, arg, (const float*)x, scale,
                       shift, mean, rstd, V, H, W, C, OH, OW, ksz, stride, pad_t, pad_l, rpb_o, partial);
    SIMCLR_CHECK_LAUNCH();
    return 0;
  }
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((bn_bwd_reduce_pool<uint16_t>), dim3(grid), dim3(256), 0, stream, (const uint16_t*)dy, arg,
                                (const uint16_t*)x, scale, shift, mean, rstd, V, H, W, C, OH, OW, ksz, stride, pad_t, pad_l,
                                rpb, partial),
             hipLaunchKernelGGL((bn_bwd_reduce_pool<float>), dim3(grid), dim3(256), 0, stream, (const float*)dy, arg,
                                (const float*)x, scale, shift, mean, rstd, V, H, W, C, OH, OW, ksz, stride, pad_t, pad_l, rpb,
                                partial));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}
int simclr_bn_bwd_apply_pool(const void* dy, const unsigned char* arg, const void* x, const float* scale,
                             const float* shift, const float* mean, const float* rstd, const float* c1, const float* c2,
                             void* dx, int V, int H, int W, int C, int OH, int OW, int ksz, int stride, int pad_t, int pad_l,
                             int dtype, hipStream_t stream) {
  // dtype | SIMCLR_FMT_PS_OUT (fp32, C a multiple of 32): dx in the pre-split block format with bf16 pieces
  const bool ps_out = (dtype & SIMCLR_FMT_PS_OUT) != 0;
  dtype &= 0xff;
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(C % epc == 0, "bn_bwd_apply_pool: C %% %d != 0", epc);
  SIMCLR_CHECK_ARG(!ps_out || (dtype == SIMCLR_DT_F32 && C % 32 == 0), "bn_bwd_apply_pool: the pre-split output needs fp32 storage and C %% 32 == 0 (C=%d)", C);
  SIMCLR_CHECK_ARG((long long)V * H * W < (1ll << 31), "bn_bwd_apply_pool: pixel count overflows int32");
  const long long total = (long long)V * H * W * (C / epc);
  // fp32 storage, 3 x 3 stride-2 pooling: a thread per 2 x 2 block of input pixels (the four windows loaded once for four pixels):
  // 1.91 -> 1.37 ms at 1024 views of 112^2 x 64, step 139.50 -> 138.94 ms in three interleaved pairs (r06_call57)
  if (dtype == SIMCLR_DT_F32 && ksz == 3 && stride == 2) {
    const int TBY = (H + pad_t + 1) / 2, TBX = (W + pad_l + 1) / 2;
    const long long total2 = (long long)V * TBY * TBX * (C / epc);
    if (ps_out) hipLaunchKernelGGL((bn_bwd_apply_pool2<float, true>), dim3(grid_for(total2)), dim3(256), 0, stream, (const float*)dy, arg,
                                   (const float*)x, scale, shift, mean, rstd, c1, c2, (float*)dx, V, H, W, C, OH, OW, pad_t, pad_l, TBY, TBX);
    else hipLaunchKernelGGL((bn_bwd_apply_pool2<float, false>), dim3(grid_for(total2)), dim3(256), 0, stream, (const float*)dy, arg,
                            (const float*)x, scale, shift, mean, rstd, c1, c2, (float*)dx, V, H, W, C, OH, OW, pad_t, pad_l, TBY, TBX);
    SIMCLR_CHECK_LAUNCH();
    return 0;
  }
  if (ps_out) {
    hipLaunchKernelGGL((bn_bwd_apply_pool<float, true>), dim3(grid_for(total)), dim3(256), 0, stream, (const float*)dy, arg,
                       (const float*)x, scale, shift, mean, rstd, c1, c2, (float*)dx, V, H, W, C, OH, OW, ksz, stride, pad_t, pad_l);
    SIMCLR_CHECK_LAUNCH();
    return 0;
  }
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((bn_bwd_apply_pool<uint16_t>), dim3(grid_for(total)), dim3(256), 0, stream, (const uint16_t*)dy,
                                arg, (const uint16_t*)x, scale, shift, mean, rstd, c1, c2, (uint16_t*)dx, V, H, W, C, OH, OW,
                                ksz, stride, pad_t, pad_l),
             hipLaunchKernelGGL((bn_bwd_apply_pool<float>), dim3(grid_for(total)), dim3(256), 0, stream, (const float*)dy, arg,
                                (const float*)x, scale, shift, mean, rstd, c1, c2, (float*)dx, V, H, W, C, OH, OW, ksz, stride,
                                pad_t, pad_l));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

int simclr_global_avgpool_bwd(const void* dy, const void* mask_src, void* dx, int V, int HW, int C,
                              int dtype, hipStream_t stream) {
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(C % epc == 0, "global_avgpool_bwd: C %% %d != 0", epc);
  const long long total = (long long)V * HW * (C / epc);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((global_avgpool_bwd<uint16_t>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const uint16_t*)dy, (const uint16_t*)mask_src, (uint16_t*)dx, V, HW, C),
             hipLaunchKernelGGL((global_avgpool_bwd<float>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const float*)dy, (const float*)mask_src, (float*)dx, V, HW, C));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}


/* batch_random_blur (tf2/data_util.py:413-440): images f32 [b,H,W,3*nviews] -> out (same shape); tmp: scratch of
 * the same size.  filt [nviews][K] (normalised Gaussian per view, K odd), selector [nviews][b] in {0,1}. */
int simclr_batch_blur(const float* images, float* tmp, float* out, const float* filt, const float* selector,
                      int b, int H, int W, int nviews, int K, hipStream_t stream) {
  SIMCLR_CHECK_ARG(b > 0 && nviews > 0 && K > 0 && (K & 1), "batch_blur: bad shape (K must be odd)");
  const long long total = (long long)b * H * W * nviews;
  hipLaunchKernelGGL((blur1d<false>), dim3(grid_for(total)), dim3(256), 0, stream, images, (const float*)nullptr, tmp,
                     filt, selector, b, H, W, nviews, K);
  hipLaunchKernelGGL((blur1d<true>), dim3(grid_for(total)), dim3(256), 0, stream, images, (const float*)tmp, out,
                     filt, selector, b, H, W, nviews, K);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

/* ResNet-D shortcut average pool (tf2/resnet.py:330-338, 400-408): 2x2, stride 1 (SAME, TF valid-count
 * divisor) or 2 (after FixedPadding(2)); OH = H (stride 1) or (H+1)/2. */
int simclr_avgpool2_fwd(const void* x, void* y, int V, int H, int W, int C, int stride, int dtype,
                        hipStream_t stream) {
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(C % epc == 0 && (stride == 1 || stride == 2), "avgpool2_fwd: bad C/stride");
  const int OH = stride == 1 ? H : (H + 1) / 2, OW = stride == 1 ? W : (W + 1) / 2;
  const long long total = (long long)V * OH * OW * (C / epc);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((avgpool2_fwd<uint16_t>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const uint16_t*)x, (uint16_t*)y, V, H, W, C, OH, OW, stride),
             hipLaunchKernelGGL((avgpool2_fwd<float>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const float*)x, (float*)y, V, H, W, C, OH, OW, stride));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}
int simclr_avgpool2_bwd(const void* dy, void* dx, int V, int H, int W, int C, int stride, int dtype,
                        hipStream_t stream) {
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(C % epc == 0 && (stride == 1 || stride == 2), "avgpool2_bwd: bad C/stride");
  const int OH = stride == 1 ? H : (H + 1) / 2, OW = stride == 1 ? W : (W + 1) / 2;
  const long long total = (long long)V * H * W * (C / epc);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((avgpool2_bwd<uint16_t>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const uint16_t*)dy, (uint16_t*)dx, V, H, W, C, OH, OW, stride),
             hipLaunchKernelGGL((avgpool2_bwd<float>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const float*)dy, (float*)dx, V, H, W, C, OH, OW, stride));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

/* SK unit (tf2/resnet.py:266-277).  a [V,HW,2f] (T): the two streams; g [V,gpitch]: pooled feature
 * (first f channels written); l [V,lpitch]: mixing logits (2f used); out [V,HW,f]. */
int simclr_sk_pool_fwd(const void* a, void* g, int V, int HW, int f, int gpitch, int dtype, hipStream_t stream) {
  SIMCLR_CHECK_ARG(f % 4 == 0 && gpitch >= f, "sk_pool_fwd: bad f/gpitch");
  const long long total = (long long)V * (f / 4);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((sk_pool_fwd<uint16_t>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const uint16_t*)a, (uint16_t*)g, V, HW, f, gpitch),
             hipLaunchKernelGGL((sk_pool_fwd<float>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const float*)a, (float*)g, V, HW, f, gpitch));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}
int simclr_sk_mix_fwd(const void* a, const void* l, void* out, int V, int HW, int f, int lpitch, int dtype,
                      hipStream_t stream) {
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(f % epc == 0 && lpitch >= 2 * f, "sk_mix_fwd: bad f/lpitch");
  const long long total = (long long)V * HW * (f / epc);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((sk_mix_fwd<uint16_t>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const uint16_t*)a, (const uint16_t*)l, (uint16_t*)out, V, HW, f, lpitch),
             hipLaunchKernelGGL((sk_mix_fwd<float>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const float*)a, (const float*)l, (float*)out, V, HW, f, lpitch));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}
/* dl [V,lpitch] = gradient wrt the mixing logits (softmax over the two streams). */
int simclr_sk_mix_bwd_logits(const void* a, const void* l, const void* dout, void* dl, int V, int HW, int f,
                             int lpitch, int dtype, hipStream_t stream) {
  SIMCLR_CHECK_ARG(f > 0 && lpitch >= 2 * f, "sk_mix_bwd_logits: bad f/lpitch");
  const long long total = (long long)V * f;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((sk_mix_bwd_logits<uint16_t>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const uint16_t*)a, (const uint16_t*)l, (const uint16_t*)dout, (uint16_t*)dl, V, HW, f, lpitch),
             hipLaunchKernelGGL((sk_mix_bwd_logits<float>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const float*)a, (const float*)l, (const float*)dout, (float*)dl, V, HW, f, lpitch));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}
/* da [V,HW,2f] = dout*m_k + dg/HW : gradient wrt both streams (mix path + pooled-feature path). */
int simclr_sk_mix_bwd_streams(const void* l, const void* dout, const void* dg, void* da, int V, int HW, int f,
                              int lpitch, int gpitch, int dtype, hipStream_t stream) {
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(f % epc == 0 && lpitch >= 2 * f && gpitch >= f, "sk_mix_bwd_streams: bad shape");
  const long long total = (long long)V * HW * (f / epc);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((sk_mix_bwd_streams<uint16_t>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const uint16_t*)l, (const uint16_t*)dout, (const uint16_t*)dg, (uint16_t*)da, V, HW, f, lpitch, gpitch),
             hipLaunchKernelGGL((sk_mix_bwd_streams<float>), dim3(grid_for(total)), dim3(256), 0, stream,
                                (const float*)l, (const float*)dout, (const float*)dg, (float*)da, V, HW, f, lpitch, gpitch));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// out[0] += mean CE loss, out[1] += top-1 accuracy (caller zeroes out[0..1])
int simclr_bias_softmax_xent(const void* z, const float* bias, const int* labels, int rows,
                             int label_rows, int nclass, int cpad, float gscale, void* dlogits, float* out,
                             int dtype, hipStream_t stream) {
  SIMCLR_CHECK_ARG(rows > 0 && nclass > 0 && cpad >= nclass && label_rows > 0, "bias_softmax_xent: bad shape");
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((bias_softmax_xent<uint16_t>), dim3(ceil_div(rows, 4)), dim3(256), 0, stream,
                                (const uint16_t*)z, bias, labels, rows, label_rows, nclass, cpad, gscale,
                                (uint16_t*)dlogits, out),
             hipLaunchKernelGGL((bias_softmax_xent<float>), dim3(ceil_div(rows, 4)), dim3(256), 0, stream,
                                (const float*)z, bias, labels, rows, label_rows, nclass, cpad, gscale,
                                (float*)dlogits, out));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

int simclr_colsum(const void* x, int rows, int C, int cvalid, float* out, int accumulate, int dtype,
                  hipStream_t stream) {
  DISPATCH_T(dtype,
             hipLaunchKernelGGL((colsum<uint16_t>), dim3(ceil_div(cvalid, 16)), dim3(256), 0, stream,
                                (const uint16_t*)x, rows, C, cvalid, out, accumulate),
             hipLaunchKernelGGL((colsum<float>), dim3(ceil_div(cvalid, 16)), dim3(256), 0, stream,
                                (const float*)x, rows, C, cvalid, out, accumulate));
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// dtype_in/dtype_out in {f32, bf16}
int simclr_cast(const void* x, void* y, long long n, int dtype_in, int dtype_out, hipStream_t stream) {
  const int grid = grid_for(n);
  if (dtype_in == SIMCLR_DT_F32 && dtype_out == SIMCLR_DT_BF16)
    hipLaunchKernelGGL((cast_kernel<float, uint16_t>), dim3(grid), dim3(256), 0, stream, (const float*)x, (uint16_t*)y, n);
  else if (dtype_in == SIMCLR_DT_BF16 && dtype_out == SIMCLR_DT_F32)
    hipLaunchKernelGGL((cast_kernel<uint16_t, float>), dim3(grid), dim3(256), 0, stream, (const uint16_t*)x, (float*)y, n);
  else if (dtype_in == SIMCLR_DT_F32 && dtype_out == SIMCLR_DT_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), dim3(grid), dim3(256), 0, stream, (const float*)x, (float*)y, n);
  else
    hipLaunchKernelGGL((cast_kernel<uint16_t, uint16_t>), dim3(grid), dim3(256), 0, stream, (const uint16_t*)x, (uint16_t*)y, n);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

int simclr_accumulate_scalars(const float* const* src, const float* scale, int n, float* dst, float* total,
                              int total_mask, float* copy, hipStream_t stream) {
  SIMCLR_CHECK_ARG(n >= 1 && n <= 16 && src, "accumulate_scalars: n=%d must be in [1, 16]", n);
  ScalarsP p = {};
  for (int i = 0; i < n; ++i) {
    SIMCLR_CHECK_ARG(src[i] != nullptr, "accumulate_scalars: null source %d", i);
    p.src[i] = src[i];
    p.scale[i] = scale ? scale[i] : 1.f;
  }
  hipLaunchKernelGGL(accumulate_scalars, dim3(1), dim3(64), 0, stream, p, n, dst, total, (unsigned)total_mask, copy);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

int simclr_axpy_f32(float a, const float* x, float* y, long long n, hipStream_t stream) {
  hipLaunchKernelGGL(axpy_f32, dim3(grid_for(n)), dim3(256), 0, stream, a, x, y, n);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}
int simclr_l2_loss_f32(const float* x, long long n, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(l2_loss_f32, dim3(min(256, ceil_div(n, 1024))), dim3(256), 0, stream, x, n, out);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
