// BatchNorm (training mode, optionally cross-replica) for gfx950: statistics
// finalisation, fused apply (+ReLU, +residual add), and the two-phase backward.
//
// Restates /root/reference/tf2/resnet.py:31-78 (BatchNormRelu over Keras
// [Sync]BatchNormalization: batch mean and BIASED variance over every axis but
// channels, eps 1e-5 (:28), moving <- moving*decay + batch*(1-decay)) and the
// residual tail `relu(inputs + shortcut)` of :382/:487.  The backward is what
// `tape.gradient` (tf2/run.py:621) derives:
//   x^ = (x-mean)*rstd,  dbeta = sum dy,  dgamma = sum dy*x^,
//   dx = gamma*rstd*(dy - mean(dy) - x^*mean(dy*x^))          (means over the GLOBAL batch)
// All tensors are NHWC with channels contiguous; every kernel moves 16-byte chunks
// (8 bf16 / 4 f32 channels per lane).  Statistics are fp32/fp64 regardless of T.
//
// Cross-replica (global_bn, resnet.py:50-60): the per-channel sums produced here are
// all-reduced by the host (RCCL) between the *_reduce and *_finalize entry points.
#include "common.h"

namespace {

// Fixed-order reduction of the partial-statistics slots: a 1024-thread workgroup owns 32 channels; thread
// (sl = tid / 32, cl = tid % 32) adds slots sl, sl+32, sl+64, ... of channel c0+cl in ascending order (coalesced
// 128-byte rows, four loads per statistic in flight), then the 32 lane sums are added in the order sl = 0..31.  The
// order never depends on timing, so the result is bit-identical from run to run whatever wrote the slots.  32 slot
// lanes: with up to 768 slots (one per persistent conv workgroup) the walk is 6 dependent round trips instead of 24
// -- these launches sit on the critical path between a convolution and its BatchNorm apply, ~130 of them per step.
// Returns the two sums of channel c0 + (tid % 32) in the threads with tid < 32 (others: partial values).
constexpr int kSlotCh = 32;
constexpr int kSlotLanes = 32;
constexpr int kSlotThreads = kSlotCh * kSlotLanes;
__device__ __forceinline__ void slot_sums(const float* __restrict__ partial, int nslot, int C, int c0,
                                          double* sh /* [2][kSlotLanes][32] */, double& s1, double& s2) {
  const int sl = threadIdx.x >> 5, cl = threadIdx.x & 31, c = c0 + cl;
  double a = 0.0, b = 0.0;
  if (c < C) {
    int s = sl;
    const long long st = 2ll * kSlotLanes * C;    // floats between slots s and s + kSlotLanes
    for (; s + 3 * kSlotLanes < nslot; s += 4 * kSlotLanes) {
      const float* q = partial + (long long)s * 2 * C + c;
      const float a0 = q[0], a1 = q[st], a2 = q[2 * st], a3 = q[3 * st];
      const float b0 = q[C], b1 = q[st + C], b2 = q[2 * st + C], b3 = q[3 * st + C];
      a += (double)a0; a += (double)a1; a += (double)a2; a += (double)a3;
      b += (double)b0; b += (double)b1; b += (double)b2; b += (double)b3;
    }
    for (; s < nslot; s += kSlotLanes) {
      a += (double)partial[(long long)s * 2 * C + c];
      b += (double)partial[(long long)s * 2 * C + C + c];
    }
  }
  sh[sl * 32 + cl] = a;
  sh[kSlotThreads + sl * 32 + cl] = b;
  __syncthreads();
  s1 = 0.0; s2 = 0.0;
  if (threadIdx.x < 32) {
#pragma unroll 8
    for (int j = 0; j < kSlotLanes; ++j) { s1 += sh[j * 32 + cl]; s2 += sh[kSlotThreads + j * 32 + cl]; }
  }
}

// sums[2][C] (double) = sum over slots of partial[slot][2][C] (float)
__global__ __launch_bounds__(kSlotThreads) void bn_reduce_slots(const float* __restrict__ partial, int nslot, int C,
                                                       double* __restrict__ sums) {
  __shared__ double sh[2 * kSlotThreads];
  const int c0 = blockIdx.x * kSlotCh;
  double s1, s2;
  slot_sums(partial, nslot, C, c0, sh, s1, s2);
  const int c = c0 + threadIdx.x;
  if (threadIdx.x < 32 && c < C) { sums[c] = s1; sums[C + c] = s2; }
}

// Slots that hold PIVOTED moments (simclr_conv2d_fwd_pivoted): S1 = sum(y - p), S2 = sum((y - p)^2) over `count` rows, p = pivot[c]
//   sum(y) = S1 + count p,   sum(y^2) = S2 + 2 p S1 + count p^2      (fp64: the cancellation of the finalize happens at 2^-53)
__global__ __launch_bounds__(kSlotThreads) void bn_reduce_slots_pivoted(const float* __restrict__ partial, int nslot, int C,
                                                               const float* __restrict__ pivot, double count,
                                                               double* __restrict__ sums) {
  __shared__ double sh[2 * kSlotThreads];
  const int c0 = blockIdx.x * kSlotCh;
  double s1, s2;
  slot_sums(partial, nslot, C, c0, sh, s1, s2);
  const int c = c0 + threadIdx.x;
  if (threadIdx.x < 32 && c < C) {
    const double p = (double)pivot[c];
    sums[c] = s1 + count * p;
    sums[C + c] = s2 + 2.0 * p * s1 + count * p * p;
  }
}

// BatchNorm statistics of c = h W (a 1x1 convolution, M rows) WITHOUT forming c: per output channel n
//   sum_m c[m][n]   = sum_k colsum(h)[k] W[k][n]
//   sum_m c[m][n]^2 = w_n^T (h^T h) w_n = sum_k GW[k][n] W[k][n],   GW = (h^T h) W
// (h^T h and colsum(h) come from one pass over h, simclr_conv2d_gram -- the same two quantities the folded BatchNorm
// backward needs, so the forward's statistics pass and the backward's Gram pass are ONE pass).  fp64 accumulation,
// fixed summation order: deterministic.
__global__ __launch_bounds__(256) void bn_sums_from_gram(const float* __restrict__ gw, const float* __restrict__ w,
                                                         const double* __restrict__ cs64, const float* __restrict__ cs32,
                                                         int K, int N, double* __restrict__ sums) {
  // 32 channels x 8 k-lanes per workgroup: lane l adds rows l, l+8, ... (ascending), the 8 lane sums are added in lane order
  __shared__ double sh[2][8][32];
  const int cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
  const int n = blockIdx.x * 32 + cl;
  double a = 0.0, b = 0.0;
  if (n < N) {
#pragma unroll 4
    for (int k = kl; k < K; k += 8) {
      const double wv = (double)w[(long long)k * N + n];
      a += (cs64 ? cs64[k] : (double)cs32[k]) * wv;
      b += (double)gw[(long long)k * N + n] * wv;
    }
  }
  sh[0][kl][cl] = a;
  sh[1][kl][cl] = b;
  __syncthreads();
  if (kl == 0 && n < N) {
#pragma unroll
    for (int l = 1; l < 8; ++l) { a += sh[0][l][cl]; b += sh[1][l][cl]; }
    sums[n] = a;
    sums[N + n] = b;
  }
}

// From global sums -> mean/rstd/scale/shift, moving-stat update.  If `partial` is given (single
// replica: no all-reduce between) the slot reduction is done here instead of a separate launch.
__global__ __launch_bounds__(kSlotThreads) void bn_finalize(const double* __restrict__ sums, const float* __restrict__ partial,
                            int nslot, double count, int C,
                            const float* __restrict__ gamma, const float* __restrict__ beta,
                            float* __restrict__ moving_mean, float* __restrict__ moving_var,
                            float decay, float eps, float* __restrict__ mean_out,
                            float* __restrict__ rstd_out, float* __restrict__ scale,
                            float* __restrict__ shift) {
  __shared__ double sh[2 * kSlotThreads];
  const int c0 = blockIdx.x * kSlotCh;
  double s1 = 0.0, s2 = 0.0;
  if (partial) slot_sums(partial, nslot, C, c0, sh, s1, s2);
  const int c = c0 + threadIdx.x;
  if (threadIdx.x >= 32 || c >= C) return;
  if (!partial) { s1 = sums[c]; s2 = sums[C + c]; }
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;  // biased variance (Keras non-fused BN)
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f;
  const float b = beta ? beta[c] : 0.f;
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  scale[c] = g * rstd;
  shift[c] = b - (float)mean * g * rstd;
  if (moving_mean) moving_mean[c] = moving_mean[c] * decay + (float)mean * (1.f - decay);
  if (moving_var) moving_var[c] = moving_var[c] * decay + (float)var * (1.f - decay);
}

// per-channel parameter vector for this thread's fixed channel chunk (16-byte loads)
template <int EPC>
__device__ __forceinline__ void load_params(const float* __restrict__ p, int c0, float* out) {
#pragma unroll
  for (int e = 0; e < EPC; e += 4) {
    const float4 v = *(const float4*)(p + c0 + e);
    out[e] = v.x; out[e + 1] = v.y; out[e + 2] = v.z; out[e + 3] = v.w;
  }
}

// y = act(x*scale + shift [+ r] [+ r*rscale + rshift])
// Streaming shape (tools/probes/bn_probe.hip, profiles/r01_notes.md): every workgroup owns ONE contiguous run
// of 256*U 16-byte chunks and every thread touches chunk base + u*256 -- a plain linear sweep of memory in
// dispatch order.  This reaches ~6.2 TB/s (read+write) on MI355X; the former grid-stride walk (a fixed grid
// striding through the tensor) stalled at ~4.4 TB/s.  U = 2 when 256 is a multiple of the chunks per row
// (every standard ResNet width), so the per-channel parameters are fetched once per thread; U = 1 otherwise.
template <bool NT> __device__ __forceinline__ u32x4 ldg16(const void* p) {
  return NT ? __builtin_nontemporal_load((const u32x4*)p) : *(const u32x4*)p;
}
template <bool NT> __device__ __forceinline__ void stg16(void* p, const u32x4& v) {
  if (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
}

template <typename T, int RES, int U, bool NT>   // RES: 0 none, 1 plain residual, 2 residual with its own BN
__global__ __launch_bounds__(256) void bn_apply(const T* __restrict__ x, const float* __restrict__ scale,
                         const float* __restrict__ shift, const T* __restrict__ res,
                         const float* __restrict__ rscale, const float* __restrict__ rshift,
                         T* __restrict__ y, long long nchunks, int C, int relu,
                         unsigned char* __restrict__ relu_bits) {
  constexpr int EPC = Elem<T>::EPC;
  const unsigned cpr = (unsigned)(C / EPC);
  const long long base = (long long)blockIdx.x * (256 * U) + threadIdx.x;
  u32x4 xv[U], rv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long i = base + u * 256;
    if (i < nchunks) {
      xv[u] = ldg16<NT>(x + i * EPC);
      if (RES) rv[u] = ldg16<NT>(res + i * EPC);
    }
  }
  // U == 2 is only launched when 256 % cpr == 0: both chunks of a thread sit in the same channel chunk
  const int c0 = (int)((unsigned long long)base % cpr) * EPC;
  float sc[EPC], sh[EPC], rsc[EPC], rsh[EPC];
  load_params<EPC>(scale, c0, sc);
  load_params<EPC>(shift, c0, sh);
  if (RES == 2) { load_params<EPC>(rscale, c0, rsc); load_params<EPC>(rshift, c0, rsh); }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long i = base + u * 256;
    if (i < nchunks) {
      float v[EPC], q[EPC];
      chunk_to_f32<T>(xv[u], v);
      if (RES) chunk_to_f32<T>(rv[u], q);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        float o = fmaf(v[e], sc[e], sh[e]);
        if (RES == 1) o += q[e];
        if (RES == 2) o += fmaf(q[e], rsc[e], rsh[e]);
        v[e] = relu ? fmaxf(o, 0.f) : o;
      }
      const u32x4 packed = f32_to_chunk<T>(v);
      stg16<NT>(y + i * EPC, packed);
      if (relu_bits) {
        // bit e = (stored y[e] > 0): the ReLU mask the backward needs, 1 byte per 16-byte chunk
        float w[EPC];
        chunk_to_f32<T>(packed, w);
        unsigned bits = 0;
#pragma unroll
        for (int e = 0; e < EPC; ++e) bits |= (w[e] > 0.f ? 1u : 0u) << e;
        relu_bits[i] = (unsigned char)bits;
      }
    }
  }
}

// mask modes for the backward: 0 none, 1 mask_src > 0, 2 recompute x*scale+shift > 0

// partial[slot][2][C] += (sum dy_masked, sum dy_masked * x^) over this block's rows
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ mask_src,
    const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ rstd, long long rows, int C,
    int mask_mode, int rows_per_block, float* __restrict__ partial, int nslot) {
  constexpr int EPC = Elem<T>::EPC;
  __shared__ float red[256 * 2 * EPC];
  const int cpr = C / EPC;
  const int cw = min(cpr, 256);
  const int rl = 256 / cw;              // row lanes
  const int tcol = threadIdx.x % cw, trow = threadIdx.x / cw;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(rows, r0 + rows_per_block);
  // nslot >= gridDim.x (simclr_bn_bwd_reduce_slots): every workgroup stores into its own slot, no float atomics
  const bool own_slot = nslot >= (int)gridDim.x;
  float* slot = partial + (long long)(own_slot ? blockIdx.x : blockIdx.x % nslot) * 2 * C;
  for (int cc0 = 0; cc0 < cpr; cc0 += cw) {
    const int cc = cc0 + tcol;
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    if (cc < cpr && trow < rl) {
      const int c0 = cc * EPC;
      float mu[EPC], rs[EPC], sc[EPC], sh[EPC];
      load_params<EPC>(mean, c0, mu);
      load_params<EPC>(rstd, c0, rs);
      if (mask_mode == 2) { load_params<EPC>(scale, c0, sc); load_params<EPC>(shift, c0, sh); }
      for (long long r = r0 + trow; r < r1; r += 2 * rl) {
        const long long i0 = (r * cpr + cc) * EPC, i1 = ((r + rl) * cpr + cc) * EPC;
        const bool two = r + rl < r1;
        u32x4 dv0 = *(const u32x4*)(dy + i0), xv0 = *(const u32x4*)(x + i0), mv0, dv1, xv1, mv1;
        if (mask_mode == 1) mv0 = *(const u32x4*)(mask_src + i0);
        if (two) {
          dv1 = *(const u32x4*)(dy + i1); xv1 = *(const u32x4*)(x + i1);
          if (mask_mode == 1) mv1 = *(const u32x4*)(mask_src + i1);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u == 1 && !two) break;
          float d[EPC], xf[EPC];
          chunk_to_f32<T>(u ? dv1 : dv0, d);
          chunk_to_f32<T>(u ? xv1 : xv0, xf);
          if (mask_mode == 1) {
            float mk[EPC];
            chunk_to_f32<T>(u ? mv1 : mv0, mk);
#pragma unroll
            for (int e = 0; e < EPC; ++e) d[e] = mk[e] > 0.f ? d[e] : 0.f;
          } else if (mask_mode == 2) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) d[e] = fmaf(xf[e], sc[e], sh[e]) > 0.f ? d[e] : 0.f;
          }
#pragma unroll
          for (int e = 0; e < EPC; ++e) { s1[e] += d[e]; s2[e] += d[e] * (xf[e] - mu[e]) * rs[e]; }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      red[(threadIdx.x * EPC + e) * 2] = s1[e];
      red[(threadIdx.x * EPC + e) * 2 + 1] = s2[e];
    }
    __syncthreads();
    if (trow == 0 && cc < cpr) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        float a = 0.f, b = 0.f;
        for (int q = 0; q < rl; ++q) {
          a += red[((q * cw + tcol) * EPC + e) * 2];
          b += red[((q * cw + tcol) * EPC + e) * 2 + 1];
        }
        if (own_slot) { slot[cc * EPC + e] = a; slot[C + cc * EPC + e] = b; }     // deterministic: one writer per entry
        else { atomicAdd(slot + cc * EPC + e, a); atomicAdd(slot + C + cc * EPC + e, b); }
      }
    }
  }
}

// local sums -> dgamma/dbeta (+=), global sums/count -> c1 = mean(dy), c2 = mean(dy*x^)
__global__ __launch_bounds__(kSlotThreads) void bn_bwd_finalize(const double* __restrict__ local_sums,
                                const double* __restrict__ global_sums, const float* __restrict__ partial,
                                int nslot, double count, int C,
                                float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                float* __restrict__ c1, float* __restrict__ c2) {
  __shared__ double sh[2 * kSlotThreads];
  const int c0 = blockIdx.x * kSlotCh;
  double l1 = 0.0, l2 = 0.0, g1, g2;
  if (partial) slot_sums(partial, nslot, C, c0, sh, l1, l2);     // single replica: local == global
  const int c = c0 + threadIdx.x;
  if (threadIdx.x >= 32 || c >= C) return;
  if (partial) { g1 = l1; g2 = l2; }
  else { l1 = local_sums[c]; l2 = local_sums[C + c]; g1 = global_sums[c]; g2 = global_sums[C + c]; }
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)l1;
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)l2;
  c1[c] = (float)(g1 / count);
  c2[c] = (float)(g2 / count);
}

// dx = scale*(dy_m - c1 - x^*c2)   [dmasked = dy_m]; same block-contiguous streaming shape as bn_apply
// PSO (fp32 only): dx is written in the pre-split block format (common.h), bf16 pieces -- its consumers are the data-gradient and
// weight-gradient GEMMs of the convolution in front of this BatchNorm, which then split nothing in their k-loops.
template <typename T, int U, bool NT, bool PSO = false>
__global__ __launch_bounds__(256) void bn_bwd_apply(const T* __restrict__ dy, const T* __restrict__ x,
                             const T* __restrict__ mask_src, const float* __restrict__ scale,
                             const float* __restrict__ shift, const float* __restrict__ mean,
                             const float* __restrict__ rstd, const float* __restrict__ c1,
                             const float* __restrict__ c2, long long nchunks, int C, int mask_mode,
                             T* __restrict__ dx, T* __restrict__ dmasked) {
  constexpr int EPC = Elem<T>::EPC;
  const unsigned cpr = (unsigned)(C / EPC);
  const long long base = (long long)blockIdx.x * (256 * U) + threadIdx.x;
  u32x4 dv[U], xv[U], mv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long i = base + u * 256;
    if (i < nchunks) {
      dv[u] = ldg16<NT>(dy + i * EPC);
      xv[u] = ldg16<NT>(x + i * EPC);
      if (mask_mode == 1) mv[u] = ldg16<NT>(mask_src + i * EPC);
    }
  }
  const int c0 = (int)((unsigned long long)base % cpr) * EPC;
  float sc[EPC], sh[EPC], mu[EPC], rs[EPC], k1[EPC], k2[EPC];
  load_params<EPC>(scale, c0, sc);
  load_params<EPC>(mean, c0, mu);
  load_params<EPC>(rstd, c0, rs);
  load_params<EPC>(c1, c0, k1);
  load_params<EPC>(c2, c0, k2);
  if (mask_mode == 2) load_params<EPC>(shift, c0, sh);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long long i = base + u * 256;
    if (i < nchunks) {
      float d[EPC], xf[EPC], o[EPC];
      chunk_to_f32<T>(dv[u], d);
      chunk_to_f32<T>(xv[u], xf);
      if (mask_mode == 1) {
        float mk[EPC];
        chunk_to_f32<T>(mv[u], mk);
#pragma unroll
        for (int e = 0; e < EPC; ++e) d[e] = mk[e] > 0.f ? d[e] : 0.f;
      } else if (mask_mode == 2) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) d[e] = fmaf(xf[e], sc[e], sh[e]) > 0.f ? d[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float xh = (xf[e] - mu[e]) * rs[e];
        o[e] = sc[e] * (d[e] - k1[e] - xh * k2[e]);
      }
      if constexpr (PSO) ps_store_quad<false, NT>(dx, i, o);
      else stg16<NT>(dx + i * EPC, f32_to_chunk<T>(o));
      if (dmasked) stg16<NT>(dmasked + i * EPC, f32_to_chunk<T>(d));
    }
  }
}

// ---- BatchNorm backward folded into the convolution that produced the BN's input (1x1 "expand" convs: K <= N) -------
// For c = h W (h [M,K], W [K,N]) followed by BN, the gradient wrt c is dh = a*dm + b*c + d per channel j
// (a = scale, b = -scale*k2*rstd, d = scale*(k2*mean*rstd - k1); k1 = mean(dm), k2 = mean(dm*x^)).  By linearity
//   dW = h^T dh = (h^T dm)*a + ((h^T h) W)*b + colsum(h) (x) d          (columns j scaled)
//   dh_in = dh W^T = dm (a*W)^T + h (W diag(b) W^T) + W d
// so neither dh nor the streaming pass that would compute it is needed: h^T dm, h^T h are weight-gradient GEMMs, the rest
// is O(K*N) work.  Kernels below do the O(K*N) parts; the GEMMs run on the conv kernels.
__global__ void bn_fold_coeffs(const float* __restrict__ scale, const float* __restrict__ mean, const float* __restrict__ rstd,
                               const float* __restrict__ c1, const float* __restrict__ c2, float* __restrict__ a,
                               float* __restrict__ b, float* __restrict__ d, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = scale[c];
  a[c] = sc;
  b[c] = -sc * c2[c] * rstd[c];
  d[c] = sc * (c2[c] * mean[c] * rstd[c] - c1[c]);
}

// sum(dm * x^) of the BatchNorm behind c = h W, WITHOUT reading c or x^: sum_m dm[m,j] c[m,j] = sum_k W[k,j] T1[k,j] with
// T1 = h^T dm (the weight-gradient GEMM that is computed anyway), so
//   sums[1][j] = rstd[j] * (sum_k W[k][j] * T1[k][j] - mean[j] * sums[0][j]),      sums[0] = sum dm (already there).
template <typename T>
__global__ __launch_bounds__(256) void bn_fold_s2(const float* __restrict__ t1, const T* __restrict__ w, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, double* __restrict__ sums, int K, int N) {
  // 32 channels x 8 k-lanes per workgroup; lane kl adds rows kl, kl+8, ... (ascending), the lanes are joined in order
  __shared__ double sh[256];
  const int jl = threadIdx.x & 31, kl = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + jl;
  double acc = 0.0;
  if (j < N) {
#pragma unroll 4
    for (int k = kl; k < K; k += 8) acc += (double)Elem<T>::ld(w + (long long)k * N + j) * (double)t1[(long long)k * N + j];
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  if (kl == 0 && j < N) {
    double a = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) a += sh[q * 32 + jl];
    sums[N + j] = (double)rstd[j] * (a - (double)mean[j] * sums[j]);
  }
}

// one workgroup per input channel i (row of W [K][N]):  wb[i][j] = W[i][j]*b[j] (fp32),  wext[i][j] = T(W[i][j]*a[j]) for
// j < N (row pitch N + K),  e[i] = sum_j W[i][j]*d[j]
// (the coefficient vectors a, b, d are computed here from the BN quantities and written out by workgroup 0)
template <typename T>
__global__ __launch_bounds__(256) void bn_fold_pre(const T* __restrict__ w, const float* __restrict__ scale,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   const float* __restrict__ c1, const float* __restrict__ c2,
                                                   float* __restrict__ a, float* __restrict__ b, float* __restrict__ d,
                                                   float* __restrict__ wb, T* __restrict__ wext,
                                                   float* __restrict__ e, int K, int N) {
  __shared__ double sh[256];
  const int i = blockIdx.x;
  double acc = 0.0;
  for (int j = threadIdx.x; j < N; j += 256) {
    const float sc = scale[j];
    const float aj = sc, bj = -sc * c2[j] * rstd[j], dj = sc * (c2[j] * mean[j] * rstd[j] - c1[j]);
    if (i == 0) { a[j] = aj; b[j] = bj; d[j] = dj; }
    const float wv = Elem<T>::ld(w + (long long)i * N + j);
    wb[(long long)i * N + j] = wv * bj;
    Elem<T>::st(wext + (long long)i * (N + K) + j, wv * aj);
    acc += (double)wv * (double)dj;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) e[i] = (float)sh[0];
}

// dw[i][j] = a[j]*t1[i][j] + b[j]*gw[i][j] + d[j]*cs[i]  (+= if accumulate);  wext[i][N + k] = T(q[k][i])
template <typename T>
__global__ __launch_bounds__(256) void bn_fold_post(const float* __restrict__ t1, const float* __restrict__ gw,
                                                    const double* __restrict__ cs, const float* __restrict__ cs32,
                                                    const float* __restrict__ a,
                                                    const float* __restrict__ b, const float* __restrict__ d,
                                                    const float* __restrict__ q, float* __restrict__ dw, T* __restrict__ wext,
                                                    int K, int N, int accumulate) {
  const long long total = (long long)K * N;
  for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total + (long long)K * K; t += gridDim.x * 256ll) {
    if (t < total) {
      const int i = (int)(t / N), j = (int)(t % N);
      const float v = fmaf(a[j], t1[t], fmaf(b[j], gw[t], d[j] * (cs ? (float)cs[i] : cs32[i])));
      dw[t] = accumulate ? dw[t] + v : v;
    } else {
      const long long u = t - total;
      const int i = (int)(u / K), k = (int)(u % K);
      Elem<T>::st(wext + (long long)i * (N + K) + N + k, q[(long long)k * K + i]);
    }
  }
}

}  // namespace

static int bwd_reduce_grid(long long rows, int C, int epc, int* rows_per_block) {
  const int cpr = C / epc;
  const int rl = max(1, 256 / min(cpr, 256));
  // ~2048 workgroups, fewer for wide layers so that the [slots][2][C] partial buffer stays around 2 M floats
  const long long want_blocks = max(256ll, min(2048ll, (1ll << 20) / max(C, 1)));
  *rows_per_block = (int)max((long long)rl * 4, (rows + want_blocks - 1) / want_blocks);
  return ceil_div(rows, *rows_per_block);
}

extern "C" {

// slots that make simclr_bn_bwd_reduce deterministic (one per workgroup)
int simclr_bn_bwd_reduce_slots(long long rows, int C, int dtype) {
  int rpb;
  return bwd_reduce_grid(rows, C, dtype == SIMCLR_DT_BF16 ? 8 : 4, &rpb);
}

// partial [nslot][2][C] fp32 -> sums [2][C] fp64 (the buffer the host all-reduces)
int simclr_bn_sums_from_gram(const float* gw, const float* w_kn, const double* cs64, const float* cs32, int K, int N,
                             double* sums, hipStream_t stream) {
  SIMCLR_CHECK_ARG(gw && w_kn && sums && ((cs64 != nullptr) != (cs32 != nullptr)) && K > 0 && N > 0,
                   "bn_sums_from_gram: bad arguments (exactly one of cs64 / cs32)");
  hipLaunchKernelGGL(bn_sums_from_gram, dim3(ceil_div(N, 32)), dim3(256), 0, stream, gw, w_kn, cs64, cs32, K, N, sums);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

int simclr_bn_reduce_slots(const float* partial, int nslot, int C, double* sums, hipStream_t stream) {
  SIMCLR_CHECK_ARG(nslot > 0 && C > 0, "bn_reduce_slots: bad shape");
  hipLaunchKernelGGL(bn_reduce_slots, dim3(ceil_div(C, kSlotCh)), dim3(kSlotThreads), 0, stream, partial, nslot, C, sums);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// the pivoted slots of simclr_conv2d_fwd_pivoted -> raw fp64 moments; count = rows this replica accumulated (V*OH*OW)
int simclr_bn_reduce_slots_pivoted(const float* partial, int nslot, int C, const float* pivot, double count, double* sums,
                                   hipStream_t stream) {
  SIMCLR_CHECK_ARG(nslot > 0 && C > 0 && count > 0, "bn_reduce_slots_pivoted: bad shape");
  SIMCLR_CHECK_ARG(partial && pivot && sums, "bn_reduce_slots_pivoted: null argument");
  hipLaunchKernelGGL(bn_reduce_slots_pivoted, dim3(ceil_div(C, kSlotCh)), dim3(kSlotThreads), 0, stream, partial, nslot, C, pivot,
                     count, sums);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// sums [2][C] (global) OR partial [nslot][2][C] (single replica; slot reduction fused here), count =
// global elements per channel.  gamma/beta nullable
// (scale=False / center=False).  moving_* nullable (no update).
int simclr_bn_finalize(const double* sums, const float* partial, int nslot, double count, int C,
                       const float* gamma, const float* beta, float* moving_mean, float* moving_var,
                       float decay, float eps, float* mean, float* rstd, float* scale, float* shift,
                       hipStream_t stream) {
  SIMCLR_CHECK_ARG(C > 0 && count > 0, "bn_finalize: bad shape");
  SIMCLR_CHECK_ARG((sums != nullptr) != (partial != nullptr), "bn_finalize: give sums OR partial slots");
  hipLaunchKernelGGL(bn_finalize, dim3(ceil_div(C, kSlotCh)), dim3(kSlotThreads), 0, stream, sums, partial, nslot, count, C,
                     gamma, beta, moving_mean, moving_var, decay, eps, mean, rstd, scale, shift);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// y = act(x*scale+shift [+ res | + res*rscale+rshift]); rows x C, T = dtype
int simclr_bn_apply(const void* x, const float* scale, const float* shift, const void* res,
                    const float* rscale, const float* rshift, void* y, unsigned char* relu_bits, long long rows,
                    int C, int relu, int dtype, hipStream_t stream) {
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(C % epc == 0, "bn_apply: C=%d must be a multiple of %d", C, epc);
  SIMCLR_CHECK_ARG(!rscale || res, "bn_apply: rscale needs res");
  // bit0: non-temporal loads / stores.  Default ON since round 5: interleaved three-fold A/B on one box (profiles/r05_notes.md section 8),
  // ms per training step: ResNet-50 63.19 -> 62.63, fp32 parity mode 186.5 -> 184.3 -- the streamed tensors are far larger than L2 / MALL
  // and are next read by a different kernel, so keeping them out of the caches leaves those to the convolutions' re-reads.
  constexpr int cfg = 1;   // bit0: non-temporal policy, bit1: one chunk per thread (settled: rounds 5 and 6 A/B, the switch SIMCLR_BN_CFG is gone)
  const bool nt = (cfg & 1) != 0;
  const int cpr = C / epc;
  const long long nchunks = rows * cpr;
  const bool two = 256 % cpr == 0 && !(cfg & 2);    // both chunks of a thread in the same channel chunk
  const long long blocks = (nchunks + (two ? 512 : 256) - 1) / (two ? 512 : 256);
  SIMCLR_CHECK_ARG(blocks < (1ll << 23), "bn_apply: tensor too large for one launch");
  const int grid = (int)blocks;
  const int mode = res ? (rscale ? 2 : 1) : 0;
#define LA(TT, RR, UU, NN)                                                                                      \
  hipLaunchKernelGGL((bn_apply<TT, RR, UU, NN>), dim3(grid), dim3(256), 0, stream, (const TT*)x, scale, shift, \
                     (const TT*)res, rscale, rshift, (TT*)y, nchunks, C, relu, relu_bits)
#define LB(TT, RR) do { if (two) { if (nt) LA(TT, RR, 2, true); else LA(TT, RR, 2, false); } \
                        else { if (nt) LA(TT, RR, 1, true); else LA(TT, RR, 1, false); } } while (0)
  if (grid > 0) {
    if (dtype == SIMCLR_DT_BF16) {
      if (mode == 0) LB(uint16_t, 0); else if (mode == 1) LB(uint16_t, 1); else LB(uint16_t, 2);
    } else {
      if (mode == 0) LB(float, 0); else if (mode == 1) LB(float, 1); else LB(float, 2);
    }
  }
#undef LB
#undef LA
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// partial [nslot][2][C] must be zeroed by the caller.  nslot >= simclr_bn_bwd_reduce_slots(rows, C, dtype): one slot
// per workgroup, plain stores, run-to-run deterministic; fewer: float atomics into slot (workgroup % nslot).
int simclr_bn_bwd_reduce(const void* dy, const void* x, const void* mask_src, const float* scale,
                         const float* shift, const float* mean, const float* rstd, long long rows, int C,
                         int mask_mode, float* partial, int nslot, int dtype, hipStream_t stream) {
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(C % epc == 0, "bn_bwd_reduce: C=%d must be a multiple of %d", C, epc);
  SIMCLR_CHECK_ARG(mask_mode != 1 || mask_src, "bn_bwd_reduce: mask_mode 1 needs mask_src");
  int rows_per_block;
  const int grid = bwd_reduce_grid(rows, C, epc, &rows_per_block);
  if (dtype == SIMCLR_DT_BF16)
    hipLaunchKernelGGL((bn_bwd_reduce<uint16_t>), dim3(grid), dim3(256), 0, stream, (const uint16_t*)dy,
                       (const uint16_t*)x, (const uint16_t*)mask_src, scale, shift, mean, rstd, rows, C,
                       mask_mode, rows_per_block, partial, nslot);
  else
    hipLaunchKernelGGL((bn_bwd_reduce<float>), dim3(grid), dim3(256), 0, stream, (const float*)dy,
                       (const float*)x, (const float*)mask_src, scale, shift, mean, rstd, rows, C, mask_mode,
                       rows_per_block, partial, nslot);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

int simclr_bn_bwd_finalize(const double* local_sums, const double* global_sums, const float* partial,
                           int nslot, double count, int C, float* dgamma, float* dbeta, int accumulate,
                           float* c1, float* c2, hipStream_t stream) {
  SIMCLR_CHECK_ARG((local_sums != nullptr && global_sums != nullptr) != (partial != nullptr),
                   "bn_bwd_finalize: give (local, global) sums OR partial slots");
  hipLaunchKernelGGL(bn_bwd_finalize, dim3(ceil_div(C, kSlotCh)), dim3(kSlotThreads), 0, stream, local_sums, global_sums,
                     partial, nslot, count, C, dgamma, dbeta, accumulate, c1, c2);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

int simclr_bn_bwd_apply(const void* dy, const void* x, const void* mask_src, const float* scale,
                        const float* shift, const float* mean, const float* rstd, const float* c1,
                        const float* c2, long long rows, int C, int mask_mode, void* dx, void* dmasked,
                        int dtype, hipStream_t stream) {
  // dtype | SIMCLR_FMT_PS_OUT (fp32 only, C a multiple of 32): dx in the pre-split block format, bf16 pieces (common.h)
  const bool ps_out = (dtype & SIMCLR_FMT_PS_OUT) != 0;
  dtype &= 0xff;
  const int epc = dtype == SIMCLR_DT_BF16 ? 8 : 4;
  SIMCLR_CHECK_ARG(C % epc == 0, "bn_bwd_apply: C=%d must be a multiple of %d", C, epc);
  SIMCLR_CHECK_ARG(!ps_out || (dtype == SIMCLR_DT_F32 && C % 32 == 0), "bn_bwd_apply: pre-split output needs fp32 storage and C %% 32 == 0 (C=%d)", C);
  constexpr int cfg = 1;      // see simclr_bn_apply
  const bool nt = (cfg & 1) != 0;
  const int cpr = C / epc;
  const long long nchunks = rows * cpr;
  const bool two = 256 % cpr == 0 && !(cfg & 2);
  const long long blocks = (nchunks + (two ? 512 : 256) - 1) / (two ? 512 : 256);
  SIMCLR_CHECK_ARG(blocks < (1ll << 23), "bn_bwd_apply: tensor too large for one launch");
  const int grid = (int)blocks;
#define LBW(TT, UU, NN) hipLaunchKernelGGL((bn_bwd_apply<TT, UU, NN>), dim3(grid), dim3(256), 0, stream, \
                       (const TT*)dy, (const TT*)x, (const TT*)mask_src, scale, shift, mean, \
                       rstd, c1, c2, nchunks, C, mask_mode, (TT*)dx, (TT*)dmasked)
#define LBX(TT) do { if (two) { if (nt) LBW(TT, 2, true); else LBW(TT, 2, false); } \
                     else { if (nt) LBW(TT, 1, true); else LBW(TT, 1, false); } } while (0)
#define LBP(UU, NN) hipLaunchKernelGGL((bn_bwd_apply<float, UU, NN, true>), dim3(grid), dim3(256), 0, stream, \
                       (const float*)dy, (const float*)x, (const float*)mask_src, scale, shift, mean, \
                       rstd, c1, c2, nchunks, C, mask_mode, (float*)dx, (float*)dmasked)
  if (grid > 0) {
    if (ps_out) { if (two) { if (nt) LBP(2, true); else LBP(2, false); } else { if (nt) LBP(1, true); else LBP(1, false); } }
    else if (dtype == SIMCLR_DT_BF16) LBX(uint16_t); else LBX(float);
  }
#undef LBP
#undef LBX
#undef LBW
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"

extern "C" {

// BN-backward coefficients of the folded form (see bn_fold_*): a, b, d [C] from scale, mean, rstd and c1 = mean(dm), c2 = mean(dm*x^)
int simclr_bn_fold_coeffs(const float* scale, const float* mean, const float* rstd, const float* c1, const float* c2,
                          float* a, float* b, float* d, int C, hipStream_t stream) {
  SIMCLR_CHECK_ARG(C > 0, "bn_fold_coeffs: bad C");
  hipLaunchKernelGGL(bn_fold_coeffs, dim3(ceil_div(C, 256)), dim3(256), 0, stream, scale, mean, rstd, c1, c2, a, b, d, C);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// w [K][N] (T: the compute copy the forward used) and the BN-backward quantities (scale, mean, rstd of the BN; c1 = mean(dm),
// c2 = mean(dm*x^) from simclr_bn_bwd_finalize): outputs a, b, d [N], wb [K][N] fp32 = w*b, wext [K][N+K] (T) columns < N = w*a,
// e [K] = w d
int simclr_bn_fold_pre(const void* w, const float* scale, const float* mean, const float* rstd, const float* c1,
                       const float* c2, float* a, float* b, float* d, float* wb, void* wext, float* e, int K, int N,
                       int dtype, hipStream_t stream) {
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "bn_fold_pre: bad dtype %d", dtype);
  if (dtype == SIMCLR_DT_BF16)
    hipLaunchKernelGGL((bn_fold_pre<uint16_t>), dim3(K), dim3(256), 0, stream, (const uint16_t*)w, scale, mean, rstd, c1, c2,
                       a, b, d, wb, (uint16_t*)wext, e, K, N);
  else
    hipLaunchKernelGGL((bn_fold_pre<float>), dim3(K), dim3(256), 0, stream, (const float*)w, scale, mean, rstd, c1, c2, a, b,
                       d, wb, (float*)wext, e, K, N);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// dw [K][N] fp32 = a*t1 + b*gw + cs (x) d  (t1 = h^T dm, gw = (h^T h) w, cs [K] = colsum h as fp64 `cs` or fp32 `cs32`);  wext columns N.. = q^T (q [K][K] = wb w^T)
int simclr_bn_fold_post(const float* t1, const float* gw, const double* cs, const float* cs32, const float* a, const float* b,
                        const float* d, const float* q, float* dw, void* wext, int K, int N, int accumulate, int dtype,
                        hipStream_t stream) {
  SIMCLR_CHECK_ARG((cs != nullptr) != (cs32 != nullptr), "bn_fold_post: give the column sums as fp64 OR fp32");
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "bn_fold_post: bad dtype %d", dtype);
  const long long total = (long long)K * N + (long long)K * K;
  const int grid = (int)min((total + 255) / 256, 1ll << 20);
  if (dtype == SIMCLR_DT_BF16)
    hipLaunchKernelGGL((bn_fold_post<uint16_t>), dim3(grid), dim3(256), 0, stream, t1, gw, cs, cs32, a, b, d, q, dw, (uint16_t*)wext, K, N, accumulate);
  else
    hipLaunchKernelGGL((bn_fold_post<float>), dim3(grid), dim3(256), 0, stream, t1, gw, cs, cs32, a, b, d, q, dw, (float*)wext, K, N, accumulate);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"

extern "C" {

// sums [2][N] fp64 with sums[0] = sum(dm) valid: fills sums[1] = sum(dm * x^) from t1 = h^T dm [K][N] and w [K][N] (T)
int simclr_bn_fold_s2(const float* t1, const void* w, const float* mean, const float* rstd, double* sums, int K, int N,
                      int dtype, hipStream_t stream) {
  SIMCLR_CHECK_ARG(dtype == SIMCLR_DT_BF16 || dtype == SIMCLR_DT_F32, "bn_fold_s2: bad dtype %d", dtype);
  if (dtype == SIMCLR_DT_BF16)
    hipLaunchKernelGGL((bn_fold_s2<uint16_t>), dim3(ceil_div(N, 32)), dim3(256), 0, stream, t1, (const uint16_t*)w, mean, rstd, sums, K, N);
  else
    hipLaunchKernelGGL((bn_fold_s2<float>), dim3(ceil_div(N, 32)), dim3(256), 0, stream, t1, (const float*)w, mean, rstd, sums, K, N);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
