// Fused NT-Xent (SimCLR contrastive loss) forward + backward for gfx950.
//
// Replaces the un-fused TF graph of /root/reference/tf2/objective.py:53-87 (four
// [n,N] matmuls, a -1e9 diagonal mask, two 2N-wide softmax cross-entropies) and the
// consumers of logits_ab in /root/reference/tf2/metrics.py:28-35 with a flash-style
// sweep: S = Q.K^T/T is produced tile by tile on the fp32-input matrix cores
// (v_mfma_f32_16x16x4_f32, exact f32), reduced online (running max / sum-exp per
// row, per lane, merged with wave shuffles) and never written to HBM.
//
// Formulation.  Q = [z1_local; z2_local]  (2n rows, this replica's views),
//               K = [z1_all;   z2_all  ]  (2N rows, all replicas; N = R*n).
// For query row q:   q <  n (view a, i=q):   masked col = rank*n+i     (the aa diagonal,
//                                              objective.py:76-77), positive col = N+rank*n+i (ab)
//                    q >= n (view b, i=q-n): masked col = N+rank*n+i   (bb diagonal, :78-79),
//                                              positive col = rank*n+i (ba)
// loss = (1/n) sum_q [ logsumexp_{col != masked} S[q,col] - S[q,pos] ]      (:83-87)
// The masked column is skipped rather than shifted by -1e9 (exp(-1e9)=0 in fp32).
//
// MFMA mapping (16x16x4 f32): a = streamed tile row fragment, b = fixed-row fragment,
// so D[streamed=(lane>>4)*4+reg][fixed=lane&15]: every lane owns ONE fixed row and 4
// streamed rows per 16x16 tile -> the online softmax is lane-local; only the final
// merge over the 4 lane groups needs shuffles (xor 16, 32).
#include "common.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kTile = 64;      // rows per LDS tile / fixed rows per workgroup
constexpr int kPartStride = 8; // floats per (split,row) partial record

__device__ __forceinline__ void row_cols(int q, int n, int N, int rank, int& mask_col, int& pos_col) {
  if (q < n) { mask_col = rank * n + q; pos_col = N + rank * n + q; }
  else { int i = q - n; mask_col = N + rank * n + i; pos_col = rank * n + i; }
}

// A streamed tile (kTile rows x D floats) travels global -> registers -> LDS: tile_fetch issues the loads of the NEXT tile
// before the MFMAs of the current one, tile_store puts them into the (XOR-swizzled) LDS tile after the barrier -- the
// global latency runs under the matrix work instead of between two barriers.
template <int D>
__device__ __forceinline__ void tile_fetch(float4* pf, const float* __restrict__ src, int row0, int nrows_total, int tid) {
  constexpr int C = D / 4;
#pragma unroll
  for (int j = 0; j < kTile * C / 256; ++j) {
    const int idx = tid + j * 256, r = idx / C, c = idx % C;
    pf[j] = (row0 + r < nrows_total) ? *(const float4*)(src + (size_t)(row0 + r) * D + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int D>
__device__ __forceinline__ void tile_store(float* lds, const float4* pf, int tid) {
  constexpr int C = D / 4;
#pragma unroll
  for (int j = 0; j < kTile * C / 256; ++j) {
    const int idx = tid + j * 256, r = idx / C, c = idx % C;
    *(float4*)(lds + r * D + ((c ^ (r & 15)) * 4)) = pf[j];
  }
}

// un-pipelined form (the backward sweeps): each 16-byte piece goes global -> LDS straight away
template <int D>
__device__ __forceinline__ void load_tile(float* lds, const float* __restrict__ src, int row0, int nrows_total, int tid) {
  constexpr int C = D / 4;
  for (int idx = tid; idx < kTile * C; idx += 256) {
    const int r = idx / C, c = idx % C;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < nrows_total) v = *(const float4*)(src + (size_t)(row0 + r) * D + c * 4);
    *(float4*)(lds + r * D + ((c ^ (r & 15)) * 4)) = v;
  }
}

// online (max,sum) merge in the base-2 domain
__device__ __forceinline__ void ml_merge(float& m, float& l, float m2, float l2) {
  float mn = fmaxf(m, m2);
  float a = (m == -INFINITY) ? 0.f : l * exp2f(m - mn);
  float b = (m2 == -INFINITY) ? 0.f : l2 * exp2f(m2 - mn);
  m = mn; l = a + b;
}
__device__ __forceinline__ void arg_merge(float& v, int& i, float v2, int i2) {
  if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}

// ------------------------------------------------------------------------------
// Last-arriver hand-off between the workgroups of ONE launch (cdna_hip_programming.md section 5 / Guideline 16): a workgroup
// that has written its partial results calls arrive_is_last(); exactly one caller per counter -- the one that draws ticket
// `expected - 1` -- gets true and may then read what all the others wrote.  Publisher: every wave drains its stores, barrier,
// ONE lane: agent-scope release fence, drained, relaxed agent-scope ticket; the last arriver: ONE agent-scope acquire fence,
// barrier, plain loads.  The last arriver also puts the counter back to zero (the ticket buffer is zero when the library
// creates it and every launch leaves it zero).  Placement-independent: correct wherever the workgroups run.
// ------------------------------------------------------------------------------
__device__ __forceinline__ bool arrive_is_last(unsigned* counter, unsigned expected, int* sh_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == expected - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *sh_flag = last;
  }
  __syncthreads();
  return *sh_flag != 0;
}

// ------------------------------------------------------------------------------
// Forward sweep: per (query row, key split) partial statistics.
// part[split][row][8] = {m0, l0, m1, l1, pos, argval, argidx(bits), 0}
//   set0 = key cols [0,N), set1 = key cols [N,2N); values are logits*log2(e).
// ------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void ntxent_fwd_partial(
    const float* __restrict__ zq, const float* __restrict__ zk, int n, int N, int rank,
    float scale2 /* log2(e)/T */, int tiles_per_split, float* __restrict__ part, int rows_pad,
    float* __restrict__ row_stats, float* __restrict__ rowterm, float* __restrict__ out, unsigned* __restrict__ tickets) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int sh_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, fl = lane & 15;
  const int q = blockIdx.x * kTile + wave * 16 + fl;
  const int two_n = 2 * n, two_N = 2 * N;
  int mask_col, pos_col;
  row_cols(q, n, N, rank, mask_col, pos_col);

  float4 ff[D / 16];
#pragma unroll
  for (int s = 0; s < D / 16; ++s) {
    ff[s] = (q < two_n) ? *(const float4*)(zq + (size_t)q * D + 16 * s + 4 * g)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float m0 = -INFINITY, l0 = 0.f, m1 = -INFINITY, l1 = 0.f, pos = -INFINITY, av = -INFINITY;
  int ai = 0x7fffffff;

  const int tile_begin = blockIdx.y * tiles_per_split;
  const int ntiles = (two_N + kTile - 1) / kTile;
  const int tile_end = min(ntiles, tile_begin + tiles_per_split);
  float4 pf[kTile * (D / 4) / 256];
  if (tile_begin < tile_end) tile_fetch<D>(pf, zk, tile_begin * kTile, two_N, tid);
  for (int kt = tile_begin; kt < tile_end; ++kt) {
    __syncthreads();
    tile_store<D>(lds, pf, tid);
    __syncthreads();
    if (kt + 1 < tile_end) tile_fetch<D>(pf, zk, (kt + 1) * kTile, two_N, tid);
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int trow = sub * 16 + fl;
#pragma unroll
      for (int s = 0; s < D / 16; ++s) {
        float4 tf = *(const float4*)(lds + trow * D + (((4 * s + g) ^ (trow & 15)) * 4));
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tf.x, ff[s].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tf.y, ff[s].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tf.z, ff[s].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tf.w, ff[s].w, acc, 0, 0, 0);
      }
      const int col0 = kt * kTile + sub * 16 + g * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = col0 + r;
        const float t = acc[r] * scale2;
        if (col == pos_col) pos = t;
        if (col < two_N && col != mask_col) {
          if (col < N) {
            float mn = fmaxf(m0, t);
            l0 = l0 * exp2f(m0 - mn) + exp2f(t - mn);
            m0 = mn;
          } else {
            float mn = fmaxf(m1, t);
            l1 = l1 * exp2f(m1 - mn) + exp2f(t - mn);
            m1 = mn;
            if (t > av || (t == av && col < ai)) { av = t; ai = col; }
          }
        }
      }
    }
  }
  // merge the 4 lane groups that share this fixed row
#pragma unroll
  for (int o = 16; o <= 32; o <<= 1) {
    float om0 = __shfl_xor(m0, o, 64), ol0 = __shfl_xor(l0, o, 64);
    float om1 = __shfl_xor(m1, o, 64), ol1 = __shfl_xor(l1, o, 64);
    float op = __shfl_xor(pos, o, 64), oav = __shfl_xor(av, o, 64);
    int oai = __shfl_xor(ai, o, 64);
    ml_merge(m0, l0, om0, ol0);
    ml_merge(m1, l1, om1, ol1);
    pos = fmaxf(pos, op);
    arg_merge(av, ai, oav, oai);
  }
  if (g == 0 && q < two_n) {
    float* p = part + ((size_t)blockIdx.y * rows_pad + q) * kPartStride;
    p[0] = m0; p[1] = l0; p[2] = m1; p[3] = l1; p[4] = pos; p[5] = av;
    p[6] = __int_as_float(ai); p[7] = 0.f;
  }
  // ---- finalize in the same launch.  Level 1: the last of this row tile's key splits merges the splits of its 64 rows
  // (16 lanes per row, fixed xor-shuffle tree: the arithmetic of the former ntxent_finalize_rows launch); level 2: the
  // last row tile adds the 2n row terms in a fixed order (the former ntxent_reduce_out launch).
  if (!arrive_is_last(tickets + blockIdx.x, gridDim.y, &sh_last)) return;
  const int nsplit = gridDim.y;
  for (int pass = 0; pass < kTile / 16; ++pass) {
    const int fq = blockIdx.x * kTile + pass * 16 + (tid >> 4);
    const int j = tid & 15;
    float fm0 = -INFINITY, fl0 = 0.f, fm1 = -INFINITY, fl1 = 0.f, fpos = -INFINITY, fav = -INFINITY;
    int fai = 0x7fffffff;
    if (fq < two_n) {
      for (int sp = j; sp < nsplit; sp += 16) {
        const float4 a = *(const float4*)(part + ((size_t)sp * rows_pad + fq) * kPartStride);
        const float4 b = *(const float4*)(part + ((size_t)sp * rows_pad + fq) * kPartStride + 4);
        ml_merge(fm0, fl0, a.x, a.y);
        ml_merge(fm1, fl1, a.z, a.w);
        fpos = fmaxf(fpos, b.x);
        arg_merge(fav, fai, b.y, __float_as_int(b.z));
      }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      float om0 = __shfl_xor(fm0, o, 64), ol0 = __shfl_xor(fl0, o, 64);
      float om1 = __shfl_xor(fm1, o, 64), ol1 = __shfl_xor(fl1, o, 64);
      float op = __shfl_xor(fpos, o, 64), oav = __shfl_xor(fav, o, 64);
      int oai = __shfl_xor(fai, o, 64);
      ml_merge(fm0, fl0, om0, ol0);
      ml_merge(fm1, fl1, om1, ol1);
      fpos = fmaxf(fpos, op);
      arg_merge(fav, fai, oav, oai);
    }
    if (j == 0 && fq < two_n) {
      const float lse_ab2 = fm1 + log2f(fl1);
      float mf = fm0, lf = fl0;
      ml_merge(mf, lf, fm1, fl1);
      const float lse2 = mf + log2f(lf);
      row_stats[2 * fq] = lse2;
      row_stats[2 * fq + 1] = lse_ab2;
      int mc, pc;
      row_cols(fq, n, N, rank, mc, pc);
      rowterm[2 * fq] = (lse2 - fpos) * kLn2;
      rowterm[2 * fq + 1] = (fq < n && fai == pc) ? 1.f : 0.f;
    }
  }
  if (!arrive_is_last(tickets + gridDim.x, gridDim.x, &sh_last)) return;
  double* shd = (double*)lds;                       // the key tile is no longer needed: 2 x 256 doubles
  double loss = 0.0, hit = 0.0;
  for (int r = tid; r < two_n; r += 256) { loss += (double)rowterm[2 * r]; hit += (double)rowterm[2 * r + 1]; }
  shd[tid] = loss;
  shd[256 + tid] = hit;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) { shd[tid] += shd[tid + st]; shd[256 + tid] += shd[256 + tid + st]; }
    __syncthreads();
  }
  if (tid == 0) {
    out[0] = (float)(shd[0] / n);
    out[1] = (float)(shd[256] / n);
  }
}

// ------------------------------------------------------------------------------
// Backward sweep (recomputes S).  FIXED_IS_QUERY=true : fixed rows = queries,
//   streamed = keys,  gpart[split][q][D]   = sum_keys dS[q,key] * K[key,:]
// FIXED_IS_QUERY=false: fixed rows = keys, streamed = queries,
//   gpart[split][key][D] = sum_q dS[q,key] * Q[q,:]
// dS = softmax - onehot(pos) (masked col -> 0); the 1/(n*T) * upstream factor is
// applied by the combine kernel.  The query-fixed instance also accumulates the
// contrast-entropy term of tf2/metrics.py:33-35 (a rows, ab block only).
// ------------------------------------------------------------------------------
template <int D, bool FIXED_IS_QUERY>
__device__ __forceinline__ void ntxent_bwd_sweep_body(
    float* lds, const float* __restrict__ fixed_mat, int fixed_rows, const float* __restrict__ stream_mat,
    int stream_rows, int n, int N, int rank, float scale2, const float* __restrict__ row_stats,
    int tiles_per_split, float* __restrict__ gpart, int rows_pad, float* __restrict__ epart) {
  float* stats_s = lds + kTile * D;  // [64][2] row stats of the streamed queries (key-fixed mode)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, fl = lane & 15;
  const int f = blockIdx.x * kTile + wave * 16 + fl;
  const int two_N = 2 * N, two_n = 2 * n;

  float4 ff[D / 16];
#pragma unroll
  for (int s = 0; s < D / 16; ++s) {
    ff[s] = (f < fixed_rows) ? *(const float4*)(fixed_mat + (size_t)f * D + 16 * s + 4 * g)
                             : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  int f_mask = -1, f_pos = -1;
  float f_lse = 0.f, f_lse_ab = 0.f;
  if (FIXED_IS_QUERY) {
    row_cols(f, n, N, rank, f_mask, f_pos);
    if (f < two_n) { f_lse = row_stats[2 * f]; f_lse_ab = row_stats[2 * f + 1]; }
  }
  f32x4 dacc[D / 16];
#pragma unroll
  for (int i = 0; i < D / 16; ++i) dacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float ent = 0.f;

  const int ntiles = (stream_rows + kTile - 1) / kTile;
  const int tile_begin = blockIdx.y * tiles_per_split;
  const int tile_end = min(ntiles, tile_begin + tiles_per_split);
  // (prefetching one tile ahead into registers, as the forward sweep does, was measured SLOWER here: on top of the 32 + 32
  // accumulator / fixed-row registers it costs a wave of occupancy or spills -- cfg3 shape 183 -> 193 us, profiles/r03_notes.md)
  for (int kt = tile_begin; kt < tile_end; ++kt) {
    __syncthreads();
    load_tile<D>(lds, stream_mat, kt * kTile, stream_rows, tid);
    if (!FIXED_IS_QUERY && tid < 2 * kTile) {
      int qq = kt * kTile + (tid >> 1);
      stats_s[tid] = (qq < two_n) ? row_stats[2 * qq + (tid & 1)] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int trow = sub * 16 + fl;
#pragma unroll
      for (int s = 0; s < D / 16; ++s) {
        float4 tf = *(const float4*)(lds + trow * D + (((4 * s + g) ^ (trow & 15)) * 4));
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tf.x, ff[s].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tf.y, ff[s].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tf.z, ff[s].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tf.w, ff[s].w, acc, 0, 0, 0);
      }
      float ds[4];
      const int s0 = kt * kTile + sub * 16 + g * 4;  // streamed row of acc[0]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = acc[r] * scale2;
        int q, col, mask_col, pos_col;
        float lse, lse_ab;
        if (FIXED_IS_QUERY) {
          q = f; col = s0 + r; mask_col = f_mask; pos_col = f_pos; lse = f_lse; lse_ab = f_lse_ab;
        } else {
          q = s0 + r; col = f;
          row_cols(q, n, N, rank, mask_col, pos_col);
          lse = stats_s[2 * (sub * 16 + g * 4 + r)];
          lse_ab = 0.f;
        }
        float d = 0.f;
        if (q < two_n && col < two_N && col != mask_col) {
          d = exp2f(t - lse);
          if (col == pos_col) d -= 1.f;
          if (FIXED_IS_QUERY && q < n && col >= N) {
            float pab = exp2f(t - lse_ab);
            ent -= pab * __logf(pab + 1e-8f);
          }
        }
        ds[r] = d;
      }
      // dF^T[d][fixed] += sum_streamed T[streamed][d] * dS[streamed][fixed]
#pragma unroll
      for (int dt = 0; dt < D / 16; ++dt) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int srow = sub * 16 + 4 * g + u;
          const int dcol = dt * 16 + fl;
          float a = lds[srow * D + ((((dcol >> 2) ^ (srow & 15)) << 2) | (dcol & 3))];
          dacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ds[u], dacc[dt], 0, 0, 0);
        }
      }
    }
  }
  if (f < fixed_rows) {
    float* gp = gpart + ((size_t)blockIdx.y * rows_pad + f) * D;
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt)
      *(float4*)(gp + dt * 16 + 4 * g) = make_float4(dacc[dt][0], dacc[dt][1], dacc[dt][2], dacc[dt][3]);
  }
  if (FIXED_IS_QUERY) {
    ent += __shfl_xor(ent, 16, 64);
    ent += __shfl_xor(ent, 32, 64);
    if (g == 0 && f < two_n) epart[(size_t)blockIdx.y * rows_pad + f] = ent;
  }
}

// Both sweeps AND their reductions in ONE launch: blockIdx.z = 0 query-fixed (gradient wrt the local rows + the entropy
// term), blockIdx.z = 1 key-fixed (gradient wrt the gathered rows); they are independent, so they share the chip.  The last
// of a row tile's splits to arrive (arrive_is_last) adds the splits of its 64 rows in split order and scales them --
// dz_local = scale * sum_split gq, dz_all = scale * sum_split gk: the arithmetic of the former ntxent_combine_all launch --
// and, on the query side, the entropy terms of its rows; the last query tile adds the per-tile entropy sums in tile order:
// out[2] = contrast entropy (tf2/metrics.py:33-35).
template <int D>
__global__ __launch_bounds__(256) void ntxent_bwd_sweeps(
    const float* __restrict__ z_local, const float* __restrict__ z_all, int n, int N, int rank, float scale2,
    const float* __restrict__ row_stats, int tiles_k, int tiles_q, float* __restrict__ gq, int rows_pad_q,
    float* __restrict__ gk, int rows_pad_k, float* __restrict__ epart, int gxq, int gyq, int gxk, int gyk,
    float scale, float* __restrict__ dz_local, float* __restrict__ dz_all, double* __restrict__ etile,
    float* __restrict__ out, unsigned* __restrict__ tickets) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int sh_last;
  const bool isq = blockIdx.z == 0;
  if (isq) {
    if ((int)blockIdx.x >= gxq || (int)blockIdx.y >= gyq) return;
    ntxent_bwd_sweep_body<D, true>(lds, z_local, 2 * n, z_all, 2 * N, n, N, rank, scale2, row_stats, tiles_k, gq,
                                   rows_pad_q, epart);
  } else {
    if ((int)blockIdx.x >= gxk || (int)blockIdx.y >= gyk) return;
    ntxent_bwd_sweep_body<D, false>(lds, z_all, 2 * N, z_local, 2 * n, n, N, rank, scale2, row_stats, tiles_q, gk,
                                    rows_pad_k, (float*)nullptr);
  }
  const int nsplit = isq ? gyq : gyk;
  if (!arrive_is_last(tickets + (isq ? 0 : gxq) + blockIdx.x, nsplit, &sh_last)) return;
  const int tid = threadIdx.x;
  const float* gp = isq ? gq : gk;
  const int rows_pad = isq ? rows_pad_q : rows_pad_k, rows = isq ? 2 * n : 2 * N;
  float* dst = isq ? dz_local : dz_all;
  const int r0 = blockIdx.x * kTile;
  for (int idx = tid; idx < kTile * (D / 4); idx += 256) {
    const int r = r0 + idx / (D / 4), c = idx % (D / 4);
    if (r >= rows) continue;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sp = 0; sp < nsplit; ++sp) {
      const float4 v = *(const float4*)(gp + ((size_t)sp * rows_pad + r) * D + c * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    acc.x *= scale; acc.y *= scale; acc.z *= scale; acc.w *= scale;
    *(float4*)(dst + (size_t)r * D + c * 4) = acc;
  }
  if (!isq) return;
  // entropy of this tile's a-rows (q < n), splits in order per row, rows by a fixed tree
  double* shd = (double*)lds;
  double e = 0.0;
  if (tid < kTile && r0 + tid < n)
    for (int sp = 0; sp < nsplit; ++sp) e += (double)epart[(size_t)sp * rows_pad_q + r0 + tid];
  __syncthreads();                                    // every wave is done with the key tile in lds
  shd[tid] = e;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) shd[tid] += shd[tid + st];
    __syncthreads();
  }
  if (tid == 0) etile[blockIdx.x] = shd[0];
  if (!arrive_is_last(tickets + gxq + gxk, gxq, &sh_last)) return;
  if (tid == 0) {
    double tot = 0.0;
    for (int t = 0; t < gxq; ++t) tot += etile[t];
    out[2] = (float)(tot / n);
  }
}

// ---- l2 normalise (tf.math.l2_normalize, tf2/objective.py:53-54) -------------
// one wave per row: z = x * rsqrt(max(sum x^2, 1e-12))
__global__ void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ z,
                                  float* __restrict__ inv, int rows, int D) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) { float v = x[(size_t)row * D + d]; ss += v * v; }
  ss = wave_sum(ss);
  const float r = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
  for (int d = lane; d < D; d += 64) z[(size_t)row * D + d] = x[(size_t)row * D + d] * r;
  if (lane == 0) inv[row] = r;
}
// dx = (dz - z * (z . dz)) * inv        (exact for sum x^2 > 1e-12)
__global__ void l2norm_bwd_kernel(const float* __restrict__ z, const float* __restrict__ inv,
                                  const float* __restrict__ dz, float* __restrict__ dx, int rows,
                                  int D) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float dot = 0.f;
  for (int d = lane; d < D; d += 64) dot += z[(size_t)row * D + d] * dz[(size_t)row * D + d];
  dot = wave_sum(dot);
  const float r = inv[row];
  for (int d = lane; d < D; d += 64)
    dx[(size_t)row * D + d] = (dz[(size_t)row * D + d] - z[(size_t)row * D + d] * dot) * r;
}

// dense logits_ab materialisation for API parity (tf2/objective.py:80,89): [n, N]
__global__ void ntxent_logits_ab_kernel(const float* __restrict__ zq, const float* __restrict__ zk,
                                        int n, int N, int D, float inv_t, float* __restrict__ out) {
  const int i = blockIdx.y;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
    const float* a = zq + (size_t)i * D;
    const float* b = zk + (size_t)(N + j) * D;
    float s = 0.f;
    for (int d = 0; d < D; ++d) s = fmaf(a[d], b[d], s);
    out[(size_t)i * N + j] = s * inv_t;
  }
}

struct Plan { int rows_pad_q, rows_pad_k, qsplit, ksplit, tiles_q, tiles_k; };
Plan make_plan(int n, int N) {
  Plan p;
  const int qtiles = ceil_div(2 * n, kTile), ktiles = ceil_div(2 * N, kTile);
  p.rows_pad_q = qtiles * kTile;
  p.rows_pad_k = ktiles * kTile;
  // split the streamed dimension so that roughly >= 512 workgroups exist
  int ks = max(1, min(ktiles, 512 / max(1, qtiles)));
  p.tiles_k = ceil_div(ktiles, ks);
  p.ksplit = ceil_div(ktiles, p.tiles_k);
  int qs = max(1, min(qtiles, 512 / max(1, ktiles)));
  p.tiles_q = ceil_div(qtiles, qs);
  p.qsplit = ceil_div(qtiles, p.tiles_q);
  return p;
}
size_t ws_floats(int n, int N, int D) {
  Plan p = make_plan(n, N);
  size_t part = (size_t)p.ksplit * p.rows_pad_q * kPartStride;
  size_t gq = (size_t)p.ksplit * p.rows_pad_q * D;
  size_t gk = (size_t)p.qsplit * p.rows_pad_k * D;
  size_t ep = (size_t)p.ksplit * p.rows_pad_q;
  size_t rowterm = (size_t)2 * p.rows_pad_q;
  size_t etile = (size_t)2 * (p.rows_pad_q / kTile) + 2;        // one double per query tile (8-byte aligned below)
  return part + gq + gk + ep + rowterm + etile;
}

// Ticket counters of the in-launch reductions: library-owned, one zeroed buffer per stream (launches of a stream are
// ordered; every launch leaves its counters at zero), created on first use.
constexpr int kTickets = 16384;
struct TicketBuf { hipStream_t stream; unsigned* buf; };
TicketBuf g_tickets[8];
int g_tickets_n = 0;
unsigned* tickets_for(hipStream_t stream) {
  for (int i = 0; i < g_tickets_n; ++i)
    if (g_tickets[i].stream == stream) return g_tickets[i].buf;
  if (g_tickets_n == 8) return nullptr;
  unsigned* b = nullptr;
  if (hipMalloc((void**)&b, kTickets * sizeof(unsigned)) != hipSuccess || hipMemset(b, 0, kTickets * sizeof(unsigned)) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  g_tickets[g_tickets_n++] = TicketBuf{stream, b};
  return b;
}

}  // namespace

extern "C" {

size_t simclr_ntxent_workspace_bytes(int n, int N, int D) { return ws_floats(n, N, D) * sizeof(float); }

int simclr_l2norm_fwd(const float* x, float* z, float* inv, int rows, int D, hipStream_t stream) {
  SIMCLR_CHECK_ARG(rows > 0 && D > 0, "l2norm_fwd: bad shape rows=%d D=%d", rows, D);
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, x, z, inv, rows, D);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}
int simclr_l2norm_bwd(const float* z, const float* inv, const float* dz, float* dx, int rows, int D,
                      hipStream_t stream) {
  SIMCLR_CHECK_ARG(rows > 0 && D > 0, "l2norm_bwd: bad shape rows=%d D=%d", rows, D);
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, z, inv, dz, dx, rows, D);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// Forward: out[0]=loss, out[1]=contrast_acc; row_stats[2n][2] kept for backward.
int simclr_ntxent_fwd(const float* z_local, const float* z_all, int n, int N, int D, int rank,
                      float temperature, float* out, float* row_stats, void* workspace,
                      hipStream_t stream) {
  SIMCLR_CHECK_ARG(n > 0 && N >= n && N % n == 0, "ntxent_fwd: need N = R*n (n=%d N=%d)", n, N);
  SIMCLR_CHECK_ARG(rank >= 0 && rank < N / n, "ntxent_fwd: rank %d out of range", rank);
  SIMCLR_CHECK_ARG(D == 64 || D == 128 || D == 256, "ntxent_fwd: D must be 64/128/256 (got %d)", D);
  SIMCLR_CHECK_ARG(temperature > 0.f, "ntxent_fwd: temperature must be > 0");
  Plan p = make_plan(n, N);
  float* part = (float*)workspace;
  const float scale2 = kLog2e / temperature;
  dim3 grid(p.rows_pad_q / kTile, p.ksplit);
  const size_t lds = (size_t)kTile * D * sizeof(float);
  float* rowterm = part + (size_t)p.ksplit * p.rows_pad_q * kPartStride + (size_t)p.ksplit * p.rows_pad_q * D +
                   (size_t)p.qsplit * p.rows_pad_k * D + (size_t)p.ksplit * p.rows_pad_q;
  unsigned* tickets = tickets_for(stream);
  SIMCLR_CHECK_ARG(tickets != nullptr, "ntxent_fwd: cannot create the ticket buffer of this stream");
  SIMCLR_CHECK_ARG((int)grid.x + 1 <= kTickets, "ntxent_fwd: 2n = %d needs more than %d ticket counters", 2 * n, kTickets);
  // ONE launch: the sweep, the per-row merge of the key splits (last split of a row tile) and the loss / accuracy sums
  // (last row tile)
#define LAUNCH_FWD(DD)                                                                          \
  hipLaunchKernelGGL((ntxent_fwd_partial<DD>), grid, dim3(256), lds, stream, z_local, z_all, n, \
                     N, rank, scale2, p.tiles_k, part, p.rows_pad_q, row_stats, rowterm, out, tickets)
  if (D == 64) LAUNCH_FWD(64); else if (D == 128) LAUNCH_FWD(128); else LAUNCH_FWD(256);
#undef LAUNCH_FWD
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// Backward: dz_local[2n][D] = d loss*grad_scale / d(query-side z_local),
//           dz_all[2N][D]   = d loss*grad_scale / d(key-side z_all) (to be reduce-scattered
//           across replicas: transpose of the concat, tf2/objective.py:114-122).
// Also writes out[2] = contrast entropy (tf2/metrics.py:33-35).
int simclr_ntxent_bwd(const float* z_local, const float* z_all, int n, int N, int D, int rank,
                      float temperature, const float* row_stats, float grad_scale, float* dz_local,
                      float* dz_all, float* out, void* workspace, hipStream_t stream) {
  SIMCLR_CHECK_ARG(n > 0 && N >= n && N % n == 0, "ntxent_bwd: need N = R*n (n=%d N=%d)", n, N);
  SIMCLR_CHECK_ARG(D == 64 || D == 128 || D == 256, "ntxent_bwd: D must be 64/128/256 (got %d)", D);
  Plan p = make_plan(n, N);
  float* part = (float*)workspace;
  float* gq = part + (size_t)p.ksplit * p.rows_pad_q * kPartStride;
  float* gk = gq + (size_t)p.ksplit * p.rows_pad_q * D;
  float* ep = gk + (size_t)p.qsplit * p.rows_pad_k * D;
  const float scale2 = kLog2e / temperature;
  const size_t lds = (size_t)(kTile * D + 2 * kTile) * sizeof(float);
  dim3 gridq(p.rows_pad_q / kTile, p.ksplit), gridk(p.rows_pad_k / kTile, p.qsplit);
  dim3 gridb(max(gridq.x, gridk.x), max(gridq.y, gridk.y), 2);
  const float scale = grad_scale / (temperature * (float)n);
  // per-tile entropy sums (double): behind the row terms of the forward, 8-byte aligned
  float* rowterm = ep + (size_t)p.ksplit * p.rows_pad_q;
  double* etile = (double*)(((uintptr_t)(rowterm + (size_t)2 * p.rows_pad_q) + 7) & ~(uintptr_t)7);
  unsigned* tickets = tickets_for(stream);
  SIMCLR_CHECK_ARG(tickets != nullptr, "ntxent_bwd: cannot create the ticket buffer of this stream");
  SIMCLR_CHECK_ARG((int)(gridq.x + gridk.x) + 1 <= kTickets, "ntxent_bwd: 2N = %d needs more than %d ticket counters", 2 * N, kTickets);
  // ONE launch: both sweeps, the split sums of every row tile (its last split) and the entropy (last query tile)
#define LAUNCH_BWD(DD)                                                                                       \
  hipLaunchKernelGGL((ntxent_bwd_sweeps<DD>), gridb, dim3(256), lds, stream, z_local, z_all, n, N, rank, scale2, \
                     row_stats, p.tiles_k, p.tiles_q, gq, p.rows_pad_q, gk, p.rows_pad_k, ep, (int)gridq.x,    \
                     (int)gridq.y, (int)gridk.x, (int)gridk.y, scale, dz_local, dz_all, etile, out, tickets)
  if (D == 64) LAUNCH_BWD(64); else if (D == 128) LAUNCH_BWD(128); else LAUNCH_BWD(256);
#undef LAUNCH_BWD
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// Optional dense logits_ab [n,N] for API parity with objective.py:89 (not on the hot path).
int simclr_ntxent_logits_ab(const float* z_local, const float* z_all, int n, int N, int D,
                            float temperature, float* logits_ab, hipStream_t stream) {
  SIMCLR_CHECK_ARG(n > 0 && N > 0 && D > 0, "ntxent_logits_ab: bad shape");
  hipLaunchKernelGGL(ntxent_logits_ab_kernel, dim3(ceil_div(N, 256), n), dim3(256), 0, stream,
                     z_local, z_all, n, N, D, 1.0f / temperature, logits_ab);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
