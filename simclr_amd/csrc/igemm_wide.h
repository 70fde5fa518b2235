// conv_igemm_wide: 256 x 256 x 64 bf16 implicit-GEMM tile for the MFMA- / L2-bound forward and dgrad launches
// (tf2/resnet.py:431-435, 446-453, 460-467: the 1x1 and 3x3 convolutions of a bottleneck block at 14^2 / 7^2, the strided
// 3x3 and projection layers), built on the eight-phase "ping-pong" schedule of cdna_hip_programming.md section 5.
// Included by conv.hip inside its anonymous namespace (uses ConvP, mma_bf16, the tile / swizzle conventions of
// conv_igemm_persistent: 128-byte LDS rows, 16-byte chunk ^= row & 7, weight fragment = MFMA A operand).
//
// Geometry.  8 waves = 2 (wr) x 4 (wc).  Wave (wr, wc) owns output rows {64 wr .. +64} of the LOWER 128-row half AND the
// same rows of the UPPER half, and columns {32 wc .. +32} of the lower and of the upper 128-column half: its 128 x 64
// accumulator is four 64 x 32 quadrants, each fed by exactly ONE A half-tile (128 rows x 128 B) and ONE B half-tile.
// A k-tile (64 reduction elements) is therefore consumed in four phases
//     0: Alo x Blo   1: Alo x Bhi   2: Ahi x Bhi   3: Ahi x Blo      (16 MFMA each)
// and a half-tile is dead -- its ring slot free for the k-tile after next -- as soon as its fragments are in registers
// (Alo after phase 0, Bhi after 1, Ahi after 2, Blo after 0: its fragments stay in registers for phase 3).
// Ring: 2 k-tiles x 4 half-tiles x 16 KB = 128 KB of LDS, filled by LDS-DMA (buffer_load ... lds, 2 x 1 KB per wave and
// half-tile; out-of-range offsets return the zeros of the padding).  One half-tile is issued per phase, five half-tiles ahead of its first read:
//     phase (t, 0) issues Blo(t+1), (t, 1) Bhi(t+1), (t, 2) Ahi(t+1), (t, 3) Alo(t+2)
// and every phase ends its load segment with a COUNTED s_waitcnt vmcnt(6): three half-tiles stay in flight, the one
// issued three phases ago has landed and may be read from the NEXT phase on (the wait sits before the phase's first
// barrier, so every other wave has passed it too).  Hazard bookkeeping (q = global phase index, both wave groups):
//   write-after-read: a slot is re-issued >= 2 phases after its last ds_read (Blo 4, Bhi 4, Ahi 4, Alo 3 phases);
//   read-after-write: Blo / Bhi / Ahi are read 4 phases after their issue, Alo 5 -- always after the retiring wait.
// The two wave groups (wr = 0 / 1; the two waves of every SIMD) run ONE BARRIER apart: wr = 1 executes an extra
// s_barrier before its first phase, wr = 0 one after its last.  A phase is  [load segment] barrier [16 MFMA] barrier,
// so while one group computes the other reads LDS, issues its DMA and waits -- the matrix pipe of a SIMD always has one
// wave in its MFMA segment.  The persistent tile walk, the XCD mapping, the split tail (ConvP::rem_*), the row-wise
// bf16 epilogue with BatchNorm statistics / fused BN-backward reduce are those of conv_igemm_persistent.
// The prefetch stream runs ACROSS output tiles: while a tile's epilogue stores its rows, the first k-tile of the
// next tile is already in the ring.  Epilogue staging (64 rows x 512 B per pass) lives in the three ring slots of the
// just-finished buffer that the stream does not touch before the next tile's first phase.
#pragma once

// FLAT: 1x1 stride-1 layers (and dense layers) -- output row m reads input pixel m, no tap decode, no border test.
template <int MODE, bool STATS, bool BNEPI, bool FLAT>
__global__ __launch_bounds__(512, 1) void conv_igemm_wide(const ConvP p) {
  typedef uint16_t T;
  constexpr int BM = 256, BN = 256, BK = 64, NTH = 512, NW = 8;
  constexpr int HT = 16384;                     // bytes of one half-tile (128 rows x 128 B)
  constexpr int SLOT_ALO = 0, SLOT_AHI = HT, SLOT_BLO = 2 * HT, SLOT_BHI = 3 * HT, BUFB = 4 * HT;
  constexpr int RING = 2 * BUFB;                // 131072
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* bnp = (float*)(smem + RING);                   // BNEPI: [4][256] scale, shift, mean, rstd of this N-tile
  long long* rowoff = (long long*)(bnp + 4 * BN);       // [256] output offsets of the tile rows
  float2* wred = (float2*)(rowoff + BM);                // [8][256] per-wave (sum, sum of squares / raw moment)
  typedef __attribute__((address_space(3))) void* lptr_t;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, fl = lane & 15;
  const int wr = wave >> 2, wc = wave & 3;
  const T* __restrict__ X = (const T*)p.x;
  const T* __restrict__ Wt = (const T*)p.w;
  T* __restrict__ Y = (T*)p.y;
  const int cls_hw = p.cls_h * p.cls_w;

  const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
  const int nt = l % p.n_tiles;
  const int mslots = gridDim.x / p.n_tiles;
  const int mslot = (l / p.n_tiles) * 8 + xcd;
  const int n0 = nt * BN;
  const int count = p.rem_parts >= 2 ? p.rem_full : ((mslot < p.m_tiles) ? (p.m_tiles - mslot + mslots - 1) / mslots : 0);
  const bool has_part = p.rem_parts >= 2 && mslot < p.rem_tiles * p.rem_parts;
  auto part_tile = [&]() __attribute__((always_inline)) { return p.rem_full * mslots + mslot / p.rem_parts; };
  auto part_index = [&]() __attribute__((always_inline)) { return mslot % p.rem_parts; };
  const int kpt = p.IC / BK;
  const int KT = p.ntaps * kpt;
  auto part_begin = [&]() __attribute__((always_inline)) { return (part_index() * KT) / p.rem_parts; };
  auto part_end = [&]() __attribute__((always_inline)) { return ((part_index() + 1) * KT) / p.rem_parts; };
  const int NT = count * KT + (has_part ? part_end() - part_begin() : 0);     // k-tiles this workgroup consumes

  const int kc = (lane & 7) ^ (lane >> 3);
  constexpr bool flat = FLAT;      // launcher: KH == KW == 1, stride 1, no padding, one class, same geometry in and out

  // ---- gather state.  Operands are fetched with buffer_load ... lds through two buffer descriptors (activation tensor,
  // weight matrix): a 32-bit byte offset per row instead of a 64-bit pointer, and a row that must read zeros (image
  // border, row >= M) simply carries an offset beyond the descriptor's size -- the hardware range check returns zeros,
  // no zero page, no select.  Row a = half * 2 + j is tile row  half * 128 + wave * 16 + j * 8 + (lane >> 3).
  constexpr unsigned OOB = 0xF0000000u;           // >= num_records of either descriptor (launcher checks), + k offsets: no wrap
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)X, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, p.N * p.K * 2, 0x00020000);
  const int pitchb = p.pixpitch * 2;
  // Per row (registers are the scarce resource of this kernel: 128 accumulators + 64 fragment registers):
  //   a_boff: byte offset of chunk kc of the row's BASE pixel (flat: of its only pixel, or OOB for rows >= M); arithmetic
  //           modulo 2^32 -- base pixels left of / above the image give wrapped values that are never used un-shifted
  //   a_yx  : (ry << 16) | (rx & 0xffff), base coordinates for the border test (non-flat only); ry = -20000: row >= M
  // The offset of the current tap is recomputed at every issue (one add + the border test) instead of being stored.
  unsigned a_boff[4];
  // a_yx lives in LDS ([4][512] ints after the statistics slots): with it in registers the non-flat instantiations exceed
  // the 256-register budget by a handful and hipcc spills loop-carried values -- every scratch reload in the k-loop nest
  // makes its conservative vmcnt scoreboard drain the DMA queue
  int* a_yx_l = (int*)(wred + NW * BN) + tid;
  int tap_dy = 0, tap_dx = 0;                     // current tap (uniform)
  unsigned tap_doff = 0;                          // its byte offset relative to the base pixel (uniform)
  auto setup_rows = [&](int mt) __attribute__((always_inline)) {
    int v0 = 0, ca0 = 0, cb0 = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int m = mt * BM + (a >> 1) * 128 + wave * 16 + (a & 1) * 8 + (lane >> 3);
      const bool ok = m < p.M;
      if (FLAT) {
        a_boff[a] = ok ? (unsigned)m * (unsigned)pitchb + kc * 16 : OOB;
      } else {
        // (image, class row, class column) of the row: ONE pair of divisions for row 0, the other three rows by the
        // launcher's decomposition of their distance (8, 128, 136 rows: ConvP::rs_*) and a single carry each -- four
        // unrolled division pairs beside 192 live accumulator / fragment registers spill
        if (a == 0) {
          v0 = m / cls_hw;
          const int rem = m - v0 * cls_hw;
          ca0 = rem / p.cls_w; cb0 = rem - ca0 * p.cls_w;
        }
        int v = v0, ca = ca0, cb = cb0;
        if (a > 0) {
          v += p.rs_dq[a - 1]; ca += p.rs_drow[a - 1]; cb += p.rs_dcol[a - 1];
          if (cb >= p.cls_w) { cb -= p.cls_w; ++ca; }
          if (ca >= p.cls_h) { ca -= p.cls_h; ++v; }
        }
        int ry, rx_;
        if (MODE == MODE_FWD) { ry = ca * p.stride - p.pad; rx_ = cb * p.stride - p.pad; }
        else { ry = ca; rx_ = cb; }
        a_boff[a] = (unsigned)((v * p.IH + ry) * p.IW + rx_) * (unsigned)pitchb + kc * 16;
        a_yx_l[a * NTH] = ((ok ? ry : -20000) << 16) | (rx_ & 0xffff);
      }
    }
  };
  auto set_tap = [&](int ti) __attribute__((always_inline)) {
    if (FLAT) return;
    tap_dy = (int)((p.dy_w >> (4 * ti)) & 15) - 8;
    tap_dx = (int)((p.dx_w >> (4 * ti)) & 15) - 8;
    tap_doff = (unsigned)((tap_dy * p.IW + tap_dx) * pitchb);
  };
  auto row_offset = [&](int a) __attribute__((always_inline)) -> unsigned {
    if (FLAT) return a_boff[a];
    const int yx = a_yx_l[a * NTH];
    const int ry = yx >> 16, rx_ = (int)(short)(yx & 0xffff);
    // branch-free on purpose: an exec-masked select here makes hipcc guard the offset register with s_waitcnt vmcnt(N)
    // against the LDS-DMA still in flight, which stalls the issue phases
    const unsigned inb = ((unsigned)(ry + tap_dy) < (unsigned)p.IH ? 1u : 0u) & ((unsigned)(rx_ + tap_dx) < (unsigned)p.IW ? 1u : 0u);
    const unsigned msk = 0u - inb;
    return ((a_boff[a] + tap_doff) & msk) | (OOB & ~msk);
  };
  // weight rows of this workgroup (N % 256 == 0: always valid): row a = b_off0 + a uniform number of rows
  const unsigned b_off0 = (unsigned)(n0 + wave * 16 + (lane >> 3)) * (unsigned)(p.K * 2) + kc * 16;

  // ---- the issue cursor: k-tile `ikt` of the workgroup's stream = (job it, tap iti, 64-channel chunk ici)
  int it = 0, iti = 0, ici = 0, ikt = 0;
  // o0 / o1: row_offset of the half's two rows, taken BEFORE the phase's fragment reads (the border coordinates come from
  // LDS; read after the fragments they would wait for all of them, LDS returns in order)
  auto issue_a = [&](int half, unsigned o0, unsigned o1) __attribute__((always_inline)) {
    unsigned char* dst = smem + (ikt & 1) * BUFB + (half ? SLOT_AHI : SLOT_ALO) + wave * 2048;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)dst, 16, (int)(o0 + (unsigned)(ici * 128)), 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lptr_t)(dst + 1024), 16, (int)(o1 + (unsigned)(ici * 128)), 0, 0, 0);
  };
  auto issue_b = [&](int half) __attribute__((always_inline)) {
    const int k0 = (int)((p.tap_w >> (4 * iti)) & 15) * p.IC + ici * BK;
    unsigned char* dst = smem + (ikt & 1) * BUFB + (half ? SLOT_BHI : SLOT_BLO) + wave * 2048;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(dst + j * 1024), 16,
                                               (int)(b_off0 + (unsigned)((half * 128 + j * 8) * p.K * 2 + k0 * 2)), 0, 0, 0);
  };
  auto advance = [&]() __attribute__((always_inline)) {       // after the four half-tiles of k-tile ikt have been issued
    ++ikt;
    if (++ici == kpt) {
      ici = 0;
      if (++iti == p.ntaps) {
        iti = 0;
        ++it;
        if (it < count) setup_rows(mslot + it * mslots);
        else if (it == count && has_part) {
          setup_rows(part_tile());
          const int pka = part_begin();
          iti = pka / kpt; ici = pka - iti * kpt;
        }
      }
      if (it < count + (has_part ? 1 : 0)) set_tap(iti);
    }
  };

  if (BNEPI) {
    for (int i = tid; i < BN; i += NTH) {
      const int n = n0 + i;
      bnp[i] = p.bn_mode == 2 ? p.bn_scale[n] : 0.f;
      bnp[BN + i] = p.bn_mode == 2 ? p.bn_shift[n] : 0.f;
      bnp[2 * BN + i] = p.bn_mean ? p.bn_mean[n] : 0.f;
      bnp[3 * BN + i] = p.bn_rstd ? p.bn_rstd[n] : 0.f;
    }
  }
  if (STATS) {
    for (int i = lane; i < BN; i += 64) wred[wave * BN + i] = make_float2(0.f, 0.f);
  }

  // ---- fragment addresses (bytes inside a half-tile): row r has r & 7 == fl & 7 for every fragment of this lane
  const int a_lane = (wr * 64 + fl) * 128, b_lane = (wc * 32 + fl) * 128;
  const int ch0 = (g ^ (fl & 7)) * 16, ch1 = ((4 + g) ^ (fl & 7)) * 16;

  // ---- prologue: half-tiles 0 .. 4 of the stream (k-tile 0 and Alo of k-tile 1); Alo(0), Blo(0) must have landed
  if (NT > 0) {
    if (count > 0) {
      setup_rows(mslot);
      set_tap(0);
    } else {
      setup_rows(part_tile());
      const int pka = part_begin();
      iti = pka / kpt; ici = pka - iti * kpt;
      set_tap(iti);
    }
    issue_a(0, row_offset(0), row_offset(1)); issue_b(0); issue_b(1); issue_a(1, row_offset(2), row_offset(3));
    advance();
    if (ikt < NT) {
      issue_a(0, row_offset(0), row_offset(1));
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

  int ckt = 0;                                  // compute cursor: k-tile index in the stream
  int skipw = 0;                                // phases left (after an epilogue) that need no DMA wait: the stream was drained

  for (int ct = 0; ct < count + (has_part ? 1 : 0); ++ct) {
    const bool part = ct == count;
    const int m0 = (part ? part_tile() : mslot + ct * mslots) * BM;
    const int nkt = part ? part_end() - part_begin() : KT;
    f32x4 acc[4][4][2];                         // [quadrant qm * 2 + qn][mi][ni]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[q][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    u32x4 af[4][2], bl[2][2], bh[2][2];

    if (wr == 1) asm volatile("s_barrier" ::: "memory");       // the second wave group runs one barrier behind
    for (int kt = 0; kt < nkt; ++kt, ++ckt) {
      const unsigned char* buf = smem + (ckt & 1) * BUFB;
      // DMA wait of a phase: every half-tile issued up to three phases ago has landed.  Near the end of the stream fewer
      // than three newer half-tiles exist; q_left = half-tiles of the stream issued after the one that must have landed.
#define WIDE_DMA_WAIT(PH)                                                                       \
      do {                                                                                       \
        if (skipw > 0) { --skipw; break; }                                                       \
        const int q_left = 4 * NT - 1 - (4 * ckt + (PH) + 2);                                    \
        if (q_left >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                        \
        else if (q_left == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                   \
        else if (q_left == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                   \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                    \
      } while (0)
#define WIDE_MMA(Q, BF)                                                                         \
      do {                                                                                       \
        asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __builtin_amdgcn_s_setprio(1);                                                           \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                         \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                         \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                         \
          acc[Q][mi][ni] = mma_bf16(BF[ni][ks], af[mi][ks], acc[Q][mi][ni]);                     \
        __builtin_amdgcn_s_setprio(0);                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        asm volatile("s_barrier" ::: "memory");                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                       \
      } while (0)
      // ---- phase 0: Alo x Blo
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        bl[ni][0] = *(const u32x4*)(buf + SLOT_BLO + b_lane + ni * 2048 + ch0);
        bl[ni][1] = *(const u32x4*)(buf + SLOT_BLO + b_lane + ni * 2048 + ch1);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        af[mi][0] = *(const u32x4*)(buf + SLOT_ALO + a_lane + mi * 2048 + ch0);
        af[mi][1] = *(const u32x4*)(buf + SLOT_ALO + a_lane + mi * 2048 + ch1);
      }
      if (ikt < NT) issue_b(0);
      WIDE_DMA_WAIT(0);
      WIDE_MMA(0, bl);
      // ---- phase 1: Alo x Bhi
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        bh[ni][0] = *(const u32x4*)(buf + SLOT_BHI + b_lane + ni * 2048 + ch0);
        bh[ni][1] = *(const u32x4*)(buf + SLOT_BHI + b_lane + ni * 2048 + ch1);
      }
      if (ikt < NT) issue_b(1);
      WIDE_DMA_WAIT(1);
      WIDE_MMA(1, bh);
      // ---- phase 2: Ahi x Bhi
      const unsigned o2 = row_offset(2), o3 = row_offset(3);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        af[mi][0] = *(const u32x4*)(buf + SLOT_AHI + a_lane + mi * 2048 + ch0);
        af[mi][1] = *(const u32x4*)(buf + SLOT_AHI + a_lane + mi * 2048 + ch1);
      }
      if (ikt < NT) { issue_a(1, o2, o3); advance(); }
      WIDE_DMA_WAIT(2);
      WIDE_MMA(3, bh);
      // ---- phase 3: Ahi x Blo (Blo fragments still in registers)
      if (ikt < NT) issue_a(0, row_offset(0), row_offset(1));
      WIDE_DMA_WAIT(3);
      WIDE_MMA(2, bl);
#undef WIDE_MMA
#undef WIDE_DMA_WAIT
    }
    if (wr == 0) asm volatile("s_barrier" ::: "memory");       // re-align the two wave groups
    // drain the prefetch stream (at most three half-tiles, issued 1-3 phases ago): the epilogue's own stores and loads
    // must not share the vmcnt queue with half-tiles somebody still has to wait for
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    skipw = 3;

    // ---- split tail (see conv_igemm_persistent): parts j > 0 publish their accumulators, part 0 adds them
    if (part) {
      const long long slot0 = ((long long)(mslot / p.rem_parts) * p.n_tiles + nt) * (p.rem_parts - 1);
      const int pj = part_index();
      if (pj != 0) {
        // write-through (sc1, aux = 16) buffer stores through a descriptor of this part's slot: one 32-bit lane offset + a
        // scalar fragment offset (32 per-fragment 64-bit lane addresses would not fit beside the accumulators)
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((f32x4*)p.part_ws + (slot0 + pj - 1) * (long long)(32 * NTH)), 0, 32 * NTH * 16, 0x00020000);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[q][mi][ni]), rp, tid * 16,
                                                     ((q * 4 + mi) * 2 + ni) * NTH * 16, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
          __hip_atomic_store(p.part_flags + slot0 + pj - 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      if (tid == 0) {
        for (int j = 1; j < p.rem_parts; ++j) {
          int spins = 0;
          while (__hip_atomic_load(p.part_flags + slot0 + j - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u &&
                 ++spins < (1 << 22))
            __builtin_amdgcn_s_sleep(8);
          if (spins >= (1 << 22)) atomicAdd(p.part_err, 1u);          // reported, never silent (see ConvP)
          else __hip_atomic_store(p.part_flags + slot0 + j - 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed: replay-safe
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      for (int j = 1; j < p.rem_parts; ++j) {
        const f32x4* src = (const f32x4*)p.part_ws + (slot0 + j - 1) * (long long)(32 * NTH);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) acc[q][mi][ni] += src[((q * 4 + mi) * 2 + ni) * NTH + tid];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    // ---- row-wise bf16 epilogue in four passes of 64 rows: pass ps = rows [64 ps, 64 ps + 64) of the tile, staged by the
    // wave group wr == (ps & 1) from its quadrants qm == ps >> 1 (both column halves), read back as whole 512-byte rows
    // Every lane-dependent address of the epilogue derives from an OPAQUE copy of the thread index: otherwise hipcc hoists
    // a dozen loop-invariant per-lane offsets / pointers into the kernel prologue, keeps them live across the k-loop and
    // spills them -- and a scratch reload anywhere in the loop nest makes its vmcnt scoreboard drain the DMA queue.
    int te = threadIdx.x;
    asm volatile("" : "+v"(te));
    const int fl_e = te & 15, g_e = (te >> 4) & 3, e_cc = te & 31, e_row = te >> 5, lane_e = te & 63;
    auto row_off = [&](int m) __attribute__((always_inline)) -> long long {
      if (m >= p.M) return -1;
      if (flat) return (long long)m * p.N;
      const int v = m / cls_hw;
      const int rem = m - v * cls_hw;
      const int ca = rem / p.cls_w, cb = rem - ca * p.cls_w;
      return (((long long)v * p.OH + ca * p.cs + p.py) * p.OW + cb * p.cs + p.px) * p.N;
    };
    // the just-consumed buffer is (ckt - 1) & 1; its Ahi / Blo slots (32 KB, contiguous) are not re-issued before the next
    // tile's first phase (Alo of that buffer may already be receiving the k-tile after next)
    unsigned char* Cs = smem + ((ckt - 1) & 1) * BUFB + SLOT_AHI;
    // Park the gather state of the issue cursor in the (equally idle) Bhi slot of that buffer for the duration of the
    // epilogue: live through it, these eight registers are what the allocator spills -- and reloads inside the k-loop,
    // where every scratch access drains the DMA queue.
    unsigned* park = (unsigned*)(smem + ((ckt - 1) & 1) * BUFB + SLOT_BHI);
#pragma unroll
    for (int a = 0; a < 4; ++a) park[a * NTH + te] = a_boff[a];
    if (te < BM) rowoff[te] = row_off(m0 + te);
    float e_s[8], e_q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { e_s[e] = 0.f; e_q[e] = 0.f; }
    if (STATS && !BNEPI) {
      // Forward statistics from the fp32 ACCUMULATORS (the source the 128-wide tiles use: the BatchNorm moments do not
      // depend on which tile ran): per (column half, 16-channel fragment) sum the eight row fragments in registers, the 16
      // row lanes by xor-shuffles, and lane fl == 0 of every 4-channel group adds into this wave's LDS slot.
#pragma unroll
      for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          float ss[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int qm = 0; qm < 2; ++qm)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float v = acc[qm * 2 + qn][mi][ni][r];
                ss[r] += v;
                sq[r] = fmaf(v, v, sq[r]);
              }
#pragma unroll
          for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) { ss[r] += __shfl_xor(ss[r], o, 64); sq[r] += __shfl_xor(sq[r], o, 64); }
          if (fl_e == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float2* slot = wred + wave * BN + qn * 128 + wc * 32 + ni * 16 + g_e * 4 + r;
              float2 o2 = *slot;
              o2.x += ss[r]; o2.y += sq[r];
              *slot = o2;
            }
          }
        }
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      __syncthreads();                          // ring reads / the previous pass's staging reads are done (rowoff visible)
      if (wr == (ps & 1)) {
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
              const int ml = mi * 16 + fl_e;                                  // row in the 64-row staging block
              const int q = qn * 32 + wc * 8 + ni * 4 + g_e;                   // 8-byte granule (4 channels) in the row
              const f32x4 v = acc[(ps >> 1) * 2 + qn][mi][ni];
              u32x2 pk;
              pk[0] = pack_bf16x2(v[0], v[1]);
              pk[1] = pack_bf16x2(v[2], v[3]);
              *(u32x2*)(Cs + ml * 512 + ((q ^ ((ml & 7) << 1)) << 3)) = pk;
            }
      }
      __syncthreads();
      const int ncol = n0 + e_cc * 8;
      // ER rows per batch: the operands of a batch are requested back to back (one memory round trip per batch).  Kept
      // small on purpose -- the accumulators of the passes still to come are live, and every register the row pass takes
      // beyond the budget is spilled INSIDE the k-loop (the spilled values are the gather offsets the DMA issue needs).
      constexpr int ER = BNEPI ? 2 : 1;
#pragma unroll 1
      for (int ib = 0; ib < 4; ib += ER) {
        long long eoff[ER];
        bool erok[ER];
#pragma unroll
        for (int i = 0; i < ER; ++i) {
          const long long off = rowoff[ps * 64 + e_row + (ib + i) * 16];
          erok[i] = off >= 0 && Y != nullptr;
          eoff[i] = erok[i] ? off + ncol : 0;
        }
        u32x4 e_ov[ER], e_xv[ER], e_mv[ER];
        if (p.accumulate) {
#pragma unroll
          for (int i = 0; i < ER; ++i) e_ov[i] = *(const u32x4*)((const uint16_t*)Y + eoff[i]);
        }
        if (BNEPI) {
          if (p.bn_mode != 4) {
#pragma unroll
            for (int i = 0; i < ER; ++i) e_xv[i] = *(const u32x4*)((const uint16_t*)p.bn_x + eoff[i]);
          } else {
#pragma unroll
            for (int i = 0; i < ER; ++i) e_xv[i] = zero16();
          }
          if (p.bn_mode == 1) {
#pragma unroll
            for (int i = 0; i < ER; ++i) e_mv[i] = *(const u32x4*)((const uint16_t*)p.bn_mask + eoff[i]);
          } else if (p.bn_mode >= 3) {
#pragma unroll
            for (int i = 0; i < ER; ++i) e_mv[i][0] = ((const unsigned char*)p.bn_mask)[eoff[i] >> 3];
          }
        }
#pragma unroll
        for (int i = 0; i < ER; ++i) {
          const int r = e_row + (ib + i) * 16;
          const u32x4 cv = *(const u32x4*)(Cs + r * 512 + (((e_cc * 2) ^ ((r & 7) << 1)) << 3));
          float v[8];
          chunk_to_f32<uint16_t>(cv, v);
          if (!erok[i]) continue;
          uint16_t* dst = (uint16_t*)Y + eoff[i];
          if (p.accumulate) {
            float o[8];
            chunk_to_f32<uint16_t>(e_ov[i], o);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += o[e];
          }
          if (BNEPI) {
            if (p.bn_mode == 4) {
              const unsigned mb = e_mv[i][0];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                v[e] = ((mb >> e) & 1u) ? v[e] : 0.f;
                e_s[e] += v[e];
              }
            } else {
              float xf[8];
              chunk_to_f32<uint16_t>(e_xv[i], xf);
              if (p.bn_mode == 1) {
                float mk[8];
                chunk_to_f32<uint16_t>(e_mv[i], mk);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = mk[e] > 0.f ? v[e] : 0.f;
              } else if (p.bn_mode == 3) {
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
          if (p.accumulate || BNEPI) *(u32x4*)dst = f32_to_chunk<uint16_t>(v);
          else *(u32x4*)dst = cv;
        }
      }
    }
    if (STATS && BNEPI) {
      // lanes l and l + 32 of a wave own the same channel chunk: fold into this wave's LDS slot (plain read-modify-write,
      // the slot belongs to one wave: deterministic)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float s1 = e_s[e] + __shfl_xor(e_s[e], 32, 64), s2 = e_q[e] + __shfl_xor(e_q[e], 32, 64);
        if (lane_e < 32) {
          float2 o = wred[wave * BN + e_cc * 8 + e];
          o.x += s1; o.y += s2;
          wred[wave * BN + e_cc * 8 + e] = o;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) a_boff[a] = park[a * NTH + te];
    // the staging block is read by every wave until here; the next tile's first DMA into it (phase 0: Blo of this buffer)
    // is issued after that phase's ds_reads, i.e. after at least the barrier below
    __syncthreads();
  }
  if (STATS) {
    __syncthreads();
    const bool own_slot = p.nslot >= mslots;
    if (tid < BN && (NT > 0 || own_slot)) {
      float s1 = 0.f, s2 = 0.f;
      for (int w = 0; w < NW; ++w) { s1 += wred[w * BN + tid].x; s2 += wred[w * BN + tid].y; }
      if (BNEPI) s2 = (s2 - bnp[2 * BN + tid] * s1) * bnp[3 * BN + tid];
      float* st = p.stats + (long long)(own_slot ? mslot : mslot % p.nslot) * 2 * p.N;
      if (own_slot) { st[n0 + tid] = s1; st[p.N + n0 + tid] = s2; }
      else { atomicAdd(st + n0 + tid, s1); atomicAdd(st + p.N + n0 + tid, s2); }
    }
  }
}
