// Fused multi-tensor LARS update for gfx950.
//
// Replaces the per-variable op groups of
// /root/reference/tf2/lars_optimizer.py:83-137 (`_resource_apply_dense`, classic
// momentum branch :99-115 and the popular-momentum branch :116-132) with two
// launches over ALL trainable tensors: (1) per-tensor squared norms, (2) trust
// ratio + momentum + weight write.  Pure HBM streaming: 16-byte loads, one
// workgroup per 8192-element chunk, chunks of every tensor in one grid.
//
// Device-side descriptor table (int64 words, built once by the host):
//   table[0*T + t] = w pointer   table[1*T + t] = g pointer   table[2*T + t] = v pointer
//   table[3*T + t] = numel       table[4*T + t] = flags (bit0 weight decay, bit1 adapt)
// chunks[2*c] = tensor id, chunks[2*c+1] = element offset of the chunk.
#include "common.h"

namespace {

constexpr int kChunk = 8192;

__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh) {
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[2 * wave] = a; sh[2 * wave + 1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double x = 0, y = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { x += sh[2 * i]; y += sh[2 * i + 1]; }
    a = x; b = y;
  }
}

// Per-chunk partial squared norms: part[2c] = sum w^2, part[2c+1] = sum u^2 over chunk c, with u = g + wd*w (classic)
// or u = momentum*v + g (+ nesterov) (popular momentum, lars_optimizer.py:117-126).  Plain stores, no atomics: the
// update kernel adds a tensor's partials in chunk order, so the trust ratio is bit-identical from run to run.
__global__ __launch_bounds__(256) void lars_norms(const long long* __restrict__ table, int T,
                                                  const long long* __restrict__ chunks,
                                                  float weight_decay, float momentum,
                                                  int classic, int nesterov,
                                                  double* __restrict__ part) {
  __shared__ double sh[8];
  const int t = (int)chunks[2 * blockIdx.x];
  const long long off = chunks[2 * blockIdx.x + 1];
  const int flags = (int)table[4 * T + t];
  if (!(flags & 2)) return;  // no layer adaptation -> norms unused
  const float* w = (const float*)table[0 * T + t];
  const float* g = (const float*)table[1 * T + t];
  const float* v = (const float*)table[2 * T + t];
  const long long numel = table[3 * T + t];
  const float wd = (flags & 1) ? weight_decay : 0.f;
  const long long end = min(numel, off + (long long)kChunk);
  double sw = 0.0, su = 0.0;
  for (long long i = off + threadIdx.x; i < end; i += 256) {
    float wi = w[i], gi = g[i] + wd * wi, ui;
    if (classic) ui = gi;
    else { float nv = momentum * v[i] + gi; ui = nesterov ? momentum * nv + gi : nv; }
    sw += (double)wi * wi;
    su += (double)ui * ui;
  }
  block_sum2(sw, su, sh);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = sw;
    part[2 * blockIdx.x + 1] = su;
  }
}

__global__ __launch_bounds__(256) void lars_update(const long long* __restrict__ table, int T,
                                                   const long long* __restrict__ chunks,
                                                   const float* __restrict__ lr_ptr, float lr_val,
                                                   float weight_decay, float momentum, float eeta,
                                                   int classic, int nesterov,
                                                   const double* __restrict__ part) {
  __shared__ double red[2 * 256];
  const int t = (int)chunks[2 * blockIdx.x];
  const long long off = chunks[2 * blockIdx.x + 1];
  const int flags = (int)table[4 * T + t];
  float* w = (float*)table[0 * T + t];
  const float* g = (const float*)table[1 * T + t];
  float* v = (float*)table[2 * T + t];
  const long long numel = table[3 * T + t];
  const float wd = (flags & 1) ? weight_decay : 0.f;
  const float lr = lr_ptr ? *lr_ptr : lr_val;
  float trust = 1.0f;
  if (flags & 2) {  // lars_optimizer.py:101-107 / :124-130
    // this tensor's chunks are consecutive in the chunk list: the first one is (off / kChunk) entries back.
    // Fixed summation order: thread i adds partials i, i+256, ... ; then a fixed binary tree over the 256 threads.
    const long long first = (long long)blockIdx.x - off / kChunk;
    const long long nch = (numel + kChunk - 1) / kChunk;
    double a = 0.0, b = 0.0;
    for (long long c = threadIdx.x; c < nch; c += 256) { a += part[2 * (first + c)]; b += part[2 * (first + c) + 1]; }
    red[threadIdx.x] = a;
    red[256 + threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) { red[threadIdx.x] += red[threadIdx.x + s]; red[256 + threadIdx.x] += red[256 + threadIdx.x + s]; }
      __syncthreads();
    }
    const float wn = (float)sqrt(red[0]);
    const float un = (float)sqrt(red[256]);
    if (wn > 0.f && un > 0.f) trust = eeta * wn / un;
  }
  const float slr = lr * trust;  // :108 / :131
  const long long end = min(numel, off + (long long)kChunk);
  for (long long i = off + threadIdx.x; i < end; i += 256) {
    const float wi = w[i];
    const float gi = g[i] + wd * wi;  // :96-97
    float nv, upd;
    if (classic) {
      nv = momentum * v[i] + slr * gi;                    // :110
      upd = nesterov ? momentum * nv + slr * gi : nv;     // :111-114
      w[i] = wi - upd;                                    // :115
    } else {
      nv = momentum * v[i] + gi;                          // :117
      upd = nesterov ? momentum * nv + gi : nv;           // :118-121
      w[i] = wi - slr * upd;                              // :132
    }
    v[i] = nv;
  }
}

}  // namespace

extern "C" {

int simclr_lars_chunk_elems(void) { return kChunk; }

// norms: device double[2*num_chunks] scratch (per-chunk partial squared norms; every used entry is rewritten by the
// call).  lr_dev may be NULL (then lr is used); a device-resident lr lets a captured hipGraph replay with a new
// learning rate.
int simclr_lars_multi_tensor(const long long* table, int num_tensors, const long long* chunks,
                             int num_chunks, const float* lr_dev, float lr, float momentum,
                             float weight_decay, float eeta, int classic_momentum, int use_nesterov,
                             double* norms, hipStream_t stream) {
  SIMCLR_CHECK_ARG(num_tensors > 0 && num_chunks > 0, "lars: empty tensor list");
  SIMCLR_CHECK_ARG(table && chunks && norms, "lars: null table/chunks/norms");
  hipLaunchKernelGGL(lars_norms, dim3(num_chunks), dim3(256), 0, stream, table, num_tensors, chunks,
                     weight_decay, momentum, classic_momentum, use_nesterov, norms);
  SIMCLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(lars_update, dim3(num_chunks), dim3(256), 0, stream, table, num_tensors, chunks,
                     lr_dev, lr, weight_decay, momentum, eeta, classic_momentum, use_nesterov, norms);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
