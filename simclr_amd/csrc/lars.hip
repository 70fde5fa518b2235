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
#include <math.h>

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

// ---- the other two branches of build_optimizer (tf2/model.py:31-34) on the same descriptor / chunk tables ----------------
// tf.keras.optimizers.SGD(lr, momentum, nesterov=True): accum = momentum * accum - lr * g;  w += momentum * accum - lr * g
// (nesterov) or w += accum.  `l2` (flags bit 0): the gradient of add_weight_decay's loss term (tf2/model.py:62-69:
// weight_decay * sum l2_loss(v) over the non-BatchNorm variables), added here as g + l2 * w instead of by a separate pass.
__global__ __launch_bounds__(256) void sgd_update(const long long* __restrict__ table, int T, const long long* __restrict__ chunks,
                                                  const float* __restrict__ lr_ptr, float lr_val, float momentum, int nesterov, float l2) {
  const int t = (int)chunks[2 * blockIdx.x];
  const long long off = chunks[2 * blockIdx.x + 1];
  const int flags = (int)table[4 * T + t];
  float* w = (float*)table[0 * T + t];
  const float* g = (const float*)table[1 * T + t];
  float* a = (float*)table[2 * T + t];
  const long long numel = table[3 * T + t];
  const float c = (flags & 1) ? l2 : 0.f;
  const float lr = lr_ptr ? *lr_ptr : lr_val;
  const long long end = min(numel, off + (long long)kChunk);
  for (long long i = off + threadIdx.x; i < end; i += 256) {
    const float wi = w[i];
    const float gi = g[i] + c * wi;
    const float na = momentum * a[i] - lr * gi;
    w[i] = wi + (nesterov ? momentum * na - lr * gi : na);
    a[i] = na;
  }
}

// tf.keras.optimizers.Adam(lr) (beta_1 0.9, beta_2 0.999, epsilon 1e-7): m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2;
// w -= lr_t * m / (sqrt(v) + eps), lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) -- `corr` is that factor for this step (host, double).
// The slot tensor of a variable is [2][numel]: m then v.
__global__ __launch_bounds__(256) void adam_update(const long long* __restrict__ table, int T, const long long* __restrict__ chunks,
                                                   const float* __restrict__ lr_ptr, float lr_val, float b1, float b2, float eps,
                                                   float corr, float l2) {
  const int t = (int)chunks[2 * blockIdx.x];
  const long long off = chunks[2 * blockIdx.x + 1];
  const int flags = (int)table[4 * T + t];
  float* w = (float*)table[0 * T + t];
  const float* g = (const float*)table[1 * T + t];
  const long long numel = table[3 * T + t];
  float* m = (float*)table[2 * T + t];
  float* v = m + numel;
  const float c = (flags & 1) ? l2 : 0.f;
  const float lr_t = (lr_ptr ? *lr_ptr : lr_val) * corr;
  const long long end = min(numel, off + (long long)kChunk);
  for (long long i = off + threadIdx.x; i < end; i += 256) {
    const float wi = w[i];
    const float gi = g[i] + c * wi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    w[i] = wi - lr_t * mi / (sqrtf(vi) + eps);
    m[i] = mi;
    v[i] = vi;
  }
}

}  // namespace

extern "C" {

int simclr_lars_chunk_elems(void) { return kChunk; }

// build_optimizer's 'momentum' branch (tf2/model.py:31-32): one launch over every trainable tensor, tables as for
// simclr_lars_multi_tensor (row 2 = the accumulator slot, flags bit 0 = add l2 * w to the gradient).
int simclr_sgd_multi_tensor(const long long* table, int num_tensors, const long long* chunks, int num_chunks,
                            const float* lr_dev, float lr, float momentum, int use_nesterov, float l2, hipStream_t stream) {
  SIMCLR_CHECK_ARG(num_tensors > 0 && num_chunks > 0 && table && chunks, "sgd: empty / null tables");
  hipLaunchKernelGGL(sgd_update, dim3(num_chunks), dim3(256), 0, stream, table, num_tensors, chunks, lr_dev, lr, momentum,
                     use_nesterov, l2);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

// build_optimizer's 'adam' branch (tf2/model.py:33-34).  step = the 1-based update count t of the bias correction; row 2 of
// the table = a slot of 2 * numel floats (m, then v).
int simclr_adam_multi_tensor(const long long* table, int num_tensors, const long long* chunks, int num_chunks,
                             const float* lr_dev, float lr, float beta1, float beta2, float epsilon, long long step, float l2,
                             hipStream_t stream) {
  SIMCLR_CHECK_ARG(num_tensors > 0 && num_chunks > 0 && table && chunks && step >= 1, "adam: empty / null tables or step < 1");
  const double corr = sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
  hipLaunchKernelGGL(adam_update, dim3(num_chunks), dim3(256), 0, stream, table, num_tensors, chunks, lr_dev, lr, beta1, beta2,
                     epsilon, (float)corr, l2);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

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
