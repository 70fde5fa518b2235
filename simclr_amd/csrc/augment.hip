// Two-view training augmentation on the device: random-resized-crop (bicubic) + flip, colour
// jitter in random order, random grayscale, clip -- the per-image part of
// /root/reference/tf2/data_util.py:443-475 (preprocess_for_train) for a whole batch and both views
// (/root/reference/tf2/data.py:52-62), so the host input pipeline only has to deliver decoded images.
//
// The random DRAWS are made by the host (simclr_amd/data_util.py: crop box via the restated
// tf.image.sample_distorted_bounding_box, flip / jitter / grayscale coins, op order, factors) and arrive as a
// parameter table [b][views][16]; the kernels do the pixel work.  Pure HBM streaming:
//   K1 crop + bicubic resize + flip : 16 taps per output pixel, source image mostly L2 resident      -> tmp [b*views][H][W][3] f32
//   K2 mean of the partially jittered image per (image, view, channel) -- tf.image.adjust_contrast needs the mean of
//      the image AS IT IS when contrast is applied, i.e. after the ops that precede it in the random order
//   K3 the jitter chain (each op followed by clip to [0,1]), grayscale, final clip                     -> out [b][H][W][3*views] f32
// TensorFlow semantics restated (kernels are not under /root/reference; see oracle/augment.py): TF2 bicubic resize =
// half-pixel centres, Keys cubic A = -0.5, weights quantised to a 1024-entry table, out-of-image taps dropped and
// the rest renormalised; adjust_saturation / adjust_hue through HSV; rgb_to_grayscale weights (0.2989, 0.5870, 0.1140).
#include "common.h"

namespace {

constexpr int kP = 16;   // floats per (image, view) parameter record:
// 0 crop_y 1 crop_x 2 crop_h 3 crop_w 4 flip 5 jitter_on 6..9 perm 10 brightness 11 contrast 12 saturation 13 hue 14 gray_on

__device__ __forceinline__ float clip01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

// Keys cubic, A = -0.5, evaluated at the 1/1024 grid point the TF table would supply
__device__ __forceinline__ void cubic_weights(int out_pos, float scale, int in_size, int* idx, float* w) {
  const float A = -0.5f;
  const float in_f = ((float)out_pos + 0.5f) * scale - 0.5f;
  const float fl = floorf(in_f);
  const int in_loc = (int)fl;
  const float delta = in_f - fl;
  const int off = (int)rintf(delta * 1024.f);
  const float x0 = (float)off * (1.0f / 1024.f), x1 = (float)(1024 - off) * (1.0f / 1024.f);
  auto near = [&](float x) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; };
  auto far = [&](float x) { const float y = x + 1.f; return ((A * y - 5.f * A) * y + 8.f * A) * y - 4.f * A; };
  w[0] = far(x0); w[1] = near(x0); w[2] = near(x1); w[3] = far(x1);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int want = in_loc - 1 + j;
    const int got = min(max(want, 0), in_size - 1);
    if (got != want) w[j] = 0.f;
    idx[j] = got;
  }
  const float s = w[0] + w[1] + w[2] + w[3];
  if (fabsf(s) >= 1000.f * 1.17549435e-38f) {
    const float r = 1.f / s;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] *= r;
  }
}

template <typename S> __device__ __forceinline__ float ld_src(const S* p);
template <> __device__ __forceinline__ float ld_src<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_src<unsigned char>(const unsigned char* p) { return (float)(*p) * (1.0f / 255.0f); }

template <typename S>
__global__ __launch_bounds__(256) void aug_crop_resize_flip(const S* __restrict__ src, const float* __restrict__ params,
                                                            float* __restrict__ tmp, int b, int views, int Hs, int Ws,
                                                            int H, int W) {
  const long long total = (long long)b * views * H * W;
  for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
    const int ox = (int)(t % W);
    const int oy = (int)((t / W) % H);
    const int iv = (int)(t / ((long long)W * H));
    const int img = iv / views;
    const float* p = params + (long long)iv * kP;
    const int cy = (int)p[0], cx = (int)p[1], ch = (int)p[2], cw = (int)p[3];
    int iy[4], ix[4];
    float wy[4], wx[4];
    cubic_weights(oy, (float)ch / (float)H, ch, iy, wy);
    cubic_weights(ox, (float)cw / (float)W, cw, ix, wx);
    const S* base = src + (long long)img * Hs * Ws * 3;
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      float row[3] = {0.f, 0.f, 0.f};
      const S* rp = base + (long long)(cy + iy[a]) * Ws * 3;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const S* q = rp + (long long)(cx + ix[c4]) * 3;
        row[0] += wx[c4] * ld_src<S>(q);
        row[1] += wx[c4] * ld_src<S>(q + 1);
        row[2] += wx[c4] * ld_src<S>(q + 2);
      }
      acc[0] += wy[a] * row[0]; acc[1] += wy[a] * row[1]; acc[2] += wy[a] * row[2];
    }
    const int dx = p[4] > 0.f ? W - 1 - ox : ox;                       // tf.image.random_flip_left_right
    float* o = tmp + (((long long)iv * H + oy) * W + dx) * 3;
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
  }
}

__device__ __forceinline__ void rgb_to_hsv(const float* c, float& h, float& s, float& v) {
  const float r = c[0], g = c[1], b = c[2];
  v = fmaxf(r, fmaxf(g, b));
  const float range = v - fminf(r, fminf(g, b));
  s = v > 0.f ? range / v : 0.f;
  const float norm = 1.0f / (6.0f * (range > 0.f ? range : 1.f));
  if (r == v) h = norm * (g - b);
  else if (g == v) h = norm * (b - r) + 2.0f / 6.0f;
  else h = norm * (r - g) + 4.0f / 6.0f;
  if (range <= 0.f) h = 0.f;
  if (h < 0.f) h += 1.f;
}
__device__ __forceinline__ void hsv_to_rgb(float h, float s, float v, float* c) {
  const float cc = s * v, m = v - cc, dh = h * 6.0f;
  const int cat = min(max((int)floorf(dh), 0), 5);
  const float fm = dh - 2.0f * floorf(dh * 0.5f);
  const float x = cc * (1.0f - fabsf(fm - 1.0f));
  float r, g, b;
  switch (cat) {
    case 0: r = cc; g = x; b = 0.f; break;
    case 1: r = x; g = cc; b = 0.f; break;
    case 2: r = 0.f; g = cc; b = x; break;
    case 3: r = 0.f; g = x; b = cc; break;
    case 4: r = x; g = 0.f; b = cc; break;
    default: r = cc; g = 0.f; b = x; break;
  }
  c[0] = r + m; c[1] = g + m; c[2] = b + m;
}

// one jitter op (0 brightness, 1 contrast, 2 saturation, 3 hue) followed by the clip of tf2/data_util.py:170-171
__device__ __forceinline__ void jitter_op(int op, float* c, const float* p, const float* mean) {
  if (op == 0) {
    c[0] *= p[10]; c[1] *= p[10]; c[2] *= p[10];
  } else if (op == 1) {
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = (c[k] - mean[k]) * p[11] + mean[k];
  } else if (op == 2) {
    float h, s, v;
    rgb_to_hsv(c, h, s, v);
    hsv_to_rgb(h, clip01(s * p[12]), v, c);
  } else {
    float h, s, v;
    rgb_to_hsv(c, h, s, v);
    h = h + p[13] + 1.0f;
    h = h - floorf(h);
    hsv_to_rgb(h, s, v, c);
  }
  c[0] = clip01(c[0]); c[1] = clip01(c[1]); c[2] = clip01(c[2]);
}

// mean[iv][3] = per-channel mean of the image after the ops that precede contrast in this view's order
__global__ __launch_bounds__(256) void aug_color_mean(const float* __restrict__ tmp, const float* __restrict__ params,
                                                      float* __restrict__ mean, int HW) {
  __shared__ double sh[3 * 256];
  const int iv = blockIdx.x;
  const float* p = params + (long long)iv * kP;
  if (!(p[5] > 0.f)) {
    if (threadIdx.x < 3) mean[iv * 3 + threadIdx.x] = 0.f;
    return;
  }
  int perm[4] = {(int)p[6], (int)p[7], (int)p[8], (int)p[9]};
  const float zero[3] = {0.f, 0.f, 0.f};
  double s[3] = {0.0, 0.0, 0.0};
  const float* im = tmp + (long long)iv * HW * 3;
  for (int i = threadIdx.x; i < HW; i += 256) {
    float c[3] = {im[i * 3], im[i * 3 + 1], im[i * 3 + 2]};
    for (int k = 0; k < 4 && perm[k] != 1; ++k) jitter_op(perm[k], c, p, zero);
    s[0] += (double)c[0]; s[1] += (double)c[1]; s[2] += (double)c[2];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) sh[k * 256 + threadIdx.x] = s[k];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
#pragma unroll
      for (int k = 0; k < 3; ++k) sh[k * 256 + threadIdx.x] += sh[k * 256 + threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) mean[iv * 3 + threadIdx.x] = (float)(sh[threadIdx.x * 256] / (double)HW);
}

__global__ __launch_bounds__(256) void aug_color_apply(const float* __restrict__ tmp, const float* __restrict__ params,
                                                       const float* __restrict__ mean, float* __restrict__ out, int b,
                                                       int views, int HW) {
  const long long total = (long long)b * views * HW;
  for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += gridDim.x * 256ll) {
    const int px = (int)(t % HW);
    const int iv = (int)(t / HW);
    const int img = iv / views, v = iv % views;
    const float* p = params + (long long)iv * kP;
    const float* q = tmp + t * 3;
    float c[3] = {q[0], q[1], q[2]};
    if (p[5] > 0.f) {                                                   // random_apply(color_jitter, p=0.8)
      const float* mu = mean + iv * 3;
#pragma unroll
      for (int k = 0; k < 4; ++k) jitter_op((int)p[6 + k], c, p, mu);
    }
    if (p[14] > 0.f) {                                                  // random_apply(to_grayscale, p=0.2)
      const float g = c[0] * 0.2989f + c[1] * 0.5870f + c[2] * 0.1140f;
      c[0] = c[1] = c[2] = g;
    }
    float* o = out + ((long long)img * HW + px) * (3 * views) + 3 * v;   // views concatenated on the channel axis
    o[0] = clip01(c[0]); o[1] = clip01(c[1]); o[2] = clip01(c[2]);       // final clip, tf2/data_util.py:473-474
  }
}

}  // namespace

extern "C" {

size_t simclr_augment_workspace_bytes(int b, int views, int H, int W) {
  return ((size_t)b * views * H * W * 3 + (size_t)b * views * 3 + 64) * sizeof(float);
}

// src [b, Hs, Ws, 3] (src_dtype SIMCLR_DT_F32 in [0,1], or 2 = uint8 0..255), params [b, views, 16] float (layout above; the
// crop box must lie inside the source canvas), out [b, H, W, 3*views] float32 in [0,1].  workspace: simclr_augment_workspace_bytes.
int simclr_augment_views(const void* src, int src_dtype, const float* params, void* workspace, float* out, int b,
                         int views, int Hs, int Ws, int H, int W, hipStream_t stream) {
  SIMCLR_CHECK_ARG(src && params && workspace && out, "augment_views: null argument");
  SIMCLR_CHECK_ARG(b > 0 && views > 0 && Hs > 0 && Ws > 0 && H > 0 && W > 0, "augment_views: bad shape");
  SIMCLR_CHECK_ARG(src_dtype == SIMCLR_DT_F32 || src_dtype == 2, "augment_views: src_dtype must be f32 (%d) or uint8 (2)", SIMCLR_DT_F32);
  float* tmp = (float*)workspace;
  float* mean = tmp + (size_t)b * views * H * W * 3;
  const long long total = (long long)b * views * H * W;
  const int grid = (int)min((total + 255) / 256, 1ll << 20);
  if (src_dtype == 2)
    hipLaunchKernelGGL((aug_crop_resize_flip<unsigned char>), dim3(grid), dim3(256), 0, stream, (const unsigned char*)src,
                       params, tmp, b, views, Hs, Ws, H, W);
  else
    hipLaunchKernelGGL((aug_crop_resize_flip<float>), dim3(grid), dim3(256), 0, stream, (const float*)src, params, tmp,
                       b, views, Hs, Ws, H, W);
  SIMCLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(aug_color_mean, dim3(b * views), dim3(256), 0, stream, tmp, params, mean, H * W);
  SIMCLR_CHECK_LAUNCH();
  hipLaunchKernelGGL(aug_color_apply, dim3(grid), dim3(256), 0, stream, tmp, params, mean, out, b, views, H * W);
  SIMCLR_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
