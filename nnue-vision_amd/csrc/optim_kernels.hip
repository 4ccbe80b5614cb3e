// Loss and step tail on gfx950: mean cross-entropy with its gradient (train.py:250-254) and
// clip_grad_norm_ + SGD(momentum, weight_decay) on one flat buffer (train.py:363-366, :457-464).
// Both are small streaming kernels; sums are staged so results are bitwise reproducible.
#include "common.h"

#include <cstdlib>

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v = fmaxf(v, __shfl_xor(v, s));
  return v;
}

// one wave per sample
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits,
                                                            const int64_t* __restrict__ labels, int B, int C,
                                                            float scale_over_b, float* __restrict__ sample_loss,
                                                            float* __restrict__ d_logits) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (b >= B) return;
  const float* __restrict__ z = logits + (size_t)b * C;
  float mx = -INFINITY;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, z[c]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < C; c += 64) se += expf(z[c] - mx);
  se = wave_sum(se);
  const int64_t y = labels[b];
  const bool ok = y >= 0 && y < C;
  const float lse = mx + logf(se);
  if (lane == 0) sample_loss[b] = ok ? lse - z[y] : 0.0f;
  if (d_logits) {
    const float inv = 1.0f / se;
    for (int c = lane; c < C; c += 64) {
      const float p = expf(z[c] - mx) * inv;
      d_logits[(size_t)b * C + c] = ok ? (p - (c == y ? 1.0f : 0.0f)) * scale_over_b : 0.0f;
    }
  }
}

// single block: fixed-order mean of the per-sample losses
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += v[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *out = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
}

// evaluation: argmax (first maximum, as numpy) and one integer atomic per sample into the confusion matrix;
// C == 1 is the reference's binary rule (output > 0.5 vs target > 0.5)
__global__ __launch_bounds__(256) void confusion_kernel(const float* __restrict__ logits,
                                                        const int64_t* __restrict__ labels, int B, int C, int K,
                                                        unsigned long long* __restrict__ confusion) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float* __restrict__ z = logits + (size_t)b * C;
  int pred, truth;
  if (C == 1) {
    pred = z[0] > 0.5f ? 1 : 0;
    truth = labels[b] > 0 ? 1 : 0;  // integer labels: > 0.5  <=>  >= 1
  } else {
    pred = 0;
    float best = z[0];
    for (int c = 1; c < C; ++c)
      if (z[c] > best) {
        best = z[c];
        pred = c;
      }
    const int64_t y = labels[b];
    if (y < 0 || y >= K) return;  // out-of-range label: not counted (caller validates)
    truth = (int)y;
  }
  atomicAdd(&confusion[(size_t)truth * K + pred], 1ull);
}

constexpr int kNormBlocks = 1024;

// Block partial of sum (g * scale)^2: 16-byte loads, four independent chains per thread (the 268 MB gradient of
// the 224x224 configuration is one streaming read; a dependent scalar chain reached only 1.25 TB/s).
__device__ __forceinline__ float sqnorm_block_partial(const float* __restrict__ g, int64_t count, float scale, float* red,
                                                      int blk, int nblk) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  const int64_t stride = (int64_t)nblk * 256;
  const int64_t tid = (int64_t)blk * 256 + threadIdx.x;
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? (count >> 2) : 0;
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
  auto sq = [&](float4 v, float a) {
    v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    return fmaf(v.w, v.w, fmaf(v.z, v.z, fmaf(v.y, v.y, fmaf(v.x, v.x, a))));
  };
  int64_t i = tid;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 v0 = g4[i], v1 = g4[i + stride], v2 = g4[i + 2 * stride], v3 = g4[i + 3 * stride];
    a0 = sq(v0, a0); a1 = sq(v1, a1); a2 = sq(v2, a2); a3 = sq(v3, a3);
  }
  for (; i < n4; i += stride) a0 = sq(g4[i], a0);
  for (int64_t j = n4 * 4 + tid; j < count; j += stride) {
    const float v = g[j] * scale;
    a1 = fmaf(v, v, a1);
  }
  float acc = wave_sum((a0 + a1) + (a2 + a3));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// The plain block partial over g[0, count) without [lo, hi) (a producer already summed that range; lo == hi: nothing
// skipped).  Two passes through the same routine; the barrier between them protects the reduction buffer.
__device__ __forceinline__ float sqnorm_block_partial_skip(const float* __restrict__ g, int64_t count, float scale, float* red, int blk,
                                                           int nblk, int64_t lo, int64_t hi) {
  if (lo >= hi) return sqnorm_block_partial(g, count, scale, red, blk, nblk);
  const float a = sqnorm_block_partial(g, lo, scale, red, blk, nblk);
  __syncthreads();
  const float b = sqnorm_block_partial(g + hi, count - hi, scale, red, blk, nblk);
  return a + b;
}

__global__ __launch_bounds__(256) void sqnorm_stage1(const float* __restrict__ g, int64_t count, float scale,
                                                     float* __restrict__ partial, int64_t lo, int64_t hi) {
  __shared__ float red[4];
  const float v = sqnorm_block_partial_skip(g, count, scale, red, blockIdx.x, gridDim.x, lo, hi);
  if (threadIdx.x == 0) partial[blockIdx.x] = v;
}

// The norm launch with the second stage of a deferred nnue_ste_conv_backward riding in it (feature_kernels.hip,
// ste_conv_backward_stage2: one wave per (channel, term), lanes stride over the output's contiguous run of partials,
// the same sums in the same order).  Workgroups [0, s2_blocks) write d_thr / d_weight -- the first `skip` elements of
// g -- and leave the squares of their four outputs in partial[nb + block] (nb = plain norm workgroups); the others are the plain norm
// workgroups over g[skip:].
__global__ __launch_bounds__(256) void sqnorm_stage1_ste(const float* __restrict__ g, int64_t count, float scale,
                                                         float* __restrict__ partial, const float* __restrict__ ste_partial,
                                                         int chunks, int fps, float* __restrict__ d_thr,
                                                         float* __restrict__ d_weight, int s2_blocks, int64_t skip, int64_t lo,
                                                         int64_t hi, int nb) {
  __shared__ float red[4];
  if ((int)blockIdx.x >= s2_blocks) {
    const float v = sqnorm_block_partial_skip(g + skip, count - skip, scale, red, (int)blockIdx.x - s2_blocks, nb,
                                              lo > skip ? lo - skip : 0, hi > skip ? hi - skip : 0);
    if (threadIdx.x == 0) partial[blockIdx.x - s2_blocks] = v;
    return;
  }
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  float sq = 0.0f;
  if (o < fps * 28) {
    const int c = o / 28, q = o - c * 28;
    const float* __restrict__ run = ste_partial + (size_t)o * chunks;
    float acc = 0.0f;
    int k = lane;
    for (; k + 7 * 64 < chunks; k += 8 * 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = run[k + 64 * u];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; k < chunks; k += 64) acc += run[k];
    acc = wave_sum(acc);
    if (lane == 0) {
      if (q == 27) d_thr[c] = -acc;
      else d_weight[c * 27 + q] = acc;
    }
    sq = (acc * scale) * (acc * scale);
  }
  if (lane == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) partial[nb + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Every block re-derives the norm from the kNormBlocks partials (4 KiB, L2-resident) in the same
// fixed order, then streams its share of the update.
__global__ __launch_bounds__(256) void sgd_apply_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, int64_t count, float lr, float momentum,
                                                        float wd, float max_norm, float scale, int first_step,
                                                        const float* __restrict__ partial, int nparts,
                                                        float* __restrict__ norm_out, const float* __restrict__ ext_partial,
                                                        int ext_count, float* __restrict__ coef_out, int64_t skip_lo, int64_t skip_hi,
                                                        const float* __restrict__ lr_dev) {
  __shared__ double red[4];
  __shared__ float coef_s;
  if (lr_dev) lr = lr_dev[0];  // the learning rate as a device scalar: a scheduler changes it without re-capturing the step
  // The first kPre passes of this thread's share are requested BEFORE the norm is re-derived: after a kernel boundary
  // both the partials and the parameters are first touches (~2 us each from a cold L2), and the two waits would
  // otherwise run one after the other.  Unconditional loads from clamped indices; results used only where valid.
  constexpr int kPre = 4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t hole = skip_hi - skip_lo;  // 0: everything is updated here
  const int64_t live = count - hole, j0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float pw[kPre], pg[kPre], pm[kPre];
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const int64_t j = j0 + u * stride < live ? j0 + u * stride : 0;
    const int64_t i = j < skip_lo ? j : j + hole;
    pw[u] = p[i];
    pg[u] = g[i];
    pm[u] = (m && !first_step) ? m[i] : 0.0f;
  }
  float clip = 1.0f;
  if (max_norm > 0.0f || norm_out || coef_out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += (double)partial[i];
    if (ext_partial) {  // a producer's sums of squares (unscaled) of the range the norm launch skipped
      const double s2 = (double)scale * (double)scale;
      for (int i = threadIdx.x; i < ext_count; i += 256) acc += (double)ext_partial[i] * s2;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
      if (norm_out && blockIdx.x == 0) *norm_out = norm;
      coef_s = max_norm > 0.0f ? fminf(max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
      if (coef_out && blockIdx.x == 0) *coef_out = coef_s;  // for the producer that applies [skip_lo, skip_hi) itself
    }
    __syncthreads();
    clip = coef_s;
  }
  const float gs = clip * scale;
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const int64_t j = j0 + u * stride;
    if (j < live) {
      const int64_t i = j < skip_lo ? j : j + hole;
      float gi = fmaf(wd, pw[u], pg[u] * gs);
      if (m) {
        gi = first_step ? gi : fmaf(momentum, pm[u], gi);
        m[i] = gi;
      }
      p[i] = pw[u] - lr * gi;
    }
  }
  for (int64_t j = j0 + kPre * stride; j < live; j += stride) {
    const int64_t i = j < skip_lo ? j : j + hole;
    const float w = p[i];
    float gi = fmaf(wd, w, g[i] * gs);
    if (m) {
      gi = first_step ? gi : fmaf(momentum, m[i], gi);
      m[i] = gi;
    }
    p[i] = w - lr * gi;
  }
}

// The same update with 16-byte accesses for buffers that are launch-sized (count, the hole and the pointers multiples of 4 /
// 16 bytes): one float4 per thread and pass, the first two passes requested before the norm is re-derived.  (The scalar kernel
// stays for big buffers: with 16-byte accesses it streamed 1.34 GB slower, 225-257 vs 211 us.)
__global__ __launch_bounds__(256) void sgd_apply_vec_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                            int64_t count, float lr, float momentum, float wd, float max_norm, float scale,
                                                            int first_step, const float* __restrict__ partial, int nparts,
                                                            float* __restrict__ norm_out, const float* __restrict__ ext_partial, int ext_count,
                                                            float* __restrict__ coef_out, int64_t skip_lo, int64_t skip_hi,
                                                            const float* __restrict__ lr_dev) {
  __shared__ double red[4];
  __shared__ float coef_s;
  if (lr_dev) lr = lr_dev[0];
  constexpr int kPre = 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t hole4 = (skip_hi - skip_lo) >> 2, lo4 = skip_lo >> 2, live4 = (count >> 2) - hole4;
  const int64_t j0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float4* __restrict__ p4 = reinterpret_cast<const float4*>(p);
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
  const float4* __restrict__ m4 = reinterpret_cast<const float4*>(m);
  float4 pw[kPre], pg[kPre], pm[kPre];
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const int64_t j = j0 + u * stride < live4 ? j0 + u * stride : 0;
    const int64_t i = j < lo4 ? j : j + hole4;
    pw[u] = p4[i];
    pg[u] = g4[i];
    pm[u] = (m && !first_step) ? m4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float clip = 1.0f;
  if (max_norm > 0.0f || norm_out || coef_out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += (double)partial[i];
    if (ext_partial) {
      const double s2 = (double)scale * (double)scale;
      for (int i = threadIdx.x; i < ext_count; i += 256) acc += (double)ext_partial[i] * s2;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
      if (norm_out && blockIdx.x == 0) *norm_out = norm;
      coef_s = max_norm > 0.0f ? fminf(max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
      if (coef_out && blockIdx.x == 0) *coef_out = coef_s;
    }
    __syncthreads();
    clip = coef_s;
  }
  const float gs = clip * scale;
  auto one = [&](float w, float gv, float mv, float& m_new) {  // the arithmetic of sgd_apply_kernel, element by element
    float gi = fmaf(wd, w, gv * gs);
    if (m) gi = first_step ? gi : fmaf(momentum, mv, gi);
    m_new = gi;
    return w - lr * gi;
  };
  auto apply = [&](int64_t i, const float4& w, const float4& gv, const float4& mv) {
    float4 mn, wn;
    wn.x = one(w.x, gv.x, mv.x, mn.x); wn.y = one(w.y, gv.y, mv.y, mn.y);
    wn.z = one(w.z, gv.z, mv.z, mn.z); wn.w = one(w.w, gv.w, mv.w, mn.w);
    if (m) reinterpret_cast<float4*>(m)[i] = mn;
    reinterpret_cast<float4*>(p)[i] = wn;
  };
#pragma unroll
  for (int u = 0; u < kPre; ++u) {
    const int64_t j = j0 + u * stride;
    if (j < live4) apply(j < lo4 ? j : j + hole4, pw[u], pg[u], pm[u]);
  }
  for (int64_t j = j0 + kPre * stride; j < live4; j += stride) {
    const int64_t i = j < lo4 ? j : j + hole4;
    apply(i, p4[i], g4[i], (m && !first_step) ? m4[i] : make_float4(0.f, 0.f, 0.f, 0.f));
  }
}

// Adam (torch.optim.Adam defaults: L2 weight decay folded into the gradient, bias-corrected moments, no
// amsgrad).  The step number lives in device memory (step_counter[0], incremented here by the norm kernel's
// first thread) so that the launch has no changing host argument and can be replayed from a hipGraph.
__global__ __launch_bounds__(256) void sqnorm_stage1_count(const float* __restrict__ g, int64_t count, float scale,
                                                           float* __restrict__ partial, int* __restrict__ step_counter) {
  __shared__ float red[4];
  if (blockIdx.x == 0 && threadIdx.x == 0) step_counter[0] += 1;
  const float v = sqnorm_block_partial(g, count, scale, red, blockIdx.x, gridDim.x);
  if (threadIdx.x == 0) partial[blockIdx.x] = v;
}

__global__ __launch_bounds__(256) void adam_apply_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v, int64_t count,
                                                         float lr, float beta1, float beta2, float eps, float wd,
                                                         float max_norm, float scale, const int* __restrict__ step_counter,
                                                         const float* __restrict__ partial, int nparts,
                                                         float* __restrict__ norm_out, const float* __restrict__ lr_dev) {
  __shared__ double red[4];
  __shared__ float coef_s;
  if (lr_dev) lr = lr_dev[0];
  double acc = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += (double)partial[i];
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
    if (norm_out && blockIdx.x == 0) *norm_out = norm;
    coef_s = max_norm > 0.0f ? fminf(max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
  }
  __syncthreads();
  const float gs = coef_s * scale;
  const int t = step_counter[0];
  const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
  const float step_size = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    const float w = p[i];
    const float gi = fmaf(wd, w, g[i] * gs);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = w - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  }
}

}  // namespace

extern "C" int nnue_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int32_t* step_counter, int64_t count,
                              float lr, float beta1, float beta2, float eps, float weight_decay, float max_norm, float grad_scale,
                              float* norm_out, void* scratch, int64_t scratch_bytes, const float* lr_dev, nnue_stream_t stream) {
  NNUE_REQUIRE(params && grads && exp_avg && exp_avg_sq && step_counter && scratch, NNUE_E_ARG, "nnue_adam_step: null pointer");
  NNUE_REQUIRE(count > 0, NNUE_E_ARG, "nnue_adam_step: count must be positive");
  NNUE_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps > 0.f, NNUE_E_ARG,
               "nnue_adam_step: betas must be in [0,1) and eps > 0");
  NNUE_REQUIRE(scratch_bytes >= nnue_sgd_scratch(count), NNUE_E_SCRATCH, "nnue_adam_step: scratch %lld < %lld bytes",
               (long long)scratch_bytes, (long long)nnue_sgd_scratch(count));
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(scratch);
  hipLaunchKernelGGL(sqnorm_stage1_count, dim3(kNormBlocks), dim3(256), 0, s, grads, count, grad_scale, partial, step_counter);
  int blocks = (int)((count + 1023) / 1024);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_apply_kernel, dim3(blocks), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, count, lr, beta1, beta2, eps,
                     weight_decay, max_norm, grad_scale, step_counter, partial, kNormBlocks, norm_out, lr_dev);
  return nnue_launch_status("nnue_adam_step");
}

extern "C" int nnue_cross_entropy(const float* logits, const int64_t* labels, int B, int C, float grad_scale,
                                  float* sample_loss, float* loss, float* d_logits, nnue_stream_t stream) {
  NNUE_REQUIRE(logits && labels && sample_loss && loss, NNUE_E_ARG, "nnue_cross_entropy: null pointer");
  NNUE_REQUIRE(B > 0 && C > 0, NNUE_E_ARG, "nnue_cross_entropy: B=%d C=%d must be positive", B, C);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(cross_entropy_kernel, dim3((B + 3) / 4), dim3(256), 0, s, logits, labels, B, C,
                     grad_scale / (float)B, sample_loss, d_logits);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, sample_loss, B, loss);
  return nnue_launch_status("nnue_cross_entropy");
}

constexpr int kSteRideBlocks = 1024;  // room for the second-stage workgroups of a deferred STE sum (fps * 28 <= 4096)

extern "C" int64_t nnue_sgd_scratch(int64_t count) {
  (void)count;
  return (kNormBlocks + kSteRideBlocks) * (int64_t)sizeof(float);
}

extern "C" int nnue_sqnorm_partials(const float* grads, int64_t count, float* partial, int nparts, nnue_stream_t stream) {
  NNUE_REQUIRE(grads && partial, NNUE_E_ARG, "nnue_sqnorm_partials: null pointer");
  NNUE_REQUIRE(count > 0 && nparts > 0 && nparts <= 65536, NNUE_E_ARG, "nnue_sqnorm_partials: count=%lld nparts=%d out of range",
               (long long)count, nparts);
  hipLaunchKernelGGL(sqnorm_stage1, dim3(nparts), dim3(256), 0, static_cast<hipStream_t>(stream), grads, count, 1.0f, partial, (int64_t)0,
                     (int64_t)0);
  return nnue_launch_status("nnue_sqnorm_partials");
}

extern "C" int nnue_sgd_step(float* params, float* grads, float* momentum_buf, int64_t count, float lr, float momentum,
                             float weight_decay, float max_norm, float grad_scale, int first_step, float* norm_out,
                             void* scratch, int64_t scratch_bytes, const float* ste_partial, int ste_chunks, int ste_fps,
                             float* ste_d_thr, float* ste_d_weight, const float* ext_partial, int ext_count, int64_t ext_lo,
                             int64_t ext_hi, float* coef_out, int ext_applied_elsewhere, const float* lr_dev, nnue_stream_t stream) {
  NNUE_REQUIRE(params && grads && scratch, NNUE_E_ARG, "nnue_sgd_step: null pointer");
  NNUE_REQUIRE(count > 0, NNUE_E_ARG, "nnue_sgd_step: count must be positive");
  NNUE_REQUIRE(momentum == 0.0f || momentum_buf, NNUE_E_ARG, "nnue_sgd_step: momentum %g needs a momentum buffer", momentum);
  NNUE_REQUIRE(scratch_bytes >= nnue_sgd_scratch(count), NNUE_E_SCRATCH, "nnue_sgd_step: scratch %lld < %lld bytes",
               (long long)scratch_bytes, (long long)nnue_sgd_scratch(count));
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(scratch);
  // norm workgroups: 16 floats per thread while that still leaves fewer than kNormBlocks (a small model's apply kernel
  // then re-derives the norm from a few hundred partials instead of 1024)
  int nb = (int)((count + 4095) / 4096);
  nb = nb < 64 ? 64 : (nb > kNormBlocks ? kNormBlocks : nb);
  int nparts = nb;
  if (ext_partial) {
    NNUE_REQUIRE(ext_count > 0 && ext_count <= 65536 && ext_lo >= 0 && ext_lo < ext_hi && ext_hi <= count && ext_lo % 4 == 0 &&
                     (ext_hi % 4 == 0 || ext_hi == count),
                 NNUE_E_ARG, "nnue_sgd_step: producer partials need 0 < count <= 65536 and a range [lo, hi) of multiples of 4 inside grads");
  } else {
    ext_lo = ext_hi = 0;
    ext_count = 0;
  }
  if (ste_partial) {
    // the deferred sums own the first `skip` elements of grads: [d_thr | d_weight] in either order, nothing else
    NNUE_REQUIRE(ste_d_thr && ste_d_weight && ste_chunks > 0 && ste_fps > 0 && ste_fps * 28 <= 4 * kSteRideBlocks, NNUE_E_ARG,
                 "nnue_sgd_step: deferred STE sums need d_thr, d_weight, chunks > 0 and fps * 28 <= %d", 4 * kSteRideBlocks);
    const float* lo = ste_d_thr < ste_d_weight ? ste_d_thr : ste_d_weight;
    const float* hi_t = ste_d_thr + ste_fps;
    const float* hi_w = ste_d_weight + (size_t)ste_fps * 27;
    const float* hi = hi_t > hi_w ? hi_t : hi_w;
    const int64_t skip = nnue_round_up(hi - grads, 4);
    NNUE_REQUIRE((!ext_partial || ext_lo >= skip) && lo == grads && skip <= count && skip <= nnue_round_up(ste_fps, 4) + nnue_round_up((int64_t)ste_fps * 27, 4) + 8, NNUE_E_ARG,
                 "nnue_sgd_step: deferred STE outputs must be the first elements of grads");
    const int s2_blocks = (ste_fps * 28 + 3) / 4;
    // padding between / after the two outputs is never written by the sums: it enters neither the norm nor is it read
    // before the update multiplies it -- keep it zero (the flat gradient buffer's padding is zero-initialised)
    hipLaunchKernelGGL(sqnorm_stage1_ste, dim3(nb + s2_blocks), dim3(256), 0, s, grads, count, grad_scale, partial, ste_partial,
                       ste_chunks, ste_fps, ste_d_thr, ste_d_weight, s2_blocks, skip, ext_lo, ext_hi, nb);
    nparts = nb + s2_blocks;
  } else if (max_norm > 0.0f || norm_out || coef_out) {
    hipLaunchKernelGGL(sqnorm_stage1, dim3(nb), dim3(256), 0, s, grads, count, grad_scale, partial, ext_lo, ext_hi);
  }
  NNUE_REQUIRE(!ext_applied_elsewhere || (ext_partial && coef_out), NNUE_E_ARG,
               "nnue_sgd_step: ext_applied_elsewhere needs the producer's partials and coef_out");
  const int64_t live = ext_applied_elsewhere ? count - (ext_hi - ext_lo) : count;
  int blocks = (int)((live + 1023) / 1024);
  blocks = blocks < 1 ? 1 : blocks;
  if (blocks > 2048) blocks = 2048;
  const int64_t skip_lo = ext_applied_elsewhere ? ext_lo : 0, skip_hi = ext_applied_elsewhere ? ext_hi : 0;
  float* mom = momentum == 0.0f ? nullptr : momentum_buf;
  static const char* novec = std::getenv("NNUE_SGD_SCALAR");
  const bool vec = !(novec && novec[0] == '1') && live <= (16ll << 20) && count % 4 == 0 && skip_lo % 4 == 0 && skip_hi % 4 == 0 &&
                   nnue_aligned16(params) && nnue_aligned16(grads) && (!mom || nnue_aligned16(mom));
  if (vec) {
    int vb = (int)((live / 4 + 511) / 512);  // two float4 passes per thread
    vb = vb < 1 ? 1 : (vb > 2048 ? 2048 : vb);
    hipLaunchKernelGGL(sgd_apply_vec_kernel, dim3(vb), dim3(256), 0, s, params, grads, mom, count, lr, momentum, weight_decay, max_norm, grad_scale,
                       first_step, partial, nparts, norm_out, ext_partial, ext_count, coef_out, skip_lo, skip_hi, lr_dev);
    return nnue_launch_status("nnue_sgd_step");
  }
  hipLaunchKernelGGL(sgd_apply_kernel, dim3(blocks), dim3(256), 0, s, params, grads, mom, count, lr, momentum, weight_decay, max_norm, grad_scale,
                     first_step, partial, nparts, norm_out, ext_partial, ext_count, coef_out, skip_lo, skip_hi, lr_dev);
  return nnue_launch_status("nnue_sgd_step");
}

extern "C" int nnue_confusion_accumulate(const float* logits, const int64_t* labels, int B, int C, uint64_t* confusion,
                                         nnue_stream_t stream) {
  NNUE_REQUIRE(logits && labels && confusion, NNUE_E_ARG, "nnue_confusion_accumulate: null pointer");
  NNUE_REQUIRE(B > 0 && C > 0, NNUE_E_ARG, "nnue_confusion_accumulate: B=%d C=%d must be positive", B, C);
  const int K = C == 1 ? 2 : C;
  hipLaunchKernelGGL(confusion_kernel, dim3((B + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), logits, labels, B, C, K,
                     reinterpret_cast<unsigned long long*>(confusion));
  return nnue_launch_status("nnue_confusion_accumulate");
}
