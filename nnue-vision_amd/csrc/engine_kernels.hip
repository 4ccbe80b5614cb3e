// Batched GPU restatement of the reference C++ engine's integer inference,
// NNUEEvaluator::evaluate_logits (engine/src/nnue_engine.cpp:704-734), on the quantised tensors of a `.nnue` file:
// what evaluate_compiled_model (evaluate.py:88-385) obtains from one `nnue_inference` subprocess per image.
// All arithmetic is the engine's integer arithmetic, so results are bit-identical to it; three of its behaviours are
// reproduced on purpose (see oracle/nnue_engine_oracle.py): the image buffer is indexed HWC, the conv weight bytes
// are read as [oc][kh][kw][ic], and the conv's dense [out_h][out_w][oc] output is read back flat with row length g.
#include "common.h"

namespace {

constexpr int kMaxColsPerThread = 8;  // L1 <= 2048

// ConvLayer::forward (nnue_engine.cpp:48-158) into the zero-filled flat [g*g*oc] buffer of nnue_engine.cpp:679-681.
// grid (ceil(F / 256), B); thread = one byte of the flat buffer.
__global__ __launch_bounds__(256) void engine_conv_kernel(const float* __restrict__ images, const int8_t* __restrict__ w,
                                                          const int32_t* __restrict__ bias, float scale, int H, int W,
                                                          int stride, int OH, int OW, int oc, int F,
                                                          int8_t* __restrict__ conv) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= F) return;
  const int b = blockIdx.y;
  int8_t result = 0;
  if (o < OH * OW * oc) {
    const int c = o % oc, pos = o / oc;
    const int oh = pos / OW, ow = pos - oh * OW;
    const float* __restrict__ img = images + (size_t)b * H * W * 3;
    int32_t acc = bias[c];
    for (int kh = 0; kh < 3; ++kh)
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = oh * stride + kh - 1, iw = ow * stride + kw - 1;
        if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
#pragma unroll
        for (int ic = 0; ic < 3; ++ic)
          acc += (int32_t)(img[(ih * W + iw) * 3 + ic] * scale) * (int32_t)w[((c * 3 + kh) * 3 + kw) * 3 + ic];
      }
    int32_t q = acc / (int32_t)scale;  // truncating division, as the engine
    q = q < -127 ? -127 : (q > 127 ? 127 : q);
    result = (int8_t)q;
  }
  conv[(size_t)b * F + o] = result;
}

__device__ __forceinline__ int32_t clamp_i(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One workgroup per image: feature grid + FeatureTransformer (int16 wrap-around) + clipped ReLU + forward_multiclass
// (nnue_engine.h:236-283, simd_scalar.cpp:78-96, nnue_engine.cpp:726-729, :480-539).
// dynamic LDS: ft [L1] i32 | pair [L1] i32 | h1 [L2] i32 | h2 [L3] i32 | counts [4] i32
__global__ __launch_bounds__(256) void engine_stack_kernel(const int8_t* __restrict__ conv, float threshold, int F, int oc,
                                                           const int16_t* __restrict__ ft_w, const int32_t* __restrict__ ft_b,
                                                           int quantized_one, const int8_t* __restrict__ l1_w,
                                                           const int32_t* __restrict__ l1_b, float l1_scale,
                                                           const int8_t* __restrict__ l2_w, const int32_t* __restrict__ l2_b,
                                                           int l2_scale, const int8_t* __restrict__ out_w,
                                                           const int32_t* __restrict__ out_b, float out_scale, int L1, int L2,
                                                           int L3, int C, float* __restrict__ logits, float* __restrict__ density) {
  extern __shared__ int32_t lds[];
  int32_t* ft = lds;
  int32_t* pair = ft + L1;
  int32_t* h1 = pair + L1;
  int32_t* h2 = h1 + L2;
  int32_t* counts = h2 + L3;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int8_t* __restrict__ cv = conv + (size_t)b * F;

  // active features, ascending: every wave forms the same ballots and adds the rows to its own columns
  int32_t acc[kMaxColsPerThread];
#pragma unroll
  for (int j = 0; j < kMaxColsPerThread; ++j) acc[j] = 0;
  int count = 0;
  for (int f0 = 0; f0 < F; f0 += 64) {
    const int f = f0 + lane;
    const bool on = f < F && (float)cv[f] > threshold && (f % oc) < 64;  // 64 channels per cell are bit-packed
    unsigned long long mask = __ballot(on);
    count += __popcll(mask);
    while (mask) {
      const int row = f0 + __builtin_ctzll(mask);
      mask &= mask - 1;
      const int16_t* __restrict__ wr = ft_w + (size_t)row * L1;
#pragma unroll
      for (int j = 0; j < kMaxColsPerThread; ++j) {
        const int col = tid + 256 * j;
        if (col < L1) acc[j] += (int32_t)wr[col];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kMaxColsPerThread; ++j) {
    const int col = tid + 256 * j;
    if (col < L1) {
      const int16_t v = (int16_t)((int32_t)(int16_t)ft_b[col] + acc[j]);  // int16 accumulator wraps
      ft[col] = clamp_i((int32_t)v, 0, quantized_one);
    }
  }
  if (tid == 0) density[b] = (float)count / (float)F;
  __syncthreads();

  // pairwise: (a * b) / 128 clamped to [0, 127] | a clamped to [0, 127]
  const int half = L1 / 2;
  for (int i = tid; i < L1; i += 256) {
    int32_t v = 0;
    if (i < half) v = clamp_i((ft[i] * ft[i + half]) / 128, 0, 127);
    else if (i < 2 * half) v = clamp_i(ft[i - half], 0, 127);
    pair[i] = v;
  }
  __syncthreads();

  // layer 1 (dense_forward_scalar: float division, truncation, clamp to [0, 127]); one wave per output
  for (int o = wave; o < L2; o += 4) {
    const int8_t* __restrict__ wr = l1_w + (size_t)o * L1;
    int32_t s = 0;
    for (int k = lane; k < L1; k += 64) s += pair[k] * (int32_t)wr[k];
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) s += __shfl_xor(s, sh);
    if (lane == 0) {
      const float r = (float)(s + l1_b[o]) / l1_scale;
      h1[o] = clamp_i((int32_t)r, 0, 127);
    }
  }
  __syncthreads();

  // layer 2: integer division, clamp to [-127, 127], ReLU; weights are [L3][2 * L2], first L2 columns used
  for (int o = tid; o < L3; o += 256) {
    const int8_t* __restrict__ wr = l2_w + (size_t)o * 2 * L2;
    int32_t s = l2_b[o];
    for (int k = 0; k < L2; ++k) s += h1[k] * (int32_t)wr[k];
    int32_t r = clamp_i(s / l2_scale, -127, 127);
    h2[o] = r > 0 ? r : 0;
  }
  __syncthreads();

  for (int c = tid; c < C; c += 256) {
    const int8_t* __restrict__ wr = out_w + (size_t)c * L3;
    int32_t s = out_b[c];
    for (int j = 0; j < L3; ++j) s += h2[j] * (int32_t)wr[j];
    logits[(size_t)b * C + c] = (float)s / out_scale;
  }
  (void)counts;
}

}  // namespace

extern "C" int64_t nnue_engine_scratch(const nnue_engine_model* m, int B) {
  if (!m || B <= 0 || m->num_features <= 0) return 0;
  return (int64_t)B * m->num_features;
}

extern "C" int nnue_engine_evaluate_logits(const nnue_engine_model* m, const float* images, int B, int H, int W, float* logits,
                                           float* density, void* scratch, int64_t scratch_bytes, nnue_stream_t stream) {
  NNUE_REQUIRE(m && images && logits && density && scratch, NNUE_E_ARG, "nnue_engine_evaluate_logits: null pointer");
  NNUE_REQUIRE(m->conv_w && m->conv_b && m->ft_w && m->ft_b && m->l1_w && m->l1_b && m->l2_w && m->l2_b && m->out_w && m->out_b,
               NNUE_E_ARG, "nnue_engine_evaluate_logits: model tensor missing");
  NNUE_REQUIRE(B > 0 && H > 0 && W > 0, NNUE_E_ARG, "nnue_engine_evaluate_logits: B=%d H=%d W=%d must be positive", B, H, W);
  const int g = m->grid, oc = m->oc, F = m->num_features;
  NNUE_REQUIRE(g > 0 && oc > 0 && F == g * g * oc, NNUE_E_SHAPE, "nnue_engine_evaluate_logits: num_features %d != %d*%d*%d", F, g, g, oc);
  NNUE_REQUIRE(m->l1 >= 2 && m->l1 <= 256 * kMaxColsPerThread && m->l2 >= 1 && m->l3 >= 1 && m->classes >= 1, NNUE_E_SHAPE,
               "nnue_engine_evaluate_logits: L1=%d (2..%d) L2=%d L3=%d C=%d", m->l1, 256 * kMaxColsPerThread, m->l2, m->l3, m->classes);
  NNUE_REQUIRE(m->conv_scale >= 1.0f && m->l2_scale >= 1.0f && m->l1_scale > 0.0f && m->out_scale > 0.0f, NNUE_E_ARG,
               "nnue_engine_evaluate_logits: scales must be positive (integer scales >= 1)");
  // the engine's own stride rule, ceil((H-1)/(g-1)) (nnue_engine.cpp:710-718) -- not the training stride
  int stride = g > 1 ? (H - 1 + g - 2) / (g - 1) : (H > 1 ? H : 1);
  if (stride < 1) stride = 1;
  const int OH = (H + 2 - 3) / stride + 1, OW = (W + 2 - 3) / stride + 1;
  NNUE_REQUIRE(OH > 0 && OW > 0 && (long long)OH * OW * oc <= F, NNUE_E_SHAPE,
               "nnue_engine_evaluate_logits: a %dx%d image gives a %dx%d map that overruns the engine's %dx%d grid buffer", H, W, OH,
               OW, g, g);
  NNUE_REQUIRE(scratch_bytes >= (int64_t)B * F, NNUE_E_SCRATCH, "nnue_engine_evaluate_logits: scratch %lld < %lld bytes",
               (long long)scratch_bytes, (long long)B * F);
  NNUE_REQUIRE((long long)B * H * W * 3 < (1ll << 40), NNUE_E_SHAPE, "nnue_engine_evaluate_logits: batch too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int8_t* conv = static_cast<int8_t*>(scratch);
  hipLaunchKernelGGL(engine_conv_kernel, dim3((F + 255) / 256, B), dim3(256), 0, s, images, m->conv_w, m->conv_b, m->conv_scale, H, W,
                     stride, OH, OW, oc, F, conv);
  const size_t lds = (size_t)(2 * m->l1 + m->l2 + m->l3 + 4) * sizeof(int32_t);
  NNUE_REQUIRE(lds <= 64 * 1024, NNUE_E_SHAPE, "nnue_engine_evaluate_logits: layer sizes need %zu bytes of LDS", lds);
  hipLaunchKernelGGL(engine_stack_kernel, dim3(B), dim3(256), lds, s, conv, m->threshold, F, oc, m->ft_w, m->ft_b,
                     (int)(int16_t)m->quantized_one, m->l1_w, m->l1_b, m->l1_scale, m->l2_w, m->l2_b, (int)m->l2_scale, m->out_w,
                     m->out_b, m->out_scale, m->l1, m->l2, m->l3, m->classes, logits, density);
  return nnue_launch_status("nnue_engine_evaluate_logits");
}
