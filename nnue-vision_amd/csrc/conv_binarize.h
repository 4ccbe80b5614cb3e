// Conv forward + StraightThroughBinary.forward as the byte {0,1} map + per-sample counts (the front of the MFMA
// FeatureTransformer path), as a device function of one workgroup (feature_kernels.hip's kernel is its only user today:
// letting the NEXT step's workgroups ride in the optimizer's apply launch of a step group -- each forming the updated conv
// weights itself, the last one to arrive publishing them -- was built on this body and measured at the CIFAR batch-512
// shape: 16.4 us for the shared launch against 7.8 + 6.0 us for the two, the ride's three dependent first touches
// (norm partials, parameters, then the map) cost more than the boundary they replace).
// Include inside the including file's anonymous namespace.
#pragma once

constexpr int kConvBinChunk = 8;  // output channels per register pass

// where the conv weights / thresholds come from: the parameters themselves ...
struct ConvParamsPlain {
  const float* __restrict__ w;    // [fps][27]
  const float* __restrict__ thr;  // [fps]
  __device__ __forceinline__ void prepare() {}
  __device__ __forceinline__ float weight(int i) const { return w[i]; }
  __device__ __forceinline__ float threshold(int c) const { return thr[c]; }
};

// Same fmaf chain as conv3x3_forward_kernel, so conv_out is bitwise the same; bits[b][c*G+hw] = conv_out > thr[c] (one
// byte); n[b] / sink[b] as nnue_ftm_binarize.  Workgroup (bx, by) of a (B, slices) grid walks positions hw = y*T + tid,
// y*T + tid + slices*T, ... of sample bx; with one slice per sample the counts are plain stores, otherwise integer-valued
// atomics into host-zeroed counters (exact in any order).  w_lds: 28 * fpad floats of LDS (fpad = fps rounded up to 8).
// prm.prepare() runs after the first patch is requested and before the weights are read (every thread; it may use
// barriers); `staged` runs once after the weights are in LDS (every thread, after the barrier).
// kFullUnroll: all 27 taps of a channel pass unrolled (the compiler then hoists every weight read: 256 registers -- right for
// the few-channel maps whose workgroups are latency-bound anyway); otherwise nine taps per unrolled pass (60 registers: the
// 64-channel map of the 224x224 shape is occupancy-bound, 28.9 -> 26.4 us).
// `patches` (or NULL): the im2col form of the images, [27][B * G] f32 term-major (patches[q][b * G + hw] = the pixel under tap q of
// position hw, 0 where the tap falls off the image) -- what nnue_ste_conv_backward_patches reads instead of the images and of
// conv_out; `out` may then be NULL (conv_out is not written: at stride 7 a 224x224 image is 5.4x its patches, and conv_out is
// 2.4x them again).
// (kPatch / kOut are compile-time: as runtime pointers tests they cost the few-channel variant 7.6 -> 12.7 us at batch 1024.)
template <bool kFullUnroll, bool kPatch, bool kOut, class Params, class Staged>
__device__ __forceinline__ void conv_binarize_body(const float* __restrict__ img, Params& prm, float* __restrict__ out,
                                                   uint8_t* __restrict__ bits, int* __restrict__ n, float* __restrict__ sink, int H, int W,
                                                   int fps, int stride, int Gh, int Gw, int F, int slices, int bx, int by,
                                                   float* __restrict__ w_lds, const Staged& staged, float* __restrict__ patches = nullptr,
                                                   size_t patch_stride = 0) {
  // weights transposed to [27][fpad] (pad columns zero): the eight channels of a register pass are two ds_read_b128 per
  // patch term instead of eight ds_read_b32 -- the kernel was LDS-issue bound (1728 reads per thread at 64 channels);
  // thr [fpad] behind them
  __shared__ int cnt_s[4], sink_s[4];
  const int fpad = (fps + 7) & ~7;
  float* thr_lds = w_lds + 27 * fpad;
  const int G = Gh * Gw;
  const int b = bx;
  // the first position's patch is requested before the weights are staged: a launch starts with cold caches, and the
  // two first-touch latencies (weights, pixels) would otherwise run one after the other
  float patch[27];
  auto load_patch = [&](int hw) {
    const int h = hw / Gw, x = hw - h * Gw;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int iy = h * stride + kh - 1, ix = x * stride + kw - 1;
          const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
          patch[ci * 9 + kh * 3 + kw] = in ? img[(((size_t)b * 3 + ci) * H + iy) * W + ix] : 0.0f;
        }
  };
  const int hw0 = by * blockDim.x + threadIdx.x;
  if (hw0 < G) load_patch(hw0);
  prm.prepare();
  for (int i = threadIdx.x; i < 27 * fpad; i += blockDim.x) {
    const int q = i / fpad, c = i - q * fpad;
    w_lds[i] = c < fps ? prm.weight(c * 27 + q) : 0.0f;
  }
  for (int i = threadIdx.x; i < fpad; i += blockDim.x) thr_lds[i] = i < fps ? prm.threshold(i) : 0.0f;
  __syncthreads();
  staged();
  int cnt = 0, snk = 0;
  for (int hw = hw0; hw < G; hw += blockDim.x * slices) {
    if (hw != hw0) load_patch(hw);
    if constexpr (kPatch) {
      // buffer stores: one 32-bit lane offset + a scalar offset per term (27 separately formed 64-bit addresses took the
      // 64-channel variant from 60 to 198 registers)
      const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(patches, 0, (unsigned)(27 * patch_stride * 4), 0x00020000);
      const int vo = (b * G + hw) * 4;
#pragma unroll
      for (int q = 0; q < 27; ++q) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(patch[q]), rsp, vo, q * (int)(patch_stride * 4), 0);
    }
    for (int c0 = 0; c0 < fps; c0 += kConvBinChunk) {
      float acc[kConvBinChunk];
#pragma unroll
      for (int u = 0; u < kConvBinChunk; ++u) acc[u] = 0.0f;
      static_assert(kConvBinChunk == 8, "two float4 per patch term");
      auto tap = [&](int q) {
        const float4 wa = *reinterpret_cast<const float4*>(&w_lds[q * fpad + c0]);
        const float4 wb = *reinterpret_cast<const float4*>(&w_lds[q * fpad + c0 + 4]);
        acc[0] = fmaf(patch[q], wa.x, acc[0]); acc[1] = fmaf(patch[q], wa.y, acc[1]);
        acc[2] = fmaf(patch[q], wa.z, acc[2]); acc[3] = fmaf(patch[q], wa.w, acc[3]);
        acc[4] = fmaf(patch[q], wb.x, acc[4]); acc[5] = fmaf(patch[q], wb.y, acc[5]);
        acc[6] = fmaf(patch[q], wb.z, acc[6]); acc[7] = fmaf(patch[q], wb.w, acc[7]);
      };
      if constexpr (kFullUnroll) {
#pragma unroll
        for (int q = 0; q < 27; ++q) tap(q);
      } else {
#pragma unroll 9
        for (int q = 0; q < 27; ++q) tap(q);
      }
#pragma unroll
      for (int u = 0; u < kConvBinChunk; ++u)
        if (c0 + u < fps) {
          const int p = (c0 + u) * G + hw;
          const size_t o = (size_t)b * fps * G + p;
          const bool on = acc[u] > thr_lds[c0 + u];
          if constexpr (kOut) out[o] = acc[u];
          bits[o] = on ? 1 : 0;
          cnt += on;
          snk += on && p >= F - 1;
        }
    }
  }
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    cnt += __shfl_xor(cnt, s);
    snk += __shfl_xor(snk, s);
  }
  if ((threadIdx.x & 63) == 0) {
    cnt_s[threadIdx.x >> 6] = cnt;
    sink_s[threadIdx.x >> 6] = snk;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int total = 0, st = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) {
      total += cnt_s[i];
      st += sink_s[i];
    }
    if (slices == 1) {
      n[b] = total;
      sink[b] = (float)st;
    } else {
      atomicAdd(&n[b], total);
      if (st) atomicAdd(&sink[b], (float)st);
    }
  }
}
