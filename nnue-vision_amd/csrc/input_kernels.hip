// Input pipeline on the GPU: what GenericVisionDataset.__getitem__ + the DataLoader's collate
// do per batch in the reference (data/datasets.py:173-195 "light" augmentation, :358-372 Normalize + ToTensorV2),
// as ONE kernel over a uint8 dataset that lives in HBM: gather by index, optional light augmentation,
// Normalize(ImageNet mean/std, max_pixel_value 255), HWC uint8 -> CHW float32, labels gathered to int64.
//
// Randomness is a counter-based hash of (seed, step, dataset index): reproducible and order-independent, but
// it is NOT albumentations' random stream -- the augmented path is "parity unpinned" (albumentations is not
// installed in this environment); the un-augmented path is an exact formula.
#include "common.h"

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ float u01(uint64_t h) { return (float)(h >> 40) * (1.0f / 16777216.0f); }  // [0, 1)

struct Aug {
  bool flip, bc, drop;
  float alpha, beta255;
  int y0, x0, hh, hw;
};

__device__ __forceinline__ Aug draw_aug(uint64_t seed, uint64_t step, int64_t index, int H, int W) {
  const uint64_t base = mix64(seed ^ mix64((uint64_t)index * 0xd1342543de82ef95ull + step));
  Aug a;
  a.flip = u01(mix64(base + 1)) < 0.5f;                       // A.HorizontalFlip(p=0.5)
  a.bc = u01(mix64(base + 2)) < 0.2f;                         // A.RandomBrightnessContrast(0.1, 0.1, p=0.2)
  a.alpha = 1.0f + (u01(mix64(base + 3)) * 0.2f - 0.1f);      //   contrast factor
  a.beta255 = (u01(mix64(base + 4)) * 0.2f - 0.1f) * 255.0f;  //   brightness shift, by max value
  a.drop = u01(mix64(base + 5)) < 0.2f;                       // A.CoarseDropout(1 hole, 5% x 5%, p=0.2)
  a.hh = max(1, (int)(0.05f * H));
  a.hw = max(1, (int)(0.05f * W));
  a.y0 = (int)(u01(mix64(base + 6)) * (float)(H - a.hh + 1));
  a.x0 = (int)(u01(mix64(base + 7)) * (float)(W - a.hw + 1));
  return a;
}

// grid (B, ceil(H*W / 256)); thread = output pixel, all three channels
__global__ __launch_bounds__(256) void load_batch_kernel(const unsigned char* __restrict__ data,
                                                         const int64_t* __restrict__ labels_all,
                                                         const int64_t* __restrict__ indices, int H, int W, int64_t N,
                                                         int augment, uint64_t seed, uint64_t step,
                                                         float* __restrict__ out, int64_t* __restrict__ labels_out) {
  const int b = blockIdx.x;
  int64_t idx = indices[b];
  idx = idx < 0 ? 0 : (idx >= N ? N - 1 : idx);  // stays in bounds; the host validates indices
  const int hw = blockIdx.y * 256 + threadIdx.x;
  if (hw == 0) labels_out[b] = labels_all[idx];
  if (hw >= H * W) return;
  const int h = hw / W, x = hw - h * W;
  Aug a{};
  if (augment) a = draw_aug(seed, step, idx, H, W);
  const int sx = (augment && a.flip) ? W - 1 - x : x;
  const unsigned char* __restrict__ px = data + (((size_t)idx * H + h) * W + sx) * 3;
  const bool hole = augment && a.drop && h >= a.y0 && h < a.y0 + a.hh && x >= a.x0 && x < a.x0 + a.hw;
  // Normalize: (v - 255*mean) * (1 / (255*std))
  const float mean255[3] = {0.485f * 255.0f, 0.456f * 255.0f, 0.406f * 255.0f};
  const float inv_std255[3] = {1.0f / (0.229f * 255.0f), 1.0f / (0.224f * 255.0f), 1.0f / (0.225f * 255.0f)};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = (float)px[c];
    if (augment && a.bc) v = floorf(fminf(fmaxf(v * a.alpha + a.beta255, 0.0f), 255.0f));  // uint8 LUT semantics
    if (hole) v = 0.0f;
    out[(((size_t)b * 3 + c) * H + h) * W + x] = (v - mean255[c]) * inv_std255[c];
  }
}

}  // namespace

extern "C" int nnue_load_batch(const uint8_t* images_u8, const int64_t* labels_all, const int64_t* indices, int B, int H, int W,
                               int64_t N, int augment, uint64_t seed, uint64_t step, float* out, int64_t* labels_out,
                               nnue_stream_t stream) {
  NNUE_REQUIRE(images_u8 && labels_all && indices && out && labels_out, NNUE_E_ARG, "nnue_load_batch: null pointer");
  NNUE_REQUIRE(B > 0 && H > 0 && W > 0 && N > 0, NNUE_E_ARG, "nnue_load_batch: B=%d H=%d W=%d N=%lld must be positive", B, H, W, (long long)N);
  NNUE_REQUIRE((long long)H * W < (1ll << 24), NNUE_E_SHAPE, "nnue_load_batch: image too large");
  hipLaunchKernelGGL(load_batch_kernel, dim3(B, (H * W + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), images_u8, labels_all,
                     indices, H, W, N, augment, seed, step, out, labels_out);
  return nnue_launch_status("nnue_load_batch");
}
