// Bucket selector + stable grouping of the samples into bucket-homogeneous 16-row tiles (bucketed layer stacks, build
// extension: SURVEY.md section 7) as a device function of ONE workgroup: the stand-alone kernel (classifier_kernels.hip)
// and an extra workgroup riding in the FeatureTransformer forward launch (ftm_kernels.hip) run the same body.
//   bucket[b] = min(K-1, n[b] * K / (P + 1))  (P = flat ids of the map; P == 0: n[b] already IS the bucket, clamped)
// then a counting sort (ballot ranks inside a wave, LDS prefix over the waves): ascending sample index inside a bucket.
// Include inside the including file's anonymous namespace.
#pragma once

constexpr int kMaxBuckets = 64;

struct GroupArgs {
  const int* n;  // NULL: no grouping requested
  int B, P, K;
  int *bucket, *rows, *tile_bucket, *seg;
  int tiles;
};

constexpr int kGroupLdsInts = 3 * kMaxBuckets + 1 + 16 * kMaxBuckets;  // cnt | start | run | per-wave totals (<= 16 waves)

template <int NT>
__device__ __forceinline__ void bucket_group_body(const GroupArgs& ga, int* __restrict__ lds) {
  static_assert(NT % 64 == 0 && NT <= 1024, "whole waves, at most 16");
  constexpr int NW = NT / 64;
  int* cnt = lds;
  int* start = lds + kMaxBuckets;           // [K + 1]
  int* run = lds + 2 * kMaxBuckets + 1;
  int* wtot = lds + 3 * kMaxBuckets + 1;    // [NW][kMaxBuckets]
  const int* __restrict__ n = ga.n;
  int* __restrict__ bucket = ga.bucket;
  int* __restrict__ rows = ga.rows;
  int* __restrict__ tile_bucket = ga.tile_bucket;
  int* __restrict__ seg = ga.seg;
  const int B = ga.B, P = ga.P, K = ga.K, tiles = ga.tiles;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto select = [&](int nb) {
    int k = P > 0 ? (int)(((long long)nb * K) / ((long long)P + 1)) : nb;
    k = k < 0 ? 0 : k;
    return k < K - 1 ? k : K - 1;
  };
  const int first = tid < B ? select(n[tid]) : -1;  // this thread's sample of the first chunk (requested before anything else)
  if (tid < K) { cnt[tid] = 0; run[tid] = 0; }
  for (int i = tid; i < tiles * 16; i += NT) rows[i] = -1;
  __syncthreads();
  for (int b = tid; b < B; b += NT) {
    const int k = b == tid ? first : select(n[b]);
    bucket[b] = k;
    atomicAdd(&cnt[k], 1);  // integer: exact in any order
  }
  __syncthreads();
  if (tid == 0) {
    start[0] = 0;
    for (int k = 0; k < K; ++k) start[k + 1] = start[k] + (cnt[k] + 15) / 16 * 16;
  }
  __syncthreads();
  if (tid <= K) seg[tid] = start[tid];
  for (int t = tid; t < tiles; t += NT) {
    int kb = -1;
    for (int k = 0; k < K; ++k)
      if (16 * t >= start[k] && 16 * t < start[k] + cnt[k]) kb = k;
    tile_bucket[t] = kb;
  }
  for (int chunk = 0; chunk < B; chunk += NT) {  // stable: ascending sample index inside a bucket
    const int b = chunk + tid;
    const int k = chunk == 0 ? first : (b < B ? select(n[b]) : -1);
    int rank = 0;
    for (int kk = 0; kk < K; ++kk) {
      const unsigned long long m = __ballot(k == kk);
      if (lane == 0) wtot[wave * kMaxBuckets + kk] = __popcll(m);
      if (k == kk) rank = __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (k >= 0) {
      int pre = 0;
      for (int w = 0; w < wave; ++w) pre += wtot[w * kMaxBuckets + k];
      rows[start[k] + run[k] + pre + rank] = b;
    }
    if (chunk + NT >= B) break;  // uniform
    __syncthreads();
    if (tid < K) {
      int add = 0;
      for (int w = 0; w < NW; ++w) add += wtot[w * kMaxBuckets + tid];
      run[tid] += add;
    }
    __syncthreads();
  }
}
