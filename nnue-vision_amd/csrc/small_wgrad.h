// The classifier's small batch-reduced gradients (d_w3, d_w2, the three bias gradients, the mean loss) as a device-side tile family
// that more than one launch can host: the classifier's own d_x launch (classifier_kernels.hip) or the merged FeatureTransformer
// backward launch (ftm_kernels.hip), where ~30 more workgroups disappear beside a few hundred.  Autograd of nnue.py:728-734 for the
// two narrow Linear layers + compute_loss's mean (train.py:250-254).
#pragma once
#include <type_traits>

#include "common.h"

namespace {
using swf32x4 = __attribute__((ext_vector_type(4))) float;

// Bucketed layer stacks (build extension, SURVEY section 7 / BASELINE configs[2]: K independent SimpleClassifier weight
// sets [K][out][in], one selected per sample by its active-feature count -- the vision analogue of the piece-count
// buckets the engine's LayerStack vector descends from, engine/src/nnue_engine.cpp:619-635).  K == 1 with null pointers
// is the reference's single stack and takes exactly the code it always took.  nnue_bucket_group sorts the samples by
// bucket into 16-row tiles, so that every MFMA tile multiplies by ONE bucket's weights ("batched per-bucket MFMA"):
//   bucket[b]         bucket of sample b
//   rows[16 t + i]    sample in row i of tile t, ascending inside a bucket; -1 = padding
//   tile_bucket[t]    bucket of tile t; -1 = unused tile
//   seg[k], seg[k+1]  row range (multiples of 16) of bucket k
struct Buckets {
  int K;
  const int* bucket;
  const int* rows;
  const int* tile_bucket;
  const int* seg;
  int tiles;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
  return v;
}

// Small batch reductions, one workgroup per 16 x 16 output tile (f32 MFMA, K = the batch):
//   d_w3 [C, L3] = d_logits^T h2 | d_w2 [L3, L2] = d_z2^T h1 | d_b3 [C] | d_b2 [L3] | d_b1 [L2] = column sums
// (The earlier form -- one wave per four outputs, lanes striding over the batch -- touched one cache line per lane and
// load: 9.1 us at the CIFAR batch-512 shape, the longest part of the d_x launch it rides in.)
struct SmallWgrad {
  const float *d_logits, *d_z2, *d_z1, *h1, *h2;
  int B, L2, L3, C;
  float *d_w3, *d_b3, *d_w2, *d_b2, *d_b1;
  const float* sample_loss;
  float* loss_out;
  int wgrad_blocks;
  const float* slabs;
  int n_slabs;
  long long slab_count;
  float* d_w1;
  Buckets bk;    // bk.rows != NULL: every output exists once per bucket and sums over that bucket's samples
  int bww_klen;  // rows per d_w1 slab slice (bucketed slab sum: slice s of bucket k is empty from s*klen >= its rows)
};

// red_mem: 1024 floats of the caller's LDS
__device__ __forceinline__ void small_wgrad_body(const SmallWgrad& a, int blk, float* __restrict__ red_mem) {
  const float* __restrict__ d_logits = a.d_logits;
  const float* __restrict__ d_z2 = a.d_z2;
  const float* __restrict__ d_z1 = a.d_z1;
  const float* __restrict__ h1 = a.h1;
  const float* __restrict__ h2 = a.h2;
  const float* __restrict__ sample_loss = a.sample_loss;
  const float* __restrict__ slabs = a.slabs;
  float* __restrict__ d_w3 = a.d_w3; float* __restrict__ d_b3 = a.d_b3; float* __restrict__ d_w2 = a.d_w2;
  float* __restrict__ d_b2 = a.d_b2; float* __restrict__ d_b1 = a.d_b1; float* __restrict__ loss_out = a.loss_out;
  float* __restrict__ d_w1 = a.d_w1;
  const int B = a.B, L2 = a.L2, L3 = a.L3, C = a.C, wgrad_blocks = a.wgrad_blocks, n_slabs = a.n_slabs;
  const long long slab_count = a.slab_count;
  const Buckets& bk = a.bk;
  if (blk >= wgrad_blocks) {  // piggy-backed pass: fixed-order sum of the d_w1 split-K slabs
    const long long i = ((long long)(blk - wgrad_blocks) * 256 + threadIdx.x) * 4;
    if (i >= slab_count) return;
    int nz = n_slabs;
    if (bk.rows) {  // slices of this element's bucket that hold rows (the others were never written)
      const int kb = (int)(i / (slab_count / bk.K));
      const int len = bk.seg[kb + 1] - bk.seg[kb];
      nz = (len + a.bww_klen - 1) / a.bww_klen;
      nz = nz < n_slabs ? nz : n_slabs;
    }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nz > 0) acc = *reinterpret_cast<const float4*>(slabs + i);
    for (int s2 = 1; s2 < nz; ++s2) {
      const float4 v = *reinterpret_cast<const float4*>(slabs + (size_t)s2 * slab_count + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(d_w1 + i) = acc;
    return;
  }
  // One workgroup per 16 x 16 output tile of out[m][n] = sum_b A[b][m] * Bm[b][n] on the f32 MFMA: the batch is walked in
  // chunks of 16 rows, chunk c by wave c % 4 (64-byte runs of each operand row per load instead of one line per lane), the
  // four waves' accumulators are added in wave order through LDS.  Bias gradients are the same tile with A == 1 (row 0 is
  // the column sum).  With layer stacks there is one workgroup per (tile, stack), contracting that stack's grouped rows
  // (zeros for a stack without samples).
  float (*red)[256] = reinterpret_cast<float (*)[256]>(red_mem);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int t3n = (L3 + 15) / 16, t2n = (L2 + 15) / 16, tcm = (C + 15) / 16;
  const int n_w3 = tcm * t3n, n_w2 = t3n * t2n;
  const int stacks = bk.rows ? bk.K : 1;
  const int kb = blk % stacks;  // a workgroup forms its tile for ONE layer stack
  int t = blk / stacks;
  const float* A;
  const float* Bm;
  float* dst;
  int lda = 0, ldb, ldd = 0, M, N, m0 = 0, n0;
  size_t stride_k;
  if (t < n_w3) {
    A = d_logits; lda = C; Bm = h2; ldb = L3; dst = d_w3; ldd = L3; M = C; N = L3; m0 = (t / t3n) * 16; n0 = (t % t3n) * 16; stride_k = (size_t)C * L3;
  } else if ((t -= n_w3) < n_w2) {
    A = d_z2; lda = L3; Bm = h1; ldb = L2; dst = d_w2; ldd = L2; M = L3; N = L2; m0 = (t / t2n) * 16; n0 = (t % t2n) * 16; stride_k = (size_t)L3 * L2;
  } else if ((t -= n_w2) < tcm) {
    A = nullptr; Bm = d_logits; ldb = C; dst = d_b3; M = 1; N = C; n0 = t * 16; stride_k = C;
  } else if ((t -= tcm) < t3n) {
    A = nullptr; Bm = d_z2; ldb = L3; dst = d_b2; M = 1; N = L3; n0 = t * 16; stride_k = L3;
  } else if ((t -= t3n) < t2n) {
    A = nullptr; Bm = d_z1; ldb = L2; dst = d_b1; M = 1; N = L2; n0 = t * 16; stride_k = L2;
  } else {
    if (t == t2n && kb == 0 && loss_out != nullptr && wave == 0) {  // one more workgroup: mean of the per-sample losses, fixed order
      float acc = 0.f;
      for (int b = lane; b < B; b += 64) acc += sample_loss[b];
      acc = wave_sum(acc);
      if (lane == 0) *loss_out = acc / (float)B;
    }
    return;
  }
  const bool m_ok = m0 + r < M, n_ok = n0 + r < N;
  const int mc = m_ok ? m0 + r : 0, nc = n_ok ? n0 + r : 0;  // clamped: every load below is unconditional and in range
  dst += (size_t)kb * stride_k;
  {
    const int g_lo = bk.rows ? bk.seg[kb] : 0, g_hi = bk.rows ? bk.seg[kb + 1] : B;
    const int nch = (g_hi - g_lo + 15) / 16;
    swf32x4 acc = (swf32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int U = 4;  // chunks of loads in flight per wave
    // straight-line load batches (no branch around a load: a conditional load drags an s_waitcnt vmcnt(0) behind it):
    // rows beyond the segment / padding rows read row 0 and are multiplied out
    auto chunk_batch = [&](int c0, auto rows_tag, auto ones_tag) {
      constexpr bool ROWS = decltype(rows_tag)::value, ONES = decltype(ones_tag)::value;
      int bi[U][4];
      bool ok[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int g = g_lo + 16 * (c0 + 4 * u) + 4 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ok[u][e] = g + e < g_hi;
          const int gi = ok[u][e] ? g + e : g_lo;
          bi[u][e] = ROWS ? bk.rows[gi] : gi;
        }
      }
      float av[U][4], bv[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (ROWS) ok[u][e] = ok[u][e] && bi[u][e] >= 0;
          const int b = ok[u][e] ? bi[u][e] : 0;
          av[u][e] = ONES ? 1.0f : A[(size_t)b * lda + mc];
          bv[u][e] = Bm[(size_t)b * ldb + nc];
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a_ = (ONES || (ok[u][e] && m_ok)) ? av[u][e] : 0.0f;
          const float b_ = (ok[u][e] && n_ok) ? bv[u][e] : 0.0f;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_, b_, acc, 0, 0, 0);
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    if (bk.rows) {
      if (A) for (int c0 = wave; c0 < nch; c0 += 4 * U) chunk_batch(c0, T{}, F{});
      else for (int c0 = wave; c0 < nch; c0 += 4 * U) chunk_batch(c0, T{}, T{});
    } else {
      if (A) for (int c0 = wave; c0 < nch; c0 += 4 * U) chunk_batch(c0, F{}, F{});
      else for (int c0 = wave; c0 < nch; c0 += 4 * U) chunk_batch(c0, F{}, T{});
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][lane * 4 + e] = acc[e];
    __syncthreads();
    if (wave == 0 && n_ok) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = lane * 4 + e, m = m0 + 4 * q + e;
        const float v = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
        if (A) {
          if (m < M) dst[(size_t)m * ldd + n0 + r] = v;
        } else if (4 * q + e == 0) {
          dst[n0 + r] = v;
        }
      }
    }
    __syncthreads();
  }
}

// number of output tiles of small_wgrad_body's five families; the launch takes tiles * K workgroups (+ 1 for the mean loss)
inline int small_wgrad_tiles(int L2, int L3, int C) {
  const int t3 = (L3 + 15) / 16, t2 = (L2 + 15) / 16, tc = (C + 15) / 16;
  return tc * t3 + t3 * t2 + tc + t3 + t2;
}

}  // namespace
