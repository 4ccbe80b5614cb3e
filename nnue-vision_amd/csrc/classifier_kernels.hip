// Pairwise product + SimpleClassifier (nnue.py:660-666, :713-738) forward and backward on gfx950.
//
//   l0 = pairwise ? cat(x[:, :h] * x[:, h:], x[:, :h]) : x           (h = L1/2; never materialised)
//   h1 = act(l0 W1^T + b1)   h2 = act(h1 W2^T + b2)   logits = h2 W3^T + b3
//
// Only the L1-wide layer is a real contraction (B x L1 x L2; 134 MFLOP at B=512, 1024 -> 128).  It
// runs on the f32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32 products, fp32 accumulate) when the
// shape allows (L1 % 32 == 0, L2 % 16 == 0), split over K so that >= 512 waves are in flight, and
// on plain-VALU "simple" kernels otherwise.  The two narrow layers (L2 -> L3 -> C) are a per-sample
// tail kernel working out of LDS.  Split-K partials are summed in fixed order: reproducible.
#include <type_traits>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "small_wgrad.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float act_fn(float z, float clip) {
  const float r = fmaxf(z, 0.0f);
  return clip > 0.0f ? fminf(r, clip) : r;
}
__device__ __forceinline__ float gate_fn(float h, float clip) {
  return (h > 0.0f && (clip <= 0.0f || h < clip)) ? 1.0f : 0.0f;
}

// element k of the (virtual) l0 row built from x row `xr`
__device__ __forceinline__ float l0_at(const float* __restrict__ xr, int k, int half, int pairwise) {
  if (!pairwise) return xr[k];
  return k < half ? xr[k] * xr[k + half] : xr[k - half];
}

// ------------------------------------------------------------------ layer 1, any shape
// wave per output (b, j): lanes stride over K; pre-activation, bias not yet added.
__global__ __launch_bounds__(256) void l1_forward_simple(const float* __restrict__ x, int pairwise,
                                                         const float* __restrict__ w1, int B, int L1, int L2,
                                                         float* __restrict__ part, Buckets bk) {
  const long long o = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= (long long)B * L2) return;
  const int b = (int)(o / L2), j = (int)(o - (long long)b * L2);
  const int lane = threadIdx.x & 63, half = L1 / 2;
  const float* __restrict__ xr = x + (size_t)b * L1;
  const float* __restrict__ wr = w1 + ((size_t)(bk.bucket ? bk.bucket[b] : 0) * L2 + j) * L1;
  float acc = 0.f;
  for (int k = lane; k < L1; k += 64) acc = fmaf(l0_at(xr, k, half, pairwise), wr[k], acc);
  acc = wave_sum(acc);
  if (lane == 0) part[o] = acc;
}

// d_w1[i, j] = sum_b d_z1[b, i] * l0[b, j]; thread per (i, j)
__global__ __launch_bounds__(256) void l1_backward_w_simple(const float* __restrict__ x, int pairwise,
                                                            const float* __restrict__ d_z1, int B, int L1, int L2,
                                                            float* __restrict__ d_w1, Buckets bk) {
  const int i = blockIdx.x % L2, kb = blockIdx.x / L2;  // grid.x = K * L2
  const int j = blockIdx.y * 256 + threadIdx.x;
  if (j >= L1) return;
  const int half = L1 / 2;
  float acc = 0.f;
  if (bk.rows) {  // the samples of bucket kb, ascending
    for (int g = bk.seg[kb]; g < bk.seg[kb + 1]; ++g) {
      const int b = bk.rows[g];
      if (b >= 0) acc = fmaf(d_z1[(size_t)b * L2 + i], l0_at(x + (size_t)b * L1, j, half, pairwise), acc);
    }
  } else {
    for (int b = 0; b < B; ++b) acc = fmaf(d_z1[(size_t)b * L2 + i], l0_at(x + (size_t)b * L1, j, half, pairwise), acc);
  }
  d_w1[((size_t)kb * L2 + i) * L1 + j] = acc;
}

// d_l0 = d_z1 W1, then the pairwise block's own backward; thread per (b, j)
__global__ __launch_bounds__(256) void l1_backward_x_simple(const float* __restrict__ x, int pairwise,
                                                            const float* __restrict__ w1,
                                                            const float* __restrict__ d_z1, int B, int L1, int L2,
                                                            float* __restrict__ d_x, Buckets bk) {
  const int b = blockIdx.x;
  const int j = blockIdx.y * 256 + threadIdx.x;
  const int half = L1 / 2;
  const float* __restrict__ dz = d_z1 + (size_t)b * L2;
  if (bk.bucket) w1 += (size_t)bk.bucket[b] * L2 * L1;
  if (pairwise) {
    if (j >= half) return;
    float lo = 0.f, hi = 0.f;
    for (int k = 0; k < L2; ++k) {
      lo = fmaf(dz[k], w1[(size_t)k * L1 + j], lo);
      hi = fmaf(dz[k], w1[(size_t)k * L1 + j + half], hi);
    }
    const float* __restrict__ xr = x + (size_t)b * L1;
    d_x[(size_t)b * L1 + j] = fmaf(lo, xr[j + half], hi);
    d_x[(size_t)b * L1 + j + half] = lo * xr[j];
  } else {
    if (j >= L1) return;
    float acc = 0.f;
    for (int k = 0; k < L2; ++k) acc = fmaf(dz[k], w1[(size_t)k * L1 + j], acc);
    d_x[(size_t)b * L1 + j] = acc;
  }
}

// ------------------------------------------------------------------ layer 1 on the f32 MFMA
// Lane map of v_mfma_f32_16x16x4_f32: lane l = 16*q + r supplies A[row r][k q] and B[k q][col r];
// accumulator register t holds D[row 4*q + t][col r].
//
// Forward ("NT": both operands contiguous along K).  A wave owns a 16-sample x 64-unit tile and a
// K slice.  Each lane loads float4 along K, i.e. K elements kb + 4q .. 4q+3, and MFMA step t uses
// element t on BOTH operands -- a permutation of the K order, which a sum does not care about.
constexpr int kFwdTileN = 4;  // 16x16 tiles along L2 per wave

__global__ __launch_bounds__(256) void l1_forward_mfma(const float* __restrict__ x, int pairwise,
                                                       const float* __restrict__ w1, int B, int L1, int L2,
                                                       int ksplit, float* __restrict__ part, Buckets bk) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int n_groups = (L2 + 16 * kFwdTileN - 1) / (16 * kFwdTileN);
  const int m_tiles = bk.rows ? bk.tiles : (B + 15) / 16;
  long long wid = (long long)blockIdx.x * 4 + wave;
  if (wid >= (long long)m_tiles * n_groups * ksplit) return;
  const int ks = (int)(wid % ksplit);
  wid /= ksplit;
  const int ng = (int)(wid % n_groups);
  const int mt = (int)(wid / n_groups);
  const int klen = L1 / ksplit;  // multiple of 16, inside one half when pairwise (ksplit even)
  const int k_lo = ks * klen;
  const int half = L1 / 2;
  int row = mt * 16 + r;
  int orow[4];  // sample behind accumulator register e (row 4 q + e of the tile)
#pragma unroll
  for (int e = 0; e < 4; ++e) orow[e] = mt * 16 + 4 * q + e;
  if (bk.rows) {  // a tile of one bucket: its rows are samples gathered by the grouping, its weights that bucket's
    const int kb = bk.tile_bucket[mt];  // wave-uniform
    if (kb < 0) return;
    w1 += (size_t)kb * L2 * L1;
    row = bk.rows[row];
#pragma unroll
    for (int e = 0; e < 4; ++e) orow[e] = bk.rows[orow[e]];
  }
  const bool row_ok = row >= 0 && row < B;
  const float* __restrict__ xr = x + (size_t)(row_ok ? row : 0) * L1;
  // pairwise: first half of l0 = x[k] * x[k + half]; second half = x[k - half]
  const bool prod = pairwise && k_lo < half;
  const int xoff = (pairwise && !prod) ? -half : 0;

  f32x4 acc[kFwdTileN];
#pragma unroll
  for (int t = 0; t < kFwdTileN; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* wrow[kFwdTileN];
  bool col_ok[kFwdTileN];
#pragma unroll
  for (int t = 0; t < kFwdTileN; ++t) {
    const int col = (ng * kFwdTileN + t) * 16 + r;
    col_ok[t] = col < L2;
    wrow[t] = w1 + (size_t)(col_ok[t] ? col : 0) * L1;
  }
  // software pipeline: the operands of K block i+1 are requested before the MFMAs of block i issue
  auto load_a = [&](int kb) {
    const int k = kb + 4 * q;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_ok) {
      a = *reinterpret_cast<const float4*>(xr + k + xoff);
      if (prod) {
        const float4 a2 = *reinterpret_cast<const float4*>(xr + k + half);
        a.x *= a2.x; a.y *= a2.y; a.z *= a2.z; a.w *= a2.w;
      }
    }
    return a;
  };
  auto load_b = [&](int kb, int t) {
    return col_ok[t] ? *reinterpret_cast<const float4*>(wrow[t] + kb + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  const int k_hi = k_lo + klen;
  float4 a = load_a(k_lo), bv[kFwdTileN];
#pragma unroll
  for (int t = 0; t < kFwdTileN; ++t) bv[t] = load_b(k_lo, t);
  for (int kb = k_lo; kb < k_hi; kb += 16) {
    const int kn = kb + 16 < k_hi ? kb + 16 : kb;  // last iteration re-reads its own block (harmless)
    const float4 an = load_a(kn);
    float4 bn[kFwdTileN];
#pragma unroll
    for (int t = 0; t < kFwdTileN; ++t) bn[t] = load_b(kn, t);
#pragma unroll
    for (int t = 0; t < kFwdTileN; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bv[t].x, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bv[t].y, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bv[t].z, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bv[t].w, acc[t], 0, 0, 0);
    }
    a = an;
#pragma unroll
    for (int t = 0; t < kFwdTileN; ++t) bv[t] = bn[t];
  }
  float* __restrict__ out = part + (size_t)ks * B * L2;
#pragma unroll
  for (int t = 0; t < kFwdTileN; ++t) {
    const int col = (ng * kFwdTileN + t) * 16 + r;
    if (col >= L2) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (orow[e] >= 0 && orow[e] < B) out[(size_t)orow[e] * L2 + col] = acc[t][e];
  }
}

// d_w1 = d_z1^T l0 ("TN": both operands contiguous along the non-K dimension; K = batch).  A wave
// owns 2 x 4 tiles (32 units x 64 columns) and a slice of the batch; per K step (4 samples) every
// lane loads one dword per tile row/column block.
constexpr int kBwTileM = 2, kBwTileN = 4;

// Bucketed (bk.rows): a wave owns the same tile of ONE bucket's d_w1 and slice ks of that bucket's row range
// [seg[kb] + ks*klen, .. + klen); a slice that starts past the bucket's end writes nothing (the slab sum knows, see
// bucket_slabs).  Slab layout [ksplit][K * L2 * L1].
__device__ __forceinline__ void l1_backward_w_body(const float* __restrict__ x, int pairwise, const float* __restrict__ d_z1,
                                                   int B, int L1, int L2, int ksplit, int klen, float* __restrict__ part,
                                                   long long block, const Buckets& bk) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int m_groups = L2 / (16 * kBwTileM);  // L2 % 32 == 0
  const int n_groups = L1 / (16 * kBwTileN);  // L1 % 64 == 0
  long long wid = block * 4 + wave;
  if (wid >= (long long)m_groups * n_groups * ksplit * bk.K) return;
  const int ks = (int)(wid % ksplit);
  wid /= ksplit;
  const int ng = (int)(wid % n_groups);
  wid /= n_groups;
  const int mg = (int)(wid % m_groups);
  const int kb = (int)(wid / m_groups);
  const int half = L1 / 2;
  int b_lo = ks * klen, b_hi = min(B, b_lo + klen);
  if (bk.rows) {
    const int s_lo = bk.seg[kb], s_hi = bk.seg[kb + 1];
    b_lo = s_lo + ks * klen;
    b_hi = min(s_hi, b_lo + klen);
    if (b_lo >= s_hi && !(ks == 0 && ksplit == 1)) return;  // empty slice; an unsplit product still writes its zeros
  }
  f32x4 acc[kBwTileM][kBwTileN];
#pragma unroll
  for (int i = 0; i < kBwTileM; ++i)
#pragma unroll
    for (int t = 0; t < kBwTileN; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // column j of l0 for each of this lane's N tiles; a 16-column tile never straddles `half` (half % 16 == 0)
  int cj[kBwTileN], cj2[kBwTileN];
#pragma unroll
  for (int t = 0; t < kBwTileN; ++t) {
    const int j = (ng * kBwTileN + t) * 16 + r;
    if (!pairwise) { cj[t] = j; cj2[t] = -1; }
    else if (j < half) { cj[t] = j; cj2[t] = j + half; }
    else { cj[t] = j - half; cj2[t] = -1; }
  }
  auto load_step = [&](int b0, float (&a)[kBwTileM], float (&bv)[kBwTileN]) {
    int b = b0 + q;
    bool ok = b < b_hi;
    if (bk.rows) {
      b = ok ? bk.rows[b] : -1;
      ok = b >= 0;
    }
    const float* __restrict__ dz = d_z1 + (size_t)(ok ? b : 0) * L2 + mg * 16 * kBwTileM + r;
    const float* __restrict__ xr = x + (size_t)(ok ? b : 0) * L1;
#pragma unroll
    for (int i = 0; i < kBwTileM; ++i) a[i] = ok ? dz[16 * i] : 0.f;
#pragma unroll
    for (int t = 0; t < kBwTileN; ++t) {
      float v = ok ? xr[cj[t]] : 0.f;
      if (cj2[t] >= 0 && ok) v *= xr[cj2[t]];
      bv[t] = v;
    }
  };
  constexpr int kSteps = 8;  // K steps (of 4 samples) whose loads are issued together
  for (int b0 = b_lo; b0 < b_hi; b0 += 4 * kSteps) {
    float a[kSteps][kBwTileM], bv[kSteps][kBwTileN];
#pragma unroll
    for (int u = 0; u < kSteps; ++u) load_step(b0 + 4 * u, a[u], bv[u]);  // rows >= b_hi load as zero
#pragma unroll
    for (int u = 0; u < kSteps; ++u)
#pragma unroll
      for (int i = 0; i < kBwTileM; ++i)
#pragma unroll
        for (int t = 0; t < kBwTileN; ++t)
          acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], bv[u][t], acc[i][t], 0, 0, 0);
  }
  float* __restrict__ out = part + ((size_t)ks * bk.K + kb) * L2 * L1;
#pragma unroll
  for (int i = 0; i < kBwTileM; ++i)
#pragma unroll
    for (int t = 0; t < kBwTileN; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int orow = (mg * kBwTileM + i) * 16 + 4 * q + e;
        const int ocol = (ng * kBwTileN + t) * 16 + r;
        out[(size_t)orow * L1 + ocol] = acc[i][t][e];
      }
}

// fixed-order sum of split-K slabs: out[i] = sum_s part[s][i]
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ part, int slabs, long long count,
                                                       float* __restrict__ out, Buckets bk, int klen) {
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= count) return;  // count % 4 == 0
  if (bk.rows) {  // only the slices of this element's bucket that hold rows were written (l1_backward_w_body)
    const int kb = (int)(i / (count / bk.K));
    const int nz = (bk.seg[kb + 1] - bk.seg[kb] + klen - 1) / klen;
    slabs = nz < slabs ? nz : slabs;
    if (slabs == 0) {
      *reinterpret_cast<float4*>(out + i) = make_float4(0.f, 0.f, 0.f, 0.f);
      return;
    }
  }
  float4 acc = *reinterpret_cast<const float4*>(part + i);
  for (int s = 1; s < slabs; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(part + (size_t)s * count + i);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(out + i) = acc;
}

// d_x = (d_z1 W1) through the pairwise block ("NN": A contiguous along K = L2, B along N).  A wave
// owns 16 samples x two 16-column tiles: columns j and j + L1/2 when pairwise (the pairwise backward
// needs both), adjacent tiles otherwise.
__device__ __forceinline__ void l1_backward_x_body(const float* __restrict__ x, int pairwise, const float* __restrict__ w1,
                                                   const float* __restrict__ d_z1, int B, int L1, int L2,
                                                   float* __restrict__ d_x, long long block, const Buckets& bk) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int half = L1 / 2;
  const int n_pairs = L1 / 32;
  const int m_tiles = bk.rows ? bk.tiles : (B + 15) / 16;
  const long long wid = block * 4 + wave;
  if (wid >= (long long)m_tiles * n_pairs) return;
  const int np = (int)(wid % n_pairs);
  const int mt = (int)(wid / n_pairs);
  const int c0 = pairwise ? np * 16 : np * 32;
  const int c1 = pairwise ? c0 + half : c0 + 16;
  int row = mt * 16 + r;
  int orows[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) orows[e] = mt * 16 + 4 * q + e;
  if (bk.rows) {  // tile of one bucket (see l1_forward_mfma)
    const int kb = bk.tile_bucket[mt];
    if (kb < 0) return;
    w1 += (size_t)kb * L2 * L1;
    row = bk.rows[row];
#pragma unroll
    for (int e = 0; e < 4; ++e) orows[e] = bk.rows[orows[e]];
  }
  const bool row_ok = row >= 0 && row < B;
  const float* __restrict__ dz = d_z1 + (size_t)(row_ok ? row : 0) * L2;
  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  // K is walked in chunks of up to 4 blocks of 16; every load of a chunk is issued before its MFMAs, so a
  // wave pays one L2 round trip per chunk (L2 = 128: two) instead of one per block
  constexpr int KC = 4;
  for (int kb0 = 0; kb0 < L2; kb0 += 16 * KC) {  // L2 % 16 == 0
    float4 a[KC];
    float b0v[KC][4], b1v[KC][4];
#pragma unroll
    for (int u = 0; u < KC; ++u) {
      const int kb = kb0 + 16 * u;
      const bool in = kb < L2;  // wave-uniform
      const int k = (in ? kb : 0) + 4 * q;
      a[u] = (row_ok && in) ? *reinterpret_cast<const float4*>(dz + k) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float* __restrict__ wk = w1 + (size_t)k * L1 + r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        b0v[u][e] = in ? wk[(size_t)e * L1 + c0] : 0.f;
        b1v[u][e] = in ? wk[(size_t)e * L1 + c1] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < KC; ++u) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b0v[u][0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b1v[u][0], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b0v[u][1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b1v[u][1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b0v[u][2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b1v[u][2], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b0v[u][3], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b1v[u][3], acc1, 0, 0, 0);
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int orow = orows[e];
    if (orow < 0 || orow >= B) continue;
    float* __restrict__ o = d_x + (size_t)orow * L1;
    if (pairwise) {
      const float* __restrict__ xr = x + (size_t)orow * L1;
      o[c0 + r] = fmaf(acc0[e], xr[c1 + r], acc1[e]);
      o[c1 + r] = acc0[e] * xr[c0 + r];
    } else {
      o[c0 + r] = acc0[e];
      o[c1 + r] = acc1[e];
    }
  }
}

__global__ __launch_bounds__(256) void l1_backward_w_mfma(const float* __restrict__ x, int pairwise,
                                                          const float* __restrict__ d_z1, int B, int L1, int L2,
                                                          int ksplit, int klen, float* __restrict__ part, Buckets bk) {
  l1_backward_w_body(x, pairwise, d_z1, B, L1, L2, ksplit, klen, part, blockIdx.x, bk);
}

__global__ __launch_bounds__(256) void l1_backward_x_mfma(const float* __restrict__ x, int pairwise,
                                                          const float* __restrict__ w1,
                                                          const float* __restrict__ d_z1, int B, int L1, int L2,
                                                          float* __restrict__ d_x, Buckets bk) {
  l1_backward_x_body(x, pairwise, w1, d_z1, B, L1, L2, d_x, blockIdx.x, bk);
}

// d_x and the d_w1 slabs in one launch: both only need d_z1, so the two small products share the chip instead of
// running back to back.  Blocks [0, w_blocks) form the weight-gradient slabs (the longer product, dispatched first),
// the rest d_x.
__global__ __launch_bounds__(256) void l1_backward_xw_mfma(const float* __restrict__ x, int pairwise,
                                                           const float* __restrict__ w1, const float* __restrict__ d_z1,
                                                           int B, int L1, int L2, float* __restrict__ d_x, int w_blocks,
                                                           int ksplit, int klen, float* __restrict__ part, Buckets bk) {
  if ((int)blockIdx.x < w_blocks) l1_backward_w_body(x, pairwise, d_z1, B, L1, L2, ksplit, klen, part, blockIdx.x, bk);
  else l1_backward_x_body(x, pairwise, w1, d_z1, B, L1, L2, d_x, (long long)blockIdx.x - w_blocks, bk);
}

// ------------------------------------------------------------------ narrow layers (per sample, LDS)
__global__ __launch_bounds__(128) void tail_forward_kernel(const float* __restrict__ part, int ksplit,
                                                           const float* __restrict__ b1, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, const float* __restrict__ w3,
                                                           const float* __restrict__ b3, float clip, int B, int L2,
                                                           int L3, int C, float* __restrict__ h1,
                                                           float* __restrict__ h2, float* __restrict__ logits, Buckets bk) {
  extern __shared__ float lds[];  // h1 [L2], h2 [L3]
  float* h1s = lds;
  float* h2s = lds + L2;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (bk.bucket) {  // this sample's layer stack
    const int kb = bk.bucket[b];
    b1 += (size_t)kb * L2; w2 += (size_t)kb * L3 * L2; b2 += (size_t)kb * L3; w3 += (size_t)kb * C * L3; b3 += (size_t)kb * C;
  }
  for (int j = tid; j < L2; j += 128) {
    float z = b1[j];
    for (int s = 0; s < ksplit; ++s) z += part[((size_t)s * B + b) * L2 + j];
    const float h = act_fn(z, clip);
    h1s[j] = h;
    h1[(size_t)b * L2 + j] = h;
  }
  __syncthreads();
  for (int j = tid; j < L3; j += 128) {
    const float* __restrict__ wr = w2 + (size_t)j * L2;
    float z = b2[j];
    for (int k = 0; k < L2; ++k) z = fmaf(wr[k], h1s[k], z);
    const float h = act_fn(z, clip);
    h2s[j] = h;
    h2[(size_t)b * L3 + j] = h;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 128) {
    const float* __restrict__ wr = w3 + (size_t)c * L3;
    float z = b3[c];
    for (int k = 0; k < L3; ++k) z = fmaf(wr[k], h2s[k], z);
    logits[(size_t)b * C + c] = z;
  }
}

__global__ __launch_bounds__(128) void tail_backward_kernel(const float* __restrict__ d_logits,
                                                            const float* __restrict__ h1, const float* __restrict__ h2,
                                                            const float* __restrict__ w2, const float* __restrict__ w3,
                                                            float clip, int L2, int L3, int C,
                                                            float* __restrict__ d_z1, float* __restrict__ d_z2, Buckets bk) {
  extern __shared__ float lds[];  // d_logits [C], d_z2 [L3]
  float* dls = lds;
  float* dz2s = lds + C;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (bk.bucket) {
    const int kb = bk.bucket[b];
    w2 += (size_t)kb * L3 * L2; w3 += (size_t)kb * C * L3;
  }
  for (int c = tid; c < C; c += 128) dls[c] = d_logits[(size_t)b * C + c];
  __syncthreads();
  for (int j = tid; j < L3; j += 128) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s = fmaf(dls[c], w3[(size_t)c * L3 + j], s);
    const float v = s * gate_fn(h2[(size_t)b * L3 + j], clip);
    dz2s[j] = v;
    d_z2[(size_t)b * L3 + j] = v;
  }
  __syncthreads();
  for (int k = tid; k < L2; k += 128) {
    float s = 0.f;
    for (int j = 0; j < L3; ++j) s = fmaf(dz2s[j], w2[(size_t)j * L2 + k], s);
    d_z1[(size_t)b * L2 + k] = s * gate_fn(h1[(size_t)b * L2 + k], clip);
  }
}

__global__ __launch_bounds__(256) void small_wgrad_kernel(SmallWgrad a) {
  __shared__ float red[1024];
  small_wgrad_body(a, (int)blockIdx.x, red);
}

// ------------------------------------------------------------------ fused narrow layers + loss (training)
// Per sample, out of LDS: h1 = act(sum of split-K slabs + b1), h2, logits, softmax cross-entropy, d_logits,
// then back through the two narrow layers to d_z2 and d_z1.  Replaces tail_forward + cross_entropy +
// tail_backward (three launches and two round trips of logits / d_logits through memory).
// Dot products use 4 threads per output (coalesced 16-byte runs of each weight row).
template <int NT>
__device__ __forceinline__ float block_reduce_nt(float v, bool is_max, float* red) {  // red: NT / 64 floats
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const float o = __shfl_xor(v, s);
    v = is_max ? fmaxf(v, o) : v + o;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
  __syncthreads();
  return r;
}

// NT threads per sample: 128 for the usual class counts, 512 when C is large (1000 classes at 224x224: the logits,
// the softmax and the d_z2 sum are then 4x wider per pass).
template <int NT, bool VEC>
__global__ __launch_bounds__(NT) void tail_train_kernel(const float* __restrict__ part, int ksplit,
                                                        const float* __restrict__ b1, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, const float* __restrict__ w3,
                                                        const float* __restrict__ b3, float clip,
                                                        const int64_t* __restrict__ labels, float scale_over_b, int B,
                                                        int L2, int L3, int C, float* __restrict__ h1,
                                                        float* __restrict__ h2, float* __restrict__ logits,
                                                        float* __restrict__ sample_loss, float* __restrict__ d_logits,
                                                        float* __restrict__ d_z1, float* __restrict__ d_z2, Buckets bk,
                                                        const float* __restrict__ x, int L1, float* __restrict__ x_g,
                                                        float* __restrict__ d_z1_g) {
  extern __shared__ float lds[];  // h1 [L2] | h2 [L3] | logits / d_logits [C] | d_z2 [L3] | red [8]
  constexpr int S = NT / 32;      // class slices of the d_z2 sum
  __shared__ float part_s[S][32];
  float* h1s = lds;
  float* h2s = h1s + L2;
  float* lgs = h2s + L3;
  float* dz2s = lgs + C;
  float* red = dz2s + L3;
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  // Grouped mode (d_z1_g != NULL; bucketed stacks whose d_w1 product rides in nnue_ftm_backward_bucketed): one workgroup per
  // GROUPED row g.  It also leaves the operands of that product in grouped row order -- x_g[g] = x[b] (the
  // FeatureTransformer output) and d_z1_g[g] = d_z1[b] -- and a padding row leaves zeros in both.
  const int g = blockIdx.x;
  if (d_z1_g) {
    b = bk.rows[g];  // workgroup-uniform
    if (b < 0) {
      for (int i = tid; i < L1; i += NT) x_g[(size_t)g * L1 + i] = 0.0f;
      for (int i = tid; i < L2; i += NT) d_z1_g[(size_t)g * L2 + i] = 0.0f;
      return;
    }
    for (int i = tid * 4; i < L1; i += NT * 4)  // L1 % 4 == 0 (the rider's shapes); nothing below waits for this copy
      *reinterpret_cast<float4*>(x_g + (size_t)g * L1 + i) = *reinterpret_cast<const float4*>(x + (size_t)b * L1 + i);
  }
  if (bk.bucket) {  // this sample's layer stack
    const int kb = bk.bucket[b];
    b1 += (size_t)kb * L2; w2 += (size_t)kb * L3 * L2; b2 += (size_t)kb * L3; w3 += (size_t)kb * C * L3; b3 += (size_t)kb * C;
  }
  const int og = tid >> 2, part4 = tid & 3;
  // VEC (L2 % 4 == 0, L3 % 4 == 0, 16-byte aligned w2 / w3): a thread's share of a weight row is float4 runs, and the
  // first pass's runs are requested before the slab sum so that their latency hides behind it
  constexpr int kPre2 = 8, kPre3 = 2;
  float4 w2v[kPre2], w3v[kPre3];
  if constexpr (VEC) {
#pragma unroll
    for (int i = 0; i < kPre2; ++i) {
      const int k = 4 * (part4 + 4 * i);
      w2v[i] = (og < L3 && k < L2) ? *reinterpret_cast<const float4*>(w2 + (size_t)og * L2 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < kPre3; ++i) {
      const int k = 4 * (part4 + 4 * i);
      w3v[i] = (og < C && k < L3) ? *reinterpret_cast<const float4*>(w3 + (size_t)og * L3 + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float b2v = (VEC && og < L3) ? b2[og] : 0.0f, b3v = (VEC && og < C) ? b3[og] : 0.0f;
  const int64_t y = labels[b];
  for (int j = tid; j < L2; j += NT) {
    float z = b1[j];
    int s = 0;
    for (; s + 8 <= ksplit; s += 8) {  // eight loads in flight, summed in slab order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[((size_t)(s + u) * B + b) * L2 + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) z += v[u];
    }
    for (; s < ksplit; ++s) z += part[((size_t)s * B + b) * L2 + j];
    const float h = act_fn(z, clip);
    h1s[j] = h;
    h1[(size_t)b * L2 + j] = h;
  }
  __syncthreads();
  auto dot4 = [](const float4& w, const float4& h, float z) { return fmaf(w.w, h.w, fmaf(w.z, h.z, fmaf(w.y, h.y, fmaf(w.x, h.x, z)))); };
  for (int j0 = 0; j0 < L3; j0 += NT / 4) {
    const int j = j0 + og;
    float z = 0.f;
    if (j < L3) {
      const float* __restrict__ wr = w2 + (size_t)j * L2;
      if constexpr (VEC) {
        int k = 4 * part4;
        if (j0 == 0) {
#pragma unroll
          for (int i = 0; i < kPre2; ++i) {
            const int kk = 4 * (part4 + 4 * i);
            if (kk < L2) z = dot4(w2v[i], *reinterpret_cast<const float4*>(h1s + kk), z);
          }
          k += 16 * kPre2;
        }
        for (; k < L2; k += 16) z = dot4(*reinterpret_cast<const float4*>(wr + k), *reinterpret_cast<const float4*>(h1s + k), z);
      } else {
        for (int k = part4; k < L2; k += 4) z = fmaf(wr[k], h1s[k], z);
      }
    }
    z += __shfl_xor(z, 1);
    z += __shfl_xor(z, 2);
    if (j < L3 && part4 == 0) {
      const float h = act_fn(z + ((VEC && j0 == 0) ? b2v : b2[j]), clip);
      h2s[j] = h;
      h2[(size_t)b * L3 + j] = h;
    }
  }
  __syncthreads();
  for (int c0 = 0; c0 < C; c0 += NT / 4) {
    const int c = c0 + og;
    float z = 0.f;
    if (c < C) {
      const float* __restrict__ wr = w3 + (size_t)c * L3;
      if constexpr (VEC) {
        int k = 4 * part4;
        if (c0 == 0) {
#pragma unroll
          for (int i = 0; i < kPre3; ++i) {
            const int kk = 4 * (part4 + 4 * i);
            if (kk < L3) z = dot4(w3v[i], *reinterpret_cast<const float4*>(h2s + kk), z);
          }
          k += 16 * kPre3;
        }
        for (; k < L3; k += 16) z = dot4(*reinterpret_cast<const float4*>(wr + k), *reinterpret_cast<const float4*>(h2s + k), z);
      } else {
        for (int k = part4; k < L3; k += 4) z = fmaf(wr[k], h2s[k], z);
      }
    }
    z += __shfl_xor(z, 1);
    z += __shfl_xor(z, 2);
    if (c < C && part4 == 0) {
      z += (VEC && c0 == 0) ? b3v : b3[c];
      lgs[c] = z;
      logits[(size_t)b * C + c] = z;
    }
  }
  __syncthreads();
  // softmax cross-entropy of this sample and its gradient
  float mx = -INFINITY, se = 0.f;
  if (C <= 64) {  // every wave forms the same two reductions by itself: no block barriers
    const int lane = tid & 63;
    const float v = lane < C ? lgs[lane] : -INFINITY;
    mx = v;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
    se = lane < C ? expf(v - mx) : 0.0f;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) se += __shfl_xor(se, sh);
  } else {
    for (int c = tid; c < C; c += NT) mx = fmaxf(mx, lgs[c]);
    mx = block_reduce_nt<NT>(mx, true, red);
    for (int c = tid; c < C; c += NT) se += expf(lgs[c] - mx);
    se = block_reduce_nt<NT>(se, false, red);
  }
  const bool ok = y >= 0 && y < C;
  if (tid == 0) sample_loss[b] = ok ? (mx + logf(se)) - lgs[y] : 0.0f;
  const float inv = 1.0f / se;
  __syncthreads();  // every thread has read lgs[y] / the logits it needs before they are overwritten
  for (int c = tid; c < C; c += NT) {
    const float g = ok ? (expf(lgs[c] - mx) * inv - (c == y ? 1.0f : 0.0f)) * scale_over_b : 0.0f;
    lgs[c] = g;
    d_logits[(size_t)b * C + c] = g;
  }
  __syncthreads();
  // d_z2[j] = sum_c d_logits[c] w3[c][j]: 32 units x S class slices per pass, four independent chains per thread
  // (a single serial chain over C = 1000 classes cost 60 us of the 224x224 configuration's step)
  for (int j0 = 0; j0 < L3; j0 += 32) {
    const int j = j0 + (tid & 31), cp = tid >> 5;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < L3) {
      const float* __restrict__ wc = w3 + j;
      int c = cp;
      for (; c + 3 * S < C; c += 4 * S) {
        s0 = fmaf(lgs[c], wc[(size_t)c * L3], s0);
        s1 = fmaf(lgs[c + S], wc[(size_t)(c + S) * L3], s1);
        s2 = fmaf(lgs[c + 2 * S], wc[(size_t)(c + 2 * S) * L3], s2);
        s3 = fmaf(lgs[c + 3 * S], wc[(size_t)(c + 3 * S) * L3], s3);
      }
      for (; c < C; c += S) s0 = fmaf(lgs[c], wc[(size_t)c * L3], s0);
    }
    part_s[cp][tid & 31] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (cp == 0 && j < L3) {
      float q[S];
#pragma unroll
      for (int i = 0; i < S; ++i) q[i] = part_s[i][tid];
#pragma unroll
      for (int w = 1; w < S; w *= 2)  // fixed pairwise tree: (p0 + p1) + (p2 + p3) ...
#pragma unroll
        for (int i = 0; i + w < S; i += 2 * w) q[i] += q[i + w];
      const float v = q[0] * gate_fn(h2s[j], clip);
      dz2s[j] = v;
      d_z2[(size_t)b * L3 + j] = v;
    }
    __syncthreads();
  }
  for (int k = tid; k < L2; k += NT) {
    float s = 0.f;
    int j = 0;
    for (; j + 16 <= L3; j += 16) {  // sixteen rows in flight, summed in row order
      float w[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) w[u] = w2[(size_t)(j + u) * L2 + k];
#pragma unroll
      for (int u = 0; u < 16; ++u) s = fmaf(dz2s[j + u], w[u], s);
    }
    for (; j < L3; ++j) s = fmaf(dz2s[j], w2[(size_t)j * L2 + k], s);
    const float v = s * gate_fn(h1s[k], clip);
    d_z1[(size_t)b * L2 + k] = v;
    if (d_z1_g) d_z1_g[(size_t)g * L2 + k] = v;
  }
  __syncthreads();
}

// ------------------------------------------------------------------ shape policy (shared by scratch + launch)
struct ClsPlan {
  bool fwd_mfma, bww_mfma, bwx_mfma;
  int fwd_ksplit;            // K slabs of the forward partial sums
  int bww_ksplit, bww_klen;  // batch slabs of d_w1
};

int bucket_tiles(int B, int K) { return (B + 15) / 16 + K; }  // sum_k ceil(c_k / 16) <= B / 16 + K

ClsPlan make_plan(int B, int L1, int L2, int pairwise, int K = 1) {
  ClsPlan p{};
  p.fwd_mfma = (L1 % 32 == 0) && (L2 % 16 == 0);
  p.fwd_ksplit = 1;
  if (p.fwd_mfma) {
    // enough K slabs for >= ~1024 waves, each slab a multiple of 16 and (pairwise) inside one half
    const int tiles = (K > 1 ? bucket_tiles(B, K) : (B + 15) / 16) * ((L2 + 63) / 64);
    int ks = 1;
    while (tiles * ks < 1024 && L1 % (ks * 2 * 16) == 0 && L1 / (ks * 2) >= 64) ks *= 2;
    if (pairwise && ks == 1) ks = 2;
    static const int force_ks = [] { const char* e = getenv("NNUE_CLS_FWD_KSPLIT"); return e ? atoi(e) : 0; }();  // developer knob
    if (force_ks >= 2 && (force_ks & (force_ks - 1)) == 0 && L1 % (force_ks * 16) == 0 && L1 / force_ks >= 32) ks = force_ks;
    p.fwd_ksplit = ks;
  }
  p.bww_mfma = (L1 % 64 == 0) && (L2 % 32 == 0);
  p.bww_ksplit = 1;
  p.bww_klen = B;
  if (p.bww_mfma && K > 1) {
    // per bucket: slices of klen rows of its own segment; a bucket that holds every sample gets ks slices, one that
    // holds B / K of them one -- the number of non-empty slices (= slabs written and summed) is <= ks + K
    const int tiles = (L2 / 32) * (L1 / 64);
    int ks = 1;
    while (tiles * ks < 512 && B / (ks * 2) >= 64) ks *= 2;
    p.bww_ksplit = ks;
    p.bww_klen = (((B + ks - 1) / ks) + 15) / 16 * 16;
  } else if (p.bww_mfma) {
    const int tiles = (L2 / 32) * (L1 / 64);
    int ks = 1;
    while (tiles * ks < 1024 && B / (ks * 2) >= 32) ks *= 2;
    p.bww_ksplit = ks;
    p.bww_klen = ((B + ks - 1) / ks + 3) / 4 * 4;
  }
  p.bwx_mfma = (L1 % 32 == 0) && (L2 % 16 == 0);
  return p;
}

int64_t plan_scratch_floats(const ClsPlan& p, int B, int L1, int L2, int L3, int K = 1) {
  const int64_t fwd = (int64_t)p.fwd_ksplit * B * L2;
  const int64_t bwd = (int64_t)B * L2 + (int64_t)B * L3 + (p.bww_ksplit > 1 ? (int64_t)p.bww_ksplit * K * L2 * L1 : 0);
  return (fwd > bwd ? fwd : bwd) + 64;
}

// ---- bucket selector + grouping (one workgroup; B a few thousand at most matters for speed, any B is correct) -----
// bucket[b] = min(K-1, n[b] * K / (P + 1))  (P = flat ids of the map; P == 0: n[b] already IS the bucket, clamped)
// then a stable counting sort of the samples into bucket-homogeneous 16-row tiles.
#include "bucket_group.h"

constexpr int kGroupThreads = 1024;  // one pass over a batch of 1024: every phase of the kernel is a latency, not work
__global__ __launch_bounds__(kGroupThreads) void bucket_group_kernel(GroupArgs ga) {
  __shared__ int lds[kGroupLdsInts];
  bucket_group_body<kGroupThreads>(ga, lds);
}

Buckets no_buckets() { return Buckets{1, nullptr, nullptr, nullptr, nullptr, 0}; }

// host-side check + conversion of the C struct; K == 1 -> the reference path whatever the pointers say
int buckets_from(const nnue_buckets* c, int B, Buckets* out, const char* who) {
  *out = no_buckets();
  if (!c || c->K <= 1) return NNUE_OK;
  NNUE_REQUIRE(c->K <= kMaxBuckets, NNUE_E_SHAPE, "%s: %d layer stacks (at most %d)", who, c->K, kMaxBuckets);
  NNUE_REQUIRE(c->bucket && c->rows && c->tile_bucket && c->seg, NNUE_E_ARG, "%s: null bucket pointer", who);
  NNUE_REQUIRE(c->tiles == bucket_tiles(B, c->K), NNUE_E_SHAPE, "%s: bucket grouping was made for another batch (tiles %d, expected %d)", who,
               c->tiles, bucket_tiles(B, c->K));
  *out = Buckets{c->K, c->bucket, c->rows, c->tile_bucket, c->seg, c->tiles};
  return NNUE_OK;
}

}  // namespace

// =============================================================================== C ABI
namespace {

int forward_impl(const float* x, int pairwise, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                 const float* b3, float clip, int B, int L1, int L2, int L3, int C, float* h1, float* h2, float* logits, void* scratch,
                 int64_t scratch_bytes, const Buckets& bk, nnue_stream_t stream) {
  NNUE_REQUIRE(x && w1 && b1 && w2 && b2 && w3 && b3 && h1 && h2 && logits && scratch, NNUE_E_ARG,
               "nnue_classifier_forward: null pointer");
  NNUE_REQUIRE(B > 0 && L1 > 0 && L2 > 0 && L3 > 0 && C > 0, NNUE_E_ARG,
               "nnue_classifier_forward: B=%d L1=%d L2=%d L3=%d C=%d must be positive", B, L1, L2, L3, C);
  NNUE_REQUIRE(!pairwise || L1 % 2 == 0, NNUE_E_SHAPE, "nnue_classifier_forward: pairwise needs an even L1 (got %d)", L1);
  NNUE_REQUIRE((int64_t)(L2 + L3) * 4 <= 64 * 1024, NNUE_E_SHAPE, "nnue_classifier_forward: L2+L3 too large for the LDS tail");
  const ClsPlan p = make_plan(B, L1, L2, pairwise, bk.K);
  NNUE_REQUIRE(scratch_bytes >= plan_scratch_floats(p, B, L1, L2, L3, bk.K) * (int64_t)sizeof(float), NNUE_E_SCRATCH,
               "nnue_classifier_forward: scratch %lld bytes too small", (long long)scratch_bytes);
  NNUE_REQUIRE(nnue_aligned16(x) && nnue_aligned16(w1) && nnue_aligned16(scratch), NNUE_E_ARG,
               "nnue_classifier_forward: pointers must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* part = static_cast<float*>(scratch);
  const int m_tiles = bk.rows ? bk.tiles : (B + 15) / 16;
  if (p.fwd_mfma) {
    const long long waves = (long long)m_tiles * ((L2 + 63) / 64) * p.fwd_ksplit;
    hipLaunchKernelGGL(l1_forward_mfma, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, pairwise, w1, B, L1, L2,
                       p.fwd_ksplit, part, bk);
  } else {
    const long long waves = (long long)B * L2;
    hipLaunchKernelGGL(l1_forward_simple, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, pairwise, w1, B, L1, L2, part, bk);
  }
  hipLaunchKernelGGL(tail_forward_kernel, dim3(B), dim3(128), (size_t)(L2 + L3) * sizeof(float), s, part, p.fwd_ksplit, b1,
                     w2, b2, w3, b3, clip, B, L2, L3, C, h1, h2, logits, bk);
  return nnue_launch_status("nnue_classifier_forward");
}

int backward_impl(const float* x, int pairwise, const float* w1, const float* w2, const float* w3, float clip, const float* h1,
                  const float* h2, const float* d_logits, int B, int L1, int L2, int L3, int C, float* d_x, float* d_w1, float* d_b1,
                  float* d_w2, float* d_b2, float* d_w3, float* d_b3, void* scratch, int64_t scratch_bytes, const Buckets& bk,
                  nnue_stream_t stream) {
  NNUE_REQUIRE(x && w1 && w2 && w3 && h1 && h2 && d_logits && scratch, NNUE_E_ARG, "nnue_classifier_backward: null input pointer");
  NNUE_REQUIRE(d_w1 && d_b1 && d_w2 && d_b2 && d_w3 && d_b3, NNUE_E_ARG, "nnue_classifier_backward: null gradient pointer");
  NNUE_REQUIRE(B > 0 && L1 > 0 && L2 > 0 && L3 > 0 && C > 0, NNUE_E_ARG,
               "nnue_classifier_backward: B=%d L1=%d L2=%d L3=%d C=%d must be positive", B, L1, L2, L3, C);
  NNUE_REQUIRE(!pairwise || L1 % 2 == 0, NNUE_E_SHAPE, "nnue_classifier_backward: pairwise needs an even L1 (got %d)", L1);
  NNUE_REQUIRE((int64_t)(C + L3) * 4 <= 64 * 1024, NNUE_E_SHAPE, "nnue_classifier_backward: C+L3 too large for the LDS tail");
  const ClsPlan p = make_plan(B, L1, L2, pairwise, bk.K);
  NNUE_REQUIRE(scratch_bytes >= plan_scratch_floats(p, B, L1, L2, L3, bk.K) * (int64_t)sizeof(float), NNUE_E_SCRATCH,
               "nnue_classifier_backward: scratch %lld bytes too small", (long long)scratch_bytes);
  NNUE_REQUIRE(nnue_aligned16(x) && nnue_aligned16(w1) && nnue_aligned16(scratch) && nnue_aligned16(d_w1), NNUE_E_ARG,
               "nnue_classifier_backward: pointers must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int K = bk.K;
  const int m_tiles = bk.rows ? bk.tiles : (B + 15) / 16;
  float* d_z1 = static_cast<float*>(scratch);
  float* d_z2 = d_z1 + nnue_round_up((int64_t)B * L2, 4);
  float* slabs = d_z2 + nnue_round_up((int64_t)B * L3, 4);
  hipLaunchKernelGGL(tail_backward_kernel, dim3(B), dim3(128), (size_t)(C + L3) * sizeof(float), s, d_logits, h1, h2, w2, w3,
                     clip, L2, L3, C, d_z1, d_z2, bk);
  {
    const int tiles = small_wgrad_tiles(L2, L3, C) * K;  // no mean loss here
    const SmallWgrad a{d_logits, d_z2, d_z1, h1, h2, B, L2, L3, C, d_w3, d_b3, d_w2, d_b2, d_b1, nullptr, nullptr, tiles,
                       nullptr, 0, 0ll, nullptr, bk, p.bww_klen};
    hipLaunchKernelGGL(small_wgrad_kernel, dim3((unsigned)tiles), dim3(256), 0, s, a);
  }
  if (p.bww_mfma) {
    const long long waves = (long long)K * (L2 / 32) * (L1 / 64) * p.bww_ksplit;
    float* target = p.bww_ksplit > 1 ? slabs : d_w1;
    hipLaunchKernelGGL(l1_backward_w_mfma, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, pairwise, d_z1, B, L1, L2,
                       p.bww_ksplit, p.bww_klen, target, bk);
    if (p.bww_ksplit > 1) {
      const long long count = (long long)K * L2 * L1;
      hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, s, slabs, p.bww_ksplit,
                         count, d_w1, bk, p.bww_klen);
    }
  } else {
    hipLaunchKernelGGL(l1_backward_w_simple, dim3(K * L2, (L1 + 255) / 256), dim3(256), 0, s, x, pairwise, d_z1, B, L1, L2, d_w1, bk);
  }
  if (d_x) {
    if (p.bwx_mfma) {
      const long long waves = (long long)m_tiles * (L1 / 32);
      hipLaunchKernelGGL(l1_backward_x_mfma, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, pairwise, w1, d_z1, B, L1,
                         L2, d_x, bk);
    } else {
      const int cols = pairwise ? L1 / 2 : L1;
      hipLaunchKernelGGL(l1_backward_x_simple, dim3(B, (cols + 255) / 256), dim3(256), 0, s, x, pairwise, w1, d_z1, B, L1, L2,
                         d_x, bk);
    }
  }
  return nnue_launch_status("nnue_classifier_backward");
}

}  // namespace

extern "C" int64_t nnue_classifier_scratch(int B, int L1, int L2, int L3) { return nnue_classifier_scratch_bucketed(B, L1, L2, L3, 1); }

extern "C" int64_t nnue_classifier_scratch_bucketed(int B, int L1, int L2, int L3, int K) {
  if (B <= 0 || L1 <= 0 || L2 <= 0 || L3 <= 0 || K <= 0) return 0;
  const ClsPlan a = make_plan(B, L1, L2, 0, K), b = make_plan(B, L1, L2, 1, K);
  const int64_t fa = plan_scratch_floats(a, B, L1, L2, L3, K), fb = plan_scratch_floats(b, B, L1, L2, L3, K);
  return (fa > fb ? fa : fb) * (int64_t)sizeof(float);
}

extern "C" int nnue_bucket_tile_count(int B, int K) { return (B > 0 && K > 0) ? bucket_tiles(B, K) : 0; }

extern "C" int nnue_bucket_group(const int32_t* n, int B, int P, int K, int32_t* bucket, int32_t* rows, int32_t* tile_bucket,
                                 int32_t* seg, nnue_stream_t stream) {
  NNUE_REQUIRE(n && bucket && rows && tile_bucket && seg, NNUE_E_ARG, "nnue_bucket_group: null pointer");
  NNUE_REQUIRE(B > 0 && P >= 0 && K >= 1, NNUE_E_ARG, "nnue_bucket_group: B=%d P=%d K=%d out of range", B, P, K);
  NNUE_REQUIRE(K <= kMaxBuckets, NNUE_E_SHAPE, "nnue_bucket_group: %d layer stacks (at most %d)", K, kMaxBuckets);
  hipLaunchKernelGGL(bucket_group_kernel, dim3(1), dim3(kGroupThreads), 0, static_cast<hipStream_t>(stream),
                     GroupArgs{n, B, P, K, bucket, rows, tile_bucket, seg, bucket_tiles(B, K)});
  return nnue_launch_status("nnue_bucket_group");
}

extern "C" int nnue_classifier_forward(const float* x, int pairwise, const float* w1, const float* b1, const float* w2,
                                       const float* b2, const float* w3, const float* b3, float clip, int B, int L1,
                                       int L2, int L3, int C, float* h1, float* h2, float* logits, void* scratch,
                                       int64_t scratch_bytes, nnue_stream_t stream) {
  return forward_impl(x, pairwise, w1, b1, w2, b2, w3, b3, clip, B, L1, L2, L3, C, h1, h2, logits, scratch, scratch_bytes, no_buckets(),
                      stream);
}

extern "C" int nnue_classifier_forward_bucketed(const float* x, int pairwise, const float* w1, const float* b1, const float* w2,
                                                const float* b2, const float* w3, const float* b3, float clip, int B, int L1, int L2,
                                                int L3, int C, float* h1, float* h2, float* logits, void* scratch,
                                                int64_t scratch_bytes, const nnue_buckets* buckets, nnue_stream_t stream) {
  Buckets bk;
  const int rc = buckets_from(buckets, B, &bk, "nnue_classifier_forward_bucketed");
  if (rc != NNUE_OK) return rc;
  return forward_impl(x, pairwise, w1, b1, w2, b2, w3, b3, clip, B, L1, L2, L3, C, h1, h2, logits, scratch, scratch_bytes, bk, stream);
}

extern "C" int nnue_classifier_backward(const float* x, int pairwise, const float* w1, const float* w2, const float* w3,
                                        float clip, const float* h1, const float* h2, const float* d_logits, int B, int L1,
                                        int L2, int L3, int C, float* d_x, float* d_w1, float* d_b1, float* d_w2,
                                        float* d_b2, float* d_w3, float* d_b3, void* scratch, int64_t scratch_bytes,
                                        nnue_stream_t stream) {
  return backward_impl(x, pairwise, w1, w2, w3, clip, h1, h2, d_logits, B, L1, L2, L3, C, d_x, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, scratch,
                       scratch_bytes, no_buckets(), stream);
}

extern "C" int nnue_classifier_backward_bucketed(const float* x, int pairwise, const float* w1, const float* w2, const float* w3,
                                                 float clip, const float* h1, const float* h2, const float* d_logits, int B, int L1,
                                                 int L2, int L3, int C, float* d_x, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                                                 float* d_w3, float* d_b3, void* scratch, int64_t scratch_bytes,
                                                 const nnue_buckets* buckets, nnue_stream_t stream) {
  Buckets bk;
  const int rc = buckets_from(buckets, B, &bk, "nnue_classifier_backward_bucketed");
  if (rc != NNUE_OK) return rc;
  return backward_impl(x, pairwise, w1, w2, w3, clip, h1, h2, d_logits, B, L1, L2, L3, C, d_x, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, scratch,
                       scratch_bytes, bk, stream);
}

// d_x and the small weight/bias gradients + mean loss in one launch: both only read what the per-sample tail kernel
// left (d_z1 resp. d_logits, d_z2, h1, h2), so the two launch-sized jobs share the chip (x blocks first).
__global__ __launch_bounds__(256) void l1_backward_x_small_wgrad(const float* __restrict__ x, int pairwise, const float* __restrict__ w1,
                                                                 const float* __restrict__ d_z1, int B, int L1, int L2,
                                                                 float* __restrict__ d_x, int x_blocks, SmallWgrad a) {
  if ((int)blockIdx.x < x_blocks) l1_backward_x_body(x, pairwise, w1, d_z1, B, L1, L2, d_x, blockIdx.x, a.bk);
  else {
    __shared__ float red[1024];
    small_wgrad_body(a, (int)blockIdx.x - x_blocks, red);
  }
}

// ---------------------------------------------------------------- fused training step of the classifier block
namespace {
struct TrainLayout {
  int64_t part, d_z1, d_z2, d_logits, slabs, d_z1_g, x_g, total;  // float offsets
};
TrainLayout train_layout(const ClsPlan& p, int B, int L1, int L2, int L3, int C, int K = 1) {
  TrainLayout t{};
  int64_t off = 0;
  auto take = [&](int64_t n) { const int64_t o = off; off += nnue_round_up(n, 4); return o; };
  // the layer-1 slabs come from this module's own forward (fwd_ksplit of them) or from the fused FeatureTransformer
  // forward (one per 64 columns of L1, phases bit 8): room for whichever is more
  const int64_t ext = (L1 % 64 == 0) ? L1 / 64 : 0;
  t.part = take((p.fwd_ksplit > ext ? (int64_t)p.fwd_ksplit : ext) * B * L2);
  t.d_z1 = take((int64_t)B * L2);
  t.d_z2 = take((int64_t)B * L3);
  t.d_logits = take((int64_t)B * C);
  t.slabs = take(p.bww_ksplit > 1 ? (int64_t)p.bww_ksplit * K * L2 * L1 : 0);
  // bucketed stacks: grouped-row copies of d_z1 and of the block's input for the d_w1 product in nnue_ftm_backward_bucketed
  const int64_t grows = K > 1 ? (int64_t)bucket_tiles(B, K) * 16 : 0;
  t.d_z1_g = take(grows * L2);
  t.x_g = take(grows * L1);
  t.total = off + 4;
  return t;
}

int train_step_impl(const float* x, int pairwise, const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                    const float* b3, float clip, const int64_t* labels, float grad_scale, int B, int L1, int L2, int L3, int C, float* h1,
                    float* h2, float* logits, float* sample_loss, float* loss, float* d_x, float* d_w1, float* d_b1, float* d_w2,
                    float* d_b2, float* d_w3, float* d_b3, void* scratch, int64_t scratch_bytes, int phases, const Buckets& bk,
                    nnue_stream_t stream) {
  NNUE_REQUIRE(x && w1 && b1 && w2 && b2 && w3 && b3 && labels && h1 && h2 && logits && sample_loss && loss && scratch,
               NNUE_E_ARG, "nnue_classifier_train_step: null pointer");
  NNUE_REQUIRE(phases >= 1 && phases <= 63 && (phases & 3) && (phases & 20) != 20 && (!(phases & 32) || (phases & 19) == 19), NNUE_E_ARG,
               "nnue_classifier_train_step: phases = 1 (activations + d_x) | 2 (weight gradients + loss) [| 4: first-layer weight product beside "
               "d_x] [| 8: layer-1 slabs already at the start of scratch] [| 16 (not with 4): d_w1 comes from nnue_ftm_backward] [| 32 (with 1, 2 "
               "and 16): the small gradients and the mean loss ride in nnue_ftm_backward's launch as well]");
  const bool ext_dw1 = (phases & 16) != 0;
  const bool ext_small = (phases & 32) != 0;
  const bool ext_slabs = (phases & 8) != 0;
  const int K = bk.K;
  NNUE_REQUIRE(K == 1 || !ext_slabs, NNUE_E_ARG,
               "nnue_classifier_train_step: phases bit 8 (layer-1 slabs formed in the FeatureTransformer forward's epilogue) exists for one layer stack only");
  NNUE_REQUIRE(K == 1 || !ext_dw1 || (pairwise && L1 % 4 == 0), NNUE_E_SHAPE,
               "nnue_classifier_train_step: phases bit 16 with K > 1 needs the pairwise block and L1 %% 4 == 0");
  NNUE_REQUIRE(!ext_slabs || (pairwise && L1 % 64 == 0), NNUE_E_SHAPE,
               "nnue_classifier_train_step: phases bit 8 needs the pairwise block and L1 %% 64 == 0 (got L1=%d)", L1);
  NNUE_REQUIRE(d_w1 && d_b1 && d_w2 && d_b2 && d_w3 && d_b3, NNUE_E_ARG, "nnue_classifier_train_step: null gradient pointer");
  NNUE_REQUIRE(B > 0 && L1 > 0 && L2 > 0 && L3 > 0 && C > 0, NNUE_E_ARG,
               "nnue_classifier_train_step: B=%d L1=%d L2=%d L3=%d C=%d must be positive", B, L1, L2, L3, C);
  NNUE_REQUIRE(!pairwise || L1 % 2 == 0, NNUE_E_SHAPE, "nnue_classifier_train_step: pairwise needs an even L1 (got %d)", L1);
  const int64_t tail_lds = ((int64_t)L2 + 2 * L3 + C + 8) * 4;
  NNUE_REQUIRE(tail_lds <= 64 * 1024, NNUE_E_SHAPE, "nnue_classifier_train_step: L2+2*L3+C too large for the LDS tail");
  const ClsPlan p = make_plan(B, L1, L2, pairwise, K);
  const TrainLayout t = train_layout(p, B, L1, L2, L3, C, K);
  NNUE_REQUIRE(scratch_bytes >= t.total * (int64_t)sizeof(float), NNUE_E_SCRATCH,
               "nnue_classifier_train_step: scratch %lld < %lld bytes", (long long)scratch_bytes, (long long)(t.total * 4));
  NNUE_REQUIRE(nnue_aligned16(x) && nnue_aligned16(w1) && nnue_aligned16(scratch) && nnue_aligned16(d_w1) && nnue_aligned16(h1) &&
                   nnue_aligned16(h2),
               NNUE_E_ARG, "nnue_classifier_train_step: pointers must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* base = static_cast<float*>(scratch);
  float *part = base + t.part, *d_z1 = base + t.d_z1, *d_z2 = base + t.d_z2, *d_logits = base + t.d_logits, *slabs = base + t.slabs;
  // K > 1 with d_w1 left to nnue_ftm_backward_bucketed: the tail kernel runs per grouped row and leaves that product's operands
  const bool grouped = K > 1 && ext_dw1;
  float* d_z1_g = grouped ? base + t.d_z1_g : nullptr;
  float* x_g = grouped ? base + t.x_g : nullptr;
  const int tail_grid = grouped ? bk.tiles * 16 : B;
  const bool slab_pass = p.bww_mfma && p.bww_ksplit > 1;
  const int m_tiles = bk.rows ? bk.tiles : (B + 15) / 16;
  // bit 4: the d_w1 product runs in phase 1's d_x launch (both MFMA forms, d_x requested); a later phase-2 call with the
  // same bit then only sums its slabs
  const bool early_bww = (phases & 4) && p.bwx_mfma && p.bww_mfma && d_x != nullptr;
  const int tail_slabs = ext_slabs ? L1 / 64 : p.fwd_ksplit;
  // one launch: the small weight/bias gradients (per layer stack), the mean loss and (piggy-backed) the d_w1 slab sum
  const int wgrad_blocks = small_wgrad_tiles(L2, L3, C) * K + 1;  // one workgroup per (tile, stack) + the mean loss
  const long long count = (long long)K * L2 * L1;
  const int slab_blocks = slab_pass && !ext_dw1 ? (int)((count / 4 + 255) / 256) : 0;
  const SmallWgrad sw{d_logits, d_z2, d_z1, h1, h2, B, L2, L3, C, d_w3, d_b3, d_w2, d_b2, d_b1, sample_loss, loss, wgrad_blocks,
                      slabs, slab_pass && !ext_dw1 ? p.bww_ksplit : 0, count, d_w1, bk, p.bww_klen};
  // both phases in one call with d_w1 left to nnue_ftm_backward: the small gradients ride in the d_x launch
  const bool wgrad_rides = (phases & 3) == 3 && ext_dw1 && p.bwx_mfma && d_x != nullptr;
  NNUE_REQUIRE(!ext_small || wgrad_rides, NNUE_E_SHAPE,
               "nnue_classifier_train_step: phases bit 32 needs the d_x launch the small gradients otherwise ride in (MFMA shapes, d_x requested)");
  if (phases & 1) {
    if (ext_slabs) {
      // part[L1/64][B][L2] was written by nnue_ftm_forward_l1 (the FeatureTransformer forward's epilogue)
    } else if (p.fwd_mfma) {
      const long long waves = (long long)m_tiles * ((L2 + 63) / 64) * p.fwd_ksplit;
      hipLaunchKernelGGL(l1_forward_mfma, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, pairwise, w1, B, L1, L2, p.fwd_ksplit, part, bk);
    } else {
      const long long waves = (long long)B * L2;
      hipLaunchKernelGGL(l1_forward_simple, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, pairwise, w1, B, L1, L2, part, bk);
    }
    const bool vec = L2 % 4 == 0 && L3 % 4 == 0 && nnue_aligned16(w2) && nnue_aligned16(w3);
#define NNUE_TAIL(NT, V)                                                                                                                  \
  hipLaunchKernelGGL((tail_train_kernel<NT, V>), dim3(tail_grid), dim3(NT), (size_t)tail_lds, s, part, tail_slabs, b1, w2, b2, w3, b3, clip, \
                     labels, grad_scale / (float)B, B, L2, L3, C, h1, h2, logits, sample_loss, d_logits, d_z1, d_z2, bk, x, L1, x_g, d_z1_g)
    if (C > 256) { if (vec) NNUE_TAIL(512, true); else NNUE_TAIL(512, false); }
    else { if (vec) NNUE_TAIL(128, true); else NNUE_TAIL(128, false); }
#undef NNUE_TAIL
    if (d_x) {
      if (p.bwx_mfma && early_bww) {
        const long long xw = (long long)m_tiles * (L1 / 32), ww = (long long)K * (L2 / 32) * (L1 / 64) * p.bww_ksplit;
        const int w_blocks = (int)((ww + 3) / 4);
        hipLaunchKernelGGL(l1_backward_xw_mfma, dim3((unsigned)(w_blocks + (xw + 3) / 4)), dim3(256), 0, s, x, pairwise, w1, d_z1, B, L1, L2,
                           d_x, w_blocks, p.bww_ksplit, p.bww_klen, slab_pass ? slabs : d_w1, bk);
      } else if (wgrad_rides) {
        const long long waves = (long long)m_tiles * (L1 / 32);
        const int x_blocks = (int)((waves + 3) / 4);
        // timing-only ablation (tools/debug, WRONG small gradients): the d_x blocks alone -- what the launch would cost if the small
        // gradients rode elsewhere
#ifdef NNUE_ABLATIONS  // compiled only with NNUE_BUILD_ABLATIONS=1 (csrc/build.py)
        static const int skip_small = [] { const char* e = getenv("NNUE_CLS_ABL_SKIP_SMALL"); return e ? atoi(e) : 0; }();
#else
        constexpr int skip_small = 0;
#endif
        hipLaunchKernelGGL(l1_backward_x_small_wgrad, dim3((unsigned)(x_blocks + ((skip_small || ext_small) ? 0 : wgrad_blocks))), dim3(256), 0, s, x, pairwise, w1,
                           (const float*)d_z1, B, L1, L2, d_x, x_blocks, sw);
      } else if (p.bwx_mfma) {
        const long long waves = (long long)m_tiles * (L1 / 32);
        hipLaunchKernelGGL(l1_backward_x_mfma, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, pairwise, w1, d_z1, B, L1, L2, d_x, bk);
      } else {
        const int cols = pairwise ? L1 / 2 : L1;
        hipLaunchKernelGGL(l1_backward_x_simple, dim3(B, (cols + 255) / 256), dim3(256), 0, s, x, pairwise, w1, d_z1, B, L1, L2, d_x, bk);
      }
    }
  }
  if (wgrad_rides) return nnue_launch_status("nnue_classifier_train_step");
  if (phases & 2) {  // needs phase 1's h1, h2, d_logits, d_z1, d_z2 (scratch) -- nothing downstream depends on it
    if (early_bww || ext_dw1) {
      // the first-layer product already ran beside d_x (phases bit 4), or rides in nnue_ftm_backward's launch (bit 16)
    } else if (p.bww_mfma) {
      const long long waves = (long long)K * (L2 / 32) * (L1 / 64) * p.bww_ksplit;
      hipLaunchKernelGGL(l1_backward_w_mfma, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, pairwise, d_z1, B, L1, L2, p.bww_ksplit,
                         p.bww_klen, slab_pass ? slabs : d_w1, bk);
    } else {
      hipLaunchKernelGGL(l1_backward_w_simple, dim3(K * L2, (L1 + 255) / 256), dim3(256), 0, s, x, pairwise, d_z1, B, L1, L2, d_w1, bk);
    }
    hipLaunchKernelGGL(small_wgrad_kernel, dim3(wgrad_blocks + slab_blocks), dim3(256), 0, s, sw);
  }
  return nnue_launch_status("nnue_classifier_train_step");
}
}  // namespace

extern "C" int64_t nnue_classifier_train_scratch(int B, int L1, int L2, int L3, int C) {
  return nnue_classifier_train_scratch_bucketed(B, L1, L2, L3, C, 1);
}

extern "C" int64_t nnue_classifier_train_scratch_bucketed(int B, int L1, int L2, int L3, int C, int K) {
  if (B <= 0 || L1 <= 0 || L2 <= 0 || L3 <= 0 || C <= 0 || K <= 0) return 0;
  const int64_t a = train_layout(make_plan(B, L1, L2, 0, K), B, L1, L2, L3, C, K).total;
  const int64_t b = train_layout(make_plan(B, L1, L2, 1, K), B, L1, L2, L3, C, K).total;
  return (a > b ? a : b) * (int64_t)sizeof(float);
}

extern "C" int64_t nnue_classifier_train_dz1_offset(int B, int L1, int L2, int L3, int C, int pairwise) {
  if (B <= 0 || L1 <= 0 || L2 <= 0 || L3 <= 0 || C <= 0) return -1;
  return train_layout(make_plan(B, L1, L2, pairwise), B, L1, L2, L3, C).d_z1 * (int64_t)sizeof(float);
}

extern "C" int64_t nnue_classifier_train_dz1_grouped_offset(int B, int L1, int L2, int L3, int C, int K) {
  if (B <= 0 || L1 <= 0 || L2 <= 0 || L3 <= 0 || C <= 0 || K <= 1 || K > kMaxBuckets) return -1;
  return train_layout(make_plan(B, L1, L2, 1, K), B, L1, L2, L3, C, K).d_z1_g * (int64_t)sizeof(float);
}

extern "C" int64_t nnue_classifier_train_x_grouped_offset(int B, int L1, int L2, int L3, int C, int K) {
  if (B <= 0 || L1 <= 0 || L2 <= 0 || L3 <= 0 || C <= 0 || K <= 1 || K > kMaxBuckets) return -1;
  return train_layout(make_plan(B, L1, L2, 1, K), B, L1, L2, L3, C, K).x_g * (int64_t)sizeof(float);
}

// The arguments of the small-gradient tile family for a step run with phases bit 32 (host side only; what train_step_impl builds
// for its own launch, with no slab pass: d_w1 belongs to nnue_ftm_backward's rider in this mode)
extern "C" int nnue_classifier_train_rider(int pairwise, int B, int L1, int L2, int L3, int C, const float* h1, const float* h2,
                                           const float* sample_loss, float* loss, float* d_b1, float* d_w2, float* d_b2, float* d_w3,
                                           float* d_b3, void* scratch, int64_t scratch_bytes, const nnue_buckets* buckets,
                                           nnue_cls_rider* out) {
  NNUE_REQUIRE(h1 && h2 && sample_loss && loss && d_b1 && d_w2 && d_b2 && d_w3 && d_b3 && scratch && out, NNUE_E_ARG,
               "nnue_classifier_train_rider: null pointer");
  NNUE_REQUIRE(B > 0 && L1 > 0 && L2 > 0 && L3 > 0 && C > 0, NNUE_E_ARG, "nnue_classifier_train_rider: sizes must be positive");
  Buckets bk;
  const int rc = buckets_from(buckets, B, &bk, "nnue_classifier_train_rider");
  if (rc != NNUE_OK) return rc;
  const ClsPlan p = make_plan(B, L1, L2, pairwise, bk.K);
  const TrainLayout t = train_layout(p, B, L1, L2, L3, C, bk.K);
  NNUE_REQUIRE(scratch_bytes >= t.total * (int64_t)sizeof(float), NNUE_E_SCRATCH, "nnue_classifier_train_rider: scratch %lld < %lld bytes",
               (long long)scratch_bytes, (long long)(t.total * 4));
  float* base = static_cast<float*>(scratch);
  const int wgrad_blocks = small_wgrad_tiles(L2, L3, C) * bk.K + 1;
  const SmallWgrad sw{base + t.d_logits, base + t.d_z2, base + t.d_z1, h1, h2, B, L2, L3, C, d_w3, d_b3, d_w2, d_b2, d_b1, sample_loss, loss,
                      wgrad_blocks, nullptr, 0, (long long)bk.K * L2 * L1, nullptr, bk, p.bww_klen};
  static_assert(sizeof(SmallWgrad) <= sizeof(nnue_cls_rider), "nnue_cls_rider is too small");
  memset(out, 0, sizeof(*out));
  memcpy(out, &sw, sizeof(sw));
  return NNUE_OK;
}

extern "C" int nnue_classifier_train_step(const float* x, int pairwise, const float* w1, const float* b1, const float* w2,
                                          const float* b2, const float* w3, const float* b3, float clip,
                                          const int64_t* labels, float grad_scale, int B, int L1, int L2, int L3, int C,
                                          float* h1, float* h2, float* logits, float* sample_loss, float* loss, float* d_x,
                                          float* d_w1, float* d_b1, float* d_w2, float* d_b2, float* d_w3, float* d_b3,
                                          void* scratch, int64_t scratch_bytes, int phases, nnue_stream_t stream) {
  return train_step_impl(x, pairwise, w1, b1, w2, b2, w3, b3, clip, labels, grad_scale, B, L1, L2, L3, C, h1, h2, logits, sample_loss, loss,
                         d_x, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, scratch, scratch_bytes, phases, no_buckets(), stream);
}

extern "C" int nnue_classifier_train_step_bucketed(const float* x, int pairwise, const float* w1, const float* b1, const float* w2,
                                                   const float* b2, const float* w3, const float* b3, float clip, const int64_t* labels,
                                                   float grad_scale, int B, int L1, int L2, int L3, int C, float* h1, float* h2,
                                                   float* logits, float* sample_loss, float* loss, float* d_x, float* d_w1, float* d_b1,
                                                   float* d_w2, float* d_b2, float* d_w3, float* d_b3, void* scratch,
                                                   int64_t scratch_bytes, int phases, const nnue_buckets* buckets, nnue_stream_t stream) {
  Buckets bk;
  const int rc = buckets_from(buckets, B, &bk, "nnue_classifier_train_step_bucketed");
  if (rc != NNUE_OK) return rc;
  return train_step_impl(x, pairwise, w1, b1, w2, b2, w3, b3, clip, labels, grad_scale, B, L1, L2, L3, C, h1, h2, logits, sample_loss, loss,
                         d_x, d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, scratch, scratch_bytes, phases, bk, stream);
}
