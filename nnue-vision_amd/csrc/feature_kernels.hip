// Front end of the NNUE hot path on gfx950: 3x3 conv, straight-through binarisation with
// active-id compaction, and the backward that reaches the threshold and the conv weight.
// Reference arithmetic: nnue.py:486-493/:640 (conv), :19-54 (StraightThroughBinary),
// :590-635 (_to_sparse_features).
#include "common.h"

#include <cstdlib>

namespace {

constexpr float kSteSharpness = 10.0f;  // k, nnue.py:41
constexpr int kConvChunk = 8;           // output channels per register pass

// ------------------------------------------------------------------ conv forward
// One thread per output position (b, h, w); the 27-value input patch sits in registers and is
// reused for every output channel; weights are broadcast from LDS.  Each output is a single
// fmaf chain over (ci, kh, kw) in that order.
__global__ __launch_bounds__(256) void conv3x3_forward_kernel(const float* __restrict__ img,
                                                              const float* __restrict__ w,
                                                              float* __restrict__ out, int B, int H, int W,
                                                              int fps, int stride, int Gh, int Gw) {
  extern __shared__ float w_lds[];  // [fps][27]
  for (int i = threadIdx.x; i < fps * 27; i += blockDim.x) w_lds[i] = w[i];
  __syncthreads();
  const int G = Gh * Gw;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)B * G) return;
  const int b = (int)(t / G);
  const int hw = (int)(t - (long long)b * G);
  const int h = hw / Gw, x = hw - h * Gw;
  float patch[27];
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
  for (int c0 = 0; c0 < fps; c0 += kConvChunk) {
    float acc[kConvChunk];
#pragma unroll
    for (int u = 0; u < kConvChunk; ++u) acc[u] = 0.0f;
#pragma unroll
    for (int q = 0; q < 27; ++q)
#pragma unroll
      for (int u = 0; u < kConvChunk; ++u) {
        const int c = (c0 + u < fps) ? c0 + u : fps - 1;  // clamp keeps the LDS read in range
        acc[u] = fmaf(patch[q], w_lds[c * 27 + q], acc[u]);
      }
#pragma unroll
    for (int u = 0; u < kConvChunk; ++u)
      if (c0 + u < fps) out[((size_t)b * fps + c0 + u) * G + hw] = acc[u];
  }
}

// Conv forward + StraightThroughBinary.forward as the byte {0,1} map + per-sample counts in one launch (the front of the
// MFMA FeatureTransformer path): conv_binarize.h.  grid (B, slices).
#include "conv_binarize.h"

template <bool kFullUnroll, bool kPatch = false, bool kOut = true>
__global__ __launch_bounds__(256) void conv_binarize_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                            const float* __restrict__ thr, float* __restrict__ out,
                                                            uint8_t* __restrict__ bits, int* __restrict__ n,
                                                            float* __restrict__ sink, int H, int W, int fps, int stride,
                                                            int Gh, int Gw, int F, int slices, float* __restrict__ patches) {
  extern __shared__ __attribute__((aligned(16))) float w_lds[];
  ConvParamsPlain prm{w, thr};
  conv_binarize_body<kFullUnroll, kPatch, kOut>(img, prm, out, bits, n, sink, H, W, fps, stride, Gh, Gw, F, slices, (int)blockIdx.x, (int)blockIdx.y,
                                                w_lds, [] {}, patches, (size_t)gridDim.x * Gh * Gw);
}

// ------------------------------------------------------------------ binarise + compact
// One workgroup per sample walks the flat ids p = c*G + hw in ascending order, 256 at a time:
// bit = conv_out > thr[c]; wave ballots + a 4-entry LDS scan give each active id its slot, so the
// list comes out ascending (bit-exact ids).  The clamp sink (ids >= F-1 all hit table row F-1,
// nnue.py:701) is counted here and stored as coefT[F-1, b].
__global__ __launch_bounds__(256) void binarize_compact_kernel(const float* __restrict__ conv_out,
                                                               const float* __restrict__ thr, int G, int P, int F,
                                                               int* __restrict__ rows, int* __restrict__ pos,
                                                               float* __restrict__ coef, int* __restrict__ n,
                                                               float* __restrict__ coefT, int ldb) {
  __shared__ int wave_count[4];
  __shared__ int sink_count[4];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ x = conv_out + (size_t)b * P;
  const size_t base = (size_t)b * P;
  int written = 0, sink = 0;
  for (int p0 = 0; p0 < P; p0 += 256) {
    const int p = p0 + tid;
    const bool on = (p < P) && (x[p] > thr[p / G]);
    const unsigned long long m = __ballot(on);
    const unsigned long long ms = __ballot(on && p >= F - 1);
    if (lane == 0) {
      wave_count[wave] = __popcll(m);
      sink_count[wave] = __popcll(ms);
    }
    __syncthreads();
    int offset = written, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = wave_count[w];
      if (w < wave) offset += c;
      total += c;
      sink += sink_count[w];
    }
    if (on) {
      const int k = offset + __popcll(m & ((1ull << lane) - 1ull));
      rows[base + k] = p < F - 1 ? p : F - 1;
      pos[base + k] = p;
      coef[base + k] = 1.0f;
    }
    written += total;
    __syncthreads();
  }
  if (tid == 0) {
    n[b] = written;
    coefT[(size_t)(F - 1) * ldb + b] = (float)sink;
  }
}

// coefT rows 0..F-2 as a 64x64 LDS-tiled transpose of the bit map: reads are coalesced along the
// flat id, writes along the sample.  Rows >= P (table rows the map cannot reach) are zero.
__global__ __launch_bounds__(256) void binarize_transpose_kernel(const float* __restrict__ conv_out,
                                                                 const float* __restrict__ thr, int B, int G, int P,
                                                                 int F, float* __restrict__ coefT, int ldb) {
  __shared__ float tile[64][65];
  const int p0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int p = p0 + tx;
  const float t = (p < P) ? thr[p / G] : 0.0f;
  for (int j = ty; j < 64; j += 4) {
    const int b = b0 + j;
    const bool on = (b < B) && (p < P) && (conv_out[(size_t)b * P + p] > t);
    tile[j][tx] = on ? 1.0f : 0.0f;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const int f = p0 + j;
    if (f < F - 1 && b0 + tx < ldb) coefT[(size_t)f * ldb + b0 + tx] = tile[tx][j];
  }
}

// ------------------------------------------------------------------ STE + conv-weight backward
// Stage 1: grid (fps, chunks); a block owns channel c and a slice of the batch, every thread
// accumulates the 27 weight-gradient terms and the threshold term over its positions, then the
// block reduces (wave shuffles + LDS) to one 28-vector.  Stage 2 sums the chunk vectors in order.
// d_conv_out is zero at inactive positions (57 % at the CIFAR configs), which are skipped.
__global__ __launch_bounds__(256) void ste_conv_backward_stage1(const float* __restrict__ img,
                                                                const float* __restrict__ conv_out,
                                                                const float* __restrict__ thr,
                                                                const float* __restrict__ d_conv_out, int B, int H,
                                                                int W, int fps, int stride, int Gh, int Gw,
                                                                int samples_per_chunk, float* __restrict__ partial) {
  __shared__ float red[4][28];
  const int c = blockIdx.x;
  const int chunk = blockIdx.y;
  const int G = Gh * Gw;
  const int b_lo = chunk * samples_per_chunk;
  const int b_hi = min(B, b_lo + samples_per_chunk);
  const float t = thr[c];
  float acc[28];
#pragma unroll
  for (int q = 0; q < 28; ++q) acc[q] = 0.0f;
  const int work = (b_hi - b_lo) * G;
  for (int i = threadIdx.x; i < work; i += 256) {
    const int b = b_lo + i / G;
    const int hw = i - (i / G) * G;
    const size_t o = ((size_t)b * fps + c) * G + hw;
    const float d = d_conv_out[o];
    if (d == 0.0f) continue;
    const float s = 1.0f / (1.0f + __expf(-kSteSharpness * (conv_out[o] - t)));
    acc[27] = fmaf(d, (kSteSharpness * s) * (1.0f - s), acc[27]);
    const int h = hw / Gw, x = hw - h * Gw;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int iy = h * stride + kh - 1, ix = x * stride + kw - 1;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W)
            acc[ci * 9 + kh * 3 + kw] = fmaf(d, img[(((size_t)b * 3 + ci) * H + iy) * W + ix], acc[ci * 9 + kh * 3 + kw]);
        }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 28; ++q) {
    float v = acc[q];
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft);
    if (lane == 0) red[wave][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < 28)
    partial[((size_t)c * 28 + threadIdx.x) * gridDim.y + chunk] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Position-tiled variant for fps <= 64 (every shipped configuration): the conv weight gradient is a dense
// contraction dW[c][q] = sum_p d[p][c] * patch[p][q] over all B*Gh*Gw positions p, so it runs on the f32 MFMA
// (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate).  A block walks 64-position tiles: the d tile
// [fps x 64] and the im2col patch tile [27 x 64] are staged once into LDS (coalesced along positions; every
// pixel is read once per block instead of once per channel), the threshold term d * k * s * (1 - s) is
// accumulated per lane while d passes through registers, and each of the four waves contracts its 16 positions
// for all channel tiles.  K (positions) is permuted identically on both operands (float4 along positions, MFMA
// step t takes element t), which a sum does not care about.  One 28-vector per channel per block goes to stage 2.
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kStePos = 64;     // positions per tile
constexpr int kSteLd = 68;      // LDS row stride in floats: 16 B aligned, rows 4 banks apart -> conflict-free b128 reads
constexpr int kSteMaxBlocks = 1024;

// kPatch: the patch terms come from the im2col buffer nnue_ftm_conv_binarize_patches wrote (`img` = patches [27][B * G], coalesced
// along positions, instead of pixels gathered at the conv stride) and conv_out -- needed only inside the threshold term's
// sigmoid -- is re-formed from the staged patch tile with the forward's own fmaf chain (`conv_out` = the conv weights [fps][27]):
// bitwise the same numbers, 48 MB read per launch at the 224x224 shape instead of 160 MB.
template <int MT, bool kPatch = false, bool kReform = false>  // 16-channel tiles, fps <= 16 * MT; kReform: conv_out = the conv weights
__global__ __launch_bounds__(256) void ste_conv_backward_mfma(const float* __restrict__ img,
                                                              const float* __restrict__ conv_out,
                                                              const float* __restrict__ thr,
                                                              const float* __restrict__ d_conv_out, int B, int H, int W,
                                                              int fps, int stride, int Gh, int Gw, int tiles,
                                                              float* __restrict__ partial, int abl) {
  // staging tiles and the final cross-wave reduction buffer share LDS (the last tile ends with a barrier)
  constexpr int kStage = (MT * 16 + 32) * kSteLd, kRed = 4 * MT * 8 * 64;
  __shared__ __attribute__((aligned(16))) float smem[kStage > kRed ? kStage : kRed];
  float (*d_lds)[kSteLd] = reinterpret_cast<float (*)[kSteLd]>(smem);
  float (*p_lds)[kSteLd] = reinterpret_cast<float (*)[kSteLd]>(smem + MT * 16 * kSteLd);
  float (*red)[MT * 8][64] = reinterpret_cast<float (*)[MT * 8][64]>(smem);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int G = Gh * Gw;
  const int NP = B * G;
  // one tile's operands in registers: d and conv_out of this wave's channels, the wave-uniform patch terms
  // (Requesting the workgroup's NEXT tile into a second register set before the current one is staged was measured at the
  // 224x224 shape and loses: 40.6 vs 34.2 us from patches, 41.7 vs 39.0 us from pixels.)
  float dv[MT * 4], cvv[MT * 4], pv[7];
  auto load_tile = [&](int tile) {
    const int p = tile * kStePos + lane;
    const bool ok = p < NP;
    const int b = ok ? p / G : 0;
    const int hw = ok ? p - b * G : 0;
#pragma unroll
    for (int j = 0; j < MT * 4; ++j) {
      const int c = wave + 4 * j;
      const size_t o = ((size_t)b * fps + (c < fps ? c : 0)) * G + hw;
#ifdef NNUE_ABLATIONS
      if (abl & 2) { dv[j] = 1.0f; cvv[j] = 0.5f; continue; }
#endif
      dv[j] = (ok && c < fps) ? d_conv_out[o] : 0.0f;
      if constexpr (!kReform) cvv[j] = (ok && c < fps) ? conv_out[o] : 0.0f;
    }
    const int h = hw / Gw, x = hw - h * Gw;
#pragma unroll
    for (int rr = 0; rr < 7; ++rr) {
      const int qq = wave + 4 * rr;  // wave-uniform patch term
      const int qc = qq < 27 ? qq : 26;
#ifdef NNUE_ABLATIONS
      if (abl & 4) { pv[rr] = 0.25f; continue; }
#endif
      if constexpr (kPatch) {
        pv[rr] = (ok && qq < 27) ? img[(size_t)qc * NP + p] : 0.0f;
      } else {
        const int ci = qc / 9, kh = (qc - ci * 9) / 3, kw = qc - ci * 9 - kh * 3;
        const int iy = h * stride + kh - 1, ix = x * stride + kw - 1;
        const bool in = ok && qq < 27 && iy >= 0 && iy < H && ix >= 0 && ix < W;
        pv[rr] = in ? img[(((size_t)b * 3 + ci) * H + iy) * W + ix] : 0.0f;
      }
    }
  };
  // the first tile is requested before LDS is prepared: a launch starts with cold caches
  if ((int)blockIdx.x < tiles) load_tile(blockIdx.x);
  for (int i = threadIdx.x; i < MT * 16 * kSteLd; i += 256) (&d_lds[0][0])[i] = 0.0f;  // rows >= fps stay zero
  for (int i = threadIdx.x; i < 32 * kSteLd; i += 256) (&p_lds[0][0])[i] = 0.0f;       // rows >= 27 stay zero
  f32x4 acc[MT][2];
  float tacc[MT * 4];  // this wave stages channels wave + 4 j
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i][0] = acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < MT * 4; ++j) tacc[j] = 0.0f;
  __syncthreads();
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    if (tile != (int)blockIdx.x) load_tile(tile);
#pragma unroll
    for (int j = 0; j < MT * 4; ++j) {
      const int c = wave + 4 * j;
      if (c < fps) {
        if constexpr (!kReform) {
          const float s = 1.0f / (1.0f + __expf(-kSteSharpness * (cvv[j] - thr[c])));  // scalar load, cached
          tacc[j] = fmaf(dv[j], (kSteSharpness * s) * (1.0f - s), tacc[j]);
        }
        d_lds[c][lane] = dv[j];
      }
    }
#pragma unroll
    for (int rr = 0; rr < 7; ++rr) {
      const int qq = wave + 4 * rr;
      if (qq < 27) p_lds[qq][lane] = pv[rr];
    }
    __syncthreads();
    if constexpr (kReform) {
      // conv_out of this lane's position for the wave's channels: conv3x3's fmaf chain (q ascending from zero) over the staged
      // patch column; the weights are wave-uniform (scalar loads)
      float pq[27];
#pragma unroll
      for (int qq = 0; qq < 27; ++qq) pq[qq] = p_lds[qq][lane];
#pragma unroll
      for (int j = 0; j < MT * 4; ++j) {
        const int c = wave + 4 * j;
        if (c < fps) {
          const float* __restrict__ wc = conv_out + c * 27;
          float cv = 0.0f;
#pragma unroll
          for (int qq = 0; qq < 27; ++qq) cv = fmaf(pq[qq], wc[qq], cv);
          const float s = 1.0f / (1.0f + __expf(-kSteSharpness * (cv - thr[c])));
          tacc[j] = fmaf(dv[j], (kSteSharpness * s) * (1.0f - s), tacc[j]);
        }
      }
    }
    const int k0 = 16 * wave + 4 * q;
    const float4 b0 = *reinterpret_cast<const float4*>(&p_lds[r][k0]);
    const float4 b1 = *reinterpret_cast<const float4*>(&p_lds[16 + r][k0]);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float4 a = *reinterpret_cast<const float4*>(&d_lds[i * 16 + r][k0]);
      acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc[i][1], 0, 0, 0);
      acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc[i][1], 0, 0, 0);
      acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b0.z, acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b1.z, acc[i][1], 0, 0, 0);
      acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b0.w, acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b1.w, acc[i][1], 0, 0, 0);
    }
    __syncthreads();
  }
  // the four waves hold disjoint position slices of the same [channel x term] tiles: sum them in wave order
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][(i * 2 + t) * 4 + e][lane] = acc[i][t][e];
  __syncthreads();
  // partial[(channel * 28 + term) * blocks + block]: stage 2 reads each output's partials as one contiguous run
#ifdef NNUE_ABLATIONS
  if (abl & 1) return;  // timing only: no partial sums written
#endif
  // Slot of this workgroup in every output's run of partials: workgroups of one XCD (equal blockIdx % 8) take CONSECUTIVE slots,
  // so that the sixteen 4-byte stores that share a 64-byte line come from one L2 and leave it as one full line.  With slot =
  // blockIdx a line was written in parts from all eight L2s -- eight masked write-backs per line: 14 of the launch's 34.6 us at
  // the 224x224 shape (timing-only ablation without these stores: 20.3 us; profiles/r03v_ste_partials.txt).
  int slot;
  {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    slot = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
  }
  float* __restrict__ out = partial + slot;
  const size_t os = gridDim.x;
  for (int o = threadIdx.x; o < MT * 16 * 32; o += 256) {
    const int c = o >> 5, qq = o & 31;
    if (c >= fps || qq >= 27) continue;
    // accumulator register e of lane 16 * qd + rr holds D[row 4 * qd + e][col rr]
    const int i = c >> 4, row = c & 15, t = qq >> 4, rr = qq & 15;
    const int slot = (i * 2 + t) * 4 + (row & 3), ln = (row >> 2) * 16 + rr;
    out[(size_t)(c * 28 + qq) * os] = (red[0][slot][ln] + red[1][slot][ln]) + (red[2][slot][ln] + red[3][slot][ln]);
  }
#pragma unroll
  for (int j = 0; j < MT * 4; ++j) {
    float v = tacc[j];
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft);
    if (lane == 0 && wave + 4 * j < fps) out[(size_t)((wave + 4 * j) * 28 + 27) * os] = v;
  }
}

// one wave per (channel, term): lanes stride over the output's contiguous run of partials (eight loads in flight),
// fixed-shape tree sum
__global__ __launch_bounds__(256) void ste_conv_backward_stage2(const float* __restrict__ partial, int chunks, int fps,
                                                                float* __restrict__ d_thr, float* __restrict__ d_weight) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= fps * 28) return;
  const int c = o / 28, q = o - c * 28;
  const float* __restrict__ run = partial + (size_t)o * chunks;
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
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) acc += __shfl_xor(acc, sft);
  if (lane != 0) return;
  if (q == 27) {
    if (d_thr) d_thr[c] = -acc;
  } else if (d_weight) {
    d_weight[c * 27 + q] = acc;
  }
}

// about two workgroups per CU: enough to fill the chip, few enough that stage 2 stays trivial
int ste_chunks(int B, int fps) {
  int chunks = 512 / (fps > 0 ? fps : 1);
  if (chunks < 1) chunks = 1;
  if (chunks > B) chunks = B;
  return chunks;
}

// about two tiles per workgroup (stage 2 reads one 28-vector per channel per workgroup), at least one workgroup per
// CU once there is that much work, never more than kSteMaxBlocks
int64_t ste_mfma_blocks(int64_t positions) {
  const int64_t tiles = (positions + kStePos - 1) / kStePos;
  if (tiles <= kSteMaxBlocks) return tiles > 0 ? tiles : 1;  // one tile per workgroup: a single load latency, all resident
  int64_t blocks = (tiles + 1) / 2;
  if (blocks < 256) blocks = tiles < 256 ? tiles : 256;
  if (blocks > kSteMaxBlocks) blocks = kSteMaxBlocks;
  return blocks > 0 ? blocks : 1;
}

// ------------------------------------------------------------------ conv backward w.r.t. the pixels
// d_images[b][ci][y][x] = sum_{c,kh,kw} d_conv_out[b][c][oh][ow] * w[c][ci][kh][kw] over the taps with
// y = oh*stride + kh - 1, x = ow*stride + kw - 1 (transposed conv; the training loop never needs it, NNUE.forward's
// autograd node offers it for callers that differentiate w.r.t. the input).  Thread = one pixel, all three input
// channels; fixed order (kh, kw, c): reproducible.
__global__ __launch_bounds__(256) void conv3x3_backward_input_kernel(const float* __restrict__ d_out, const float* __restrict__ w,
                                                                     float* __restrict__ d_img, int B, int H, int W, int fps,
                                                                     int stride, int Gh, int Gw) {
  extern __shared__ float w_lds[];  // [fps][27]
  for (int i = threadIdx.x; i < fps * 27; i += blockDim.x) w_lds[i] = w[i];
  __syncthreads();
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)B * H * W) return;
  const int b = (int)(t / ((long long)H * W));
  const int yx = (int)(t - (long long)b * H * W);
  const int y = yx / W, x = yx - y * W;
  const int G = Gh * Gw;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ny = y + 1 - kh;
    if (ny < 0 || ny % stride) continue;
    const int oh = ny / stride;
    if (oh >= Gh) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int nx = x + 1 - kw;
      if (nx < 0 || nx % stride) continue;
      const int ow = nx / stride;
      if (ow >= Gw) continue;
      const float* __restrict__ d = d_out + (size_t)b * fps * G + oh * Gw + ow;
      const int q = kh * 3 + kw;
      for (int c = 0; c < fps; ++c) {
        const float g = d[(size_t)c * G];
        a0 = fmaf(g, w_lds[c * 27 + q], a0);
        a1 = fmaf(g, w_lds[c * 27 + 9 + q], a1);
        a2 = fmaf(g, w_lds[c * 27 + 18 + q], a2);
      }
    }
  }
  float* __restrict__ o = d_img + (size_t)b * 3 * H * W + yx;
  o[0] = a0;
  o[(size_t)H * W] = a1;
  o[(size_t)2 * H * W] = a2;
}

// ------------------------------------------------------------------ _to_sparse_features: values and their gradient
// val[b][i] = map[b][idx[b][i]] for idx >= 0, else 0 (nnue.py:628-633: the values stay attached to the map)
__global__ __launch_bounds__(256) void sparse_values_gather_kernel(const float* __restrict__ map, const int64_t* __restrict__ idx,
                                                                   long long total, int P, int M, float* __restrict__ val) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long long b = t / M;
  const int64_t p = idx[t];
  val[t] = (p >= 0 && p < P) ? map[b * P + p] : 0.0f;
}
// d_map[b][idx[b][i]] = d_val[b][i]; ids of one sample are distinct (they come from nonzero()), so plain stores
__global__ __launch_bounds__(256) void sparse_values_scatter_kernel(const float* __restrict__ d_val, const int64_t* __restrict__ idx,
                                                                    long long total, int P, int M, float* __restrict__ d_map) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const long long b = t / M;
  const int64_t p = idx[t];
  if (p >= 0 && p < P) d_map[b * P + p] = d_val[t];
}

}  // namespace

// =============================================================================== C ABI
extern "C" int nnue_conv3x3_forward(const float* images, const float* weight, float* conv_out, int B, int H, int W,
                                    int fps, int stride, nnue_stream_t stream) {
  NNUE_REQUIRE(images && weight && conv_out, NNUE_E_ARG, "nnue_conv3x3_forward: null pointer");
  NNUE_REQUIRE(B > 0 && H > 0 && W > 0 && fps > 0 && stride > 0, NNUE_E_ARG,
               "nnue_conv3x3_forward: B=%d H=%d W=%d fps=%d stride=%d must be positive", B, H, W, fps, stride);
  NNUE_REQUIRE(fps * 27 * 4 <= 64 * 1024, NNUE_E_SHAPE, "nnue_conv3x3_forward: fps=%d too large for the LDS weight tile", fps);
  const int Gh = (H - 1) / stride + 1, Gw = (W - 1) / stride + 1;
  const long long total = (long long)B * Gh * Gw;
  NNUE_REQUIRE(total < (1ll << 31) * 256, NNUE_E_SHAPE, "nnue_conv3x3_forward: too many outputs");
  hipLaunchKernelGGL(conv3x3_forward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), fps * 27 * sizeof(float),
                     static_cast<hipStream_t>(stream), images, weight, conv_out, B, H, W, fps, stride, Gh, Gw);
  return nnue_launch_status("nnue_conv3x3_forward");
}

namespace {
int conv_binarize_impl(const char* who, const float* images, const float* weight, const float* thr, int B, int H, int W, int fps, int stride, int F,
                       float* conv_out, float* patches, uint8_t* bits, int32_t* n, float* sink, nnue_stream_t stream) {
  NNUE_REQUIRE(images && weight && thr && (conv_out || patches) && bits && n && sink, NNUE_E_ARG, "%s: null pointer", who);
  NNUE_REQUIRE(B > 0 && H > 0 && W > 0 && fps > 0 && stride > 0 && F > 0, NNUE_E_ARG, "%s: B=%d H=%d W=%d fps=%d stride=%d F=%d must be positive", who,
               B, H, W, fps, stride, F);
  NNUE_REQUIRE(((fps + 7) & ~7) * 28 * 4 <= 64 * 1024, NNUE_E_SHAPE, "%s: fps=%d too large for the LDS weight tile", who, fps);
  const int Gh = (H - 1) / stride + 1, Gw = (W - 1) / stride + 1;
  const long long G = (long long)Gh * Gw;
  NNUE_REQUIRE(G * fps < (1ll << 30) && (long long)B * G * fps < (1ll << 40), NNUE_E_SHAPE, "%s: map too large", who);
  NNUE_REQUIRE(!patches || 27ll * B * G * 4 < (1ll << 31), NNUE_E_SHAPE, "%s: the im2col buffer must stay below 2 GiB", who);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int threads = G <= 64 ? 64 : (G <= 128 ? 128 : 256);
  // one workgroup per sample when the batch alone fills the chip; otherwise split samples (at least one position per
  // thread and slice) until there are about two workgroups per CU
  int slices = (512 + B - 1) / B;
  if (slices > (int)(G / threads)) slices = (int)(G / threads);
  slices = slices < 1 ? 1 : (slices > 16 ? 16 : slices);
  if (slices > 1) nnue_zero_counters(n, sink, B, s);  // a kernel, not a memset node (common.h)
  const size_t lds = (size_t)(((fps + 7) & ~7) * 28) * sizeof(float);
#define NNUE_CONV_LAUNCH(FULL, PATCH, OUT)                                                                                                      \
  hipLaunchKernelGGL((conv_binarize_kernel<FULL, PATCH, OUT>), dim3(B, slices), dim3(threads), lds, s, images, weight, thr, conv_out, bits, n, sink, H, \
                     W, fps, stride, Gh, Gw, F, slices, patches)
  if (fps <= 16) {
    if (!patches) NNUE_CONV_LAUNCH(true, false, true);
    else if (conv_out) NNUE_CONV_LAUNCH(true, true, true);
    else NNUE_CONV_LAUNCH(true, true, false);
  } else {
    if (!patches) NNUE_CONV_LAUNCH(false, false, true);
    else if (conv_out) NNUE_CONV_LAUNCH(false, true, true);
    else NNUE_CONV_LAUNCH(false, true, false);
  }
#undef NNUE_CONV_LAUNCH
  return nnue_launch_status(who);
}
}  // namespace

extern "C" int nnue_ftm_conv_binarize(const float* images, const float* weight, const float* thr, int B, int H, int W, int fps,
                                      int stride, int F, float* conv_out, uint8_t* bits, int32_t* n, float* sink,
                                      nnue_stream_t stream) {
  NNUE_REQUIRE(conv_out, NNUE_E_ARG, "nnue_ftm_conv_binarize: null pointer");
  return conv_binarize_impl("nnue_ftm_conv_binarize", images, weight, thr, B, H, W, fps, stride, F, conv_out, nullptr, bits, n, sink, stream);
}

extern "C" int nnue_ftm_conv_binarize_patches(const float* images, const float* weight, const float* thr, int B, int H, int W, int fps,
                                              int stride, int F, float* patches, float* conv_out, uint8_t* bits, int32_t* n, float* sink,
                                              nnue_stream_t stream) {
  NNUE_REQUIRE(patches, NNUE_E_ARG, "nnue_ftm_conv_binarize_patches: null pointer");
  return conv_binarize_impl("nnue_ftm_conv_binarize_patches", images, weight, thr, B, H, W, fps, stride, F, conv_out, patches, bits, n, sink, stream);
}

extern "C" int nnue_binarize_features(const float* conv_out, const float* thr, int B, int fps, int Gh, int Gw, int F,
                                      int32_t* rows, int32_t* pos, float* coef, int32_t* n, float* coefT, int ldb,
                                      nnue_stream_t stream) {
  NNUE_REQUIRE(conv_out && thr && rows && pos && coef && n && coefT, NNUE_E_ARG, "nnue_binarize_features: null pointer");
  NNUE_REQUIRE(B > 0 && fps > 0 && Gh > 0 && Gw > 0 && F > 0, NNUE_E_ARG,
               "nnue_binarize_features: B=%d fps=%d Gh=%d Gw=%d F=%d must be positive", B, fps, Gh, Gw, F);
  NNUE_REQUIRE(ldb >= B && ldb % 64 == 0, NNUE_E_SHAPE, "nnue_binarize_features: ldb=%d must be a multiple of 64 and >= B=%d", ldb, B);
  const long long P64 = (long long)fps * Gh * Gw;
  NNUE_REQUIRE(P64 < (1ll << 30), NNUE_E_SHAPE, "nnue_binarize_features: fps*Gh*Gw too large");
  const int P = (int)P64, G = Gh * Gw;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (F > 1)
    hipLaunchKernelGGL(binarize_transpose_kernel, dim3((F - 1 + 63) / 64, ldb / 64), dim3(256), 0, s, conv_out, thr, B, G,
                       P, F, coefT, ldb);
  hipLaunchKernelGGL(binarize_compact_kernel, dim3(B), dim3(256), 0, s, conv_out, thr, G, P, F, rows, pos, coef, n, coefT,
                     ldb);
  return nnue_launch_status("nnue_binarize_features");
}

extern "C" int64_t nnue_ste_conv_backward_scratch(int B, int fps, int Gh, int Gw) {
  if (B <= 0 || fps <= 0) return 0;
  int64_t blocks = ste_chunks(B, fps);
  if (fps <= 64 && Gh > 0 && Gw > 0) blocks = ste_mfma_blocks((int64_t)B * Gh * Gw);
  return blocks * fps * 28 * (int64_t)sizeof(float);
}

extern "C" int64_t nnue_ste_conv_backward_chunks(int B, int fps, int Gh, int Gw) {
  if (B <= 0 || fps <= 0 || Gh <= 0 || Gw <= 0) return 0;
  return fps <= 64 ? ste_mfma_blocks((int64_t)B * Gh * Gw) : ste_chunks(B, fps);
}

namespace {
// patches != NULL: the im2col form (images unused, `weight` needed to re-form conv_out); else pixels + conv_out
int ste_impl(const char* who, const float* images, const float* conv_out, const float* patches, const float* weight, const float* thr,
             const float* d_conv_out, int B, int H, int W, int fps, int stride, int Gh, int Gw, float* d_thr, float* d_weight, void* scratch,
             int64_t scratch_bytes, int stages, nnue_stream_t stream) {
  NNUE_REQUIRE(d_thr || d_weight, NNUE_E_ARG, "%s: both outputs are null", who);
  NNUE_REQUIRE(stages >= 1 && stages <= 3, NNUE_E_ARG, "%s: stages = 1 (partials) | 2 (final sums)", who);
  NNUE_REQUIRE(scratch_bytes >= nnue_ste_conv_backward_scratch(B, fps, Gh, Gw), NNUE_E_SCRATCH, "%s: scratch %lld < %lld bytes", who,
               (long long)scratch_bytes, (long long)nnue_ste_conv_backward_scratch(B, fps, Gh, Gw));
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* partial = static_cast<float*>(scratch);
  int chunks;
  if (fps <= 64) {
    const long long NP = (long long)B * Gh * Gw;
    NNUE_REQUIRE(NP < (1ll << 31) - 64 && (long long)B * fps * Gh * Gw < (1ll << 40), NNUE_E_SHAPE, "%s: too many positions", who);
    const int tiles = (int)((NP + kStePos - 1) / kStePos);
    chunks = (int)ste_mfma_blocks(NP);
    const char* abl_env = std::getenv("NNUE_STE_ABL");  // timing-only ablations (NNUE_ABLATIONS builds)
    const int abl = abl_env ? std::atoi(abl_env) : 0;
    if (stages & 1) {
#define NNUE_STE_LAUNCH(MT)                                                                                                          \
  do {                                                                                                                               \
    if (patches && conv_out)                                                                                                         \
      hipLaunchKernelGGL((ste_conv_backward_mfma<MT, true, false>), dim3(chunks), dim3(256), 0, s, patches, conv_out, thr, d_conv_out, B, H, W, \
                         fps, stride, Gh, Gw, tiles, partial, abl);                                                                  \
    else if (patches)                                                                                                                \
      hipLaunchKernelGGL((ste_conv_backward_mfma<MT, true, true>), dim3(chunks), dim3(256), 0, s, patches, weight, thr, d_conv_out, B, H, W, fps, \
                         stride, Gh, Gw, tiles, partial, abl);                                                                       \
    else                                                                                                                             \
      hipLaunchKernelGGL((ste_conv_backward_mfma<MT, false>), dim3(chunks), dim3(256), 0, s, images, conv_out, thr, d_conv_out, B, H, W, \
                         fps, stride, Gh, Gw, tiles, partial, abl);                                                                  \
  } while (0)
    switch ((fps + 15) / 16) {
      case 1: NNUE_STE_LAUNCH(1); break;
      case 2: NNUE_STE_LAUNCH(2); break;
      case 3: NNUE_STE_LAUNCH(3); break;
      default: NNUE_STE_LAUNCH(4); break;
    }
#undef NNUE_STE_LAUNCH
    }
  } else {
    NNUE_REQUIRE(!patches, NNUE_E_SHAPE, "%s: the im2col form needs fps <= 64 (fps = %d)", who, fps);
    chunks = ste_chunks(B, fps);
    const int spc = (B + chunks - 1) / chunks;
    NNUE_REQUIRE((long long)spc * Gh * Gw < (1ll << 31), NNUE_E_SHAPE, "%s: chunk too large", who);
    if (stages & 1)
      hipLaunchKernelGGL(ste_conv_backward_stage1, dim3(fps, chunks), dim3(256), 0, s, images, conv_out, thr, d_conv_out, B, H,
                         W, fps, stride, Gh, Gw, spc, partial);
  }
  if (stages & 2)
    hipLaunchKernelGGL(ste_conv_backward_stage2, dim3((fps * 28 + 3) / 4), dim3(256), 0, s, partial, chunks, fps, d_thr, d_weight);
  return nnue_launch_status(who);
}
}  // namespace

extern "C" int nnue_ste_conv_backward(const float* images, const float* conv_out, const float* thr,
                                      const float* d_conv_out, int B, int H, int W, int fps, int stride, float* d_thr,
                                      float* d_weight, void* scratch, int64_t scratch_bytes, int stages, nnue_stream_t stream) {
  NNUE_REQUIRE(images && conv_out && thr && d_conv_out && scratch, NNUE_E_ARG, "nnue_ste_conv_backward: null pointer");
  NNUE_REQUIRE(B > 0 && H > 0 && W > 0 && fps > 0 && stride > 0, NNUE_E_ARG,
               "nnue_ste_conv_backward: B=%d H=%d W=%d fps=%d stride=%d must be positive", B, H, W, fps, stride);
  const int Gh = (H - 1) / stride + 1, Gw = (W - 1) / stride + 1;
  return ste_impl("nnue_ste_conv_backward", images, conv_out, nullptr, nullptr, thr, d_conv_out, B, H, W, fps, stride, Gh, Gw, d_thr, d_weight,
                  scratch, scratch_bytes, stages, stream);
}

extern "C" int nnue_ste_conv_backward_patches(const float* patches, const float* weight, const float* conv_out, const float* thr,
                                              const float* d_conv_out, int B, int fps, int Gh, int Gw, float* d_thr, float* d_weight,
                                              void* scratch, int64_t scratch_bytes, int stages, nnue_stream_t stream) {
  NNUE_REQUIRE(patches && (weight || conv_out) && thr && d_conv_out && scratch, NNUE_E_ARG, "nnue_ste_conv_backward_patches: null pointer");
  NNUE_REQUIRE(B > 0 && fps > 0 && Gh > 0 && Gw > 0, NNUE_E_ARG, "nnue_ste_conv_backward_patches: B=%d fps=%d Gh=%d Gw=%d must be positive", B, fps,
               Gh, Gw);
  return ste_impl("nnue_ste_conv_backward_patches", nullptr, conv_out, patches, weight, thr, d_conv_out, B, 0, 0, fps, 1, Gh, Gw, d_thr, d_weight,
                  scratch, scratch_bytes, stages, stream);
}

extern "C" int nnue_conv3x3_backward_input(const float* d_conv_out, const float* weight, int B, int H, int W, int fps, int stride,
                                           float* d_images, nnue_stream_t stream) {
  NNUE_REQUIRE(d_conv_out && weight && d_images, NNUE_E_ARG, "nnue_conv3x3_backward_input: null pointer");
  NNUE_REQUIRE(B > 0 && H > 0 && W > 0 && fps > 0 && stride > 0, NNUE_E_ARG,
               "nnue_conv3x3_backward_input: B=%d H=%d W=%d fps=%d stride=%d must be positive", B, H, W, fps, stride);
  NNUE_REQUIRE(fps * 27 * 4 <= 64 * 1024, NNUE_E_SHAPE, "nnue_conv3x3_backward_input: fps=%d too large for the LDS weight tile", fps);
  const int Gh = (H - 1) / stride + 1, Gw = (W - 1) / stride + 1;
  const long long total = (long long)B * H * W;
  NNUE_REQUIRE(total < (1ll << 31) * 256, NNUE_E_SHAPE, "nnue_conv3x3_backward_input: too many pixels");
  hipLaunchKernelGGL(conv3x3_backward_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), fps * 27 * sizeof(float),
                     static_cast<hipStream_t>(stream), d_conv_out, weight, d_images, B, H, W, fps, stride, Gh, Gw);
  return nnue_launch_status("nnue_conv3x3_backward_input");
}

extern "C" int nnue_sparse_values(const float* map, const int64_t* idx, int B, int P, int M, float* val, nnue_stream_t stream) {
  NNUE_REQUIRE(map && idx && val, NNUE_E_ARG, "nnue_sparse_values: null pointer");
  NNUE_REQUIRE(B > 0 && P > 0 && M > 0, NNUE_E_ARG, "nnue_sparse_values: B=%d P=%d M=%d must be positive", B, P, M);
  const long long total = (long long)B * M;
  hipLaunchKernelGGL(sparse_values_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), map,
                     idx, total, P, M, val);
  return nnue_launch_status("nnue_sparse_values");
}

extern "C" int nnue_sparse_values_backward(const float* d_val, const int64_t* idx, int B, int P, int M, float* d_map,
                                           nnue_stream_t stream) {
  NNUE_REQUIRE(d_val && idx && d_map, NNUE_E_ARG, "nnue_sparse_values_backward: null pointer");
  NNUE_REQUIRE(B > 0 && P > 0 && M > 0, NNUE_E_ARG, "nnue_sparse_values_backward: B=%d P=%d M=%d must be positive", B, P, M);
  hipStream_t s = static_cast<hipStream_t>(stream);
  nnue_zero_floats(d_map, (size_t)B * P, s);
  const long long total = (long long)B * M;
  hipLaunchKernelGGL(sparse_values_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_val, idx, total, P, M, d_map);
  return nnue_launch_status("nnue_sparse_values_backward");
}
