// FeatureTransformer for the binary grid features of NNUE.forward as dense products on the f32 MFMA.
//
// Inside NNUE.forward the feature values are {0,1} and, at the reference's threshold, 43 % of them are set
// (n = 414 of 968 ids at 32x32, 28.1 k of 65.5 k at 224x224).  At that density "gather the active
// rows" (nnue.py:694-708) does not skip enough to beat streaming every row once: the three products
//     out      = A  W  + bias          A[b][f] = membership of table row f in sample b   (nnue.py:686-710)
//     d_weight = A^T d_out             (autograd of nnue.py:702-708; d_bias = column sums of d_out)
//     d_value  = (d_out W^T) . A       (autograd of nnue.py:705-707 + :628-633, masked to the active positions)
// are genuine matrix products, so they run on v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate --
// the same arithmetic as the gather kernels, different summation order).  A is the binary map itself, written
// once per step as a byte {0,1} matrix bits[B][P] by nnue_ftm_binarize (nnue.py:19-25) and widened to float while a
// tile is written to LDS (a quarter of the bytes through the L2 -> LDS fill path, which is what bounds the small
// shapes); no masks or id lists.
// Ids >= F-1 clamp to row F-1 (nnue.py:701): that row's membership is the count sink[b], a rank-one term the
// epilogues add (forward) or a separate column reduction forms (weight gradient); rows the map cannot reach get 0.
//
// One LDS-tiled kernel serves all three: block tile BM x BN, K tile BK, waves in a 2 x 2 grid, operands staged
// with 16-byte loads along whichever axis is contiguous in memory:
//   KC  source contiguous along k    -> LDS [row][k]   (16-byte chunks XOR-swizzled by row), fragment = one ds_read_b128
//   RC  source contiguous along rows -> LDS [k][row]   (stride BR+4), fragment = four ds_read_b32
// both conflict-free.  K is permuted identically on both operands (a lane supplies k = 4q..4q+3 of each 16-k
// block; MFMA step t takes element t), which a sum does not care about.  The next K tile's global loads are
// issued right after the barrier that publishes the current one and stay in flight during its MFMAs.
#include "common.h"
#include "ftv_kernels.h"
#include "small_wgrad.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

#include "bucket_group.h"  // the bucketed stacks' selector + grouping: rides as one extra workgroup of the forward launch

// Row-major matrix seen as (outer, inner), read through a buffer descriptor: 16-byte loads at (outer, inner..inner+3),
// hardware range check instead of branches -- rows past the end of the `bytes` window read as zero (that is how K
// and M tails along the outer index vanish).  Rows past `clamp` read row `clamp` (ids >= F-1 share table row F-1);
// along the inner index a K tail is zeroed by `inner_k` (KC operands whose partner is not zero there).  What a tile
// reads past N or M along the inner index is real neighbouring data; those outputs are never stored.
struct Mat {
  const void* p;   // float elements, or bytes for the binary map (operand A of forward / weight gradient)
  unsigned bytes;  // window of valid rows, in bytes (< 2^31)
  int ld, clamp, inner_k;
};
using u32x4 = __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned;

// The raw 16 bytes (float operand) or 4 bytes (byte operand, in .x) of (outer, inner .. inner+3).  Nothing is done to
// the value here: widening / splitting happens when the registers are stored to LDS, one K tile later, so the load
// stays in flight across the MFMA phase.  A K-tail mask moves the offset out of the window (range check -> zeros).
template <bool U8>
__device__ __forceinline__ u32x4 mat_load(__amdgpu_buffer_rsrc_t rsrc, const Mat& m, int outer, int inner) {
  const int row = outer < m.clamp ? outer : m.clamp;
  const int elem = row * m.ld + inner;
  if (U8) {
    u32x4 raw = {0u, 0u, 0u, 0u};
    raw[0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, inner < m.inner_k ? elem : 0x7ffffff0, 0, 0);
    return raw;
  }
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, inner < m.inner_k ? elem * 4 : 0x7ffffff0, 0, 0);
}

template <bool U8>
__device__ __forceinline__ float4 widen(const u32x4& raw) {
  if (U8) {
    const unsigned w = raw[0];
    return make_float4((float)(w & 0xffu), (float)((w >> 8) & 0xffu), (float)((w >> 16) & 0xffu), (float)(w >> 24));
  }
  return make_float4(__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]), __uint_as_float(raw[3]));
}

// ---- epilogues: col(n) gives per-column values, pre(m, n) a per-element operand loaded before any store ---------
struct FwdEpi {  // out = acc + bias + sink[b] * weight[F-1]   (or a split-K slab, finished by ftm_finish_kernel)
  static constexpr bool kAU8 = true;  // operand A is the byte map
  static constexpr bool kFusedL1 = false;
  static constexpr bool kBPair = false;
  static constexpr bool kSq = false;
  const float* __restrict__ bias;
  const float* __restrict__ w_last;  // weight row F-1
  const float* __restrict__ sink;
  float* __restrict__ out;  // ksplit == 1: out [B][L1]; else partial [ksplit][B][L1]
  int B, L1, ksplit;
  __device__ __forceinline__ float2 col(int n) const { return ksplit == 1 ? make_float2(bias[n], w_last[n]) : make_float2(0.f, 0.f); }
  __device__ __forceinline__ float pre(int m, int) const { return ksplit == 1 ? sink[m] : 0.0f; }
  __device__ __forceinline__ void store(int m, int n, float v, float2 c, float s, int ks) const {
    if (ksplit == 1) v += fmaf(s, c.y, c.x);
    out[((size_t)ks * B + m) * L1 + n] = v;
  }
};

struct BwwEpi {  // d_weight rows with a position of their own
  static constexpr bool kAU8 = true;
  static constexpr bool kFusedL1 = false;
  static constexpr bool kBPair = false;
  static constexpr bool kSq = true;  // can leave the sum of squares of its tile (gradient-norm partial)
  float* __restrict__ d_weight;
  int L1;
  float* __restrict__ sq;  // NULL or one float per tile: sum of the squares of the tile's stored elements
  __device__ __forceinline__ float2 col(int) const { return make_float2(0.f, 0.f); }
  __device__ __forceinline__ float pre(int, int) const { return 0.0f; }
  __device__ __forceinline__ void store(int m, int n, float v, float2, float, int) const { d_weight[(size_t)m * L1 + n] = v; }
};

// The weight gradient consumed where it is produced: clip_grad_norm_ + SGD(momentum, weight decay) (train.py:363-366,
// :457-464) applied to the table rows in place, element by element exactly as sgd_apply_kernel does -- d_weight is never
// written (268 MB written and read back per step at the 224x224 configuration).  The clip coefficient comes from the
// device scalar the optimizer's norm pass left (its table part formed without the gradient: nnue_ftm_gram_sqnorm).
struct BwwSgdEpi {
  static constexpr bool kAU8 = true;
  static constexpr bool kFusedL1 = false;
  static constexpr bool kBPair = false;
  static constexpr bool kSq = false;
  static constexpr bool kRmw = true;  // read-modify-write epilogue: goes through LDS so that rows are touched as whole 16-byte runs
  float* __restrict__ weight;    // table rows [0, direct), updated in place
  float* __restrict__ momentum;  // matching momentum rows or NULL
  const float* __restrict__ coef;  // clip coefficient (device scalar)
  int L1;
  float lr, mom, wd, scale;
  int first_step;
  int xcd_remap;  // 1: workgroups that share an XCD (equal blockIdx % 8) take consecutive tiles
  const float* __restrict__ lr_dev;  // when non-NULL the learning rate is read from this device float (a scheduler's hook: no re-capture)
  __device__ __forceinline__ float2 col(int) const { return make_float2(coef[0] * scale, 0.f); }
  __device__ __forceinline__ float pre(int m, int n) const { return weight[(size_t)m * L1 + n]; }
  __device__ __forceinline__ void store(int m, int n, float v, float2 c, float w, int) const {
    const size_t i = (size_t)m * L1 + n;
    float gi = fmaf(wd, w, v * c.x);
    if (momentum) {
      gi = first_step ? gi : fmaf(mom, momentum[i], gi);
      momentum[i] = gi;
    }
    weight[i] = w - (lr_dev ? lr_dev[0] : lr) * gi;
  }
};

struct ValEpi {  // d_conv_out = acc where the position is active, else 0
  static constexpr bool kAU8 = false;
  static constexpr bool kFusedL1 = false;
  static constexpr bool kBPair = false;
  static constexpr bool kSq = false;
  const uint8_t* __restrict__ bits;
  float* __restrict__ d_conv_out;
  int P;
  __device__ __forceinline__ float2 col(int) const { return make_float2(0.f, 0.f); }
  __device__ __forceinline__ float pre(int m, int p) const { return (float)bits[(size_t)m * P + p]; }
  __device__ __forceinline__ void store(int m, int p, float v, float2, float bit, int) const {
    d_conv_out[(size_t)m * P + p] = bit != 0.0f ? v : 0.0f;
  }
};

// Weight gradient of the classifier's first Linear, d_w1 = d_z1^T l0 (autograd of nnue.py:728-730 through the pairwise
// block nnue.py:660-666), as a third tile family of the merged backward launch: A = d_z1^T, B = l0 formed from ft while
// its tile is stored to LDS (l0[:, c] = ft[:, c] * ft[:, c + L1/2] for c < L1/2, ft[:, c - L1/2] above; a 64-column tile
// lies in one half because L1 % 128 == 0).
struct CwEpi {
  static constexpr bool kAU8 = false;
  static constexpr bool kFusedL1 = false;
  static constexpr bool kBPair = true;
  static constexpr bool kSq = false;
  float* __restrict__ d_w1;  // [L2][L1]
  int L1, half;
  __device__ __forceinline__ float2 col(int) const { return make_float2(0.f, 0.f); }
  __device__ __forceinline__ float pre(int, int) const { return 0.0f; }
  __device__ __forceinline__ void store(int m, int n, float v, float2, float, int) const { d_w1[(size_t)m * L1 + n] = v; }
};

// FeatureTransformer forward whose epilogue also forms this column tile's share of the classifier's first layer
// (pairwise block nnue.py:660-666 + Linear(L1, L2) nnue.py:728-730): tile t owns table columns 32t .. 32t+31 AND
// L1/2 + 32t .. L1/2 + 32t+31, i.e. both factors of 32 pairwise products, so it can write the slab
//   part[t][b][j] = sum_{c<32} ft[b][32t+c] * ft[b][L1/2+32t+c] * w1[j][32t+c]  +  ft[b][32t+c] * w1[j][L1/2+32t+c]
// that nnue_classifier_train_step otherwise gets from its own layer-1 launch (phases bit 8).
struct FwdL1Epi {
  static constexpr bool kAU8 = true;
  static constexpr bool kFusedL1 = true;
  static constexpr bool kBPair = false;
  const float* __restrict__ bias;
  const float* __restrict__ w_last;
  const float* __restrict__ sink;
  float* __restrict__ out;   // ft [B][L1]
  const float* __restrict__ w1;  // [L2][L1]
  float* __restrict__ part;  // [L1/64][B][L2]
  int B, L1, L2, half;
  // table / ft column of tile-local column index n_abs = 64 * tile + n_local
  __device__ __forceinline__ int col(int n_abs) const { return ((n_abs >> 6) << 5) + (n_abs & 31) + ((n_abs & 32) ? half : 0); }
};

// LDS floats one tile needs (both staged operands)
template <int BM, int BN, int BK, bool AKC, bool BKC>
constexpr int gemm_lds_floats() {
  return (AKC ? BM * BK : BK * (BM + 4)) + (BKC ? BN * BK : BK * (BN + 4));
}

// Epilogue of the fused forward (BN = 64, waves 2 x 2): ft tile -> memory and LDS, pairwise products in LDS, then the
// [BM x 64] x [64 x L2] product on the MFMA with the w1 fragments read straight from memory (L2-resident, 32 KB per
// workgroup).  smem: BM x 68 floats for the ft tile + BM x 68 for the l0 tile.
constexpr int kL1Ld = 68;
// Everything the epilogue reads from memory is requested before the K loop (FusedL1Pre), so that none of its
// latencies is left at the end of the tile: sink / bias / last table row, and the w1 fragments of this wave's first
// two 16-unit column tiles.
constexpr int kL1PreTiles = 2;
template <int BM>
struct FusedL1Pre {
  float sk[BM / 32][4];
  float bv[2], wl[2];
  float4 bq[kL1PreTiles][4];
};

template <int BM>
__device__ __forceinline__ void fused_l1_prefetch(const FwdL1Epi& e, FusedL1Pre<BM>& p, int m_base, int tile_n, int m0, int n0, int r,
                                                  int q, int wave) {
#pragma unroll
  for (int i = 0; i < BM / 32; ++i)
#pragma unroll
    for (int ee = 0; ee < 4; ++ee) {
      const int m = m_base + m0 + 16 * i + 4 * q + ee;
      p.sk[i][ee] = m < e.B ? e.sink[m] : 0.0f;
    }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int colg = e.col(tile_n * 64 + n0 + 16 * t + r);
    p.bv[t] = e.bias[colg];
    p.wl[t] = e.w_last[colg];
  }
#pragma unroll
  for (int s = 0; s < kL1PreTiles; ++s) {
    const int j = (wave + 4 * s) * 16 + r;
    const bool jok = j < e.L2;
    const float* __restrict__ wrow = e.w1 + (size_t)(jok ? j : 0) * e.L1;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int k = kb * 16 + 4 * q;  // tile-local l0 column; 4 consecutive columns stay inside one 32-column run
      const int colg = k < 32 ? tile_n * 32 + k : e.half + tile_n * 32 + (k - 32);
      p.bq[s][kb] = jok ? *reinterpret_cast<const float4*>(wrow + colg) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

template <int BM>
__device__ __forceinline__ void fused_l1_epilogue(const FwdL1Epi& e, const FusedL1Pre<BM>& p, float* __restrict__ smem,
                                                  const f32x4 (&acc)[BM / 32][2], int m_base, int tile_n, int m0, int n0, int r, int q,
                                                  int wave, int tid) {
  float* __restrict__ T = smem;
  float* __restrict__ L0 = smem + BM * kL1Ld;
  constexpr int TM = BM / 32;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n_loc = n0 + 16 * t + r;
    const int colg = e.col(tile_n * 64 + n_loc);
    const float bv = p.bv[t], wl = p.wl[t];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int ee = 0; ee < 4; ++ee) {
        const int m_loc = m0 + 16 * i + 4 * q + ee, m = m_base + m_loc;
        const float v = acc[i][t][ee] + fmaf(p.sk[i][ee], wl, bv);  // same arithmetic as FwdEpi
        if (m < e.B) e.out[(size_t)m * e.L1 + colg] = v;
        T[m_loc * kL1Ld + n_loc] = m < e.B ? v : 0.0f;
      }
  }
  __syncthreads();
  for (int idx = tid; idx < BM * 64; idx += 256) {
    const int m_loc = idx >> 6, c = idx & 63;
    const float* __restrict__ row = T + m_loc * kL1Ld;
    L0[m_loc * kL1Ld + c] = c < 32 ? row[c] * row[32 + c] : row[c - 32];
  }
  __syncthreads();
  const int n_tiles = (e.L2 + 15) / 16;
  auto column_tile = [&](int nt, const float4 (&bq)[4]) {
    const int j = nt * 16 + r;
    f32x4 z[BM / 16];
#pragma unroll
    for (int mt = 0; mt < BM / 16; ++mt) z[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int mt = 0; mt < BM / 16; ++mt) {
        const float4 a = *reinterpret_cast<const float4*>(&L0[(mt * 16 + r) * kL1Ld + kb * 16 + 4 * q]);
        z[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bq[kb].x, z[mt], 0, 0, 0);
        z[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bq[kb].y, z[mt], 0, 0, 0);
        z[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bq[kb].z, z[mt], 0, 0, 0);
        z[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bq[kb].w, z[mt], 0, 0, 0);
      }
    }
    if (j < e.L2) {
#pragma unroll
      for (int mt = 0; mt < BM / 16; ++mt)
#pragma unroll
        for (int ee = 0; ee < 4; ++ee) {
          const int m = m_base + mt * 16 + 4 * q + ee;
          if (m < e.B) e.part[((size_t)tile_n * e.B + m) * e.L2 + j] = z[mt][ee];
        }
    }
  };
#pragma unroll
  for (int s = 0; s < kL1PreTiles; ++s)
    if (wave + 4 * s < n_tiles) column_tile(wave + 4 * s, p.bq[s]);
  for (int nt = wave + 4 * kL1PreTiles; nt < n_tiles; nt += 4) {
    const int j = nt * 16 + r;
    const bool jok = j < e.L2;
    const float* __restrict__ wrow = e.w1 + (size_t)(jok ? j : 0) * e.L1;
    float4 bq[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int k = kb * 16 + 4 * q;
      const int colg = k < 32 ? tile_n * 32 + k : e.half + tile_n * 32 + (k - 32);
      bq[kb] = jok ? *reinterpret_cast<const float4*>(wrow + colg) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    column_tile(nt, bq);
  }
}

// Epilogue shared by the f32 and the bf16-split tiles (the C/D lane map of the MFMA is the same for both): accumulator
// register e of lane 16 q + r holds C[row 4 q + e][col r]; every epilogue operand is loaded before the first store so
// that the loads overlap.  `smem` is free when this runs (the K loop ended with a barrier).
template <int BM, int BN, class Epi>
__device__ __forceinline__ void store_tile(float* __restrict__ smem, const Epi& epi, const f32x4 (&acc)[BM / 32][BN / 32], int M, int N,
                                           int m_base, int n_base, int m0, int n0, int tile, int ks) {
  constexpr int TM = BM / 32, TN = BN / 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  float2 cv[TN];
  float pre[TM][TN][4];
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int n = n_base + n0 + 16 * t + r;
    cv[t] = n < N ? epi.col(n) : make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m_base + m0 + 16 * i + 4 * q + e;
        pre[i][t][e] = (m < M && n < N) ? epi.pre(m, n) : 0.0f;
      }
  }
  float sqacc = 0.0f;
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int n = n_base + n0 + 16 * t + r;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m_base + m0 + 16 * i + 4 * q + e;
        if (m < M && n < N) {
          epi.store(m, n, acc[i][t][e], cv[t], pre[i][t][e], ks);
          if constexpr (Epi::kSq) sqacc = fmaf(acc[i][t][e], acc[i][t][e], sqacc);
        }
      }
  }
  if constexpr (Epi::kSq) {
    if (epi.sq) {  // uniform; fixed order: lanes by butterfly, waves 0..3
#pragma unroll
      for (int sh = 32; sh >= 1; sh >>= 1) sqacc += __shfl_xor(sqacc, sh);
      if (lane == 0) smem[wave] = sqacc;
      __syncthreads();
      if (tid == 0) epi.sq[tile] = (smem[0] + smem[1]) + (smem[2] + smem[3]);
    }
  }
}

// streaming (non-temporal) 16-byte accesses: data touched once per step should not displace the L2's working set
__device__ __forceinline__ float4 nt_load4(const float* p) {
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store4(float* p, const float4& v) {
  __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p));
}

template <class Epi, class = void>
struct is_rmw { static constexpr bool value = false; };
template <class Epi>
struct is_rmw<Epi, decltype((void)Epi::kRmw)> { static constexpr bool value = Epi::kRmw; };

// Epilogue of a tile whose results update memory in place (BwwSgdEpi): the accumulators are 16 consecutive floats per
// lane group and row -- 64-byte pieces, a poor shape for a read-modify-write of two 268 MB arrays -- so the tile goes
// through LDS and every thread then owns whole float4 runs of a row: all of the tile's parameter and momentum loads are
// issued before the first result is combined (64 KB in flight per workgroup at 128 x 64).
// (Requesting the tile's parameters and momentum before the K loop instead -- 64 more live registers -- was measured and
// loses: 234 vs 200 us at the 224x224 shape with the 128-deep tiles, 191-202 vs 188 us with the 64-deep ones (228 registers);
// 64-row tiles, whose 160 / 108 registers allow three / four workgroups per CU with early / late loads: 197 / 200-204 vs 188 us.)
template <int BM, int BN, class Epi>
__device__ __forceinline__ void rmw_tile(float* __restrict__ smem, const Epi& epi, const f32x4 (&acc)[BM / 32][BN / 32], int M, int N,
                                         int m_base, int n_base, int m0, int n0) {
  constexpr int TM = BM / 32, TN = BN / 32, LD = BN + 4, PER = BM * BN / 4 / 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const float gs = epi.coef[0] * epi.scale;
  const float lr = epi.lr_dev ? epi.lr_dev[0] : epi.lr;
  float4 w[PER], mo[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int idx = tid + 256 * u, row = idx / (BN / 4), c4 = idx % (BN / 4);
    const int m = m_base + row, n = n_base + 4 * c4;
    const bool ok = m < M && n < N;
    const size_t i = (size_t)m * epi.L1 + n;
    w[u] = ok ? nt_load4(epi.weight + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    mo[u] = (ok && epi.momentum && !epi.first_step) ? nt_load4(epi.momentum + i) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) smem[(m0 + 16 * i + 4 * q + e) * LD + n0 + 16 * t + r] = acc[i][t][e];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int idx = tid + 256 * u, row = idx / (BN / 4), c4 = idx % (BN / 4);
    const int m = m_base + row, n = n_base + 4 * c4;
    if (m < M && n < N) {
      const float4 v = *reinterpret_cast<const float4*>(smem + row * LD + 4 * c4);
      const size_t i = (size_t)m * epi.L1 + n;
      float4 g;  // the arithmetic of sgd_apply_kernel, element by element
      g.x = fmaf(epi.wd, w[u].x, v.x * gs); g.y = fmaf(epi.wd, w[u].y, v.y * gs);
      g.z = fmaf(epi.wd, w[u].z, v.z * gs); g.w = fmaf(epi.wd, w[u].w, v.w * gs);
      if (epi.momentum) {
        if (!epi.first_step) {
          g.x = fmaf(epi.mom, mo[u].x, g.x); g.y = fmaf(epi.mom, mo[u].y, g.y);
          g.z = fmaf(epi.mom, mo[u].z, g.z); g.w = fmaf(epi.mom, mo[u].w, g.w);
        }
        nt_store4(epi.momentum + i, g);
      }
      nt_store4(epi.weight + i, make_float4(w[u].x - lr * g.x, w[u].y - lr * g.y, w[u].z - lr * g.z, w[u].w - lr * g.w));
    }
  }
}

// One BM x BN output tile (linear tile index `tile`, K slab `ks`) by the 256 threads of a workgroup; `smem` is the
// workgroup's LDS (gemm_lds_floats() floats, 16-byte aligned).
// The tile contracts k in [k_lo, k_hi); `ks` only names the split-K slab the epilogue stores to.
template <int BM, int BN, int BK, bool AKC, bool BKC, class Epi>
__device__ __forceinline__ void gemm_tile(float* __restrict__ smem, const Mat& ma, const Mat& mb, const Epi& epi, int M, int N, int k_lo,
                                          int k_hi, int tiles_n, int tile, int ks) {
  constexpr int LDA = AKC ? BK : BM + 4, LDB = BKC ? BK : BN + 4;
  float* __restrict__ As = smem;
  float* __restrict__ Bs = smem + (AKC ? BM : BK) * LDA;
  // KC image: unpadded rows with the 16-byte chunk index XOR-swizzled by the row, so that every 16-lane group of a
  // fragment ds_read_b128 (16 different rows, two neighbouring chunks) lands on 16 different 16-byte slots
  auto kc = [](int row, int k) { return row * BK + ((((k >> 2) ^ (BK == 32 ? (row >> 1) & 7 : row & 15))) << 2); };
  constexpr bool AU8 = Epi::kAU8;
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ma.p), 0, ma.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(mb.p), 0, mb.bytes, 0x00020000);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
  const int m_base = tile_m * BM, n_base = tile_n * BN;
  constexpr int AG = BM * BK / 1024, BG = BN * BK / 1024;  // float4 groups per thread
  constexpr int TM = BM / 32, TN = BN / 32;
  // staging coordinates of float4 group i: (row, k) with the memory-contiguous axis walked by consecutive lanes
  auto a_row = [&](int i) { const int g = tid + 256 * i; return AKC ? g / (BK / 4) : (g % (BM / 4)) * 4; };
  auto a_k = [&](int i) { const int g = tid + 256 * i; return AKC ? (g % (BK / 4)) * 4 : g / (BM / 4); };
  auto b_row = [&](int i) { const int g = tid + 256 * i; return BKC ? g / (BK / 4) : (g % (BN / 4)) * 4; };
  auto b_k = [&](int i) { const int g = tid + 256 * i; return BKC ? (g % (BK / 4)) * 4 : g / (BN / 4); };
  auto b_col = [&](int n_abs) {  // fused forward: a tile's 64 columns are two 32-column runs (see FwdL1Epi)
    if constexpr (Epi::kFusedL1) return epi.col(n_abs);
    else return n_abs;
  };
  // pairwise B operand (CwEpi): the tile's columns sit in the product half (two factors) or in the copy half
  bool b_prod = false;
  int b_shift = 0;
  if constexpr (Epi::kBPair) {
    b_prod = n_base < epi.half;
    b_shift = b_prod ? 0 : epi.half;
  }
  u32x4 ra[AG], rb[BG], rb2[Epi::kBPair ? BG : 1];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AG; ++i)
      ra[i] = AKC ? mat_load<AU8>(rsa, ma, m_base + a_row(i), k0 + a_k(i)) : mat_load<AU8>(rsa, ma, k0 + a_k(i), m_base + a_row(i));
#pragma unroll
    for (int i = 0; i < BG; ++i)
    {
      rb[i] = BKC ? mat_load<false>(rsb, mb, n_base + b_row(i), k0 + b_k(i))
                  : mat_load<false>(rsb, mb, k0 + b_k(i), b_col(n_base + b_row(i)) - b_shift);
      if constexpr (Epi::kBPair)
        if (b_prod) rb2[i] = mat_load<false>(rsb, mb, k0 + b_k(i), n_base + b_row(i) + epi.half);
    }
  };
  const int m0 = (wave >> 1) * (BM / 2), n0 = (wave & 1) * (BN / 2);
  // fragments of the 16-k block at kb: a lane supplies k = kb + 4q .. 4q+3 of row r of each of its tiles
  auto frags = [&](int kb, float4 (&a)[TM], float4 (&b)[TN]) {
    const int k = kb + 4 * q;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int x = m0 + 16 * i + r;
      if (AKC) a[i] = *reinterpret_cast<const float4*>(&As[kc(x, k)]);
      else a[i] = make_float4(As[k * LDA + x], As[(k + 1) * LDA + x], As[(k + 2) * LDA + x], As[(k + 3) * LDA + x]);
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int x = n0 + 16 * t + r;
      if (BKC) b[t] = *reinterpret_cast<const float4*>(&Bs[kc(x, k)]);
      else b[t] = make_float4(Bs[k * LDB + x], Bs[(k + 1) * LDB + x], Bs[(k + 2) * LDB + x], Bs[(k + 3) * LDB + x]);
    }
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  FusedL1Pre<Epi::kFusedL1 ? BM : 32> l1pre;
  if constexpr (Epi::kFusedL1) fused_l1_prefetch<BM>(epi, l1pre, m_base, tile_n, m0, n0, r, q, wave);
  // (Two K tiles of loads in flight -- two register sets, the loop written out twice, every load unconditional with a dead request's
  // offset outside its window, no branch between the halves, scheduling barriers between the sets' requests: each of these was
  // needed before the compiler's waits became vmcnt(12) instead of vmcnt(0), which round 2's attempt never reached -- was measured at
  // the CIFAR batch-512 forward with the waits verified in the ISA: 20.3-20.4 us against 18.5-18.6 us.  Per-tile load latency is
  // not what bounds the launch-sized products; the variant was removed again, the refactoring it needed cost the shipped loop 0.4 us.)
  // (Rotating the K loop per tile -- tile t starts at K tile (5 t) mod n and wraps, so that the workgroups of a launch do not all
  // walk the same 256-byte columns of the shared operands at the same time -- was measured at the CIFAR shapes and changes nothing:
  // forward 18.1 vs 18.2 us, merged backward 25.7 vs 25.4 us; profiles/r03n_krot_ab.txt.)
  fetch(k_lo);
  for (int k0 = k_lo; k0 < k_hi; k0 += BK) {
#pragma unroll
    for (int i = 0; i < AG; ++i) *reinterpret_cast<float4*>(&As[AKC ? kc(a_row(i), a_k(i)) : a_k(i) * LDA + a_row(i)]) = widen<AU8>(ra[i]);
#pragma unroll
    for (int i = 0; i < BG; ++i) {
      float4 v = widen<false>(rb[i]);
      if constexpr (Epi::kBPair)
        if (b_prod) {
          const float4 w = widen<false>(rb2[i]);
          v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w;
        }
      *reinterpret_cast<float4*>(&Bs[BKC ? kc(b_row(i), b_k(i)) : b_k(i) * LDB + b_row(i)]) = v;
    }
    __syncthreads();
    if (k0 + BK < k_hi) fetch(k0 + BK);
    float4 a[2][TM], b[2][TN];
    frags(0, a[0], b[0]);
#pragma unroll
    for (int kb = 0; kb < BK; kb += 16) {
      const int cur = (kb >> 4) & 1;
      // the next block's LDS reads are issued before this block's MFMAs and land while they run
      if (kb + 16 < BK) frags(kb + 16, a[cur ^ 1], b[cur ^ 1]);
      __builtin_amdgcn_sched_barrier(0);
      // consecutive MFMAs go to different accumulators
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][i].x, b[cur][t].x, acc[i][t], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][i].y, b[cur][t].y, acc[i][t], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][i].z, b[cur][t].z, acc[i][t], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][i].w, b[cur][t].w, acc[i][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  if constexpr (Epi::kFusedL1) {
    fused_l1_epilogue<BM>(epi, l1pre, smem, acc, m_base, tile_n, m0, n0, r, q, wave, tid);
    return;
  } else {
    store_tile<BM, BN, Epi>(smem, epi, acc, M, N, m_base, n_base, m0, n0, tile, ks);
  }
}

// ---- the two products whose A operand is the binary map, on the bf16 matrix unit -----------------------------------
// out = A W and d_W = A^T d_out multiply a {0,1} matrix -- exact in bf16 -- by an f32 one.  A float splits EXACTLY into
// three bf16 terms by truncation (hi = top 8 significant bits, mid = top 8 bits of the remainder, lo = what is left: at
// most 8 bits, so nothing is dropped): x = hi + mid + lo identically (for |x| >= 2^-103; below that lo, then mid, is a bf16
// denormal, which the matrix unit flushes: an absolute error below 2^-126, tests/test_gpu_ftm.py).  Every product 1 * term is exact, so three
// v_mfma_f32_16x16x32_bf16 (f32 accumulate) over the three planes compute the same sum of the same numbers as the f32
// MFMA -- in another order, like every other path here -- at 3 x 16 cycles per 32 k against 8 x 32 cycles for
// v_mfma_f32_16x16x4_f32 (MI355X_MICROARCH.md: the f32-input MFMA runs at 1/16 of the bf16 rate).
//
// Tile BM x BN x 128, waves 2 x 2.  Both operands sit in LDS as [row][k] bf16 images (256-byte rows, 16-byte chunk index
// XOR-swizzled by the row -- the layout the f32 kernel's KC images use, conflict-free for ds_read_b128): A one image, the
// f32 operand three (planes).  The f32 operand is contiguous along its rows in memory (table rows in the forward, d_out
// rows in the weight gradient), i.e. along n, while the MFMA wants 8 consecutive k per lane: a thread therefore owns an
// 8 k x 4 n block (eight 16-byte loads), splits it in registers and writes, per n, one 16-byte chunk of 8 k per plane.
// The split is 4 VALU per value + 1.5 for packing; it is amortised over the BM rows of the tile, so tall tiles matter
// more here than for the f32 kernel.
using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
constexpr int kBfK = 128;  // K tile
template <int BM, int BN>
constexpr int gemm_bf_lds_bytes() { return (BM + 3 * BN) * kBfK * 2; }

__device__ __forceinline__ int bf_img(int row, int chunk) { return row * (kBfK * 2) + ((chunk ^ (row & 15)) << 4); }  // byte offset

// four {0,1} bytes -> four bf16 (two words)
__device__ __forceinline__ void bytes_to_bf16(unsigned x, unsigned& w0, unsigned& w1) {
  w0 = ((x & 0xffu) | ((x & 0xff00u) << 8)) * 0x3f80u;
  w1 = (((x >> 16) & 0xffu) | ((x >> 8) & 0xff0000u)) * 0x3f80u;
}

template <int BM, int BN, bool AKC, class Epi>
__device__ __forceinline__ void gemm_tile_bf(unsigned char* __restrict__ smem, const Mat& ma, const Mat& mb, const Epi& epi, int M, int N,
                                             int k_lo, int k_hi, int tiles_n, int tile, int ks) {
  static_assert(BN == 64, "the f32 operand's staging assigns one 8 k x 4 n block per thread: 128 x 64 per K tile");
  static_assert(BM == 32 || BM == 64 || BM == 128, "tile heights");
  constexpr int TM = BM / 32, TN = BN / 32;
  unsigned char* __restrict__ As = smem;
  unsigned char* __restrict__ Bs = smem + BM * kBfK * 2;  // plane p at + p * BN * 256
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ma.p), 0, ma.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(mb.p), 0, mb.bytes, 0x00020000);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
  const int m_base = tile_m * BM, n_base = tile_n * BN;
  const int m0 = (wave >> 1) * (BM / 2), n0 = (wave & 1) * (BN / 2);
  auto b_col = [&](int n_abs) {
    if constexpr (Epi::kFusedL1) return epi.col(n_abs);
    else return n_abs;
  };
  // ---- staging coordinates
  // f32 operand: thread = (k block of 8, n block of 4).  Inside a 16-lane group the lanes take 4 n blocks x 4 k blocks, so
  // that their 16-byte LDS stores (row n, chunk k/8, chunk XOR-swizzled by the row) fall into 16 different 16-byte slots:
  // with 16 n blocks x 1 k block per group the rows 4 j + e repeat modulo 16 and the stores were 4-way bank-conflicted.
  // A wave still reads 256 contiguous bytes of each of its k rows.
  // (ds_write_b128 is served in groups of 8 consecutive lanes over a 128-byte window: those take 2 n blocks x 4 k blocks, rows
  // r and r + 4 with four consecutive chunks each -- eight different slots; 4 n blocks x 2 k blocks put rows r and r + 8 on
  // the same slot: a third of the 64-row forward's LDS cycles were conflicts)
  const int bn4 = ((lane & 1) | ((lane >> 3) << 1)) * 4, bk8 = (((lane >> 1) & 3) | (wave << 2)) * 8;
  const int b_off = b_col(n_base + bn4) * 4;  // byte offset inside a row
  // A, forward (bytes contiguous along k): 16-byte groups, (row, 16 k); BM * 8 groups
  // A, weight gradient (bytes contiguous along m): thread = (k block of 8, m block of 4); BM / 4 x 16 blocks
  const bool stream_b = mb.bytes > (64u << 20);  // uniform
  // (same 4 x 4 arrangement inside a 16-lane group as for the f32 operand: block index -> (m block, k block))
  auto a_block = [&](int g) {
    constexpr int MB = BM / 4;  // m blocks per k block
    const int grp = g >> 4, l = g & 15;  // 16 consecutive blocks = 4 m blocks x 4 k blocks
    const int groups_m = MB / 4;         // groups along m per 4 k blocks
    const int gm = grp % groups_m, gk = grp / groups_m;
    return ((gk * 4 + ((l >> 1) & 3)) * MB) + gm * 4 + ((l & 1) | ((l >> 3) << 1));  // 8 lanes: 2 m blocks x 4 k blocks (see above)
  };
  // forward: 16 consecutive groups are two rows x eight 16-k groups, rows alternating lane by lane, so that the eight lanes of a
  // ds_write_b128 group hold rows r, r + 1 with chunks {0, 2, 4, 6} (+1 for the second store): even and odd slots
  auto akc_row = [](int g) { return ((g >> 4) << 1) | (g & 1); };
  auto akc_grp = [](int g) { return (g >> 1) & 7; };
  constexpr int AGK = BM * 8 / 256;                 // forward: 16-byte loads per thread
  constexpr int AGR = (BM / 4) * 16 / 256 ? (BM / 4) * 16 / 256 : 1;  // weight gradient: 8 x 4 blocks per thread (BM = 32: half the threads)
  struct Regs {
    u32x4 ra[AKC ? AGK : 1];
    unsigned rat[AKC ? 1 : AGR][8];
    u32x4 rb[8];
  };
  Regs set0;
  auto fetch = [&](int k0, Regs& R) {
    auto& ra = R.ra; auto& rat = R.rat; auto& rb = R.rb;
    if constexpr (AKC) {
#pragma unroll
      for (int i = 0; i < AGK; ++i) {
        const int g = tid + 256 * i, row = akc_row(g), k = k0 + akc_grp(g) * 16;
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsa, (m_base + row) * ma.ld + k, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < AGR; ++i) {
        const int g = a_block(tid + 256 * i);
        const int m4 = (g % (BM / 4)) * 4, k8 = (g / (BM / 4)) * 8;
        const bool on = g < (BM / 4) * 16;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          rat[i][j] = __builtin_amdgcn_raw_buffer_load_b32(rsa, on ? (k0 + k8 + j) * ma.ld + m_base + m4 : 0x7ffffff0, 0, 0);
      }
    }
    // a table larger than the caches is streamed once per launch: non-temporal, so that it does not displace the map,
    // d_out and the split-K slabs in L2
    if (stream_b) {
#pragma unroll
      for (int j = 0; j < 8; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsb, (k0 + bk8 + j) * mb.ld * 4 + b_off, 0, 2);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsb, (k0 + bk8 + j) * mb.ld * 4 + b_off, 0, 0);
    }
  };
  auto stage = [&](const Regs& R) {
    const auto& ra = R.ra; const auto& rat = R.rat; const auto& rb = R.rb;
    if constexpr (AKC) {
#pragma unroll
      for (int i = 0; i < AGK; ++i) {
        const int g = tid + 256 * i, row = akc_row(g), c = akc_grp(g) * 2;
        u32x4 lo, hi;
        unsigned a, b;
        bytes_to_bf16(ra[i][0], a, b); lo[0] = a; lo[1] = b;
        bytes_to_bf16(ra[i][1], a, b); lo[2] = a; lo[3] = b;
        bytes_to_bf16(ra[i][2], a, b); hi[0] = a; hi[1] = b;
        bytes_to_bf16(ra[i][3], a, b); hi[2] = a; hi[3] = b;
        *reinterpret_cast<u32x4*>(As + bf_img(row, c)) = lo;
        *reinterpret_cast<u32x4*>(As + bf_img(row, c + 1)) = hi;
      }
    } else {
#pragma unroll
      for (int i = 0; i < AGR; ++i) {
        const int g = a_block(tid + 256 * i);
        if (g < (BM / 4) * 16) {
          const int m4 = (g % (BM / 4)) * 4, c = g / (BM / 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {  // byte e of the eight words = 8 consecutive k of row m4 + e
            u32x4 v;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const unsigned sel = 0x0c000c00u | (unsigned)e | ((unsigned)(4 + e) << 16);  // [lo.byte e, 0, hi.byte e, 0]
              v[t] = __builtin_amdgcn_perm(rat[i][2 * t + 1], rat[i][2 * t], sel) * 0x3f80u;
            }
            *reinterpret_cast<u32x4*>(As + bf_img(m4 + e, c)) = v;
          }
        }
      }
    }
    // exact three-way split of the 8 k x 4 n block, packed along k
    u32x4 pl[3][4];  // [plane][n]: 8 bf16
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      unsigned h[2][4], m[2][4], l[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = __uint_as_float(rb[2 * t + u][e]);
          const unsigned hb = __float_as_uint(x) & 0xffff0000u;
          const float r1 = x - __uint_as_float(hb);
          const unsigned mb_ = __float_as_uint(r1) & 0xffff0000u;
          const float r2 = r1 - __uint_as_float(mb_);
          h[u][e] = hb; m[u][e] = mb_; l[u][e] = __float_as_uint(r2);
        }
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // word t of column e: k = 2 t (low half) and 2 t + 1 (high half)
        pl[0][e][t] = __builtin_amdgcn_perm(h[1][e], h[0][e], 0x07060302u);
        pl[1][e][t] = __builtin_amdgcn_perm(m[1][e], m[0][e], 0x07060302u);
        pl[2][e][t] = __builtin_amdgcn_perm(l[1][e], l[0][e], 0x07060302u);
      }
    }
#pragma unroll
    for (int pnum = 0; pnum < 3; ++pnum)
#pragma unroll
      for (int e = 0; e < 4; ++e) *reinterpret_cast<u32x4*>(Bs + pnum * (BN * kBfK * 2) + bf_img(bn4 + e, bk8 >> 3)) = pl[pnum][e];
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  FusedL1Pre<Epi::kFusedL1 ? BM : 32> l1pre;
  if constexpr (Epi::kFusedL1) fused_l1_prefetch<BM>(epi, l1pre, m_base, tile_n, m0, n0, r, q, wave);
  auto contract = [&]() {
#pragma unroll
    for (int kb = 0; kb < kBfK / 32; ++kb) {
      const int c = kb * 4 + q;  // this lane's 8 k of the 32-k block
      bf16x8 a[TM], b[3][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(As + bf_img(m0 + 16 * i + r, c));
#pragma unroll
      for (int pnum = 0; pnum < 3; ++pnum)
#pragma unroll
        for (int t = 0; t < TN; ++t) b[pnum][t] = *reinterpret_cast<const bf16x8*>(Bs + pnum * (BN * kBfK * 2) + bf_img(n0 + 16 * t + r, c));
      // smallest terms first; consecutive MFMAs go to different accumulators
#pragma unroll
      for (int pnum = 2; pnum >= 0; --pnum)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int t = 0; t < TN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[pnum][t], acc[i][t], 0, 0, 0);
    }
  };
  // (two register sets with the loads of tile t+2 in flight were measured and lose: 83 -> 108 us for the 224x224 forward,
  // 109 -> 193 us for its weight gradient, whose K = batch is a single tile; nothing at the CIFAR shapes)
  fetch(k_lo, set0);
  for (int k0 = k_lo; k0 < k_hi; k0 += kBfK) {
    stage(set0);
    __syncthreads();
    if (k0 + kBfK < k_hi) fetch(k0 + kBfK, set0);
    contract();
    __syncthreads();
  }
  if constexpr (Epi::kFusedL1) {
    fused_l1_epilogue<BM>(epi, l1pre, reinterpret_cast<float*>(smem), acc, m_base, tile_n, m0, n0, r, q, wave, tid);
  } else if constexpr (is_rmw<Epi>::value) {
    rmw_tile<BM, BN, Epi>(reinterpret_cast<float*>(smem), epi, acc, M, N, m_base, n_base, m0, n0);
  } else {
    store_tile<BM, BN, Epi>(reinterpret_cast<float*>(smem), epi, acc, M, N, m_base, n_base, m0, n0, tile, ks);
  }
}

// ---- gemm_tile_bf with K tiles of 64 (BM = 128 or 64, BN = 64): half the LDS and fewer live registers, for the kernels
// whose occupancy is what they wait for -- the two big-table kernels (two workgroups per CU at 80 KB / ~200 registers) and
// the merged backward launch (its 64 KB were these tiles').  Same arithmetic as gemm_tile_bf; the f32 operand is staged in
// 4 k x 4 n blocks (four 16-byte loads, 8-byte LDS stores).
constexpr int kBf64K = 64;
__device__ __forceinline__ int bf64_img(int row, int chunk) { return row * (kBf64K * 2) + ((chunk ^ ((row >> 1) & 7)) << 4); }
template <int BM>
constexpr int gemm_bf64_lds_bytes() { return (BM + 3 * 64) * kBf64K * 2; }

template <int BM, bool AKC, class Epi>
__device__ __forceinline__ void gemm_tile_bf64(unsigned char* __restrict__ smem, const Mat& ma, const Mat& mb, const Epi& epi, int M, int N,
                                               int k_lo, int k_hi, int tiles_n, int tile, int ks) {
  static_assert(BM == 64 || BM == 128, "tile heights");
  constexpr int BN = 64, KT = kBf64K;
  constexpr int TM = BM / 32, TN = BN / 32;
  unsigned char* __restrict__ As = smem;
  unsigned char* __restrict__ Bs = smem + BM * KT * 2;  // plane p at + p * BN * KT * 2
  constexpr int PB = BN * KT * 2;
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ma.p), 0, ma.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(mb.p), 0, mb.bytes, 0x00020000);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
  const int m_base = tile_m * BM, n_base = tile_n * BN;
  const int m0 = (wave >> 1) * (BM / 2), n0 = (wave & 1) * (BN / 2);
  // f32 operand: thread = (k block of 4: 16 of them, n block of 4: 16 of them); lanes of a 16-lane group take 4 n x 4 k blocks
  const int bn4 = (((lane & 3) | ((lane >> 4) << 2))) * 4, bk4 = (((lane >> 2) & 3) | (wave << 2)) * 4;
  const int b_off = (n_base + bn4) * 4;
  const bool stream_b = mb.bytes > (64u << 20);  // uniform
  // A, forward (bytes contiguous along k): 16-byte groups (row, 16 k): 4 per row, two per thread
  // A, weight gradient (bytes contiguous along m): one 8 k x 4 m block per thread (8 x 32 blocks)
  auto a_block = [&](int g) {  // as in gemm_tile_bf: 16 consecutive blocks = 4 m blocks x 4 k blocks
    constexpr int MB = BM / 4;
    const int grp = g >> 4, l = g & 15;
    const int groups_m = MB / 4;
    const int gm = grp % groups_m, gk = grp / groups_m;
    return ((gk * 4 + (l >> 2)) * MB) + gm * 4 + (l & 3);
  };
  // forward: the four groups of a row go to four consecutive lanes; the eight lanes of a ds_write_b128 group then hold two
  // rows, which must differ in row / 2 (the swizzle term) or their chunks collide: bits 0 and 1 of the row index are swapped
  auto a_row_of = [](int g) { const int x = g >> 2; return (x & ~3) | ((x & 1) << 1) | ((x >> 1) & 1); };
  constexpr int AGK = BM * 4 / 256;           // forward: 16-byte groups per thread (2 or 1)
  constexpr int ABLK = (BM / 4) * (KT / 8);   // weight gradient: 8 k x 4 m blocks (256 or 128: the first threads)
  u32x4 ra[AGK];
  unsigned rat[8];
  u32x4 rb[4];
  auto fetch = [&](int k0) {
    if constexpr (AKC) {
#pragma unroll
      for (int i = 0; i < AGK; ++i) {
        const int g = tid + 256 * i, row = a_row_of(g), k = k0 + (g & 3) * 16;
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsa, (m_base + row) * ma.ld + k, 0, 0);
      }
    } else {
      const int g = a_block(tid);
      const int m4 = (g % (BM / 4)) * 4, k8 = (g / (BM / 4)) * 8;
      const bool on = tid < ABLK;
#pragma unroll
      for (int j = 0; j < 8; ++j) rat[j] = __builtin_amdgcn_raw_buffer_load_b32(rsa, on ? (k0 + k8 + j) * ma.ld + m_base + m4 : 0x7ffffff0, 0, 0);
    }
    if (stream_b) {
#pragma unroll
      for (int j = 0; j < 4; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsb, (k0 + bk4 + j) * mb.ld * 4 + b_off, 0, 2);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(rsb, (k0 + bk4 + j) * mb.ld * 4 + b_off, 0, 0);
    }
  };
  auto stage = [&]() {
    if constexpr (AKC) {
#pragma unroll
      for (int i = 0; i < AGK; ++i) {
        const int g = tid + 256 * i, row = a_row_of(g), c = (g & 3) * 2;
        u32x4 lo, hi;
        unsigned a, b;
        bytes_to_bf16(ra[i][0], a, b); lo[0] = a; lo[1] = b;
        bytes_to_bf16(ra[i][1], a, b); lo[2] = a; lo[3] = b;
        bytes_to_bf16(ra[i][2], a, b); hi[0] = a; hi[1] = b;
        bytes_to_bf16(ra[i][3], a, b); hi[2] = a; hi[3] = b;
        *reinterpret_cast<u32x4*>(As + bf64_img(row, c)) = lo;
        *reinterpret_cast<u32x4*>(As + bf64_img(row, c + 1)) = hi;
      }
    } else if (tid < ABLK) {
      const int g = a_block(tid);
      const int m4 = (g % (BM / 4)) * 4, c = g / (BM / 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {  // byte e of the eight words = 8 consecutive k of row m4 + e
        u32x4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const unsigned sel = 0x0c000c00u | (unsigned)e | ((unsigned)(4 + e) << 16);
          v[t] = __builtin_amdgcn_perm(rat[2 * t + 1], rat[2 * t], sel) * 0x3f80u;
        }
        *reinterpret_cast<u32x4*>(As + bf64_img(m4 + e, c)) = v;
      }
    }
    // exact three-way split of the 4 k x 4 n block, packed along k: per n and plane one 8-byte half chunk
    using u32x2 = __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned;
    u32x2 pl[3][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      unsigned h[2][4], m[2][4], l[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = __uint_as_float(rb[2 * t + u][e]);
          const unsigned hb = __float_as_uint(x) & 0xffff0000u;
          const float r1 = x - __uint_as_float(hb);
          const unsigned mb_ = __float_as_uint(r1) & 0xffff0000u;
          const float r2 = r1 - __uint_as_float(mb_);
          h[u][e] = hb; m[u][e] = mb_; l[u][e] = __float_as_uint(r2);
        }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pl[0][e][t] = __builtin_amdgcn_perm(h[1][e], h[0][e], 0x07060302u);
        pl[1][e][t] = __builtin_amdgcn_perm(m[1][e], m[0][e], 0x07060302u);
        pl[2][e][t] = __builtin_amdgcn_perm(l[1][e], l[0][e], 0x07060302u);
      }
    }
    const int half = (bk4 & 4) ? 8 : 0;
#pragma unroll
    for (int pnum = 0; pnum < 3; ++pnum)
#pragma unroll
      for (int e = 0; e < 4; ++e) *reinterpret_cast<u32x2*>(Bs + pnum * PB + bf64_img(bn4 + e, bk4 >> 3) + half) = pl[pnum][e];
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto contract = [&]() {
#pragma unroll
    for (int kb = 0; kb < KT / 32; ++kb) {
      const int c = kb * 4 + q;
      bf16x8 a[TM], b[3][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8*>(As + bf64_img(m0 + 16 * i + r, c));
#pragma unroll
      for (int pnum = 0; pnum < 3; ++pnum)
#pragma unroll
        for (int t = 0; t < TN; ++t) b[pnum][t] = *reinterpret_cast<const bf16x8*>(Bs + pnum * PB + bf64_img(n0 + 16 * t + r, c));
#pragma unroll
      for (int pnum = 2; pnum >= 0; --pnum)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int t = 0; t < TN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[pnum][t], acc[i][t], 0, 0, 0);
    }
  };
  fetch(k_lo);
  for (int k0 = k_lo; k0 < k_hi; k0 += KT) {
    stage();
    __syncthreads();
    if (k0 + KT < k_hi) fetch(k0 + KT);
    contract();
    __syncthreads();
  }
  if constexpr (is_rmw<Epi>::value) rmw_tile<BM, BN, Epi>(reinterpret_cast<float*>(smem), epi, acc, M, N, m_base, n_base, m0, n0);
  else store_tile<BM, BN, Epi>(reinterpret_cast<float*>(smem), epi, acc, M, N, m_base, n_base, m0, n0, tile, ks);
}

template <bool AKC, class Epi>
__global__ __launch_bounds__(256) void ftm_gemm_bf64_kernel(Mat ma, Mat mb, Epi epi, int M, int N, int K, int klen, int tiles_n) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[gemm_bf64_lds_bytes<128>()];
  const int ks = blockIdx.y, k_lo = ks * klen;
  int tile = blockIdx.x;
  if constexpr (is_rmw<Epi>::value) {
    if (epi.xcd_remap) {  // see ftm_gemm_bf_kernel
      const int nwg = gridDim.x, xcd = tile & 7, q8 = nwg >> 3, r8 = nwg & 7;
      tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3);
    }
  }
  gemm_tile_bf64<128, AKC, Epi>(smem, ma, mb, epi, M, N, k_lo, (k_lo + klen < K) ? k_lo + klen : K, tiles_n, tile, ks);
}

#include "update_forward.h"  // the big table's update of step t + the forward of step t+1 in one pass over the table

// ---- both operands f32 (the value gradient d_out W^T), on the bf16 matrix unit ------------------------------------------
// Both operands are split into the three truncation planes; the six plane products whose weight is >= 2^-16
// (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi) are accumulated, smallest first.  Left out: mid lo + lo mid (<= 2^-23 of a
// product, the size of an f32 rounding) and lo lo (2^-32).  Both operands are contiguous along k in memory, so a thread owns
// runs of 8 k of one row (two 16-byte loads), splits them in registers and writes one 16-byte chunk per plane; images are
// [row][KT k] bf16 (KT = 64: 128-byte rows, chunk index XOR-swizzled by row / 2 like the f32 kernel's 32-float rows; KT = 32:
// 64-byte rows, swizzled by row / 4).  What decides is how many workgroups a CU holds, i.e. how much of one workgroup's split /
// LDS / MFMA phases another's can hide: the first attempt of the round (K tiles of 128, 144 KB, one workgroup per CU) lost
// against the f32 MFMA, 178 vs 153 us at the 224x224 shape; K tiles of 64 (72 KB, two per CU) take 115 us, K tiles of 32
// (36 KB and 120 registers: four per CU) 100 us.  (Splitting d_out -- the same rows for every workgroup, two thirds of a tile's
// split work -- once per launch into planes in memory instead: 114.8 + 4.6 us for the extra kernel against 117; the split
// VALU is not what the tile waits for.)
constexpr int kBf6K = 64;
template <int BM, int BN, int KT = kBf6K>
constexpr int gemm_bf6_lds_bytes() { return 3 * (BM + BN) * KT * 2; }
// rows of KT bf16 (128 or 64 bytes): two or four rows per 256-byte bank row, chunk index XOR-swizzled accordingly
template <int KT>
__device__ __forceinline__ int bf6_img(int row, int chunk) {
  // KT == 32: a ds_read_b128 lane group holds rows {0-3, 12-15} of one chunk and rows {4-11} of the next (the hardware's
  // groups are {0-3, 12-15, 20-27}, ...), and four 64-byte rows share a 256-byte bank row, so the XOR term per block of four
  // rows is {0, 3, 2, 1} -- with the obvious {0, 1, 2, 3} a third of the kernel's LDS cycles were conflicts (SQ_LDS_BANK_CONFLICT)
  return row * (KT * 2) + ((chunk ^ (KT == 64 ? (row >> 1) & 7 : (4 - (row >> 2)) & 3)) << 4);
}

// 8 consecutive k (two float4) -> one 16-byte chunk of 8 bf16 per plane
__device__ __forceinline__ void split8(const u32x4& v0, const u32x4& v1, u32x4& hi, u32x4& mid, u32x4& lo) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned xb = e < 4 ? v0[e] : v1[e - 4];
    const float x = __uint_as_float(xb);
    const unsigned hb = xb & 0xffff0000u;
    const float r1 = x - __uint_as_float(hb);
    const unsigned mb_ = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb_);
    h[e] = hb; m[e] = mb_; l[e] = __float_as_uint(r2);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {  // word t: k = 2 t (low half), 2 t + 1 (high half)
    hi[t] = __builtin_amdgcn_perm(h[2 * t + 1], h[2 * t], 0x07060302u);
    mid[t] = __builtin_amdgcn_perm(m[2 * t + 1], m[2 * t], 0x07060302u);
    lo[t] = __builtin_amdgcn_perm(l[2 * t + 1], l[2 * t], 0x07060302u);
  }
}

// ABL: timing-only ablations for tools/debug (results are WRONG unless 0): 1 no split of A (raw words stored), 2 no split at
// all, 3 no MFMAs (fragments still read), 4 no staging (no split, no LDS stores), 5 no global loads
// APL: operand A arrives already split -- ma.p = K-tile-major planes [K / 32][3][M][32] bf16 (hi, mid, lo; ftv_split_planes), KT = 32: three 16-byte loads per 8-k
// run and no split VALU for it (the d_out planes of nnue_ftm_backward_values_ws).  This kernel is issue-bound (MFMA and VALU issue
// add up on a SIMD, tools/micro/mfma_bf16_tile.hip), and two thirds of its split work is the same d_out rows in every workgroup.
template <int BM, int BN, class Epi, int KT = kBf6K, int ABL = 0, bool APL = false>
__device__ __forceinline__ void gemm_tile_bf6(unsigned char* __restrict__ smem, const Mat& ma, const Mat& mb, const Epi& epi, int M, int N,
                                              int k_lo, int k_hi, int tiles_n, int tile, int ks) {
  static_assert((BM == 32 || BM == 64 || BM == 128) && (BN == 64 || BN == 128), "tile shapes");
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int PA = BM * KT * 2, PB = BN * KT * 2;  // bytes per plane
  constexpr int RUNS = KT / 8;                      // 8-k runs per row of a K tile
  unsigned char* __restrict__ As = smem;
  unsigned char* __restrict__ Bs = smem + 3 * PA;
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ma.p), 0, ma.bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(mb.p), 0, mb.bytes, 0x00020000);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
  const int m_base = tile_m * BM, n_base = tile_n * BN;
  const int m0 = (wave >> 1) * (BM / 2), n0 = (wave & 1) * (BN / 2);
  constexpr int GA = BM * RUNS / 256, GB = BN * RUNS / 256;  // 8-k runs per thread
  static_assert(GA >= 1 && GB >= 1, "every thread stages at least one run of each operand");
  const bool stream_b = mb.bytes > (64u << 20);        // uniform: a table larger than the caches is read non-temporally
  u32x4 ra[GA][APL ? 3 : 2], rb[GB][2];
  auto fetch = [&](int k0) {
    if constexpr (ABL == 5) {
      if (k0 != k_lo) return;
    }
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      const int g = tid + 256 * i, row = g / RUNS, k = k0 + (g % RUNS) * 8;
      if constexpr (APL) {  // rows past M re-read row M - 1: their accumulators are never stored
        const int rr = m_base + row < M ? m_base + row : M - 1;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          ra[i][pl] = __builtin_amdgcn_raw_buffer_load_b128(rsa, ((((k >> 5) * 3 + pl) * M + rr) * 32 + (k & 31)) * 2, 0, 0);
      } else {
        ra[i][0] = mat_load<false>(rsa, ma, m_base + row, k);
        ra[i][1] = mat_load<false>(rsa, ma, m_base + row, k + 4);
      }
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const int g = tid + 256 * i, row = g / RUNS, k = k0 + (g % RUNS) * 8;
      const int rr = n_base + row < mb.clamp ? n_base + row : mb.clamp;
      const int off0 = k < mb.inner_k ? (rr * mb.ld + k) * 4 : 0x7ffffff0, off1 = k + 4 < mb.inner_k ? (rr * mb.ld + k + 4) * 4 : 0x7ffffff0;
      if (stream_b) {
        rb[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rsb, off0, 0, 2);
        rb[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rsb, off1, 0, 2);
      } else {
        rb[i][0] = __builtin_amdgcn_raw_buffer_load_b128(rsb, off0, 0, 0);
        rb[i][1] = __builtin_amdgcn_raw_buffer_load_b128(rsb, off1, 0, 0);
      }
    }
  };
  auto stage = [&]() {
    if constexpr (ABL == 4) return;
#pragma unroll
    for (int i = 0; i < GA; ++i) {
      const int g = tid + 256 * i, row = g / RUNS, c = g % RUNS;
      u32x4 hi, mid, lo;
      if constexpr (APL) { hi = ra[i][0]; mid = ra[i][1]; lo = ra[i][2]; }
      else if constexpr (ABL == 1 || ABL == 2) { hi = ra[i][0]; mid = ra[i][1]; lo = ra[i][0] ^ ra[i][1]; }
      else split8(ra[i][0], ra[i][1], hi, mid, lo);
      *reinterpret_cast<u32x4*>(As + bf6_img<KT>(row, c)) = hi;
      *reinterpret_cast<u32x4*>(As + PA + bf6_img<KT>(row, c)) = mid;
      *reinterpret_cast<u32x4*>(As + 2 * PA + bf6_img<KT>(row, c)) = lo;
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const int g = tid + 256 * i, row = g / RUNS, c = g % RUNS;
      u32x4 hi, mid, lo;
      if constexpr (ABL == 2) { hi = rb[i][0]; mid = rb[i][1]; lo = rb[i][0] ^ rb[i][1]; }
      else split8(rb[i][0], rb[i][1], hi, mid, lo);
      *reinterpret_cast<u32x4*>(Bs + bf6_img<KT>(row, c)) = hi;
      *reinterpret_cast<u32x4*>(Bs + PB + bf6_img<KT>(row, c)) = mid;
      *reinterpret_cast<u32x4*>(Bs + 2 * PB + bf6_img<KT>(row, c)) = lo;
    }
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int t = 0; t < TN; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto contract = [&]() {
#pragma unroll
    for (int kb = 0; kb < KT / 32; ++kb) {
      const int c = kb * 4 + q;  // this lane's 8 k of the 32-k block
      bf16x8 a[3][TM], b[3][TN];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[pl][i] = *reinterpret_cast<const bf16x8*>(As + pl * PA + bf6_img<KT>(m0 + 16 * i + r, c));
#pragma unroll
        for (int t = 0; t < TN; ++t) b[pl][t] = *reinterpret_cast<const bf16x8*>(Bs + pl * PB + bf6_img<KT>(n0 + 16 * t + r, c));
      }
      // smallest terms first (plane 0 = hi, 1 = mid, 2 = lo); consecutive MFMAs go to different accumulators
      constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
      if constexpr (ABL == 3) {  // keep the fragment reads alive without the matrix unit
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int t = 0; t < TN; ++t) acc[i][t][0] += (float)a[pa[s6]][i][0] + (float)b[pb[s6]][t][0];
      } else {
#pragma unroll
      for (int s6 = 0; s6 < 6; ++s6)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int t = 0; t < TN; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pa[s6]][i], b[pb[s6]][t], acc[i][t], 0, 0, 0);
      }
    }
  };
  fetch(k_lo);
  for (int k0 = k_lo; k0 < k_hi; k0 += KT) {
    stage();
    __syncthreads();
    if (k0 + KT < k_hi) fetch(k0 + KT);
    contract();
    __syncthreads();
  }
  store_tile<BM, BN, Epi>(reinterpret_cast<float*>(smem), epi, acc, M, N, m_base, n_base, m0, n0, tile, ks);
}

template <int BM, int BN, class Epi, int KT, int ABL = 0, bool APL = false>
__global__ __launch_bounds__(256) void ftm_gemm_bf6_kernel(Mat ma, Mat mb, Epi epi, int M, int N, int K, int tiles_n) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[gemm_bf6_lds_bytes<BM, BN, KT>()];
  gemm_tile_bf6<BM, BN, Epi, KT, ABL, APL>(smem, ma, mb, epi, M, N, 0, K, tiles_n, blockIdx.x, 0);
}

template <int BM, int BN, bool AKC, class Epi>
__global__ __launch_bounds__(256) void ftm_gemm_bf_kernel(Mat ma, Mat mb, Epi epi, int M, int N, int K, int klen, int tiles_n, GroupArgs ga) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[gemm_bf_lds_bytes<BM, BN>()];
  if (ga.n && blockIdx.x == gridDim.x - 1) {  // see ftm_gemm_kernel
    if (blockIdx.y == 0) bucket_group_body<256>(ga, reinterpret_cast<int*>(smem));
    return;
  }
  const int ks = blockIdx.y, k_lo = ks * klen;
  int tile = blockIdx.x;
  // (dealing the K slices of the split-K forward to the XCDs instead of its column tiles -- one L2 per map slice -- was
  // measured at the 224x224 shape and changes nothing: 68.7 vs 69.1 us)
  if constexpr (is_rmw<Epi>::value) {
    // Blocks are dealt round-robin over the 8 XCDs (observed, never relied on for results): with this remap the blocks of one
    // XCD walk a contiguous run of tiles, so the 16 column tiles that share a map tile meet in ONE L2 instead of eight.
    // Bijective for any grid size (cdna_hip_programming.md, XCD swizzle).
    if (epi.xcd_remap) {
      const int nwg = gridDim.x, xcd = tile & 7, q8 = nwg >> 3, r8 = nwg & 7;
      tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (tile >> 3);
    }
  }
  gemm_tile_bf<BM, BN, AKC, Epi>(smem, ma, mb, epi, M, N, k_lo, (k_lo + klen < K) ? k_lo + klen : K, tiles_n, tile, ks);
}

template <int BM>
__global__ __launch_bounds__(256) void ftm_forward_l1_bf_kernel(Mat ma, Mat mb, FwdL1Epi epi, int M, int N, int K, int tiles_n) {
  constexpr int kGemm = gemm_bf_lds_bytes<BM, 64>(), kEpi = 2 * BM * kL1Ld * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kGemm > kEpi ? kGemm : kEpi];
  gemm_tile_bf<BM, 64, true, FwdL1Epi>(smem, ma, mb, epi, M, N, 0, K, tiles_n, blockIdx.x, 0);
}

// ga.n != NULL: the launch has one workgroup more than tiles (grid.x - 1 tiles); it groups the batch by layer-stack bucket
// (bucket_group.h) -- work that only needs the binarise kernel's counts, like this product, and would otherwise be a launch
// of its own between them.
template <int BM, int BN, int BK, bool AKC, bool BKC, class Epi>
__global__ __launch_bounds__(256) void ftm_gemm_kernel(Mat ma, Mat mb, Epi epi, int M, int N, int K, int klen, int tiles_n, GroupArgs ga) {
  __shared__ __attribute__((aligned(16))) float smem[gemm_lds_floats<BM, BN, BK, AKC, BKC>()];
  static_assert(gemm_lds_floats<BM, BN, BK, AKC, BKC>() >= kGroupLdsInts, "the grouping workgroup borrows the tile's LDS");
  if (ga.n && blockIdx.x == gridDim.x - 1) {
    if (blockIdx.y == 0) bucket_group_body<256>(ga, reinterpret_cast<int*>(smem));
    return;
  }
  const int k_lo = blockIdx.y * klen;
  gemm_tile<BM, BN, BK, AKC, BKC, Epi>(smem, ma, mb, epi, M, N, k_lo, (k_lo + klen < K) ? k_lo + klen : K, tiles_n, blockIdx.x, blockIdx.y);
}

template <int BM, int BK>
__global__ __launch_bounds__(256) void ftm_forward_l1_kernel(Mat ma, Mat mb, FwdL1Epi epi, int M, int N, int K, int tiles_n) {
  constexpr int kGemm = gemm_lds_floats<BM, 64, BK, true, false>(), kEpi = 2 * BM * kL1Ld;
  __shared__ __attribute__((aligned(16))) float smem[kGemm > kEpi ? kGemm : kEpi];
  gemm_tile<BM, 64, BK, true, false, FwdL1Epi>(smem, ma, mb, epi, M, N, 0, K, tiles_n, blockIdx.x, 0);
}

// out = bias + sink[b] * weight[F-1] + sum of the split-K slabs (fixed order)
__global__ __launch_bounds__(256) void ftm_finish_kernel(const float* __restrict__ partial, int ksplit, int64_t count4,
                                                         const float* __restrict__ bias, const float* __restrict__ w_last,
                                                         const float* __restrict__ sink, int L1, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count4) return;
  const int64_t e = i * 4;
  const int n = (int)(e % L1);
  const float s = sink[e / L1];
  const float4 bv = *reinterpret_cast<const float4*>(bias + n), wv = *reinterpret_cast<const float4*>(w_last + n);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (; k + 8 <= ksplit; k += 8) {  // eight slabs in flight, summed in slab order
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(partial + (size_t)(k + u) * count4 * 4 + e);
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  for (; k < ksplit; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(partial + (size_t)k * count4 * 4 + e);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  acc.x += fmaf(s, wv.x, bv.x); acc.y += fmaf(s, wv.y, bv.y); acc.z += fmaf(s, wv.z, bv.z); acc.w += fmaf(s, wv.w, bv.w);
  *reinterpret_cast<float4*>(out + e) = acc;
}

// The rows of the weight gradient that are not positions of the map, and the bias gradient:
//   d_bias = sum_b d_out[b], d_weight[F-1] = sum_b sink[b] d_out[b], d_weight[direct .. F-2] = 0.
// grid (ceil(L1 / 16), 1 + zero-fill slices); block y = 0 reduces 16 columns over the batch in sixteen fixed slices.
struct TailRows {
  const float* d_out;
  const float* sink;
  int B, L1, direct, F;
  float* d_weight;
  float* d_bias;
  int col_blocks, zero_slices;  // block grid: col_blocks x (1 + zero_slices)
};

// red_mem: 512 floats of the caller's LDS (the merged launches lend their tile buffer: a static array of its own would
// push them over a third of a CU's LDS)
__device__ __forceinline__ void tail_rows_block(const TailRows& t, int bx, int by, float* __restrict__ red_mem) {
  float (*red)[16][16] = reinterpret_cast<float (*)[16][16]>(red_mem);
  const float* __restrict__ d_out = t.d_out;
  const float* __restrict__ sink = t.sink;
  float* __restrict__ d_weight = t.d_weight;
  float* __restrict__ d_bias = t.d_bias;
  const int B = t.B, L1 = t.L1, direct = t.direct, F = t.F;
  const int c = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int col = bx * 16 + c;
  if (by > 0) {  // zero rows the map cannot reach
    if (d_weight && col < L1)
      for (int f = direct + (by - 1) * 16 + part; f < F - 1; f += 16 * t.zero_slices) d_weight[(size_t)f * L1 + col] = 0.0f;
    return;
  }
  float sb = 0.f, sw = 0.f;
  if (col < L1) {
    const int per = (B + 15) / 16, lo = part * per, hi = lo + per < B ? lo + per : B;
    int b = lo;
    for (; b + 8 <= hi; b += 8) {  // eight independent loads in flight; the sums keep their order
      float d[8], s[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        d[u] = d_out[(size_t)(b + u) * L1 + col];
        s[u] = sink[b + u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        sb += d[u];
        sw = fmaf(s[u], d[u], sw);
      }
    }
    for (; b < hi; ++b) {
      const float d = d_out[(size_t)b * L1 + col];
      sb += d;
      sw = fmaf(sink[b], d, sw);
    }
  }
  red[0][part][c] = sb;
  red[1][part][c] = sw;
  __syncthreads();
  if (part < 2 && col < L1) {  // part 0 -> bias row, part 1 -> table row F-1; sixteen partials in order
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[part][i][c];
    if (part == 0) {
      if (d_bias) d_bias[col] = s;
    } else if (d_weight) {
      d_weight[(size_t)(F - 1) * L1 + col] = s;
    }
  }
}

__global__ __launch_bounds__(256) void ftm_tail_rows_kernel(TailRows t) {
  __shared__ float red[512];
  tail_rows_block(t, blockIdx.x, blockIdx.y, red);
}

// Weight gradient, value gradient and the tail rows in ONE launch: they are independent (all three read d_out), so
// their workgroups share the chip instead of queueing behind two kernel boundaries.  Blocks [0, n_v) are value-gradient
// tiles, [n_v, n_v + n_w) weight-gradient tiles, the rest tail-row blocks.
struct CwArgs {  // classifier first-layer weight gradient riding in the same launch (n_c == 0: absent)
  Mat a, b;
  CwEpi e;
  int M, N, K, tiles_n, n_c;
  // bucketed layer stacks (seg != NULL): d_z1 and ft are in GROUPED row order (nnue_bucket_group), bucket k owns rows
  // seg[k] .. seg[k+1]; the tile family repeats per bucket (per_bucket tiles each), tile (k, t) contracts only that row
  // range into d_w1[k] -- an empty bucket's tiles store zeros
  const int* seg;
  int per_bucket;
};

template <int WM, int WN, int WK, int VM, int VN, int VK>
__global__ __launch_bounds__(256) void ftm_backward_kernel(Mat wa, Mat wb, BwwEpi we, int wM, int wN, int wK, int w_tiles_n, int n_w,
                                                           Mat va, Mat vb, ValEpi ve, int vM, int vN, int vK, int v_tiles_n, int n_v,
                                                           CwArgs c, TailRows t, SmallWgrad sw) {
  constexpr int kW = gemm_lds_floats<WM, WN, WK, false, false>(), kV = gemm_lds_floats<VM, VN, VK, true, true>();
  constexpr int kC = gemm_lds_floats<WM, WN, WK, false, false>();  // the rider's tiles have the weight tiles' shape
  constexpr int kWV = kW > kV ? kW : kV;
  __shared__ __attribute__((aligned(16))) float smem[kWV > kC ? kWV : kC];
  const int blk = blockIdx.x;
  if (blk < n_v) {  // the longer tiles (K = L1) are dispatched first
    gemm_tile<VM, VN, VK, true, true, ValEpi>(smem, va, vb, ve, vM, vN, 0, vK, v_tiles_n, blk, 0);
  } else if (blk < n_v + n_w) {
    gemm_tile<WM, WN, WK, false, false, BwwEpi>(smem, wa, wb, we, wM, wN, 0, wK, w_tiles_n, blk - n_v, 0);
  } else if (blk < n_v + n_w + c.n_c) {
    const int t = blk - n_v - n_w;
    if (c.seg) {
      const int kb = t / c.per_bucket;
      const int k_lo = c.seg[kb], k_hi = c.seg[kb + 1];
      Mat a = c.a, b = c.b;  // rows from k_hi on belong to the next bucket: end both windows there (reads as zero)
      a.bytes = (unsigned)k_hi * (unsigned)a.ld * 4u;
      b.bytes = (unsigned)k_hi * (unsigned)b.ld * 4u;
      CwEpi e = c.e;
      e.d_w1 += (size_t)kb * c.M * c.N;
      gemm_tile<WM, WN, WK, false, false, CwEpi>(smem, a, b, e, c.M, c.N, k_lo, k_hi, c.tiles_n, t - kb * c.per_bucket, 0);
    } else {
      gemm_tile<WM, WN, WK, false, false, CwEpi>(smem, c.a, c.b, c.e, c.M, c.N, 0, c.K, c.tiles_n, t, 0);
    }
  } else {
    const int i = blk - n_w - n_v - c.n_c, n_t = t.col_blocks * (1 + t.zero_slices);
    if (i < n_t) tail_rows_block(t, i % t.col_blocks, i / t.col_blocks, smem);
    else small_wgrad_body(sw, i - n_t, smem);  // the classifier's small gradients + mean loss (blocks exist only when they ride here)
  }
}

// The same launch with the weight-gradient tiles on the bf16 matrix unit (gemm_tile_bf, WM x 64 x 128); value-gradient
// tiles (both operands f32) and the rider keep the f32 MFMA.
template <int WM, int VM, int VN, int VK, bool V6 = false, bool W64 = false>
__global__ __launch_bounds__(256) void ftm_backward_bf_kernel(Mat wa, Mat wb, BwwEpi we, int wM, int wN, int wK, int w_tiles_n, int n_w,
                                                              Mat va, Mat vb, ValEpi ve, int vM, int vN, int vK, int v_tiles_n, int n_v,
                                                              CwArgs c, TailRows t, SmallWgrad sw) {
  constexpr int kW = W64 ? gemm_bf64_lds_bytes<WM>() : gemm_bf_lds_bytes<WM, 64>(), kV = V6 ? gemm_bf6_lds_bytes<VM, VN>() : gemm_lds_floats<VM, VN, VK, true, true>() * 4;
  constexpr int kC = gemm_lds_floats<32, 64, 128, false, false>() * 4;
  constexpr int kWV = kW > kV ? kW : kV;
  __shared__ __attribute__((aligned(16))) unsigned char smem_b[kWV > kC ? kWV : kC];
  float* smem = reinterpret_cast<float*>(smem_b);
  const int blk = blockIdx.x;
  if (blk < n_v) {
    if constexpr (V6) gemm_tile_bf6<VM, VN, ValEpi>(smem_b, va, vb, ve, vM, vN, 0, vK, v_tiles_n, blk, 0);
    else gemm_tile<VM, VN, VK, true, true, ValEpi>(smem, va, vb, ve, vM, vN, 0, vK, v_tiles_n, blk, 0);
  } else if (blk < n_v + n_w) {
    if constexpr (W64) gemm_tile_bf64<WM, false, BwwEpi>(smem_b, wa, wb, we, wM, wN, 0, wK, w_tiles_n, blk - n_v, 0);
    else gemm_tile_bf<WM, 64, false, BwwEpi>(smem_b, wa, wb, we, wM, wN, 0, wK, w_tiles_n, blk - n_v, 0);
  } else if (blk < n_v + n_w + c.n_c) {
    const int t2 = blk - n_v - n_w;
    if (c.seg) {
      const int kb = t2 / c.per_bucket;
      const int k_lo = c.seg[kb], k_hi = c.seg[kb + 1];
      Mat a = c.a, b = c.b;
      a.bytes = (unsigned)k_hi * (unsigned)a.ld * 4u;
      b.bytes = (unsigned)k_hi * (unsigned)b.ld * 4u;
      CwEpi e = c.e;
      e.d_w1 += (size_t)kb * c.M * c.N;
      gemm_tile<32, 64, 128, false, false, CwEpi>(smem, a, b, e, c.M, c.N, k_lo, k_hi, c.tiles_n, t2 - kb * c.per_bucket, 0);
    } else {
      gemm_tile<32, 64, 128, false, false, CwEpi>(smem, c.a, c.b, c.e, c.M, c.N, 0, c.K, c.tiles_n, t2, 0);
    }
  } else {
    const int i = blk - n_w - n_v - c.n_c, n_t = t.col_blocks * (1 + t.zero_slices);
    if (i < n_t) tail_rows_block(t, i % t.col_blocks, i / t.col_blocks, smem);
    else small_wgrad_body(sw, i - n_t, smem);  // the classifier's small gradients + mean loss (blocks exist only when they ride here)
  }
}

// bits[b][p] = conv_out[b][p] > thr[channel] as a byte, n[b] = active positions, sink[b] = active positions >= F-1.
// grid (B, slices); integer-valued atomics into host-zeroed counters when a sample is split (exact in any order).
__global__ __launch_bounds__(256) void ftm_binarize_kernel(const float* __restrict__ conv_out, const float* __restrict__ thr,
                                                           int P, int G, int F, int slices, uint8_t* __restrict__ bits,
                                                           int* __restrict__ n, float* __restrict__ sink) {
  __shared__ int cnt_s[4], sink_s[4];
  const int b = blockIdx.x;
  const float4* __restrict__ x4 = reinterpret_cast<const float4*>(conv_out + (size_t)b * P);
  unsigned* __restrict__ o4 = reinterpret_cast<unsigned*>(bits + (size_t)b * P);
  int cnt = 0, snk = 0;
  for (int g = blockIdx.y * 256 + threadIdx.x; g < P / 4; g += 256 * slices) {
    const float4 x = x4[g];
    const int p = g * 4;
    int c = p / G, rem = p - c * G;
    float4 v;
    v.x = x.x > thr[c] ? 1.0f : 0.0f;
    if (++rem == G) { rem = 0; ++c; }
    v.y = x.y > thr[c] ? 1.0f : 0.0f;
    if (++rem == G) { rem = 0; ++c; }
    v.z = x.z > thr[c] ? 1.0f : 0.0f;
    if (++rem == G) { rem = 0; ++c; }
    v.w = x.w > thr[c] ? 1.0f : 0.0f;
    o4[g] = (unsigned)v.x | ((unsigned)v.y << 8) | ((unsigned)v.z << 16) | ((unsigned)v.w << 24);
    cnt += (int)(v.x + v.y + v.z + v.w);
    snk += (p >= F - 1 ? (int)v.x : 0) + (p + 1 >= F - 1 ? (int)v.y : 0) + (p + 2 >= F - 1 ? (int)v.z : 0) + (p + 3 >= F - 1 ? (int)v.w : 0);
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
    const int total = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3], st = sink_s[0] + sink_s[1] + sink_s[2] + sink_s[3];
    if (slices == 1) {
      n[b] = total;
      sink[b] = (float)st;
    } else {
      atomicAdd(&n[b], total);
      if (st) atomicAdd(&sink[b], (float)st);
    }
  }
}

// ---- ||A^T D||_F^2 without A^T D ---------------------------------------------------------------------------------
// The squared norm of the table's weight gradient d_W = A^T D (A [B][direct] the binary map, D = d_out [B][L1]) is
//   sum_{f,n} (sum_b A_bf D_bn)^2 = sum_{b,b'} (A A^T)_{bb'} (D D^T)_{bb'}
// -- two B x B Gram matrices instead of the [direct][L1] gradient, which lets clip_grad_norm_ (train.py:363-366) know
// its norm before that gradient exists, so the product that forms it can apply the update in its epilogue (BwwSgdEpi).
// A A^T: counts of common active positions, on the i8 matrix unit straight from the byte map.  Lane l of
// v_mfma_i32_16x16x64_i8 holds 16 consecutive k of row l & 15 for both operands -- exactly a 16-byte load of a map row, so
// there is no LDS image and no conversion; both operands are fetched the same way, so whatever order the instruction
// gives the 64 k of a step, the two sides agree.  A workgroup owns one 32 x 32 tile pair (tm >= tn: the matrix is
// symmetric) and one K slice, its four waves a quarter of the slice each; their accumulators are added through LDS and
// stored as one int32 slab, and gram_finish_kernel adds the slabs in slice order and writes both mirror positions.
// (The first form -- bf16 MFMA over LDS images, 128 x 128 tiles in 128 slices added with float atomics into a zeroed
// matrix -- spent 12.8 of its 23.5 us at the 224x224 shape on 2 M atomic additions into one 64 KB matrix; 64 x 64 tiles
// over more workgroups only moved the same additions around: 21-50 us for 1-4 M of them.)
using i32x4 = __attribute__((ext_vector_type(4))) int;
constexpr int kGramU = 4;  // 64-k steps whose loads are issued together

__device__ __forceinline__ void gram_pair(int x, int& tm, int& tn) {  // x enumerates the pairs tm >= tn
  tm = 0;
  while ((tm + 1) * (tm + 2) / 2 <= x) ++tm;
  tn = x - tm * (tm + 1) / 2;
}

// (grid rows at and beyond `slices` carry riders: the tail rows of the table's weight gradient, nnue_ftm_gram_sqnorm_tail -- work
// that like this product only needs d_out / the map and would otherwise be a launch of its own)
__global__ __launch_bounds__(256) void gram_i8_kernel(const uint8_t* __restrict__ bits, unsigned bytes, int B, int P, int direct, int wave_steps,
                                                      int* __restrict__ slabs, int slices, TailRows tr, int n_tail) {
  __shared__ int red[4][4][256];
  if ((int)blockIdx.y >= slices) {  // uniform
    const int idx = ((int)blockIdx.y - slices) * (int)gridDim.x + (int)blockIdx.x;
    if (idx < n_tail) tail_rows_block(tr, idx % tr.col_blocks, idx / tr.col_blocks, reinterpret_cast<float*>(&red[0][0][0]));
    return;
  }
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(bits), 0, bytes, 0x00020000);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  int tm, tn;
  gram_pair((int)blockIdx.x, tm, tn);
  const bool diag = tm == tn;  // uniform
  const int row_m = (tm * 32 + r) * P, row_n = (tn * 32 + r) * P;  // rows beyond the batch lie beyond the buffer: they read as zero
  const int k_lo = ((int)blockIdx.y * 4 + wave) * wave_steps * 64;
  const int k_end = (direct + 63) & ~63;
  const int k_hi = k_lo + wave_steps * 64 < k_end ? k_lo + wave_steps * 64 : k_end;
  i32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[i][t] = (i32x4){0, 0, 0, 0};
  auto clip = [&](i32x4 v, int k) {  // bytes at k .. k + 15 of a row: those at or beyond `direct` do not count
    i32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int valid = direct - (k + 4 * j);
      o[j] = valid >= 4 ? v[j] : valid <= 0 ? 0 : (int)((unsigned)v[j] & ((1u << (8 * valid)) - 1u));
    }
    return o;
  };
  for (int k0 = k_lo; k0 < k_hi; k0 += 64 * kGramU) {
    i32x4 am[kGramU][2], an[kGramU][2];
#pragma unroll
    for (int u = 0; u < kGramU; ++u) {
      const int k = k0 + 64 * u + 16 * q;
      const bool in = k0 + 64 * u < k_hi;  // uniform; a step past the slice reads beyond the buffer (zeros)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        am[u][i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, in ? row_m + 16 * i * P + k : 0x7ffffff0, 0, 0));
        if (!diag) an[u][i] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, in ? row_n + 16 * i * P + k : 0x7ffffff0, 0, 0));
      }
    }
#pragma unroll
    for (int u = 0; u < kGramU; ++u) {
      const int k = k0 + 64 * u + 16 * q;
      if (k0 + 64 * u + 64 > direct) {  // uniform: the step that holds position `direct`
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          am[u][i] = clip(am[u][i], k);
          if (!diag) an[u][i] = clip(an[u][i], k);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          acc[i][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(am[u][i], diag ? am[u][t] : an[u][t], acc[i][t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][i * 2 + t][lane * 4 + e] = acc[i][t][e];
  __syncthreads();
  int* __restrict__ out = slabs + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 1024;
#pragma unroll
  for (int tile = 0; tile < 4; ++tile) out[tile * 256 + tid] = (red[0][tile][tid] + red[1][tile][tid]) + (red[2][tile][tid] + red[3][tile][tid]);
}

// G[m][n] = G[n][m] = sum over the K slices of the pair's slab element; grid (pairs * 4): one workgroup per 16 x 16 sub-tile
__global__ __launch_bounds__(256) void gram_finish_kernel(const int* __restrict__ slabs, int pairs, int slices, int B, float* __restrict__ G) {
  const int pair = blockIdx.x >> 2, tile = blockIdx.x & 3, tid = threadIdx.x;
  int tm, tn;
  gram_pair(pair, tm, tn);
  const int* __restrict__ src = slabs + (size_t)pair * 1024 + tile * 256 + tid;
  int acc = 0, sl = 0;
  for (; sl + 8 <= slices; sl += 8) {
    int v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(sl + u) * pairs * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; sl < slices; ++sl) acc += src[(size_t)sl * pairs * 1024];
  // slab element [tile = 2 i + t][lane * 4 + e]: accumulator register e of lane 16 q + r is row 4 q + e, column r
  const int ln = tid >> 2, e = tid & 3, i = tile >> 1, t = tile & 1;
  const int m = tm * 32 + 16 * i + 4 * (ln >> 4) + e, n = tn * 32 + 16 * t + (ln & 15);
  if (m < B && n < B) {
    G[(size_t)m * B + n] = (float)acc;
    if (tm != tn) G[(size_t)n * B + m] = (float)acc;
  }
}

// sum_{b,b'} G_A[b][b'] (D D^T)[b][b'] = sum_{b,n} (G_A D)[b][n] D[b][n]: one wave per 16 x 16 tile of T = G_A D (f32 MFMA,
// K = B), multiplied element-wise with the same tile of D and summed in float64 -> partial[tile].  Unlike D D^T this
// splits over the L1 columns too (B/16 x L1/16 independent waves), and D D^T is never formed.
__global__ __launch_bounds__(256) void gram_apply_kernel(const float* __restrict__ G, const float* __restrict__ D, int B, int L1,
                                                         float* __restrict__ partial) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int tiles_m = (B + 15) / 16, tiles_n = (L1 + 15) / 16;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= tiles_m * tiles_n) return;
  const int ti = tile / tiles_n, tj = tile % tiles_n;
  const int row = ti * 16 + r, col = tj * 16 + r;
  const float* __restrict__ ga = G + (size_t)(row < B ? row : 0) * B;  // A operand: G_A[row][k], k contiguous (B % 4 == 0 not needed: scalar tail)
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;  // k blocks of 4 whose loads are issued together
  for (int k0 = 0; k0 < B; k0 += 4 * U) {
    float a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 4 * u + q;
      const bool in = k < B;
      a[u] = (in && row < B) ? ga[k] : 0.0f;
      b[u] = (in && col < L1) ? D[(size_t)k * L1 + col] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int m = ti * 16 + 4 * q + e;
    if (m < B && col < L1) s += (double)acc[e] * (double)D[(size_t)m * L1 + col];
  }
#pragma unroll
  for (int sh = 32; sh >= 1; sh >>= 1) s += __shfl_xor(s, sh);
  if (lane == 0) partial[tile] = (float)s;
}

// ---- launch policy -------------------------------------------------------------------------------------------
int env_int(const char* name, int fallback) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : fallback;
}

// Tile shapes (BM, BN, BK).  BK grows as the tile shrinks so that a K tile always carries >= 2048 MFMA cycles/wave.
struct Shape {
  int cfg;  // 0: 32x64x128   1: 64x64x64   2: 128x64x32   3: 64x128x32
  int bm, bn, bk, tiles_m, tiles_n, ksplit, klen;
};
constexpr int kCfg[9][3] = {{32, 64, 128}, {64, 64, 64}, {128, 64, 32}, {64, 128, 32}, {128, 64, 64}, {64, 128, 64},
                            {32, 64, 128}, {64, 64, 128}, {128, 64, 128}};  // 6..8: bf16-split tiles (gemm_tile_bf)

// The two products over the binary map run on the bf16 matrix unit (exact three-way split of the f32 operand) unless
// NNUE_FTM_BF16=0; NNUE_FTM_BF_BM forces the bf16 tile height (developer knob).
bool use_bf16() { return env_int("NNUE_FTM_BF16", 1) != 0; }  // read per call: tests switch it inside one process

// prefer_m: the operand re-read per M tile is the big one (the table, in forward and value gradient), so take tall
// tiles; otherwise (weight gradient: the map is re-read per N tile) take wide ones.  The largest preferred shape
// that still gives every CU a workgroup wins; the forward of a small batch (M <= 128: one tall tile covers it, the
// table is then read exactly once) splits K until there are about two workgroups per CU.
Shape plan(int M, int N, int K, bool prefer_m, bool allow_split, bool bf_ok = false) {
  static const int force_cfg = env_int("NNUE_FTM_CFG", -1), force_split = env_int("NNUE_FTM_KSPLIT", 0);
  const int force_bf_bm = env_int("NNUE_FTM_BF_BM", 0);
  const int order_m[3] = {2, 1, 0}, order_n[3] = {3, 1, 0};
  const int* order = prefer_m ? order_m : order_n;
  auto tiles = [&](int c) { return (long long)((M + kCfg[c][0] - 1) / kCfg[c][0]) * ((N + kCfg[c][1] - 1) / kCfg[c][1]); };
  int cfg = order[2];
  for (int i = 0; i < 3; ++i)
    if (tiles(order[i]) >= 256) { cfg = order[i]; break; }
  const bool small_m = allow_split && M <= 128;
  if (small_m) cfg = 2;
  if (force_cfg >= 0 && force_cfg < 6) cfg = force_cfg;
  if (bf_ok && use_bf16()) {
    // The split of the f32 operand (5.5 VALU per value) is amortised over the tile's rows, so only tall tiles pay:
    // 128 or 64 rows where they still fill the chip, the split-K forward of a small batch (128 rows).  Launch-sized
    // products that would need 32-row tiles stay on the f32 MFMA (measured at the CIFAR batch-512 shape: forward 21.8 us
    // with 32-row bf16 tiles, 25.7 with 64-row ones that leave half the CUs idle, 20.3 with the f32 kernel).
    for (int c = 8; c >= 7; --c)
      if (tiles(c) >= 256) { cfg = c; break; }
    // split-K forward of a small batch: only where K still gives the 128-deep bf16 tiles many slabs (the 65 536-row table);
    // at K = 799 the f32 tiles (K tile 32: six slabs) fill more of the chip than one workgroup walking seven bf16 K tiles
    if (small_m) cfg = (K >= 4096) ? 8 : 2;
    if (force_bf_bm == 32 || force_bf_bm == 64 || force_bf_bm == 128) cfg = force_bf_bm == 32 ? 6 : force_bf_bm == 64 ? 7 : 8;
  }
  Shape s;
  s.cfg = cfg;
  s.bm = kCfg[cfg][0]; s.bn = kCfg[cfg][1]; s.bk = kCfg[cfg][2];
  s.tiles_m = (M + s.bm - 1) / s.bm;
  s.tiles_n = (N + s.bn - 1) / s.bn;
  const long long blocks = (long long)s.tiles_m * s.tiles_n;
  const int ktiles = (K + s.bk - 1) / s.bk;
  int split = 1;
  if (allow_split && (small_m || force_split)) {
    split = force_split ? force_split : (int)((512 + blocks - 1) / blocks);
    if (split > ktiles / 4) split = ktiles / 4;  // at least four K tiles per slab
    if (split > 64) split = 64;
    if (split < 1) split = 1;
  }
  const int tiles_per = (ktiles + split - 1) / split;
  s.klen = tiles_per * s.bk;
  s.ksplit = (ktiles + tiles_per - 1) / tiles_per;
  return s;
}

template <bool AKC, bool BKC, class Epi>
void launch(hipStream_t st, const Shape& s, Mat ma, Mat mb, Epi epi, int M, int N, int K, GroupArgs ga = GroupArgs{}) {
  const dim3 grid((unsigned)(s.tiles_m * s.tiles_n) + (ga.n ? 1u : 0u), (unsigned)s.ksplit);
#define NNUE_FTM_LAUNCH(BM, BN, BK)                                                                                           \
  hipLaunchKernelGGL((ftm_gemm_kernel<BM, BN, BK, AKC, BKC, Epi>), grid, dim3(256), 0, st, ma, mb, epi, M, N, K, s.klen, s.tiles_n, ga)
  switch (s.cfg) {
    case 0: NNUE_FTM_LAUNCH(32, 64, 128); break;
    case 1: NNUE_FTM_LAUNCH(64, 64, 64); break;
    case 2: NNUE_FTM_LAUNCH(128, 64, 32); break;
    case 3: NNUE_FTM_LAUNCH(64, 128, 32); break;
    case 4: NNUE_FTM_LAUNCH(128, 64, 64); break;
    case 5: NNUE_FTM_LAUNCH(64, 128, 64); break;
    default:
      if constexpr (!BKC && Epi::kAU8) {  // bf16-split tiles: the map times an f32 operand that is contiguous along its rows
#define NNUE_FTM_LAUNCH_BF(BM) \
  hipLaunchKernelGGL((ftm_gemm_bf_kernel<BM, 64, AKC, Epi>), grid, dim3(256), 0, st, ma, mb, epi, M, N, K, s.klen, s.tiles_n, ga)
        // 128-row tiles of the big-table kernels (split-K forward, update in the epilogue) with K tiles of 64: 40 KB and
        // 124 / 152 registers instead of 80 KB and ~190 / 200 -- four / three workgroups per CU instead of two: forward 67.5 -> 64 us,
        // update 192.5 -> 184 us at the 224x224 shape (A/B in one run)
        static const int kt64 = env_int("NNUE_FTM_BF_KT64", 1);  // developer knob
        if (s.cfg == 6) NNUE_FTM_LAUNCH_BF(32);
        else if (s.cfg == 7) NNUE_FTM_LAUNCH_BF(64);
        else if (kt64 && !ga.n && (std::is_same<Epi, FwdEpi>::value || is_rmw<Epi>::value))
          hipLaunchKernelGGL((ftm_gemm_bf64_kernel<AKC, Epi>), grid, dim3(256), 0, st, ma, mb, epi, M, N, K, s.klen, s.tiles_n);
        else NNUE_FTM_LAUNCH_BF(128);
#undef NNUE_FTM_LAUNCH_BF
      }
      break;
  }
#undef NNUE_FTM_LAUNCH
}

// every operand is addressed with 32-bit byte offsets (tile overhang included)
TailRows tail_rows(const float* d_out, const float* sink, int B, int L1, int direct, int F, float* d_weight, float* d_bias) {
  int zero_slices = d_weight ? (F - 1 - direct + 255) / 256 : 0;
  zero_slices = zero_slices > 256 ? 256 : zero_slices;
  return TailRows{d_out, sink, B, L1, direct, F, d_weight, d_bias, (L1 + 15) / 16, zero_slices};
}

bool shape_ok(int B, int F, int P, int L1) {
  const long long lim = (1ll << 31) - 1;
  return B > 0 && F > 0 && P > 0 && L1 > 0 && ((long long)B + 256) * P * 4 < lim && ((long long)F + 256) * L1 * 4 < lim &&
         ((long long)B + 256) * L1 * 4 < lim && ((long long)P + 256) * L1 * 4 < lim;
}

constexpr int kIntMax = 0x7fffffff;

}  // namespace

// =============================================================================== C ABI
extern "C" int nnue_ftm_supported(int F, int P, int L1) {
  // 16-byte staging along positions and along table columns; 32-bit byte offsets into the table (tile overhang included)
  const long long lim = (1ll << 31) - 1;
  return F > 0 && P > 0 && P % 4 == 0 && L1 > 0 && L1 % 4 == 0 && ((long long)F + 256) * L1 * 4 < lim && ((long long)P + 256) * L1 * 4 < lim;
}

extern "C" int64_t nnue_ftm_scratch(int B, int F, int P, int L1) {
  if (B <= 0 || F <= 0 || P <= 0 || L1 <= 0) return 0;
  const int direct = (F - 1 < P) ? F - 1 : P;
  const Shape s = plan(B, L1, direct > 0 ? direct : 1, true, true, true);
  return s.ksplit > 1 ? (int64_t)s.ksplit * B * L1 * (int64_t)sizeof(float) : 0;
}

extern "C" int nnue_ftm_binarize(const float* conv_out, const float* thr, int B, int fps, int Gh, int Gw, int F, uint8_t* bits,
                                 int32_t* n, float* sink, nnue_stream_t stream) {
  NNUE_REQUIRE(conv_out && thr && bits && n && sink, NNUE_E_ARG, "nnue_ftm_binarize: null pointer");
  NNUE_REQUIRE(B > 0 && fps > 0 && Gh > 0 && Gw > 0 && F > 0 && (long long)fps * Gh * Gw < (1ll << 30), NNUE_E_ARG,
               "nnue_ftm_binarize: B=%d fps=%d Gh=%d Gw=%d F=%d out of range", B, fps, Gh, Gw, F);
  const int P = fps * Gh * Gw;
  NNUE_REQUIRE(P % 4 == 0, NNUE_E_SHAPE, "nnue_ftm_binarize: P=%d must be a multiple of 4", P);
  NNUE_REQUIRE(nnue_aligned16(conv_out) && nnue_aligned16(bits), NNUE_E_ARG, "nnue_ftm_binarize: pointers must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  int slices = P / 8192;
  slices = slices < 1 ? 1 : (slices > 32 ? 32 : slices);
  if (slices > 1) nnue_zero_counters(n, sink, B, s);  // a kernel, not a memset node (common.h)
  hipLaunchKernelGGL(ftm_binarize_kernel, dim3(B, slices), dim3(256), 0, s, conv_out, thr, P, Gh * Gw, F, slices, bits, n, sink);
  return nnue_launch_status("nnue_ftm_binarize");
}

namespace {
int ftm_forward_impl(const uint8_t* bits, const float* sink, const float* weight, const float* bias, int B, int F, int P, int L1, float* out,
                     void* scratch, int64_t scratch_bytes, GroupArgs ga, nnue_stream_t stream);
}
extern "C" int nnue_ftm_forward(const uint8_t* bits, const float* sink, const float* weight, const float* bias, int B, int F, int P,
                                int L1, float* out, void* scratch, int64_t scratch_bytes, nnue_stream_t stream) {
  return ftm_forward_impl(bits, sink, weight, bias, B, F, P, L1, out, scratch, scratch_bytes, GroupArgs{}, stream);
}
extern "C" int nnue_ftm_forward_grouping(const uint8_t* bits, const float* sink, const float* weight, const float* bias, int B, int F, int P,
                                         int L1, float* out, void* scratch, int64_t scratch_bytes, const int32_t* n, int K, int32_t* bucket,
                                         int32_t* rows, int32_t* tile_bucket, int32_t* seg, nnue_stream_t stream) {
  NNUE_REQUIRE(n && bucket && rows && tile_bucket && seg, NNUE_E_ARG, "nnue_ftm_forward_grouping: null pointer");
  NNUE_REQUIRE(K >= 1 && K <= kMaxBuckets && B > 0 && P > 0, NNUE_E_ARG, "nnue_ftm_forward_grouping: K=%d B=%d P=%d out of range", K, B, P);
  return ftm_forward_impl(bits, sink, weight, bias, B, F, P, L1, out, scratch, scratch_bytes,
                          GroupArgs{n, B, P, K, bucket, rows, tile_bucket, seg, (B + 15) / 16 + K}, stream);
}
namespace {
int ftm_forward_impl(const uint8_t* bits, const float* sink, const float* weight, const float* bias, int B, int F, int P, int L1, float* out,
                     void* scratch, int64_t scratch_bytes, GroupArgs ga, nnue_stream_t stream) {
  NNUE_REQUIRE(bits && sink && weight && bias && out, NNUE_E_ARG, "nnue_ftm_forward: null pointer");
  NNUE_REQUIRE(shape_ok(B, F, P, L1), NNUE_E_ARG, "nnue_ftm_forward: B=%d F=%d P=%d L1=%d out of range", B, F, P, L1);
  NNUE_REQUIRE(nnue_ftm_supported(F, P, L1), NNUE_E_SHAPE, "nnue_ftm_forward: P=%d and L1=%d must be multiples of 4", P, L1);
  NNUE_REQUIRE(nnue_aligned16(bits) && nnue_aligned16(weight) && nnue_aligned16(bias) && nnue_aligned16(out), NNUE_E_ARG,
               "nnue_ftm_forward: pointers must be 16-byte aligned");
  const int direct = (F - 1 < P) ? F - 1 : P;
  const int K = direct > 0 ? direct : 1;  // F == 1: only the sink row; the product runs over one zero column
  const Shape s = plan(B, L1, K, true, true, true);
  const int64_t need = s.ksplit > 1 ? (int64_t)s.ksplit * B * L1 * (int64_t)sizeof(float) : 0;
  NNUE_REQUIRE(scratch_bytes >= need && (need == 0 || (scratch && nnue_aligned16(scratch))), NNUE_E_SCRATCH,
               "nnue_ftm_forward: scratch %lld < %lld bytes", (long long)scratch_bytes, (long long)need);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* dst = s.ksplit > 1 ? static_cast<float*>(scratch) : out;
  const float* w_last = weight + (size_t)(F - 1) * L1;
  // A = the map (its K tail needs no zeroing: the table window ends at row `direct`, so B is zero there)
  launch<true, false>(st, s, Mat{bits, (unsigned)((size_t)B * P), P, kIntMax, kIntMax},
                      Mat{weight, (unsigned)((size_t)direct * L1 * 4), L1, kIntMax, kIntMax}, FwdEpi{bias, w_last, sink, dst, B, L1, s.ksplit}, B,
                      L1, K, ga);
  if (s.ksplit > 1) {
    const int64_t count4 = (int64_t)B * L1 / 4;
    hipLaunchKernelGGL(ftm_finish_kernel, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, st, dst, s.ksplit, count4, bias, w_last, sink,
                       L1, out);
  }
  return nnue_launch_status("nnue_ftm_forward");
}
}  // namespace

namespace {
int backward_weight_impl(const uint8_t* bits, const float* sink, const float* d_out, int B, int F, int P, int L1, float* d_weight,
                         float* d_bias, float* sq, nnue_stream_t stream);
}
extern "C" int nnue_ftm_backward_weight(const uint8_t* bits, const float* sink, const float* d_out, int B, int F, int P, int L1,
                                        float* d_weight, float* d_bias, nnue_stream_t stream) {
  return backward_weight_impl(bits, sink, d_out, B, F, P, L1, d_weight, d_bias, nullptr, stream);
}
namespace {
int backward_weight_impl(const uint8_t* bits, const float* sink, const float* d_out, int B, int F, int P, int L1, float* d_weight,
                         float* d_bias, float* sq, nnue_stream_t stream) {
  NNUE_REQUIRE(bits && sink && d_out, NNUE_E_ARG, "nnue_ftm_backward_weight: null pointer");
  NNUE_REQUIRE(d_weight || d_bias, NNUE_E_ARG, "nnue_ftm_backward_weight: both outputs are null");
  NNUE_REQUIRE(shape_ok(B, F, P, L1), NNUE_E_ARG, "nnue_ftm_backward_weight: B=%d F=%d P=%d L1=%d out of range", B, F, P, L1);
  NNUE_REQUIRE(nnue_ftm_supported(F, P, L1), NNUE_E_SHAPE, "nnue_ftm_backward_weight: P=%d and L1=%d must be multiples of 4", P, L1);
  NNUE_REQUIRE(nnue_aligned16(bits) && nnue_aligned16(d_out), NNUE_E_ARG, "nnue_ftm_backward_weight: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int direct = (F - 1 < P) ? F - 1 : P;
  if (d_weight && direct > 0) {
    const Shape s = plan(direct, L1, B, false, false, true);
    launch<false, false>(st, s, Mat{bits, (unsigned)((size_t)B * P), P, kIntMax, kIntMax},
                         Mat{d_out, (unsigned)((size_t)B * L1 * 4), L1, kIntMax, kIntMax}, BwwEpi{d_weight, L1, s.ksplit == 1 ? sq : nullptr},
                         direct, L1, B);
  }
  const TailRows t = tail_rows(d_out, sink, B, L1, direct, F, d_weight, d_bias);
  hipLaunchKernelGGL(ftm_tail_rows_kernel, dim3(t.col_blocks, 1 + t.zero_slices), dim3(256), 0, st, t);
  return nnue_launch_status("nnue_ftm_backward_weight");
}
}  // namespace

namespace {
// the stand-alone value gradient as six bf16 plane products (gemm_tile_bf6): the big-map shape's 128 x 64 tiles
bool values_bf6(int B, int P, int L1) {
  static const int bf6 = env_int("NNUE_FTM_VAL_BF6", 1);  // developer knob
  const Shape s = plan(B, P, L1, true, false);
  return bf6 && use_bf16() && s.cfg == 2 && s.ksplit == 1 && L1 % 8 == 0;
}
}  // namespace

extern "C" int nnue_ftm_backward_values(const uint8_t* bits, const float* d_out, const float* weight, int B, int F, int P, int L1,
                                        float* d_conv_out, nnue_stream_t stream) {
  NNUE_REQUIRE(bits && d_out && weight && d_conv_out, NNUE_E_ARG, "nnue_ftm_backward_values: null pointer");
  NNUE_REQUIRE(shape_ok(B, F, P, L1), NNUE_E_ARG, "nnue_ftm_backward_values: B=%d F=%d P=%d L1=%d out of range", B, F, P, L1);
  NNUE_REQUIRE(nnue_ftm_supported(F, P, L1), NNUE_E_SHAPE, "nnue_ftm_backward_values: P=%d and L1=%d must be multiples of 4", P, L1);
  NNUE_REQUIRE(nnue_aligned16(d_out) && nnue_aligned16(weight), NNUE_E_ARG, "nnue_ftm_backward_values: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const Shape s = plan(B, P, L1, true, false);
  // Stays on the f32 MFMA: both operands are f32, so the bf16 unit needs both splits and six plane products per pair
  // (hh, hm, mh, hl, lh, mm).  Built and measured in round 2 with the truncation split (4.5 VALU per value), 128 x 64 x 128
  // tiles, six LDS images (144 KB, one workgroup per CU): 178 us against 153 us for this kernel at the 224x224 shape --
  // with one workgroup per CU nothing overlaps the split with the MFMA phase.  (Round 1's attempt with a rounding split: 145-160.)
  // K = L1 runs along the inner index of both operands: A is zeroed past it, the table rows are clamped to F-1
  const Mat ma{d_out, (unsigned)((size_t)B * L1 * 4), L1, kIntMax, L1}, mb{weight, (unsigned)((size_t)F * L1 * 4), L1, F - 1, kIntMax};
  const ValEpi epi{bits, d_conv_out, P};
  // (128 x 128 x 32 tiles -- every workgroup stages the same d_out rows beside its table rows, so wider tiles cut the bytes
  // through the CUs' load path from 3x to 2x the table's -- were measured at the 224x224 shape and lose: 152.0 vs 147.1 us;
  // 184 registers leave two workgroups per CU instead of four.)
  if (values_bf6(B, P, L1)) {  // the big-map shape: 128 x 64 tiles, six bf16 plane products (147 -> 117 us at the 224x224 shape)
    // K tiles of 32: six images of a 128 x 64 tile are 36 KB and 120 registers -- four workgroups per CU -- 100 us against
    // 115 us with K tiles of 64 (72 KB, two per CU) and 147 us on the f32 MFMA at the 224x224 shape
    static const int kt = env_int("NNUE_FTM_BF6_KT", 32);  // developer knob
    const dim3 grid((unsigned)(s.tiles_m * s.tiles_n));
#ifdef NNUE_ABLATIONS  // timing-only ablations (WRONG results), tools/debug: compiled only with NNUE_BUILD_ABLATIONS=1 (csrc/build.py)
    static const int abl = env_int("NNUE_FTM_BF6_ABL", 0);
#define NNUE_ABL(N) if (abl == N) { hipLaunchKernelGGL((ftm_gemm_bf6_kernel<128, 64, ValEpi, 32, N>), grid, dim3(256), 0, st, ma, mb, epi, B, P, L1, s.tiles_n); return nnue_launch_status("nnue_ftm_backward_values"); }
    NNUE_ABL(1) NNUE_ABL(2) NNUE_ABL(3) NNUE_ABL(4) NNUE_ABL(5)
#undef NNUE_ABL
#endif
    static const int bn = env_int("NNUE_FTM_BF6_BN", 64);  // developer knob: 128-column tiles stage d_out half as often
    if (bn == 128) {
      const int tn = (P + 127) / 128;
      hipLaunchKernelGGL((ftm_gemm_bf6_kernel<128, 128, ValEpi, 32>), dim3((unsigned)(s.tiles_m * tn)), dim3(256), 0, st, ma, mb, epi, B, P, L1, tn);
      return nnue_launch_status("nnue_ftm_backward_values");
    }
    if (kt == 64) hipLaunchKernelGGL((ftm_gemm_bf6_kernel<128, 64, ValEpi, 64>), grid, dim3(256), 0, st, ma, mb, epi, B, P, L1, s.tiles_n);
    else hipLaunchKernelGGL((ftm_gemm_bf6_kernel<128, 64, ValEpi, 32>), grid, dim3(256), 0, st, ma, mb, epi, B, P, L1, s.tiles_n);
    return nnue_launch_status("nnue_ftm_backward_values");
  }
  launch<true, true>(st, s, ma, mb, epi, B, P, L1);
  return nnue_launch_status("nnue_ftm_backward_values");
}

// The same value gradient with a workspace: big maps take ftv_kernels.hip (d_out split once into bf16 planes in the
// workspace and staged by LDS-DMA, table fragments straight into registers); every other shape, or a call without enough
// workspace for it, runs nnue_ftm_backward_values unchanged.
namespace {
// the six-plane tile kernel fed with d_out planes from the workspace (K in whole tiles of 32; offsets fit 31 bits)
bool values_planes(int B, int P, int L1) {
  // Off by default: measured in the 224x224 step this is SLOWER than splitting d_out in every workgroup (125.6 vs 98.8 us): three
  // 16-byte plane loads per run instead of two float loads put 1.36x the bytes through the CUs' load path, which is what the tile
  // kernel waits for -- not its split VALU.  Read per call (tests switch it).
  const char* e = getenv("NNUE_FTM_VAL_PLANES");
  return (e ? atoi(e) : 0) && values_bf6(B, P, L1) && L1 % 32 == 0 && (long long)3 * B * L1 * 2 < (1ll << 31);
}
}  // namespace

extern "C" int64_t nnue_ftm_backward_values_scratch(int B, int F, int P, int L1) {
  if (!nnue_ftm_supported(F, P, L1) || !shape_ok(B, F, P, L1)) return 0;
  if (values_bf6(B, P, L1) && ftv_supported(B, F, P, L1)) return ftv_scratch_bytes(B, L1);
  return values_planes(B, P, L1) ? ftv_scratch_bytes(B, L1) : 0;
}

extern "C" int nnue_ftm_backward_values_ws(const uint8_t* bits, const float* d_out, const float* weight, int B, int F, int P, int L1,
                                           float* d_conv_out, void* scratch, int64_t scratch_bytes, nnue_stream_t stream) {
  const int64_t need = nnue_ftm_backward_values_scratch(B, F, P, L1);
  if (need == 0) return nnue_ftm_backward_values(bits, d_out, weight, B, F, P, L1, d_conv_out, stream);
  NNUE_REQUIRE(bits && d_out && weight && d_conv_out, NNUE_E_ARG, "nnue_ftm_backward_values_ws: null pointer");
  NNUE_REQUIRE(scratch && scratch_bytes >= need, NNUE_E_SCRATCH, "nnue_ftm_backward_values_ws: workspace of %lld bytes needed, %lld given",
               (long long)need, (long long)scratch_bytes);
  NNUE_REQUIRE(nnue_aligned16(d_out) && nnue_aligned16(weight) && nnue_aligned16(scratch), NNUE_E_ARG,
               "nnue_ftm_backward_values_ws: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (values_bf6(B, P, L1) && ftv_supported(B, F, P, L1)) {
    ftv_launch(bits, d_out, weight, B, F, P, L1, d_conv_out, scratch, st);
    return nnue_launch_status("nnue_ftm_backward_values_ws");
  }
  // d_out split once per launch; the 128 x 64 x 32 six-plane tiles take its planes as they are (no split VALU for operand A)
  ftv_split_planes(d_out, B, L1, scratch, st);
  const Shape s = plan(B, P, L1, true, false);
  const Mat ma{scratch, (unsigned)((size_t)3 * B * L1 * 2), L1, kIntMax, L1}, mb{weight, (unsigned)((size_t)F * L1 * 4), L1, F - 1, kIntMax};
  const ValEpi epi{bits, d_conv_out, P};
  hipLaunchKernelGGL((ftm_gemm_bf6_kernel<128, 64, ValEpi, 32, 0, true>), dim3((unsigned)(s.tiles_m * s.tiles_n)), dim3(256), 0, st, ma, mb, epi, B, P,
                     L1, s.tiles_n);
  return nnue_launch_status("nnue_ftm_backward_values_ws");
}

// Both gradients of the binary-map FeatureTransformer in one launch (see ftm_backward_kernel); falls back to the two
// separate launches for tile-shape pairs that are not instantiated.
namespace {
// the shape pair the merged launch is used for (also the condition for the d_w1 tile family to ride along)
// weight-gradient tile height of the merged launch when it runs on the bf16 matrix unit (0: f32 tiles)
int merged_bf_wm() {
  const int wm = env_int("NNUE_FTM_BF_WM", 64);
  return use_bf16() ? (wm == 32 ? 32 : 64) : 0;
}

bool merged_backward_shape(int B, int F, int P, int L1, bool* big) {
  const int direct = (F - 1 < P) ? F - 1 : P;
  if (direct <= 0) return false;
  static const int split_launch = env_int("NNUE_FTM_SPLIT_BACKWARD", 0);
  static const int force_pair = env_int("NNUE_FTM_BWD_PAIR", 0);  // developer knob: 11 forces 64x64x64, 1 forbids it
  const Shape sw = plan(direct, L1, B, false, false), sv = plan(B, P, L1, true, false);
  const long long tiles64 = (long long)((direct + 63) / 64) * ((L1 + 63) / 64) + (long long)((B + 63) / 64) * ((P + 63) / 64);
  const bool small_pair = sw.cfg == 0 && (sv.cfg == 0 || sv.cfg == 1);
  *big = force_pair == 11 || (force_pair != 1 && small_pair && tiles64 >= 448);
  return !split_launch && (small_pair || *big);
}
// value tiles of the merged launch as six bf16 plane products: for the 64 x 64 tiles (C3 shapes: 43.5 vs 46.0 us for the
// launch); the 32 x 64 tiles of the batch-512 shape are latency-bound per K tile and lose with the 64-deep bf16 tiles (27.2 vs 26.3 us)
bool merged_values_bf6(int B, int F, int P, int L1) {
  static const int bf6 = env_int("NNUE_FTM_BWD_BF6", 1);  // developer knob
  bool big = false;
  if (!merged_backward_shape(B, F, P, L1, &big) || !merged_bf_wm()) return false;
  const bool v_small = !big && plan(B, P, L1, true, false).cfg == 0;
  return bf6 && !v_small && L1 % 8 == 0;
}
}  // namespace

// Number of squared-norm partials nnue_ftm_backward leaves (one per weight-gradient tile), 0 when the product is split
// along K (never for this operand order today) or the shape is not taken.
extern "C" int64_t nnue_ftm_backward_sq_count(int B, int F, int P, int L1) {
  if (!nnue_ftm_supported(F, P, L1) || !shape_ok(B, F, P, L1)) return 0;
  const int direct = (F - 1 < P) ? F - 1 : P;
  if (direct <= 0) return 0;
  bool big = false;
  if (merged_backward_shape(B, F, P, L1, &big)) {
    if (const int wm = merged_bf_wm()) return (int64_t)((direct + wm - 1) / wm) * ((L1 + 63) / 64);
    if (big) return (int64_t)((direct + 63) / 64) * ((L1 + 63) / 64);
    const Shape s = plan(direct, L1, B, false, false);
    return (int64_t)s.tiles_m * s.tiles_n;
  }
  const Shape s = plan(direct, L1, B, false, false, true);
  return s.ksplit == 1 ? (int64_t)s.tiles_m * s.tiles_n : 0;
}

extern "C" int nnue_ftm_backward_cw_supported(int B, int F, int P, int L1, int L2) {
  bool big = false;
  return nnue_ftm_supported(F, P, L1) && shape_ok(B, F, P, L1) && merged_backward_shape(B, F, P, L1, &big) && L1 % 128 == 0 && L2 > 0 &&
         L2 % 4 == 0 && ((long long)B + 256) * L2 * 4 < (1ll << 31);
}

namespace {
int ftm_backward_impl(const uint8_t* bits, const float* sink, const float* d_out, const float* weight, int B, int F, int P, int L1,
                      float* d_weight, float* d_bias, float* d_conv_out, const float* ft, const float* d_z1, int L2, float* d_w1,
                      float* sq_partial, int K, const int32_t* seg, int grouped_rows, const nnue_cls_rider* small, nnue_stream_t stream);
}
extern "C" int nnue_ftm_backward(const uint8_t* bits, const float* sink, const float* d_out, const float* weight, int B, int F, int P,
                                 int L1, float* d_weight, float* d_bias, float* d_conv_out, const float* ft, const float* d_z1, int L2,
                                 float* d_w1, float* sq_partial, const nnue_cls_rider* small, nnue_stream_t stream) {
  return ftm_backward_impl(bits, sink, d_out, weight, B, F, P, L1, d_weight, d_bias, d_conv_out, ft, d_z1, L2, d_w1, sq_partial, 1, nullptr, 0,
                           small, stream);
}
extern "C" int nnue_ftm_backward_bucketed(const uint8_t* bits, const float* sink, const float* d_out, const float* weight, int B, int F,
                                          int P, int L1, float* d_weight, float* d_bias, float* d_conv_out, const float* ft_grouped,
                                          const float* d_z1_grouped, int L2, float* d_w1, float* sq_partial, int K, const int32_t* seg,
                                          int grouped_rows, const nnue_cls_rider* small, nnue_stream_t stream) {
  NNUE_REQUIRE(K >= 1 && K <= 64, NNUE_E_ARG, "nnue_ftm_backward_bucketed: K=%d out of range", K);
  NNUE_REQUIRE(K == 1 || !d_w1 || (seg && grouped_rows >= B && grouped_rows % 16 == 0), NNUE_E_ARG,
               "nnue_ftm_backward_bucketed: d_w1 for K > 1 needs seg and the grouped row count (a multiple of 16, >= B)");
  return ftm_backward_impl(bits, sink, d_out, weight, B, F, P, L1, d_weight, d_bias, d_conv_out, ft_grouped, d_z1_grouped, L2, d_w1, sq_partial,
                           K, K > 1 ? seg : nullptr, K > 1 ? grouped_rows : 0, small, stream);
}
namespace {
int ftm_backward_impl(const uint8_t* bits, const float* sink, const float* d_out, const float* weight, int B, int F, int P, int L1,
                      float* d_weight, float* d_bias, float* d_conv_out, const float* ft, const float* d_z1, int L2, float* d_w1,
                      float* sq_partial, int K, const int32_t* seg, int grouped_rows, const nnue_cls_rider* small, nnue_stream_t stream) {
  NNUE_REQUIRE(bits && sink && d_out && weight && d_weight && d_bias && d_conv_out, NNUE_E_ARG, "nnue_ftm_backward: null pointer");
  const bool want_cw = d_w1 != nullptr;
  NNUE_REQUIRE(!want_cw || (ft && d_z1 && nnue_ftm_backward_cw_supported(B, F, P, L1, L2) && nnue_aligned16(ft) && nnue_aligned16(d_z1)),
               NNUE_E_SHAPE, "nnue_ftm_backward: d_w1 requested for a shape / pointers the merged launch does not take (nnue_ftm_backward_cw_supported)");
  NNUE_REQUIRE(shape_ok(B, F, P, L1), NNUE_E_ARG, "nnue_ftm_backward: B=%d F=%d P=%d L1=%d out of range", B, F, P, L1);
  NNUE_REQUIRE(nnue_ftm_supported(F, P, L1), NNUE_E_SHAPE, "nnue_ftm_backward: P=%d and L1=%d must be multiples of 4", P, L1);
  NNUE_REQUIRE(nnue_aligned16(bits) && nnue_aligned16(d_out) && nnue_aligned16(weight), NNUE_E_ARG,
               "nnue_ftm_backward: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int direct = (F - 1 < P) ? F - 1 : P;
  const Shape sw = plan(direct > 0 ? direct : 1, L1, B, false, false), sv = plan(B, P, L1, true, false);
  // measured: one launch wins where the products are launch-sized (C2 -10 %, C3 -8 % of the two launches) and loses
  // 7 % at the 224x224 shapes, where each product fills the chip by itself and the two tile shapes fight over L2.
  // Tile shapes for the shared launch: the products sit near the L2 -> LDS fill rate at these sizes (32x64 tiles: 10.7
  // flop per staged byte), so 64x64x64 tiles (16 flop/B) are taken for both as soon as the two products together still
  // give ~1.75 workgroups per CU (C3: 45.5 vs 49.3 us); below that the small tiles win (C2: 28 vs 37 us).
  bool big_pair = false;
  const bool pair_ok = merged_backward_shape(B, F, P, L1, &big_pair);
  Shape sw2 = sw, sv2 = sv;
  for (Shape* q : {&sw2, &sv2}) { q->cfg = 1; q->bm = 64; q->bn = 64; q->bk = 64; }
  sw2.tiles_m = (direct + 63) / 64; sw2.tiles_n = (L1 + 63) / 64; sv2.tiles_m = (B + 63) / 64; sv2.tiles_n = (P + 63) / 64;
  NNUE_REQUIRE(!small || pair_ok, NNUE_E_SHAPE, "nnue_ftm_backward: the classifier's small gradients can ride only in the merged launch");
  if (!pair_ok) {
    const int rc = backward_weight_impl(bits, sink, d_out, B, F, P, L1, d_weight, d_bias, sq_partial, stream);
    return rc != NNUE_OK ? rc : nnue_ftm_backward_values(bits, d_out, weight, B, F, P, L1, d_conv_out, stream);
  }
  const Mat wa{bits, (unsigned)((size_t)B * P), P, kIntMax, kIntMax}, wb{d_out, (unsigned)((size_t)B * L1 * 4), L1, kIntMax, kIntMax};
  const Mat va{d_out, (unsigned)((size_t)B * L1 * 4), L1, kIntMax, L1}, vb{weight, (unsigned)((size_t)F * L1 * 4), L1, F - 1, kIntMax};
  const BwwEpi we{d_weight, L1, sq_partial};
  const ValEpi ve{bits, d_conv_out, P};
  const TailRows t = tail_rows(d_out, sink, B, L1, direct, F, d_weight, d_bias);
  const int bf_wm = merged_bf_wm();
  Shape sw3 = sw;  // bf16 weight-gradient tiles: bf_wm x 64 x 128
  if (bf_wm) { sw3.bm = bf_wm; sw3.bn = 64; sw3.bk = 128; sw3.tiles_m = (direct + bf_wm - 1) / bf_wm; sw3.tiles_n = (L1 + 63) / 64; }
  const Shape& swr = bf_wm ? sw3 : big_pair ? sw2 : sw;
  // (value-gradient tiles of 32 x 32 x 128 -- twice the workgroups, two resident per CU -- were measured at the CIFAR
  // batch-512 shape and lose: 30.5 vs 28.9 us for the launch)
  const Shape& svr = big_pair ? sv2 : sv;
  const int n_w = swr.tiles_m * swr.tiles_n, n_v = svr.tiles_m * svr.tiles_n, n_t = t.col_blocks * (1 + t.zero_slices);
  CwArgs cw{};
  if (want_cw) {  // d_w1 [L2][L1] = d_z1^T [L2 x B] l0 [B x L1]   (per bucket over its own grouped rows when seg != NULL)
    const int rows = seg ? grouped_rows : B;
    cw.a = Mat{d_z1, (unsigned)((size_t)rows * L2 * 4), L2, kIntMax, kIntMax};
    cw.b = Mat{ft, (unsigned)((size_t)rows * L1 * 4), L1, kIntMax, kIntMax};
    cw.e = CwEpi{d_w1, L1, L1 / 2};
    cw.M = L2; cw.N = L1; cw.K = B; cw.tiles_n = (L1 + 63) / 64;
    const int cw_bm = bf_wm ? 32 : swr.bm;  // the rider keeps f32 tiles: the weight-gradient shape, or 32 x 64 x 128 beside bf16 tiles
    cw.per_bucket = ((L2 + cw_bm - 1) / cw_bm) * cw.tiles_n;
    cw.seg = seg;
    cw.n_c = cw.per_bucket * (seg ? K : 1);
  }
  // the classifier's small gradients + mean loss as the last tile family (nnue_classifier_train_rider filled the arguments)
  SmallWgrad sgw{};
  static_assert(sizeof(SmallWgrad) <= sizeof(nnue_cls_rider), "nnue_cls_rider is too small for the tile family's arguments");
  if (small) memcpy(&sgw, small, sizeof(SmallWgrad));
  const int n_s = small ? sgw.wgrad_blocks : 0;
  const dim3 grid((unsigned)(n_w + n_v + cw.n_c + n_t + n_s));
#define NNUE_FTM_BWD(WM, WN, WK, VM, VN, VK)                                                                                          \
  hipLaunchKernelGGL((ftm_backward_kernel<WM, WN, WK, VM, VN, VK>), grid, dim3(256), 0, st, wa, wb, we, direct, L1, B, swr.tiles_n, n_w, va, \
                     vb, ve, B, P, L1, svr.tiles_n, n_v, cw, t, sgw)
#define NNUE_FTM_BWD_BF(WM, VM, VN, VK)                                                                                                  \
  do {                                                                                                                                   \
    if (w64 && WM == 64 && !v6) hipLaunchKernelGGL((ftm_backward_bf_kernel<64, VM, VN, VK, false, true>), grid, dim3(256), 0, st, wa, wb, we, direct, L1, B, \
                               swr.tiles_n, n_w, va, vb, ve, B, P, L1, svr.tiles_n, n_v, cw, t, sgw);                                    \
    else if (w64 && WM == 64) hipLaunchKernelGGL((ftm_backward_bf_kernel<64, VM, VN, VK, true, true>), grid, dim3(256), 0, st, wa, wb, we, direct, L1, B, \
                               swr.tiles_n, n_w, va, vb, ve, B, P, L1, svr.tiles_n, n_v, cw, t, sgw);                                    \
    else if (v6) hipLaunchKernelGGL((ftm_backward_bf_kernel<WM, VM, VN, VK, true>), grid, dim3(256), 0, st, wa, wb, we, direct, L1, B, swr.tiles_n, n_w, va, \
                               vb, ve, B, P, L1, svr.tiles_n, n_v, cw, t, sgw);                                                           \
    else hipLaunchKernelGGL((ftm_backward_bf_kernel<WM, VM, VN, VK>), grid, dim3(256), 0, st, wa, wb, we, direct, L1, B, swr.tiles_n, n_w, va, vb, \
                            ve, B, P, L1, svr.tiles_n, n_v, cw, t, sgw);                                                                  \
  } while (0)
  const bool v_small = !big_pair && sv.cfg == 0;
  const bool v6 = merged_values_bf6(B, F, P, L1);
  // weight tiles with K tiles of 64 (gemm_tile_bf64): the launch's LDS drops from 64 KB to the rider's 52 KB -- three
  // workgroups per CU instead of two (the 168 registers allow exactly that): 43.5 -> 37.7 us at the C3 shapes, 26.2 -> 25.1 us at C2
  static const int w64 = env_int("NNUE_FTM_BWD_W64", 1);  // developer knob
  if (bf_wm == 64) { if (v_small) NNUE_FTM_BWD_BF(64, 32, 64, 128); else NNUE_FTM_BWD_BF(64, 64, 64, 64); }
  else if (bf_wm == 32) { if (v_small) NNUE_FTM_BWD_BF(32, 32, 64, 128); else NNUE_FTM_BWD_BF(32, 64, 64, 64); }
  else if (big_pair) NNUE_FTM_BWD(64, 64, 64, 64, 64, 64);
  else if (sv.cfg == 0) NNUE_FTM_BWD(32, 64, 128, 32, 64, 128);
  else NNUE_FTM_BWD(32, 64, 128, 64, 64, 64);
#undef NNUE_FTM_BWD_BF
#undef NNUE_FTM_BWD
  return nnue_launch_status("nnue_ftm_backward");
}
}  // namespace

// The forward with the classifier's layer-1 slabs formed in its epilogue (FwdL1Epi).  Taken for the shapes whose
// forward is one launch without split-K (tile 32x64x128 or 64x64x64), an even pairwise split that falls on 32-column
// runs (L1 % 64 == 0) and float4-readable w1 rows.
extern "C" int nnue_ftm_forward_l1_supported(int B, int F, int P, int L1, int L2) {
  if (!nnue_ftm_supported(F, P, L1) || !shape_ok(B, F, P, L1) || L1 % 64 != 0 || L2 <= 0) return 0;
  const int direct = (F - 1 < P) ? F - 1 : P;
  if (direct <= 0) return 0;
  const Shape s = plan(B, L1, direct, true, true, true);
  return (s.cfg == 0 || s.cfg == 1 || s.cfg == 6 || s.cfg == 7) && s.ksplit == 1;
}

extern "C" int nnue_ftm_forward_l1(const uint8_t* bits, const float* sink, const float* weight, const float* bias, const float* w1, int B,
                                   int F, int P, int L1, int L2, float* out, float* part, nnue_stream_t stream) {
  NNUE_REQUIRE(bits && sink && weight && bias && w1 && out && part, NNUE_E_ARG, "nnue_ftm_forward_l1: null pointer");
  NNUE_REQUIRE(nnue_ftm_forward_l1_supported(B, F, P, L1, L2), NNUE_E_SHAPE,
               "nnue_ftm_forward_l1: B=%d F=%d P=%d L1=%d L2=%d is not a fused-forward shape (nnue_ftm_forward_l1_supported)", B, F, P, L1, L2);
  NNUE_REQUIRE(nnue_aligned16(bits) && nnue_aligned16(weight) && nnue_aligned16(bias) && nnue_aligned16(out) && nnue_aligned16(w1), NNUE_E_ARG,
               "nnue_ftm_forward_l1: pointers must be 16-byte aligned");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int direct = (F - 1 < P) ? F - 1 : P;
  const Shape s = plan(B, L1, direct, true, true, true);
  const Mat ma{bits, (unsigned)((size_t)B * P), P, kIntMax, kIntMax}, mb{weight, (unsigned)((size_t)direct * L1 * 4), L1, kIntMax, kIntMax};
  const FwdL1Epi epi{bias, weight + (size_t)(F - 1) * L1, sink, out, w1, part, B, L1, L2, L1 / 2};
  const dim3 grid((unsigned)(s.tiles_m * s.tiles_n));
  if (s.cfg == 0) hipLaunchKernelGGL((ftm_forward_l1_kernel<32, 128>), grid, dim3(256), 0, st, ma, mb, epi, B, L1, direct, s.tiles_n);
  else if (s.cfg == 1) hipLaunchKernelGGL((ftm_forward_l1_kernel<64, 64>), grid, dim3(256), 0, st, ma, mb, epi, B, L1, direct, s.tiles_n);
  else if (s.cfg == 6) hipLaunchKernelGGL((ftm_forward_l1_bf_kernel<32>), grid, dim3(256), 0, st, ma, mb, epi, B, L1, direct, s.tiles_n);
  else hipLaunchKernelGGL((ftm_forward_l1_bf_kernel<64>), grid, dim3(256), 0, st, ma, mb, epi, B, L1, direct, s.tiles_n);
  return nnue_launch_status("nnue_ftm_forward_l1");
}


// ---- the table's weight gradient consumed in place (single rank + SGD; see BwwSgdEpi and the Gram kernels) ----------
extern "C" int64_t nnue_ftm_gram_sq_count(int B, int L1) { return (B > 0 && L1 > 0) ? (int64_t)((B + 15) / 16) * ((L1 + 15) / 16) : 0; }

namespace {
// K slicing of the A A^T product: about 512 workgroups, every wave at least four 64-k steps (a multiple of four: one
// batch of loads), slices = workgroups per tile pair
struct GramPlan { int pairs, slices, wave_steps; };
GramPlan gram_plan(int B, int direct) {
  const int tiles_b = (B + 31) / 32, pairs = tiles_b * (tiles_b + 1) / 2;
  const int steps = direct > 0 ? (direct + 63) / 64 : 1;
  static const int want_wgs = env_int("NNUE_FTM_GRAM_WGS", 512);  // developer knob
  int want = want_wgs / pairs;
  want = want < 1 ? 1 : want;
  int per = (steps + want * 4 - 1) / (want * 4);
  per = ((per + 3) / 4) * 4;
  const int slices = (steps + per * 4 - 1) / (per * 4);
  return GramPlan{pairs, slices, per};
}
}  // namespace

// floats of scratch nnue_ftm_gram_sqnorm needs behind `gram`: the B x B matrix, then the K slices' int32 slabs
extern "C" int64_t nnue_ftm_gram_scratch(int B, int F, int P) {
  if (B <= 0 || F <= 1 || P <= 0) return 0;
  const GramPlan g = gram_plan(B, (F - 1 < P) ? F - 1 : P);
  return (int64_t)B * B + (int64_t)g.pairs * g.slices * 1024;
}

namespace {
int gram_sqnorm_impl(const char* who, const uint8_t* bits, const float* sink, const float* d_out, int B, int F, int P, int L1, float* gram,
                     float* sq_partial, float* d_weight, float* d_bias, nnue_stream_t stream) {
  NNUE_REQUIRE(bits && d_out && gram && sq_partial, NNUE_E_ARG, "%s: null pointer", who);
  NNUE_REQUIRE(shape_ok(B, F, P, L1), NNUE_E_ARG, "%s: B=%d F=%d P=%d L1=%d out of range", who, B, F, P, L1);
  NNUE_REQUIRE(nnue_ftm_supported(F, P, L1), NNUE_E_SHAPE, "%s: P=%d and L1=%d must be multiples of 4", who, P, L1);
  NNUE_REQUIRE(nnue_ftm_gram_sq_count(B, L1) <= 65536, NNUE_E_SHAPE, "%s: B=%d x L1=%d gives more than 65536 partials", who, B, L1);
  NNUE_REQUIRE(nnue_aligned16(bits) && nnue_aligned16(d_out), NNUE_E_ARG, "%s: pointers must be 16-byte aligned", who);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int direct = (F - 1 < P) ? F - 1 : P;
  const GramPlan g = gram_plan(B, direct);
  int* slabs = reinterpret_cast<int*>(gram + (size_t)B * B);
  const bool tail = d_weight || d_bias;
  const TailRows tr = tail ? tail_rows(d_out, sink, B, L1, direct, F, d_weight, d_bias) : TailRows{};
  const int n_tail = tail ? tr.col_blocks * (1 + tr.zero_slices) : 0;
  if (direct > 0) {
    const int extra = (n_tail + g.pairs - 1) / g.pairs;  // grid rows of riders
    hipLaunchKernelGGL(gram_i8_kernel, dim3((unsigned)g.pairs, (unsigned)(g.slices + extra)), dim3(256), 0, st, bits, (unsigned)((size_t)B * P), B, P,
                       direct, g.wave_steps, slabs, g.slices, tr, n_tail);
    hipLaunchKernelGGL(gram_finish_kernel, dim3((unsigned)(g.pairs * 4)), dim3(256), 0, st, (const int*)slabs, g.pairs, g.slices, B, gram);
  } else {
    nnue_zero_floats(gram, (size_t)B * B, st);
    if (tail) hipLaunchKernelGGL(ftm_tail_rows_kernel, dim3(tr.col_blocks, 1 + tr.zero_slices), dim3(256), 0, st, tr);
  }
  const int64_t tiles16 = nnue_ftm_gram_sq_count(B, L1);
  hipLaunchKernelGGL(gram_apply_kernel, dim3((unsigned)((tiles16 + 3) / 4)), dim3(256), 0, st, gram, d_out, B, L1, sq_partial);
  return nnue_launch_status(who);
}
}  // namespace

extern "C" int nnue_ftm_gram_sqnorm(const uint8_t* bits, const float* d_out, int B, int F, int P, int L1, float* gram, float* sq_partial,
                                    nnue_stream_t stream) {
  return gram_sqnorm_impl("nnue_ftm_gram_sqnorm", bits, nullptr, d_out, B, F, P, L1, gram, sq_partial, nullptr, nullptr, stream);
}

extern "C" int nnue_ftm_gram_sqnorm_tail(const uint8_t* bits, const float* sink, const float* d_out, int B, int F, int P, int L1, float* gram,
                                         float* sq_partial, float* d_weight, float* d_bias, nnue_stream_t stream) {
  NNUE_REQUIRE(sink && (d_weight || d_bias), NNUE_E_ARG, "nnue_ftm_gram_sqnorm_tail: null pointer");
  return gram_sqnorm_impl("nnue_ftm_gram_sqnorm_tail", bits, sink, d_out, B, F, P, L1, gram, sq_partial, d_weight, d_bias, stream);
}

extern "C" int nnue_ftm_backward_tail_rows(const float* sink, const float* d_out, int B, int F, int P, int L1, float* d_weight, float* d_bias,
                                           nnue_stream_t stream) {
  NNUE_REQUIRE(sink && d_out && (d_weight || d_bias), NNUE_E_ARG, "nnue_ftm_backward_tail_rows: null pointer");
  NNUE_REQUIRE(shape_ok(B, F, P, L1), NNUE_E_ARG, "nnue_ftm_backward_tail_rows: B=%d F=%d P=%d L1=%d out of range", B, F, P, L1);
  const int direct = (F - 1 < P) ? F - 1 : P;
  const TailRows t = tail_rows(d_out, sink, B, L1, direct, F, d_weight, d_bias);
  hipLaunchKernelGGL(ftm_tail_rows_kernel, dim3(t.col_blocks, 1 + t.zero_slices), dim3(256), 0, static_cast<hipStream_t>(stream), t);
  return nnue_launch_status("nnue_ftm_backward_tail_rows");
}

extern "C" int nnue_ftm_backward_weight_update(const uint8_t* bits, const float* d_out, int B, int F, int P, int L1, float* weight,
                                               float* momentum_rows, const float* coef, float lr, float momentum, float weight_decay,
                                               float grad_scale, int first_step, const float* lr_dev, nnue_stream_t stream) {
  NNUE_REQUIRE(bits && d_out && weight && coef, NNUE_E_ARG, "nnue_ftm_backward_weight_update: null pointer");
  NNUE_REQUIRE(momentum == 0.0f || momentum_rows, NNUE_E_ARG, "nnue_ftm_backward_weight_update: momentum %g needs the momentum rows", momentum);
  NNUE_REQUIRE(shape_ok(B, F, P, L1), NNUE_E_ARG, "nnue_ftm_backward_weight_update: B=%d F=%d P=%d L1=%d out of range", B, F, P, L1);
  NNUE_REQUIRE(nnue_ftm_supported(F, P, L1), NNUE_E_SHAPE, "nnue_ftm_backward_weight_update: P=%d and L1=%d must be multiples of 4", P, L1);
  NNUE_REQUIRE(nnue_aligned16(bits) && nnue_aligned16(d_out) && nnue_aligned16(weight), NNUE_E_ARG,
               "nnue_ftm_backward_weight_update: pointers must be 16-byte aligned");
  const int direct = (F - 1 < P) ? F - 1 : P;
  if (direct <= 0) return NNUE_OK;
  Shape s = plan(direct, L1, B, false, false, true);
  if (s.cfg < 6) {  // launch-sized product: a 64-row bf16 tile (the in-place epilogue lives in the bf16 tile)
    s.cfg = 7; s.bm = 64; s.bn = 64; s.bk = kBfK;
    s.tiles_m = (direct + 63) / 64; s.tiles_n = (L1 + 63) / 64; s.ksplit = 1; s.klen = (B + kBfK - 1) / kBfK * kBfK;
  }
  NNUE_REQUIRE(s.ksplit == 1, NNUE_E_SHAPE, "nnue_ftm_backward_weight_update: the product must not be split along K");
  static const int xcd = env_int("NNUE_FTM_XCD_REMAP", 1);  // developer knob
  launch<false, false>(static_cast<hipStream_t>(stream), s, Mat{bits, (unsigned)((size_t)B * P), P, kIntMax, kIntMax},
                       Mat{d_out, (unsigned)((size_t)B * L1 * 4), L1, kIntMax, kIntMax},
                       BwwSgdEpi{weight, momentum == 0.0f ? nullptr : momentum_rows, coef, L1, lr, momentum, weight_decay, grad_scale, first_step, xcd, lr_dev},
                       direct, L1, B);
  return nnue_launch_status("nnue_ftm_backward_weight_update");
}

// nnue_ftm_backward_weight_update + nnue_ftm_forward of the NEXT step's map in one pass over the table (update_forward.h).
// B: rows of the gradient's factors (the batch, or the global batch under the factor exchange); B_next: rows of the next map.
extern "C" int nnue_ftm_update_forward_supported(int B, int B_next, int F, int P, int L1) {
  if (!nnue_ftm_supported(F, P, L1) || !shape_ok(B, F, P, L1) || B_next <= 0 || B_next > 128 || L1 % 64 != 0) return 0;
  const int nk = (B + 63) / 64;  // K tiles of the weight-gradient product: 1, 2, 4, 8 or 16
  if (nk > 16 || (nk & (nk - 1)) != 0) return 0;
  const int direct = (F - 1 < P) ? F - 1 : P;
  if (direct <= 0 || !use_bf16()) return 0;
  // the forward must be the split-K product of 128 x 64 bf16-split tiles with 64-deep K tiles, whole 128-row table tiles per slab
  const Shape s = plan(B_next, L1, direct, true, true, true);
  static const int kt64 = env_int("NNUE_FTM_BF_KT64", 1);
  return s.cfg == 8 && kt64 && s.ksplit > 1 && s.klen % 128 == 0 && plan(direct, L1, B, false, false, true).cfg == 8;
}

extern "C" int nnue_ftm_backward_weight_update_forward(const uint8_t* bits, const float* d_out, int B, int F, int P, int L1, float* weight,
                                                       float* momentum_rows, const float* coef, float lr, float momentum, float weight_decay,
                                                       float grad_scale, int first_step, const float* lr_dev, const uint8_t* bits_next,
                                                       const float* sink_next, int B_next, const float* bias, float* out_next, void* scratch,
                                                       int64_t scratch_bytes, nnue_stream_t stream) {
  NNUE_REQUIRE(bits && d_out && weight && coef && bits_next && sink_next && bias && out_next && scratch, NNUE_E_ARG,
               "nnue_ftm_backward_weight_update_forward: null pointer");
  NNUE_REQUIRE(momentum == 0.0f || momentum_rows, NNUE_E_ARG, "nnue_ftm_backward_weight_update_forward: momentum %g needs the momentum rows", momentum);
  NNUE_REQUIRE(nnue_ftm_update_forward_supported(B, B_next, F, P, L1), NNUE_E_SHAPE,
               "nnue_ftm_backward_weight_update_forward: B=%d B_next=%d F=%d P=%d L1=%d is not a split-K forward over a big table (use the two separate "
               "calls)", B, B_next, F, P, L1);
  NNUE_REQUIRE(nnue_aligned16(bits) && nnue_aligned16(bits_next) && nnue_aligned16(d_out) && nnue_aligned16(weight) && nnue_aligned16(bias) &&
                   nnue_aligned16(out_next) && nnue_aligned16(scratch) && (!momentum_rows || nnue_aligned16(momentum_rows)),
               NNUE_E_ARG, "nnue_ftm_backward_weight_update_forward: pointers must be 16-byte aligned");
  NNUE_REQUIRE(bits != bits_next, NNUE_E_ARG, "nnue_ftm_backward_weight_update_forward: the two maps must be different buffers");
  const int direct = (F - 1 < P) ? F - 1 : P;
  const Shape s = plan(B_next, L1, direct, true, true, true);
  const int64_t need = (int64_t)s.ksplit * B_next * L1 * (int64_t)sizeof(float);
  NNUE_REQUIRE(scratch_bytes >= need, NNUE_E_SCRATCH, "nnue_ftm_backward_weight_update_forward: scratch %lld < %lld bytes", (long long)scratch_bytes,
               (long long)need);
  hipStream_t st = static_cast<hipStream_t>(stream);
  static const int xcd = env_int("NNUE_FTM_XCD_REMAP", 1);  // developer knob
  const int blocks = s.tiles_n * s.ksplit;
  UpdFwd a{bits, bits_next, d_out, weight, momentum == 0.0f ? nullptr : momentum_rows, coef, lr_dev, static_cast<float*>(scratch),
           B, B_next, P, L1, direct, s.klen, s.tiles_n, (xcd && s.ksplit % 8 == 0) ? 1 : 0, lr, momentum, weight_decay, grad_scale, env_int("NNUE_FTM_UF_ABL", 0)};
  const bool mom = a.momentum != nullptr, first = mom && first_step;
#define NNUE_UF_LAUNCH(NK)                                                                                                         \
  do {                                                                                                                             \
    if (!mom) hipLaunchKernelGGL((ftm_update_forward_kernel<NK, false, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);      \
    else if (first) hipLaunchKernelGGL((ftm_update_forward_kernel<NK, true, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);  \
    else hipLaunchKernelGGL((ftm_update_forward_kernel<NK, true, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);            \
  } while (0)
  switch ((B + 63) / 64) {
    case 1: NNUE_UF_LAUNCH(1); break;
    case 2: NNUE_UF_LAUNCH(2); break;
    case 4: NNUE_UF_LAUNCH(4); break;
    case 8: NNUE_UF_LAUNCH(8); break;
    default: NNUE_UF_LAUNCH(16); break;
  }
#undef NNUE_UF_LAUNCH
  const int64_t count4 = (int64_t)B_next * L1 / 4;
  hipLaunchKernelGGL(ftm_finish_kernel, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, st, static_cast<const float*>(scratch), s.ksplit, count4,
                     bias, weight + (size_t)(F - 1) * L1, sink_next, L1, out_next);
  return nnue_launch_status("nnue_ftm_backward_weight_update_forward");
}

// Which matrix unit a product of this shape runs on (the launch policy above, for reporting: bench.py prices a kernel
// against the peak of the unit it used).  which: 0 forward, 1 stand-alone weight gradient, 2 weight-gradient tiles of the
// merged backward launch, 3 weight gradient with the update in its epilogue, 4 / 5 value gradient (stand-alone / tiles of
// the merged launch: six plane products).  1 = bf16-split tiles, 0 = f32 MFMA.
extern "C" int nnue_ftm_uses_bf16(int which, int B, int F, int P, int L1) {
  if (!nnue_ftm_supported(F, P, L1) || !shape_ok(B, F, P, L1)) return 0;
  const int direct = (F - 1 < P) ? F - 1 : P;
  if (direct <= 0) return 0;
  if (which == 0) return plan(B, L1, direct, true, true, true).cfg >= 6;
  if (which == 1) return plan(direct, L1, B, false, false, true).cfg >= 6;
  if (which == 2) { bool big = false; return merged_backward_shape(B, F, P, L1, &big) ? merged_bf_wm() != 0 : plan(direct, L1, B, false, false, true).cfg >= 6; }
  if (which == 4) return values_bf6(B, P, L1);            // stand-alone value gradient: six bf16 plane products
  if (which == 5) return merged_values_bf6(B, F, P, L1);  // value tiles of the merged launch
  return 1;
}
