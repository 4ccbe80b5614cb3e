// FeatureTransformer for BINARY grid features, LDS-staged and batch-tiled (gfx950).
//
// Inside NNUE.forward the feature values are exactly {0,1} and the ids ascend (nnue.py:590-635), so
// the act list collapses to bit masks and the three FT products become bit-driven gathers out of LDS:
//
//   forward        out[b]  = bias + sum_{p active, p < F-1} W[p] + sink[b] * W[F-1]      (nnue.py:686-710)
//   weight grad    dW[f]   = sum_{b: bit(b,f)} dOut[b] ; dW[F-1] = sum_b sink[b]*dOut[b] ; db = sum_b dOut[b]
//   value grad     dX[b,p] = bit(b,p) ? <dOut[b], W[min(p,F-1)]> : 0                      (= d conv_out)
//
// where sink[b] counts the active ids >= F-1 (the clamp of nnue.py:701).  Instead of every sample
// re-reading its ~400 table rows from L2/HBM (gather kernels in ft_kernels.hip: 854 MB of L2 traffic per
// launch at the CIFAR shapes), a workgroup stages a tile of the table (or of dOut) in LDS ONCE and all of
// its samples (rows) gather from there: memory-side traffic drops by the sample-tile factor and the
// inner loop runs at LDS bandwidth.  Accumulation order per output element is ascending row / sample
// order: results are bitwise reproducible and equal the list kernels' up to the sink term.
//
// Layouts:  maskW [B][pw64]  u64, bit p of sample b  (flat position bits, p < P; pw64 even)
//           maskT [F+1][bw64] u64, bit b of table row f; row F-1 = (sink[b] != 0), row F = all samples
//           sink  [B] float
#include "common.h"

namespace {

using u64 = unsigned long long;

constexpr int kTC = 64;    // columns per workgroup: 16 lanes x float4, 4 lane groups per wave
constexpr int kRT = 128;   // staged rows per LDS tile (two 64-bit mask words)
constexpr int kTileFloats = kRT * kTC + kTC;  // + one all-zero row that exhausted lanes read

// ------------------------------------------------------------------ bit masks from conv_out
// per sample: position words, active count, sink count
__global__ __launch_bounds__(256) void bits_rows_kernel(const float* __restrict__ conv_out,
                                                        const float* __restrict__ thr, int G, int P, int F,
                                                        u64* __restrict__ maskW, int pw64, float* __restrict__ sink,
                                                        int* __restrict__ n) {
  __shared__ int cnt_s[4], sink_s[4];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ x = conv_out + (size_t)b * P;
  int cnt = 0, snk = 0;
  for (int p0 = 0; p0 < pw64 * 64; p0 += 256) {
    const int p = p0 + tid;
    const bool on = (p < P) && (x[p] > thr[p / G]);
    const u64 m = __ballot(on);
    const u64 ms = __ballot(on && p >= F - 1);
    const int word = p0 / 64 + wave;
    if (lane == 0 && word < pw64) maskW[(size_t)b * pw64 + word] = m;
    cnt += __popcll(m);
    snk += __popcll(ms);
  }
  if (lane == 0) {
    cnt_s[wave] = cnt;
    sink_s[wave] = snk;
  }
  __syncthreads();
  if (tid == 0) {
    n[b] = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3];
    sink[b] = (float)(sink_s[0] + sink_s[1] + sink_s[2] + sink_s[3]);
  }
}

// transposed words: 64 table rows x 64 samples per workgroup through an LDS byte tile + ballots
__global__ __launch_bounds__(256) void bits_transpose_kernel(const float* __restrict__ conv_out,
                                                             const float* __restrict__ thr,
                                                             const float* __restrict__ sink, int B, int G, int P,
                                                             int F, u64* __restrict__ maskT, int bw64) {
  __shared__ unsigned char tile[64][68];
  const int f0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int f = f0 + tx;
  const int direct = (F - 1 < P) ? F - 1 : P;  // rows with a position bit of their own
  const float t = (f < direct) ? thr[f / G] : 0.0f;
  for (int j = ty; j < 64; j += 4) {
    const int b = b0 + j;
    bool on = false;
    if (b < B) {
      if (f < direct) on = conv_out[(size_t)b * P + f] > t;
      else if (f == F - 1) on = sink[b] != 0.0f;
      else if (f == F) on = true;
    }
    tile[j][tx] = on ? 1 : 0;
  }
  __syncthreads();
  for (int j = ty; j < 64; j += 4) {
    const u64 m = __ballot(tile[tx][j] != 0);  // lane = sample
    if (tx == 0 && f0 + j <= F) maskT[(size_t)(f0 + j) * bw64 + blockIdx.y] = m;
  }
}

// ------------------------------------------------------------------ bit-driven gather out of LDS
// MODE 0 (forward):     outputs = samples,    gathered rows = table rows, masks = maskW, bias + valued sink row
// MODE 1 (weight grad): outputs = table rows, gathered rows = d_out rows, masks = maskT, row F-1 valued by sink[]
// Workgroup = 4 waves; a wave = 4 lane groups of 16 lanes (64 columns as float4), each group owns SG
// outputs.  The gathered matrix streams through two 32 KB LDS tiles (register-staged: loads for tile
// t+1 are issued before tile t is consumed).  Per tile and output, the set bits of two mask words
// select the LDS rows to add; groups that run out of bits read an all-zero row, so the loop has no
// divergence.
template <int SG, int MODE>
__global__ __launch_bounds__(256) void ftb_gather_kernel(const float* __restrict__ src,      // [n_src][L1]
                                                         const float* __restrict__ bias,     // MODE 0
                                                         const u64* __restrict__ mask, int mw64,
                                                         const float* __restrict__ sink,     // [B]
                                                         int n_out, int n_src, int sink_row, int L1,
                                                         float* __restrict__ out,            // [n_out][L1] (MODE 1: d_weight)
                                                         float* __restrict__ out_extra) {    // MODE 1: d_bias
  __shared__ __attribute__((aligned(16))) float tile[2][kTileFloats];
  __shared__ float coef_t[2][kRT];  // MODE 1: sink[] of the staged samples
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, l16 = lane & 15;
  const int c0 = blockIdx.y * kTC + l16 * 4;
  const int o0 = blockIdx.x * (16 * SG) + (wave * 4 + grp) * SG;

  float4 acc[SG];
  bool ok[SG];
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    ok[s] = o0 + s < n_out;
    acc[s] = (MODE == 0) ? *reinterpret_cast<const float4*>(bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (tid < 2 * (kTC / 4)) {  // the zero rows
    float4* z = reinterpret_cast<float4*>(&tile[tid / (kTC / 4)][kRT * kTC]);
    z[tid % (kTC / 4)] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int ntiles = (n_src + kRT - 1) / kRT;
  const int srow = tid >> 4, scol = (tid & 15) * 4;  // staging role: 16 rows x 64 columns per pass, 8 passes

  float4 st[8];
  float st_coef = 0.f;
  auto stage_load = [&](int t) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = t * kRT + srow + 16 * i;
      st[i] = (r < n_src) ? *reinterpret_cast<const float4*>(src + (size_t)r * L1 + blockIdx.y * kTC + scol)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (MODE == 1 && tid < kRT) st_coef = (t * kRT + tid < n_src) ? sink[t * kRT + tid] : 0.f;
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(&tile[buf][(srow + 16 * i) * kTC + scol]) = st[i];
    if (MODE == 1 && tid < kRT) coef_t[buf][tid] = st_coef;
  };
  auto load_words = [&](int t, u64 (&w)[SG][2]) {
#pragma unroll
    for (int s = 0; s < SG; ++s)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int word = 2 * t + h;
        u64 m = (ok[s] && word < mw64) ? mask[(size_t)(o0 + s) * mw64 + word] : 0ull;
        const int base = word * 64;  // rows this word covers: [base, base + 64)
        if (base + 64 > n_src) m = (base >= n_src) ? 0ull : (m & ((1ull << (n_src - base)) - 1ull));
        w[s][h] = m;
      }
  };

  u64 mw[SG][2], nw[SG][2];
  if (ntiles > 0) {
    stage_load(0);
    load_words(0, mw);
    stage_store(0);
  }
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) {
      stage_load(t + 1);
      load_words(t + 1, nw);
    }
    const float* __restrict__ tb = &tile[buf][0];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u64 m[SG];
      u64 any = 0;
#pragma unroll
      for (int s = 0; s < SG; ++s) {
        m[s] = mw[s][h];
        any |= m[s];
      }
      while (__any(any != 0)) {
        constexpr int U = (SG >= 4) ? 1 : (SG == 2 ? 2 : 4);  // rows per output per iteration: 4 LDS reads in flight
        float4 v[SG][U];
        float c[SG][U];
#pragma unroll
        for (int s = 0; s < SG; ++s)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const bool has = m[s] != 0;
            const int j = has ? (h * 64 + __builtin_ctzll(m[s])) : kRT;  // kRT = the zero row
            m[s] &= m[s] - 1;
            v[s][u] = *reinterpret_cast<const float4*>(tb + j * kTC + l16 * 4);
            c[s][u] = 1.0f;
            if (MODE == 1 && o0 + s == sink_row) c[s][u] = has ? coef_t[buf][j] : 0.f;
          }
        any = 0;
#pragma unroll
        for (int s = 0; s < SG; ++s) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            acc[s].x = fmaf(c[s][u], v[s][u].x, acc[s].x);
            acc[s].y = fmaf(c[s][u], v[s][u].y, acc[s].y);
            acc[s].z = fmaf(c[s][u], v[s][u].z, acc[s].z);
            acc[s].w = fmaf(c[s][u], v[s][u].w, acc[s].w);
          }
          any |= m[s];
        }
      }
    }
    if (t + 1 < ntiles) stage_store(buf ^ 1);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < SG; ++s) {
      mw[s][0] = nw[s][0];
      mw[s][1] = nw[s][1];
    }
  }
#pragma unroll
  for (int s = 0; s < SG; ++s) {
    if (!ok[s]) continue;
    const int o = o0 + s;
    if (MODE == 0) {
      if (sink_row >= 0) {  // folded ids: sink[b] copies of table row F-1
        const float sv = sink[o];
        if (sv != 0.f) {
          const float4 w = *reinterpret_cast<const float4*>(src + (size_t)sink_row * L1 + c0);
          acc[s].x = fmaf(sv, w.x, acc[s].x);
          acc[s].y = fmaf(sv, w.y, acc[s].y);
          acc[s].z = fmaf(sv, w.z, acc[s].z);
          acc[s].w = fmaf(sv, w.w, acc[s].w);
        }
      }
      *reinterpret_cast<float4*>(out + (size_t)o * L1 + c0) = acc[s];
    } else {
      float* dst = (o == n_out - 1) ? out_extra : out;  // last output row = bias gradient
      if (dst != nullptr) *reinterpret_cast<float4*>(dst + (o == n_out - 1 ? (size_t)0 : (size_t)o * L1) + c0) = acc[s];
    }
  }
}

// ------------------------------------------------------------------ value gradient
// Workgroup = (sample tile) x (32 table rows).  The 32 rows (32 x L1 floats, 128 KB at L1 = 1024) are
// staged in LDS once; each wave then walks its samples: d_out[b] sits in registers (prefetched one sample
// ahead), the set bits of the sample's 32-bit mask slice pick LDS rows, 16 dot products at a time are
// transposed-and-reduced across the wave with a butterfly, and ranks map them back to bit positions, so
// the 32 outputs leave as ONE coalesced 128-byte store (zeros included: no separate zero fill).
template <int S>
__global__ __launch_bounds__(256) void ftb_values_kernel(const float* __restrict__ d_out,
                                                         const float* __restrict__ W,
                                                         const u64* __restrict__ maskW, int pw64, int B, int F,
                                                         int P, int samples_per_block, float* __restrict__ dst) {
  constexpr int L1 = 256 * S;
  __shared__ __attribute__((aligned(16))) float rows[32 * L1];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool has_sink = F - 1 < P;                   // ids >= F-1 exist and fold into row F-1
  const int p_lim = has_sink ? F - 1 : P;            // positions below p_lim have a table row of their own
  const int r_last = has_sink ? F - 1 : P - 1;       // highest row that can be referenced
  const int r0 = blockIdx.y * 32;
  const int nrows = min(32, r_last + 1 - r0);
  for (int i = tid; i < nrows * (L1 / 4); i += 256) {
    const int r = i / (L1 / 4), c4 = i - r * (L1 / 4);
    reinterpret_cast<float4*>(rows)[i] = *reinterpret_cast<const float4*>(W + (size_t)(r0 + r) * L1 + c4 * 4);
  }
  __syncthreads();
  const bool sink_block = has_sink && (F - 1 >= r0) && (F - 1 < r0 + 32);
  const int b_lo = blockIdx.x * samples_per_block;
  const int b_hi = min(B, b_lo + samples_per_block);
  const int nd = p_lim - r0;  // direct positions in this block's 32-wide slice
  const unsigned keep = nd >= 32 ? 0xffffffffu : (nd <= 0 ? 0u : ((1u << nd) - 1u));
  const float* __restrict__ rl = rows + lane * 4;

  float4 g[S], gn[S];
  int b = b_lo + wave;
  if (b < b_hi) {
#pragma unroll
    for (int s = 0; s < S; ++s) g[s] = *reinterpret_cast<const float4*>(d_out + (size_t)b * L1 + s * 256 + lane * 4);
  }
  for (; b < b_hi; b += 4) {
    const int bn = b + 4;
    if (bn < b_hi) {
#pragma unroll
      for (int s = 0; s < S; ++s) gn[s] = *reinterpret_cast<const float4*>(d_out + (size_t)bn * L1 + s * 256 + lane * 4);
    }
    const u64 word = maskW[(size_t)b * pw64 + (r0 >> 6)];
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(word >> (r0 & 63))) & keep;
    const int L = lane & 31;
    const int rank = __popc(m & ((1u << L) - 1u));
    const bool bit_mine = (m >> L) & 1u;
    float res = 0.f;
    unsigned mm = m;
    for (int grp = 0; mm != 0; ++grp) {  // 16 set bits per round (at most two rounds)
      float p[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        float a = 0.f;
        if (mm != 0) {  // wave-uniform
          const int j = __builtin_ctz(mm);
          mm &= mm - 1;
          const float* __restrict__ wr = rl + j * L1;
#pragma unroll
          for (int s = 0; s < S; ++s) {
            const float4 v = *reinterpret_cast<const float4*>(wr + s * 256);
            a = fmaf(v.x, g[s].x, a);
            a = fmaf(v.y, g[s].y, a);
            a = fmaf(v.z, g[s].z, a);
            a = fmaf(v.w, g[s].w, a);
          }
        }
        p[u] = a;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool hi = lane & 8;
        const float keepv = hi ? p[u + 8] : p[u], send = hi ? p[u] : p[u + 8];
        p[u] = keepv + __shfl_xor(send, 8);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool hi = lane & 4;
        const float keepv = hi ? p[u + 4] : p[u], send = hi ? p[u] : p[u + 4];
        p[u] = keepv + __shfl_xor(send, 4);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const bool hi = lane & 2;
        const float keepv = hi ? p[u + 2] : p[u], send = hi ? p[u] : p[u + 2];
        p[u] = keepv + __shfl_xor(send, 2);
      }
      {
        const bool hi = lane & 1;
        const float keepv = hi ? p[1] : p[0], send = hi ? p[0] : p[1];
        p[0] = keepv + __shfl_xor(send, 1);
      }
      float tot = p[0];
      tot += __shfl_xor(tot, 16);
      tot += __shfl_xor(tot, 32);  // every lane: dot product of entry (lane & 15) of this round
      const float v = __shfl(tot, rank & 15);
      if (bit_mine && (rank >> 4) == grp) res = v;
    }
    if (lane < 32 && lane < nd) dst[(size_t)b * P + r0 + lane] = res;
    if (sink_block) {  // every active id >= F-1 receives <d_out[b], W[F-1]>
      const float* __restrict__ wr = rl + (F - 1 - r0) * L1;
      float a = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(wr + s * 256);
        a = fmaf(v.x, g[s].x, a);
        a = fmaf(v.y, g[s].y, a);
        a = fmaf(v.z, g[s].z, a);
        a = fmaf(v.w, g[s].w, a);
      }
#pragma unroll
      for (int sft = 32; sft >= 1; sft >>= 1) a += __shfl_xor(a, sft);
      for (int p = F - 1 + lane; p < P; p += 64) {
        const bool on = (maskW[(size_t)b * pw64 + (p >> 6)] >> (p & 63)) & 1ull;
        dst[(size_t)b * P + p] = on ? a : 0.f;
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) g[s] = gn[s];
  }
}

int pick_sg(int n_out, int col_tiles) {
  // largest SG (outputs per lane group) that still leaves >= 256 workgroups; more outputs per staged tile =
  // less staging traffic per output
  for (int sg = 4; sg > 1; sg >>= 1)
    if ((long long)((n_out + 16 * sg - 1) / (16 * sg)) * col_tiles >= 256) return sg;
  return 1;
}

template <int MODE>
void launch_gather(int sg, dim3 block, hipStream_t s, const float* src, const float* bias, const u64* mask, int mw64,
                   const float* sink, int n_out, int n_src, int sink_row, int L1, float* out, float* extra) {
  const dim3 grid((n_out + 16 * sg - 1) / (16 * sg), L1 / kTC);
  if (sg == 4)
    hipLaunchKernelGGL((ftb_gather_kernel<4, MODE>), grid, block, 0, s, src, bias, mask, mw64, sink, n_out, n_src, sink_row, L1, out, extra);
  else if (sg == 2)
    hipLaunchKernelGGL((ftb_gather_kernel<2, MODE>), grid, block, 0, s, src, bias, mask, mw64, sink, n_out, n_src, sink_row, L1, out, extra);
  else
    hipLaunchKernelGGL((ftb_gather_kernel<1, MODE>), grid, block, 0, s, src, bias, mask, mw64, sink, n_out, n_src, sink_row, L1, out, extra);
}

}  // namespace

// =============================================================================== C ABI
extern "C" int nnue_ftb_supported(int L1) { return (L1 == 256 || L1 == 512 || L1 == 1024) ? 1 : 0; }

extern "C" int nnue_binarize_bits(const float* conv_out, const float* thr, int B, int fps, int Gh, int Gw, int F,
                                  uint64_t* maskW, int pw64, uint64_t* maskT, int bw64, float* sink, int32_t* n,
                                  nnue_stream_t stream) {
  NNUE_REQUIRE(conv_out && thr && maskW && maskT && sink && n, NNUE_E_ARG, "nnue_binarize_bits: null pointer");
  NNUE_REQUIRE(B > 0 && fps > 0 && Gh > 0 && Gw > 0 && F > 0, NNUE_E_ARG,
               "nnue_binarize_bits: B=%d fps=%d Gh=%d Gw=%d F=%d must be positive", B, fps, Gh, Gw, F);
  const long long P64 = (long long)fps * Gh * Gw;
  NNUE_REQUIRE(P64 < (1ll << 30), NNUE_E_SHAPE, "nnue_binarize_bits: fps*Gh*Gw too large");
  const int P = (int)P64;
  NNUE_REQUIRE(pw64 % 2 == 0 && (long long)pw64 * 64 >= P, NNUE_E_SHAPE,
               "nnue_binarize_bits: pw64=%d must be even and cover %d positions", pw64, P);
  NNUE_REQUIRE((long long)bw64 * 64 >= B && bw64 % 2 == 0, NNUE_E_SHAPE, "nnue_binarize_bits: bw64=%d must be even and cover %d samples", bw64, B);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(bits_rows_kernel, dim3(B), dim3(256), 0, s, conv_out, thr, Gh * Gw, P, F,
                     reinterpret_cast<u64*>(maskW), pw64, sink, n);
  hipLaunchKernelGGL(bits_transpose_kernel, dim3((F + 1 + 63) / 64, bw64), dim3(256), 0, s, conv_out, thr, sink, B, Gh * Gw,
                     P, F, reinterpret_cast<u64*>(maskT), bw64);
  return nnue_launch_status("nnue_binarize_bits");
}

extern "C" int nnue_ftb_forward(const float* weight, const float* bias, const uint64_t* maskW, int pw64, const float* sink,
                                int B, int F, int P, int L1, float* out, nnue_stream_t stream) {
  NNUE_REQUIRE(weight && bias && maskW && sink && out, NNUE_E_ARG, "nnue_ftb_forward: null pointer");
  NNUE_REQUIRE(B > 0 && F > 0 && P > 0, NNUE_E_ARG, "nnue_ftb_forward: B=%d F=%d P=%d must be positive", B, F, P);
  NNUE_REQUIRE(nnue_ftb_supported(L1), NNUE_E_SHAPE, "nnue_ftb_forward: L1=%d not supported (256/512/1024)", L1);
  NNUE_REQUIRE(pw64 % 2 == 0 && (long long)pw64 * 64 >= P, NNUE_E_SHAPE, "nnue_ftb_forward: pw64=%d does not cover P=%d", pw64, P);
  NNUE_REQUIRE(nnue_aligned16(weight) && nnue_aligned16(bias) && nnue_aligned16(out), NNUE_E_ARG,
               "nnue_ftb_forward: pointers must be 16-byte aligned");
  const bool has_sink = F - 1 < P;
  const int direct = has_sink ? F - 1 : P;  // table rows selected by a position bit of their own
  launch_gather<0>(pick_sg(B, L1 / kTC), dim3(256), static_cast<hipStream_t>(stream), weight, bias,
                   reinterpret_cast<const u64*>(maskW), pw64, sink, B, direct, has_sink ? F - 1 : -1, L1, out, nullptr);
  return nnue_launch_status("nnue_ftb_forward");
}

extern "C" int nnue_ftb_backward_weight(const float* d_out, const uint64_t* maskT, int bw64, const float* sink, int B, int F,
                                        int L1, float* d_weight, float* d_bias, nnue_stream_t stream) {
  NNUE_REQUIRE(d_out && maskT && sink, NNUE_E_ARG, "nnue_ftb_backward_weight: null pointer");
  NNUE_REQUIRE(d_weight || d_bias, NNUE_E_ARG, "nnue_ftb_backward_weight: both outputs are null");
  NNUE_REQUIRE(B > 0 && F > 0, NNUE_E_ARG, "nnue_ftb_backward_weight: B=%d F=%d must be positive", B, F);
  NNUE_REQUIRE(nnue_ftb_supported(L1), NNUE_E_SHAPE, "nnue_ftb_backward_weight: L1=%d not supported (256/512/1024)", L1);
  NNUE_REQUIRE(bw64 % 2 == 0 && (long long)bw64 * 64 >= B, NNUE_E_SHAPE, "nnue_ftb_backward_weight: bw64=%d does not cover B=%d", bw64, B);
  NNUE_REQUIRE(nnue_aligned16(d_out) && nnue_aligned16(d_weight) && nnue_aligned16(d_bias), NNUE_E_ARG,
               "nnue_ftb_backward_weight: pointers must be 16-byte aligned");
  // outputs: F table rows + the bias row; row F-1 is valued by sink[], rows the map cannot reach have empty masks
  launch_gather<1>(pick_sg(F + 1, L1 / kTC), dim3(256), static_cast<hipStream_t>(stream), d_out, nullptr,
                   reinterpret_cast<const u64*>(maskT), bw64, sink, F + 1, B, F - 1, L1, d_weight, d_bias);
  return nnue_launch_status("nnue_ftb_backward_weight");
}

extern "C" int nnue_ftb_backward_values(const float* d_out, const float* weight, const uint64_t* maskW, int pw64, int B,
                                        int F, int P, int L1, float* d_conv_out, nnue_stream_t stream) {
  NNUE_REQUIRE(d_out && weight && maskW && d_conv_out, NNUE_E_ARG, "nnue_ftb_backward_values: null pointer");
  NNUE_REQUIRE(B > 0 && F > 0 && P > 0, NNUE_E_ARG, "nnue_ftb_backward_values: B=%d F=%d P=%d must be positive", B, F, P);
  NNUE_REQUIRE(nnue_ftb_supported(L1), NNUE_E_SHAPE, "nnue_ftb_backward_values: L1=%d not supported (256/512/1024)", L1);
  NNUE_REQUIRE(pw64 % 2 == 0 && (long long)pw64 * 64 >= P, NNUE_E_SHAPE, "nnue_ftb_backward_values: pw64=%d does not cover P=%d", pw64, P);
  NNUE_REQUIRE(nnue_aligned16(d_out) && nnue_aligned16(weight), NNUE_E_ARG, "nnue_ftb_backward_values: pointers must be 16-byte aligned");
  const int r_last = (F - 1 < P) ? F - 1 : P - 1;
  const int row_blocks = r_last / 32 + 1;
  int spb = 64;  // samples per workgroup: fewer when the grid would be too small
  while (spb > 16 && (long long)((B + spb - 1) / spb) * row_blocks < 256) spb >>= 1;
  const dim3 grid((B + spb - 1) / spb, row_blocks), block(256);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const u64* mw = reinterpret_cast<const u64*>(maskW);
  if (L1 == 1024)
    hipLaunchKernelGGL(ftb_values_kernel<4>, grid, block, 0, s, d_out, weight, mw, pw64, B, F, P, spb, d_conv_out);
  else if (L1 == 512)
    hipLaunchKernelGGL(ftb_values_kernel<2>, grid, block, 0, s, d_out, weight, mw, pw64, B, F, P, spb, d_conv_out);
  else
    hipLaunchKernelGGL(ftb_values_kernel<1>, grid, block, 0, s, d_out, weight, mw, pw64, B, F, P, spb, d_conv_out);
  return nnue_launch_status("nnue_ftb_backward_values");
}
