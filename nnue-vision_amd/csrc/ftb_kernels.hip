// FeatureTransformer for BINARY grid features, LDS-staged and batch-tiled (gfx950).
//
// Inside NNUE.forward the feature values are exactly {0,1} and the ids ascend (nnue.py:590-635), so
// the three FT products become gathers out of LDS tiles driven by bit masks / per-tile byte lists:
//
//   forward        out[b]  = bias + sum_{p active, p < F-1} W[p] + sink[b] * W[F-1]      (nnue.py:686-710)
//   weight grad    dW[f]   = sum_{b: bit(b,f)} dOut[b] ; dW[F-1] = sum_b sink[b]*dOut[b] ; db = sum_b dOut[b]
//   value grad     dX[b,p] = bit(b,p) ? <dOut[b], W[min(p,F-1)]> : 0                      (= d conv_out)
//
// where sink[b] counts the active ids >= F-1 (the clamp of nnue.py:701).  Instead of every sample
// re-reading its ~400 table rows from L2/HBM (the list kernels in ft_kernels.hip), a workgroup stages a
// tile of the table (or of dOut) in LDS ONCE and all of its samples (rows) gather from there: memory-side
// traffic drops by the sample-tile factor and the inner loop runs out of LDS.
//
// Layouts (include/nnue_hip.h):
//   maskW [B][pw64] u64      bit p of sample b (flat position bits)
//   maskT [F+1][bw64] u64    bit b of table row f; row F-1 = (sink[b] != 0), row F = every sample (bias)
//   sink  [B] float
//   tile lists  tl [R][NT][128] u16 + tc [R][NT] u8: for output r and tile t (128 staged rows) the LDS byte
//               offsets (local row * 256) of the staged rows to add, ascending, padded with 32768 (= an
//               all-zero LDS row).  Entry e sits at slot ((e>>2)&3)*32 + (e>>4)*4 + (e&3): lane group g of a
//               wave reads its entries {16c + 4g + k} as 64 contiguous bytes.
//     forward lists  : R = B samples,      tiles over table rows   (tlW / tcW)
//     backward lists : R = F + 1 outputs,  tiles over batch samples (tlT / tcT)
#include "common.h"

namespace {

using u64 = unsigned long long;

constexpr int kTC = 64;                       // columns per workgroup: 16 lanes x float4
constexpr int kRT = 128;                      // staged rows per LDS tile
constexpr int kWaves = 16;                    // waves per gather workgroup (4 per SIMD)
constexpr int kRecBytes = kRT * 2;            // one list record: 128 u16 offsets

__device__ __forceinline__ int list_pos(int e) { return ((e >> 2) & 3) * 32 + (e >> 4) * 4 + (e & 3); }

// Writes the tile list of one (output, tile) from its two 64-bit membership words.  One wave; lane = bit.
__device__ __forceinline__ void emit_tile_list(u64 m0, u64 m1, int lane, unsigned short* __restrict__ lst,
                                               unsigned char* __restrict__ cnt_out) {
  const u64 lt = (1ull << lane) - 1ull;
  const int c0 = __popcll(m0), cnt = c0 + __popcll(m1);
  if ((m0 >> lane) & 1ull) lst[list_pos(__popcll(m0 & lt))] = (unsigned short)(lane * (kTC * 4));
  if ((m1 >> lane) & 1ull) lst[list_pos(c0 + __popcll(m1 & lt))] = (unsigned short)((64 + lane) * (kTC * 4));
  if (lane >= cnt) lst[list_pos(lane)] = (unsigned short)(kRT * kTC * 4);  // padding -> zero row
  if (64 + lane >= cnt) lst[list_pos(64 + lane)] = (unsigned short)(kRT * kTC * 4);
  if (lane == 0) *cnt_out = (unsigned char)cnt;
}

// ------------------------------------------------------------------ bit masks + lists from conv_out
// grid (sample, slice); wave w of slice y takes position tiles 4y + w, 4y + w + 4 * slices, ... (128 positions =
// two ballots).  With more than one slice per sample (large maps: 512 tiles at 224x224) the per-sample counts are
// integer-valued atomics into buffers the host zeroes first -- exact and order-independent.
__global__ __launch_bounds__(256) void bits_rows_kernel(const float* __restrict__ conv_out,
                                                        const float* __restrict__ thr, int G, int P, int F,
                                                        u64* __restrict__ maskW, int pw64, float* __restrict__ sink,
                                                        int* __restrict__ n, unsigned short* __restrict__ tlW,
                                                        unsigned char* __restrict__ tcW, int ntW, int slices) {
  __shared__ int cnt_s[4], sink_s[4];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* __restrict__ x = conv_out + (size_t)b * P;
  const int direct = (F - 1 < P) ? F - 1 : P;  // positions with a table row of their own
  int cnt = 0, snk = 0;
  for (int t = blockIdx.y * 4 + wave; t < pw64 / 2; t += 4 * slices) {
    const int p0 = t * 128 + lane, p1 = p0 + 64;
    const bool on0 = (p0 < P) && (x[p0] > thr[p0 / G]);
    const bool on1 = (p1 < P) && (x[p1] > thr[p1 / G]);
    const u64 m0 = __ballot(on0), m1 = __ballot(on1);
    if (lane == 0) {
      maskW[(size_t)b * pw64 + 2 * t] = m0;
      maskW[(size_t)b * pw64 + 2 * t + 1] = m1;
    }
    cnt += __popcll(m0) + __popcll(m1);
    snk += __popcll(__ballot(on0 && p0 >= F - 1)) + __popcll(__ballot(on1 && p1 >= F - 1));
    if (t < ntW) {
      const u64 d0 = __ballot(on0 && p0 < direct), d1 = __ballot(on1 && p1 < direct);
      emit_tile_list(d0, d1, lane, tlW + ((size_t)b * ntW + t) * kRT, tcW + (size_t)b * ntW + t);
    }
  }
  if (lane == 0) {
    cnt_s[wave] = cnt;
    sink_s[wave] = snk;
  }
  __syncthreads();
  if (tid == 0) {
    const int total = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3], snk_total = sink_s[0] + sink_s[1] + sink_s[2] + sink_s[3];
    if (slices == 1) {
      n[b] = total;
      sink[b] = (float)snk_total;
    } else {
      atomicAdd(&n[b], total);
      if (snk_total) atomicAdd(&sink[b], (float)snk_total);  // small integers: exact in any order
    }
  }
}

// Transposed membership: 16 table rows x 128 samples per workgroup through an LDS byte tile + ballots.
constexpr int kTrRows = 16;
__global__ __launch_bounds__(256) void bits_transpose_kernel(const float* __restrict__ conv_out,
                                                             const float* __restrict__ thr,
                                                             const float* __restrict__ sink, int B, int G, int P,
                                                             int F, u64* __restrict__ maskT, int bw64,
                                                             unsigned short* __restrict__ tlT,
                                                             unsigned char* __restrict__ tcT, int ntT) {
  __shared__ unsigned char tile[128][kTrRows + 4];
  const int f0 = blockIdx.x * kTrRows, b0 = blockIdx.y * 128;
  const int tx = threadIdx.x & (kTrRows - 1), ty = threadIdx.x / kTrRows;  // 16 rows x 16 sample lanes
  const int f = f0 + tx;
  const int direct = (F - 1 < P) ? F - 1 : P;
  const float t = (f < direct) ? thr[f / G] : 0.0f;
  for (int j = ty; j < 128; j += 256 / kTrRows) {
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
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < kTrRows; j += 4) {
    const int fo = f0 + j;
    const u64 m0 = __ballot(tile[lane][j] != 0), m1 = __ballot(tile[64 + lane][j] != 0);  // lane = sample
    if (fo > F) continue;  // wave-uniform
    if (lane == 0) {
      maskT[(size_t)fo * bw64 + 2 * blockIdx.y] = m0;
      maskT[(size_t)fo * bw64 + 2 * blockIdx.y + 1] = m1;
    }
    emit_tile_list(m0, m1, lane, tlT + ((size_t)fo * ntT + blockIdx.y) * kRT, tcT + (size_t)fo * ntT + blockIdx.y);
  }
}

// ------------------------------------------------------------------ list-driven gather out of LDS
// MODE 0 (forward):     outputs = samples,    staged rows = table rows, lists tlW; bias + valued sink row
// MODE 1 (weight grad): outputs = table rows + bias row, staged rows = d_out rows, lists tlT; row F-1 valued by sink[]
//
// Workgroup = 8 waves x SW outputs each, 64 columns (grid.y), a slice of the tiles (grid.z).
//
// Staging is asynchronous LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write): a ring of NBUF slots,
// each holding one tile = 128 staged rows x 256 B, the 128-byte list records of the workgroup's outputs for
// that tile and (MODE 1) the staged samples' sink values.  With NBUF = 3 two tiles are always in flight while a
// third is consumed; arrival is tracked with a counted s_waitcnt vmcnt and ONE raw s_barrier per tile -- the
// loop contains no ordinary global load, so nothing drains the DMA queue early.  NBUF = 1 (single-tile
// problems) trades the ring for three workgroups per CU.
//
// Consumption is wave-uniform: a wave works on ONE output at a time; its four 16-lane groups fetch four
// DIFFERENT staged rows with one ds_read_b128 (1 KiB).  Each lane group therefore holds a partial sum over
// its quarter of the entries; the four partials are combined in a fixed order at the end.
constexpr int kSlotData = kRT * kTC * 4 + kTC * 4;  // 128 rows + the all-zero row, bytes
constexpr int kCoefBytes = 528;                     // 129 floats, padded

__device__ __forceinline__ void dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}

using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int SW, int MODE, int NBUF>
__global__ __launch_bounds__(1024) void ftb_gather_kernel(const float* __restrict__ src,   // [n_src][L1]
                                                          const float* __restrict__ bias,  // MODE 0
                                                          const unsigned short* __restrict__ tl, int nt,
                                                          const float* __restrict__ sink, int n_out, int n_src,
                                                          int sink_row, int L1, int tiles_per_split,
                                                          float* __restrict__ out, float* __restrict__ out_extra,
                                                          float* __restrict__ slabs) {
  constexpr int kListBytes = kWaves * SW * kRecBytes;  // list records of one tile (SW * 4 KiB)
  constexpr int kListPieces = kListBytes / 1024;       // 1 KiB DMA pieces: 4 * SW
  constexpr int kSlot = kSlotData + kListBytes + (MODE == 1 ? kCoefBytes : 0);
  __shared__ __attribute__((aligned(16))) char lds[NBUF * kSlot];  // ONE shared object (see guide: glds + 2nd object)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = lane >> 4, l16 = lane & 15;
  const int col = blockIdx.y * kTC + l16 * 4;
  const int obase = blockIdx.x * kWaves * SW;
  const int o0 = obase + wave * SW;
  const int t_lo = blockIdx.z * tiles_per_split;
  const int t_hi = min(nt, t_lo + tiles_per_split);
  // DMA instructions this wave issues per tile: 2 data pieces (+1 list piece, +1 coefficient piece)
  const bool lists_mine = wave < kListPieces;
  const bool coef_mine = (MODE == 1) && wave >= kWaves - 2;

  for (int b = 0; b < NBUF; ++b) {  // zero rows (index kRT) and the zero coefficient, never overwritten by DMA
    if (tid < kTC / 4) reinterpret_cast<float4*>(lds + b * kSlot + kRT * kTC * 4)[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 1 && tid == 0) reinterpret_cast<float*>(lds + b * kSlot + kSlotData + kListBytes)[kRT] = 0.f;
  }

  auto issue_tile = [&](int t, int slot) {
    char* base = lds + slot * kSlot;
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // one piece = 4 staged rows x 256 B = 1 KiB, lane-linear in LDS
      const int piece = wave * 2 + i;
      int r = t * kRT + piece * 4 + (lane >> 4);
      r = r < n_src ? r : max(n_src - 1, 0);  // rows past the end are never referenced by a list (row 0 always exists)
      dma16(src + (size_t)r * L1 + blockIdx.y * kTC + l16 * 4, base + piece * 1024);
    }
    if (lists_mine) {  // piece = 4 list records of 256 B
      int rec = obase + wave * 4 + (lane >> 4);
      rec = rec < n_out ? rec : n_out - 1;
      dma16(reinterpret_cast<const char*>(tl) + ((size_t)rec * nt + t) * kRecBytes + l16 * 16, base + kSlotData + wave * 1024);
    }
    if (coef_mine) {
      const int h = wave - (kWaves - 2);
      int r = t * kRT + h * 64 + lane;
      r = r < n_src ? r : max(n_src - 1, 0);
      dma4(sink + r, base + kSlotData + kListBytes + h * 256);
    }
  };
  auto wait_tiles_in_flight = [&](bool one_behind) {  // returns when every DMA except the youngest tile's has landed
    if (!one_behind) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (lists_mine && coef_mine) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if (lists_mine || coef_mine) {
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    }
  };

  f32x2 acc[SW][2];
#pragma unroll
  for (int s = 0; s < SW; ++s) acc[s][0] = acc[s][1] = (f32x2){0.f, 0.f};

  if (t_lo < t_hi) issue_tile(t_lo, 0);
  if (NBUF > 1 && t_lo + 1 < t_hi) issue_tile(t_lo + 1, 1);
  for (int t = t_lo; t < t_hi; ++t) {
    const int slot = (NBUF > 1) ? (t - t_lo) % NBUF : 0;
    wait_tiles_in_flight(NBUF > 1 && t + 1 < t_hi);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's own LDS traffic (zero rows, tile t-1 reads) has landed
    __builtin_amdgcn_s_barrier();  // tile t is complete in LDS; every wave is done with tile t-1
    asm volatile("" ::: "memory");
    if (NBUF > 1 && t + 2 < t_hi) issue_tile(t + 2, (t + 2 - t_lo) % NBUF);  // refills the slot tile t-1 used
    const char* __restrict__ sb = lds + slot * kSlot;
    const char* __restrict__ tb = sb + l16 * 16;
    const float* __restrict__ coef_t = reinterpret_cast<const float*>(sb + kSlotData + kListBytes);
    // Outputs are consumed G = 2 at a time, chunk by chunk: 8 ds_read_b128 in flight per wave, 16 waves per
    // CU, and the register count stays under the 128 a 1024-thread workgroup may use.  A list that has ended
    // keeps reading the zero row.
    constexpr int G = SW < 2 ? SW : 2;
    constexpr unsigned kPad = 0x80008000u;  // two padding offsets (32768)
#pragma unroll
    for (int s0 = 0; s0 < SW; s0 += G) {
      if (o0 + s0 >= n_out) break;  // wave-uniform
      unsigned words[G][16];
      bool valued[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const bool live = o0 + s0 + g < n_out;  // wave-uniform
        valued[g] = (MODE == 1) && (o0 + s0 + g == sink_row);
        const uint4* rec = reinterpret_cast<const uint4*>(sb + kSlotData + (wave * SW + s0 + g) * kRecBytes + grp * 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint4 w = rec[q];
          words[g][4 * q + 0] = live ? w.x : kPad; words[g][4 * q + 1] = live ? w.y : kPad;
          words[g][4 * q + 2] = live ? w.z : kPad; words[g][4 * q + 3] = live ? w.w : kPad;
        }
      }
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        bool any_live = false;
#pragma unroll
        for (int g = 0; g < G; ++g) any_live |= words[g][2 * ch] != kPad;  // a chunk's first pair is padding only if all of it is
        if (!__any(any_live)) break;  // both lists have ended
        float4 v[G][4];
        unsigned off[G][4];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const unsigned e0 = words[g][2 * ch], e1 = words[g][2 * ch + 1];
          off[g][0] = e0 & 0xffffu; off[g][1] = e0 >> 16; off[g][2] = e1 & 0xffffu; off[g][3] = e1 >> 16;
#pragma unroll
          for (int k = 0; k < 4; ++k) v[g][k] = *reinterpret_cast<const float4*>(tb + off[g][k]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (!valued[g]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              acc[s0 + g][0] += (f32x2){v[g][k].x, v[g][k].y};
              acc[s0 + g][1] += (f32x2){v[g][k].z, v[g][k].w};
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const float c = coef_t[off[g][k] >> 8];
              acc[s0 + g][0] += (f32x2){c * v[g][k].x, c * v[g][k].y};
              acc[s0 + g][1] += (f32x2){c * v[g][k].z, c * v[g][k].w};
            }
          }
        }
      }
    }
    if (NBUF == 1) __builtin_amdgcn_s_barrier();  // single slot: everyone is done before it is refilled
    if (NBUF == 1 && t + 1 < t_hi) issue_tile(t + 1, 0);
  }
  // combine the four lane-group partials (fixed order), then lanes 0..15 hold the 64-column result
#pragma unroll
  for (int s = 0; s < SW; ++s) {
    float4 a = make_float4(acc[s][0].x, acc[s][0].y, acc[s][1].x, acc[s][1].y);
    a.x += __shfl_xor(a.x, 16); a.y += __shfl_xor(a.y, 16); a.z += __shfl_xor(a.z, 16); a.w += __shfl_xor(a.w, 16);
    a.x += __shfl_xor(a.x, 32); a.y += __shfl_xor(a.y, 32); a.z += __shfl_xor(a.z, 32); a.w += __shfl_xor(a.w, 32);
    const int o = o0 + s;
    if (o >= n_out || grp != 0) continue;
    if (slabs != nullptr) {  // tiles were split over grid.z: partial sums, finished by ftb_finish_kernel
      *reinterpret_cast<float4*>(slabs + ((size_t)blockIdx.z * n_out + o) * L1 + col) = a;
      continue;
    }
    if (MODE == 0) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + col);
      a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
      if (sink_row >= 0) {  // folded ids: sink[b] copies of table row F-1
        const float sv = sink[o];
        if (sv != 0.f) {
          const float4 w = *reinterpret_cast<const float4*>(src + (size_t)sink_row * L1 + col);
          a.x = fmaf(sv, w.x, a.x); a.y = fmaf(sv, w.y, a.y); a.z = fmaf(sv, w.z, a.z); a.w = fmaf(sv, w.w, a.w);
        }
      }
      *reinterpret_cast<float4*>(out + (size_t)o * L1 + col) = a;
    } else {
      float* dst = (o == n_out - 1) ? out_extra : out;  // last output row = bias gradient
      if (dst != nullptr) *reinterpret_cast<float4*>(dst + (o == n_out - 1 ? (size_t)0 : (size_t)o * L1) + col) = a;
    }
  }
}

// Sums the split slabs in order and applies what the unsplit kernel does in its epilogue.
template <int MODE>
__global__ __launch_bounds__(256) void ftb_finish_kernel(const float* __restrict__ slabs, int splits, int n_out, int L1,
                                                         const float* __restrict__ src, const float* __restrict__ bias,
                                                         const float* __restrict__ sink, int sink_row,
                                                         float* __restrict__ out, float* __restrict__ out_extra) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // float4 index
  const int q = L1 / 4;
  if (i >= (long long)n_out * q) return;
  const int o = (int)(i / q), col = (int)(i - (long long)o * q) * 4;
  float4 a = *reinterpret_cast<const float4*>(slabs + (size_t)o * L1 + col);
  for (int s = 1; s < splits; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(slabs + ((size_t)s * n_out + o) * L1 + col);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (MODE == 0) {
    const float4 bv = *reinterpret_cast<const float4*>(bias + col);
    a.x += bv.x; a.y += bv.y; a.z += bv.z; a.w += bv.w;
    if (sink_row >= 0) {
      const float sv = sink[o];
      if (sv != 0.f) {
        const float4 w = *reinterpret_cast<const float4*>(src + (size_t)sink_row * L1 + col);
        a.x = fmaf(sv, w.x, a.x); a.y = fmaf(sv, w.y, a.y); a.z = fmaf(sv, w.z, a.z); a.w = fmaf(sv, w.w, a.w);
      }
    }
    *reinterpret_cast<float4*>(out + (size_t)o * L1 + col) = a;
  } else {
    float* dst = (o == n_out - 1) ? out_extra : out;
    if (dst != nullptr) *reinterpret_cast<float4*>(dst + (o == n_out - 1 ? (size_t)0 : (size_t)o * L1) + col) = a;
  }
}

// ------------------------------------------------------------------ value gradient
// Workgroup (8 waves) = (sample tile) x (RB table rows).  The rows (RB x L1 floats, 64 KB at L1 = 1024) are
// staged in LDS once; each wave then walks its samples: d_out[b] sits in registers (prefetched one sample
// ahead together with the sample's mask word), the set bits of the 16-bit mask slice pick LDS rows, the dot
// products are transposed-and-reduced across the wave with a butterfly, and ranks map them back to bit
// positions, so the RB outputs leave as ONE coalesced store (zeros included: no separate zero fill).
constexpr int kValWaves = 8;  // waves per value-gradient workgroup
template <int S>
__global__ __launch_bounds__(512) void ftb_values_kernel(const float* __restrict__ d_out,
                                                         const float* __restrict__ W,
                                                         const u64* __restrict__ maskW, int pw64, int B, int F,
                                                         int P, int samples_per_block, float* __restrict__ dst) {
  constexpr int L1 = 256 * S;
  constexpr int RB = 16;  // rows per workgroup = one butterfly round
  __shared__ __attribute__((aligned(16))) float rows[(RB + 1) * L1];  // + an all-zero row
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool has_sink = F - 1 < P;              // ids >= F-1 exist and fold into row F-1
  const int p_lim = has_sink ? F - 1 : P;       // positions below p_lim have a table row of their own
  const int r_last = has_sink ? F - 1 : P - 1;  // highest row that can be referenced
  const int r0 = blockIdx.y * RB;
  const int nrows = min(RB, r_last + 1 - r0);
  for (int i = tid; i < nrows * (L1 / 4); i += 512) {
    const int r = i / (L1 / 4), c4 = i - r * (L1 / 4);
    reinterpret_cast<float4*>(rows)[i] = *reinterpret_cast<const float4*>(W + (size_t)(r0 + r) * L1 + c4 * 4);
  }
  for (int i = tid; i < L1 / 4; i += 512) reinterpret_cast<float4*>(rows + RB * L1)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  const bool sink_block = has_sink && (F - 1 >= r0) && (F - 1 < r0 + RB);
  const int b_lo = blockIdx.x * samples_per_block;
  const int b_hi = min(B, b_lo + samples_per_block);
  const int nd = p_lim - r0;  // direct positions in this block's RB-wide slice
  const unsigned keep = nd >= RB ? ((1u << RB) - 1u) : (nd <= 0 ? 0u : ((1u << nd) - 1u));
  const float* __restrict__ rl = rows + lane * 4;
  const int L = lane & (RB - 1);

  float4 g[S], gn[S];
  u64 word = 0, wordn = 0;
  int b = b_lo + wave;
  if (b < b_hi) {
#pragma unroll
    for (int s = 0; s < S; ++s) g[s] = *reinterpret_cast<const float4*>(d_out + (size_t)b * L1 + s * 256 + lane * 4);
    word = maskW[(size_t)b * pw64 + (r0 >> 6)];
  }
  for (; b < b_hi; b += kValWaves) {
    const int bn = b + kValWaves;
    if (bn < b_hi) {
#pragma unroll
      for (int s = 0; s < S; ++s) gn[s] = *reinterpret_cast<const float4*>(d_out + (size_t)bn * L1 + s * 256 + lane * 4);
      wordn = maskW[(size_t)bn * pw64 + (r0 >> 6)];
    }
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(word >> (r0 & 63))) & keep;
    float p[RB];
    unsigned mm = m;
    const int cnt = __popc(m);
#pragma unroll
    for (int u0 = 0; u0 < RB; u0 += 4) {
      if (u0 >= cnt) {  // wave-uniform: nothing left
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u0 + u] = 0.f;
        continue;
      }
      // four entries at a time, no branches inside: entries past the end read the all-zero row RB
      int j[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        j[u] = mm ? __builtin_ctz(mm) : RB;
        mm &= mm - 1;
      }
      float4 v[4][S];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int s = 0; s < S; ++s) v[u][s] = *reinterpret_cast<const float4*>(rl + j[u] * L1 + s * 256);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float a = 0.f;
#pragma unroll
        for (int s = 0; s < S; ++s) a += fmaf(v[u][s].w, g[s].w, fmaf(v[u][s].z, g[s].z, fmaf(v[u][s].y, g[s].y, v[u][s].x * g[s].x)));
        p[u0 + u] = a;
      }
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
    tot += __shfl_xor(tot, 32);  // every lane: dot product of the (lane & 15)-th set bit
    const int rank = __popc(m & ((1u << L) - 1u));
    const float v = __shfl(tot, rank);
    const float res = ((m >> L) & 1u) ? v : 0.f;
    if (lane < RB && lane < nd) dst[(size_t)b * P + r0 + lane] = res;
    if (sink_block) {  // every active id >= F-1 receives <d_out[b], W[F-1]>
      const float* __restrict__ wr = rl + (F - 1 - r0) * L1;
      float a = 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(wr + s * 256);
        a += fmaf(v.w, g[s].w, fmaf(v.z, g[s].z, fmaf(v.y, g[s].y, v.x * g[s].x)));
      }
#pragma unroll
      for (int sft = 32; sft >= 1; sft >>= 1) a += __shfl_xor(a, sft);
      for (int pp = F - 1 + lane; pp < P; pp += 64) {
        const bool on = (maskW[(size_t)b * pw64 + (pp >> 6)] >> (pp & 63)) & 1ull;
        dst[(size_t)b * P + pp] = on ? a : 0.f;
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) g[s] = gn[s];
    word = wordn;
  }
}

struct GatherPlan {
  int sw, splits, tiles_per_split;
};

GatherPlan plan_gather(int n_out, int nt, int col_tiles) {
  // Outputs per wave (SW): more outputs per staged tile = less staging traffic, as long as about one
  // workgroup per CU remains (the ring version runs one 16-wave workgroup per CU).  When the outputs alone
  // cannot fill the chip and there are many tiles (big tables), keep SW = 2 and split the tiles over grid.z
  // instead (partial slabs + a finish pass).
  GatherPlan p{4, 1, nt > 0 ? nt : 1};
  auto groups = [&](int sw) { return (long long)((n_out + kWaves * sw - 1) / (kWaves * sw)) * col_tiles; };
  if (groups(2) < 192 && nt >= 8) {
    p.sw = 2;
    int want = (int)((256 + groups(2) - 1) / groups(2));
    if (want > nt / 2) want = nt / 2;
    if (want > 16) want = 16;
    p.tiles_per_split = (nt + want - 1) / want;
    p.splits = (nt + p.tiles_per_split - 1) / p.tiles_per_split;
    return p;
  }
  while (p.sw > 1 && groups(p.sw) < 200) p.sw >>= 1;
  return p;
}

template <int MODE>
int launch_gather(hipStream_t s, const float* src, const float* bias, const unsigned short* tl, int nt, const float* sink, int n_out,
                  int n_src, int sink_row, int L1, float* out, float* extra, void* scratch, int64_t scratch_bytes, const char* who) {
  const int col_tiles = L1 / kTC;
  const GatherPlan p = plan_gather(n_out, nt, col_tiles);
  float* slabs = nullptr;
  if (p.splits > 1) {
    const int64_t need = (int64_t)p.splits * n_out * L1 * (int64_t)sizeof(float);
    NNUE_REQUIRE(scratch && scratch_bytes >= need, NNUE_E_SCRATCH, "%s: scratch %lld < %lld bytes", who, (long long)scratch_bytes,
                 (long long)need);
    NNUE_REQUIRE(nnue_aligned16(scratch), NNUE_E_ARG, "%s: scratch must be 16-byte aligned", who);
    slabs = static_cast<float*>(scratch);
  }
  const dim3 grid((n_out + kWaves * p.sw - 1) / (kWaves * p.sw), col_tiles, p.splits), block(64 * kWaves);
  const bool ring = p.tiles_per_split > 1;  // single-tile problems: one slot, three workgroups per CU
#define NNUE_LAUNCH_GATHER(SWV, NB)                                                                                            \
  hipLaunchKernelGGL((ftb_gather_kernel<SWV, MODE, NB>), grid, block, 0, s, src, bias, tl, nt, sink, n_out, n_src, sink_row, L1, \
                     p.tiles_per_split, out, extra, slabs)
#define NNUE_LAUNCH_GATHER_SW(SWV) \
  do {                             \
    if (ring) NNUE_LAUNCH_GATHER(SWV, 3); \
    else NNUE_LAUNCH_GATHER(SWV, 1);      \
  } while (0)
  if (p.sw == 4) NNUE_LAUNCH_GATHER_SW(4);
  else if (p.sw == 2) NNUE_LAUNCH_GATHER_SW(2);
  else NNUE_LAUNCH_GATHER_SW(1);
#undef NNUE_LAUNCH_GATHER_SW
#undef NNUE_LAUNCH_GATHER
  if (p.splits > 1) {
    const long long q = (long long)n_out * (L1 / 4);
    hipLaunchKernelGGL(ftb_finish_kernel<MODE>, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, s, slabs, p.splits, n_out, L1, src,
                       bias, sink, sink_row, out, extra);
  }
  return nnue_launch_status(who);
}

int direct_rows(int F, int P) { return (F - 1 < P) ? F - 1 : P; }

}  // namespace

// =============================================================================== C ABI
extern "C" int nnue_ftb_supported(int L1) { return (L1 == 256 || L1 == 512 || L1 == 1024) ? 1 : 0; }

extern "C" int nnue_ftb_list_tiles(int B, int F, int P, int* tiles_fwd, int* tiles_bwd) {
  if (B <= 0 || F <= 0 || P <= 0 || !tiles_fwd || !tiles_bwd) return NNUE_E_ARG;
  const int d = direct_rows(F, P);
  *tiles_fwd = d > 0 ? (d + kRT - 1) / kRT : 1;
  *tiles_bwd = (B + kRT - 1) / kRT;
  return NNUE_OK;
}

extern "C" int64_t nnue_ftb_scratch(int B, int F, int P, int L1) {
  if (B <= 0 || F <= 0 || P <= 0 || L1 <= 0 || L1 % kTC) return 0;
  int ntf = 0, ntb = 0;
  nnue_ftb_list_tiles(B, F, P, &ntf, &ntb);
  const GatherPlan a = plan_gather(B, ntf, L1 / kTC), b = plan_gather(F + 1, ntb, L1 / kTC);
  const int64_t fa = a.splits > 1 ? (int64_t)a.splits * B * L1 : 0, fb = b.splits > 1 ? (int64_t)b.splits * (F + 1) * L1 : 0;
  return ((fa > fb ? fa : fb) + 4) * (int64_t)sizeof(float);
}

extern "C" int nnue_binarize_bits(const float* conv_out, const float* thr, int B, int fps, int Gh, int Gw, int F,
                                  uint64_t* maskW, int pw64, uint64_t* maskT, int bw64, float* sink, int32_t* n,
                                  uint8_t* tlW, uint8_t* tcW, uint8_t* tlT, uint8_t* tcT, int stages, nnue_stream_t stream) {
  NNUE_REQUIRE(conv_out && thr && maskW && maskT && sink && n && tlW && tcW && tlT && tcT, NNUE_E_ARG,
               "nnue_binarize_bits: null pointer");
  NNUE_REQUIRE(B > 0 && fps > 0 && Gh > 0 && Gw > 0 && F > 0, NNUE_E_ARG,
               "nnue_binarize_bits: B=%d fps=%d Gh=%d Gw=%d F=%d must be positive", B, fps, Gh, Gw, F);
  const long long P64 = (long long)fps * Gh * Gw;
  NNUE_REQUIRE(P64 < (1ll << 30), NNUE_E_SHAPE, "nnue_binarize_bits: fps*Gh*Gw too large");
  const int P = (int)P64;
  NNUE_REQUIRE(pw64 % 2 == 0 && (long long)pw64 * 64 >= P, NNUE_E_SHAPE,
               "nnue_binarize_bits: pw64=%d must be even and cover %d positions", pw64, P);
  NNUE_REQUIRE(bw64 % 2 == 0 && (long long)bw64 * 64 >= B, NNUE_E_SHAPE,
               "nnue_binarize_bits: bw64=%d must be even and cover %d samples", bw64, B);
  NNUE_REQUIRE(nnue_aligned16(tlW) && nnue_aligned16(tlT), NNUE_E_ARG, "nnue_binarize_bits: list buffers must be 16-byte aligned");
  int ntf = 0, ntb = 0;
  nnue_ftb_list_tiles(B, F, P, &ntf, &ntb);
  hipStream_t s = static_cast<hipStream_t>(stream);
  NNUE_REQUIRE(stages >= 1 && stages <= 3, NNUE_E_ARG, "nnue_binarize_bits: stages must be 1 (per-sample), 2 (transposed) or 3 (both)");
  if (stages & 1) {
    int slices = (pw64 / 2) / 32;  // >= 8 tiles per wave before a sample is split
    slices = slices < 1 ? 1 : (slices > 32 ? 32 : slices);
    if (slices > 1) nnue_zero_counters(n, sink, B, s);  // a kernel, not a memset node (common.h)
    hipLaunchKernelGGL(bits_rows_kernel, dim3(B, slices), dim3(256), 0, s, conv_out, thr, Gh * Gw, P, F, reinterpret_cast<u64*>(maskW), pw64,
                       sink, n, reinterpret_cast<unsigned short*>(tlW), tcW, ntf, slices);
  }
  if (stages & 2)
    hipLaunchKernelGGL(bits_transpose_kernel, dim3((F + 1 + kTrRows - 1) / kTrRows, ntb), dim3(256), 0, s, conv_out, thr, sink, B, Gh * Gw, P, F,
                     reinterpret_cast<u64*>(maskT), bw64, reinterpret_cast<unsigned short*>(tlT), tcT, ntb);
  return nnue_launch_status("nnue_binarize_bits");
}

extern "C" int nnue_ftb_forward(const float* weight, const float* bias, const uint8_t* tlW, const uint8_t* tcW, const float* sink,
                                int B, int F, int P, int L1, float* out, void* scratch, int64_t scratch_bytes,
                                nnue_stream_t stream) {
  NNUE_REQUIRE(weight && bias && tlW && tcW && sink && out, NNUE_E_ARG, "nnue_ftb_forward: null pointer");
  NNUE_REQUIRE(B > 0 && F > 0 && P > 0, NNUE_E_ARG, "nnue_ftb_forward: B=%d F=%d P=%d must be positive", B, F, P);
  NNUE_REQUIRE(nnue_ftb_supported(L1), NNUE_E_SHAPE, "nnue_ftb_forward: L1=%d not supported (256/512/1024)", L1);
  NNUE_REQUIRE(nnue_aligned16(weight) && nnue_aligned16(bias) && nnue_aligned16(out) && nnue_aligned16(tlW), NNUE_E_ARG,
               "nnue_ftb_forward: pointers must be 16-byte aligned");
  int ntf = 0, ntb = 0;
  nnue_ftb_list_tiles(B, F, P, &ntf, &ntb);
  const int d = direct_rows(F, P);
  (void)tcW;  // counts are implied by the padding; kept in the ABI for consumers that want them
  return launch_gather<0>(static_cast<hipStream_t>(stream), weight, bias, reinterpret_cast<const unsigned short*>(tlW), ntf, sink, B, d, (F - 1 < P) ? F - 1 : -1, L1, out,
                          nullptr, scratch, scratch_bytes, "nnue_ftb_forward");
}

extern "C" int nnue_ftb_backward_weight(const float* d_out, const uint8_t* tlT, const uint8_t* tcT, const float* sink, int B, int F,
                                        int P, int L1, float* d_weight, float* d_bias, void* scratch, int64_t scratch_bytes,
                                        nnue_stream_t stream) {
  NNUE_REQUIRE(d_out && tlT && tcT && sink, NNUE_E_ARG, "nnue_ftb_backward_weight: null pointer");
  NNUE_REQUIRE(d_weight || d_bias, NNUE_E_ARG, "nnue_ftb_backward_weight: both outputs are null");
  NNUE_REQUIRE(B > 0 && F > 0 && P > 0, NNUE_E_ARG, "nnue_ftb_backward_weight: B=%d F=%d P=%d must be positive", B, F, P);
  NNUE_REQUIRE(nnue_ftb_supported(L1), NNUE_E_SHAPE, "nnue_ftb_backward_weight: L1=%d not supported (256/512/1024)", L1);
  NNUE_REQUIRE(nnue_aligned16(d_out) && nnue_aligned16(d_weight) && nnue_aligned16(d_bias) && nnue_aligned16(tlT), NNUE_E_ARG,
               "nnue_ftb_backward_weight: pointers must be 16-byte aligned");
  int ntf = 0, ntb = 0;
  nnue_ftb_list_tiles(B, F, P, &ntf, &ntb);
  // outputs: F table rows + the bias row; row F-1 is valued by sink[]; rows the map cannot reach have empty lists
  (void)tcT;
  return launch_gather<1>(static_cast<hipStream_t>(stream), d_out, nullptr, reinterpret_cast<const unsigned short*>(tlT), ntb, sink, F + 1, B, F - 1, L1, d_weight, d_bias, scratch,
                          scratch_bytes, "nnue_ftb_backward_weight");
}

extern "C" int nnue_ftb_backward_values(const float* d_out, const float* weight, const uint64_t* maskW, int pw64, int B,
                                        int F, int P, int L1, float* d_conv_out, nnue_stream_t stream) {
  NNUE_REQUIRE(d_out && weight && maskW && d_conv_out, NNUE_E_ARG, "nnue_ftb_backward_values: null pointer");
  NNUE_REQUIRE(B > 0 && F > 0 && P > 0, NNUE_E_ARG, "nnue_ftb_backward_values: B=%d F=%d P=%d must be positive", B, F, P);
  NNUE_REQUIRE(nnue_ftb_supported(L1), NNUE_E_SHAPE, "nnue_ftb_backward_values: L1=%d not supported (256/512/1024)", L1);
  NNUE_REQUIRE(pw64 % 2 == 0 && (long long)pw64 * 64 >= P, NNUE_E_SHAPE, "nnue_ftb_backward_values: pw64=%d does not cover P=%d", pw64, P);
  NNUE_REQUIRE(nnue_aligned16(d_out) && nnue_aligned16(weight), NNUE_E_ARG, "nnue_ftb_backward_values: pointers must be 16-byte aligned");
  const int r_last = (F - 1 < P) ? F - 1 : P - 1;
  const int row_blocks = r_last / 16 + 1;
  int spb = 128;  // samples per workgroup: fewer when the grid would be too small
  while (spb > 16 && (long long)((B + spb - 1) / spb) * row_blocks < 512) spb >>= 1;
  const dim3 grid((B + spb - 1) / spb, row_blocks), block(512);
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
