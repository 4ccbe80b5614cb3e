// FeatureTransformer value gradient for big maps (the 224x224 shape: 65 536 positions, batch 128):
//     d_conv_out[b][p] = bit[b][p] * sum_k d_out[b][k] * W[min(p, F-1)][k]         (autograd of nnue.py:702-708, :628-633)
// i.e. C[M = B][N = P] = D[M][K = L1] * W^T with W streamed from HBM exactly once (268 MB) and D (0.5 MB) shared by
// every workgroup.  Both operands are f32, so the product runs on the bf16 matrix unit as six plane products of the exact
// three-way truncation split (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi -- smallest first; see gemm_tile_bf6 in
// ftm_kernels.hip, whose arithmetic this kernel repeats term for term).
//
// What round 3's ablations of gemm_tile_bf6 showed (profiles/r03c_val_ablation.txt): that kernel is not waiting for the
// matrix unit (without any MFMA it still takes 90 of its 108 us) but for its staging -- 36 KB of ds_write_b128 per
// workgroup and K tile, two thirds of them the SAME d_out planes every one of the 1024 workgroups re-splits and re-stores,
// behind two barriers per K tile.  This kernel removes the register->LDS path altogether:
//   * D is split ONCE per launch into three bf16 planes in a workspace (split_planes_kernel, 0.75 MB) and its K tiles
//     arrive in LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no VALU, no ds_write), double-buffered, the XOR swizzle
//     of the image applied to the per-lane SOURCE address;
//   * W never touches LDS: the four waves of a workgroup sit side by side along N (each 128 rows x 32 columns), so a table
//     fragment is used by exactly one wave -- it is loaded in fragment layout straight into registers (lane = (column r, k
//     octet q): two 16-byte loads, 128 contiguous bytes per table row and K tile), split there (44 VALU per 8 values) and
//     multiplied against all eight row blocks of the tile;
//   * 128 x 128 tiles: the D planes are staged half as often per column as with 128 x 64; one barrier per K tile;
//   * three K tiles in flight per wave (the first version, one tile ahead behind a vmcnt(0), was latency-bound on the table
//     stream: 32 KB in flight per CU; without any MFMA it still took 76 of its 101 us): the table loads are inline assembly so
//     that the compiler does not drain the queue, and a counted s_waitcnt hands their registers back.
// Per wave and K tile of 32: 96 MFMAs (1536 matrix cycles), 24 ds_read_b128, ~120 VALU, 6 DMA pieces, 4 loads.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "ftv_kernels.h"

namespace {
using u32x4 = __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned;
using f32x4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;
using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;

constexpr int kBM = 128, kKT = 32;  // tile rows, K tile; the tile is 64 * NB columns wide (NB = 16-column blocks per wave: 2 or 4)
constexpr int kPlane = kBM * kKT * 2;  // bytes of one plane of a K tile: 128 rows x 64 bytes
constexpr int kBuf = 3 * kPlane;       // 24 KB per K tile; kDepth of them

// [row][32 k] bf16 image, 64-byte rows, 16-byte chunk XOR-swizzled per block of four rows with {0, 3, 2, 1}: conflict-free
// for the fragment read (16 rows x 4 chunks per ds_read_b128) under gfx950's lane groups (same image as bf6_img<32>)
__device__ __forceinline__ int img(int row, int chunk) { return row * 64 + ((chunk ^ ((4 - (row >> 2)) & 3)) << 4); }

// exact three-way truncation split of 8 consecutive k (two float4) into one 16-byte chunk of 8 bf16 per plane
__device__ __forceinline__ void split8(const u32x4& v0, const u32x4& v1, u32x4& hi, u32x4& mid, u32x4& lo) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned xb = e < 4 ? v0[e] : v1[e - 4];
    const float x = __uint_as_float(xb);
    const unsigned hb = xb & 0xffff0000u;
    const float r1 = x - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    h[e] = hb; m[e] = mb; l[e] = __float_as_uint(r2);
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {  // word t: k = 2 t (low half), 2 t + 1 (high half)
    hi[t] = __builtin_amdgcn_perm(h[2 * t + 1], h[2 * t], 0x07060302u);
    mid[t] = __builtin_amdgcn_perm(m[2 * t + 1], m[2 * t], 0x07060302u);
    lo[t] = __builtin_amdgcn_perm(l[2 * t + 1], l[2 * t], 0x07060302u);
  }
}

// d_out [rows][cols] f32 -> planes, K-TILE-MAJOR: [cols / 32][3 planes][rows][32 k] bf16 (hi, mid, lo); one thread per 8 values.
// Every workgroup of the value gradient reads the same K tile of these planes at about the same time; in the row-major form a K
// tile is 128 pieces of 64 bytes at a stride of one row (2 or 4 KB), i.e. one L2 channel for the whole tile.  Tile-major, a K tile
// is 24 KB of consecutive bytes spread over every channel.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, int rows, int cols, unsigned short* __restrict__ planes) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // 8-value run i: row i / (cols / 8), k = 8 * (i % (cols / 8))
  const int runs = cols / 8;
  if (i >= (long long)rows * runs) return;
  const int m = (int)(i / runs), k = (int)(i % runs) * 8;
  const u32x4 v0 = *reinterpret_cast<const u32x4*>(src + i * 8), v1 = *reinterpret_cast<const u32x4*>(src + i * 8 + 4);
  u32x4 hi, mid, lo;
  split8(v0, v1, hi, mid, lo);
  const size_t plane = (size_t)rows * 32, base = ((size_t)(k >> 5) * 3 * rows + m) * 32 + (k & 31);
  *reinterpret_cast<u32x4*>(planes + base) = hi;
  *reinterpret_cast<u32x4*>(planes + base + plane) = mid;
  *reinterpret_cast<u32x4*>(planes + base + 2 * plane) = lo;
}

__device__ __forceinline__ void dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// One 16-byte table load whose completion the COMPILER does not track (inline assembly): with LDS-DMA in flight hipcc would
// otherwise wait vmcnt(0) at the first use of any ordinary load result and so drain the whole prefetch queue each K tile.
// The registers are handed back by wait_tiles(), which carries them as in/out operands of the counted s_waitcnt.
template <bool NT>
__device__ __forceinline__ void load16(u32x4& dst, const float* base, unsigned off) {
  if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(off), "s"(base) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory");
}
template <bool NT>
__device__ __forceinline__ void load16_hi(u32x4& dst, const float* base, unsigned off) {
  if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, %2 offset:16 nt" : "=v"(dst) : "v"(off), "s"(base) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(dst) : "v"(off), "s"(base) : "memory");
}

constexpr int kDepth = 3;            // K tiles in flight: tile t computes while t + 1 and t + 2 travel
// per wave and K tile: six DMA pieces, then 2 * NB table loads (issue order = vmcnt order)

// ABL: timing-only ablations (tools/debug, wrong results unless 0): 1 no DMA after the prologue, 2 no table loads after the
// prologue.  (A "no MFMA" variant spilled 109 registers to scratch beside the asm loads and faulted: not kept.)
template <int ABL, bool NT, int NB>
__global__ __launch_bounds__(256, NB == 2 ? 2 : 1) void ftv_values_kernel(const unsigned short* __restrict__ planes,  // [K / 32][3][M][32] bf16
                                                         const float* __restrict__ weight, const uint8_t* __restrict__ bits,
                                                         float* __restrict__ out, int M, int N, int K, int F, int prio_mode) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[kDepth * kBuf];
  // Two workgroups share a CU, one wave of each per SIMD, both running this same program: left alone they fall into
  // lockstep (both in their MFMA phase, each at half rate, then both in their split / wait / barrier phase with the matrix
  // pipe idle: SQ counters of the first version -- issue-stalled 52 % of the wave cycles, pipe busy 51 %).  A static priority
  // for one of the two lets it take the pipe whole, so that the phases interleave.  Which workgroups share a CU is the
  // dispatcher's business; the guesses below are for speed only (results do not depend on them).
  if (prio_mode == 1) { if (blockIdx.x >= (gridDim.x >> 1)) __builtin_amdgcn_s_setprio(2); }
  else if (prio_mode == 2) { if (blockIdx.x & 1) __builtin_amdgcn_s_setprio(2); }
  else if (prio_mode == 3) { if ((blockIdx.x >> 3) & 1) __builtin_amdgcn_s_setprio(2); }
  else if (prio_mode == 4) { if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(2); }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  constexpr int kBN = 64 * NB;
  const int n_base = blockIdx.x * kBN, m_base = blockIdx.y * kBM;
  const int n_wave = n_base + 16 * NB * wave;
  const int tiles = K / kKT;

  // ---- D planes by LDS-DMA.  Piece = 1 KiB = 16 rows x 64 bytes of one plane; 24 pieces per K tile, six per wave.  Lane l
  // lands at byte 16 l of the piece = (row 16 g + l / 4, physical chunk l % 4) and therefore fetches the logical chunk
  // (l % 4) ^ swizzle(row).  Rows past M re-read row M - 1 (their accumulators are never stored).
  const unsigned short* src_piece[6];
  int dst_piece[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int piece = wave + 4 * i, plane = piece >> 3, g = piece & 7;
    const int row = 16 * g + (lane >> 2);
    const int chunk = (lane & 3) ^ ((4 - (row >> 2)) & 3);
    const int m = m_base + row < M ? m_base + row : M - 1;
    src_piece[i] = planes + ((size_t)plane * M + m) * 32 + chunk * 8;  // K-tile-major planes: tile t at + t * 3 * M * 32
    dst_piece[i] = plane * kPlane + g * 1024;
  }
  // ---- table fragments straight into registers: column block j of this wave, lane (r, q) = 8 consecutive k of row
  // min(n, F - 1): two 16-byte loads, 128 contiguous bytes per table row and K tile
  unsigned w_off[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n_wave + 16 * j + r;
    const int row = n < F - 1 ? n : F - 1;
    w_off[j] = (unsigned)(row * K + 8 * q) * 4u;
  }
  u32x4 rw[kDepth][NB][2];
  // Tile t into slot t % kDepth: DMA pieces first, table loads after them.  Issued UNCONDITIONALLY, with t clamped to the last
  // tile: past the end the queue is kept full with re-reads of the last tile into slots nobody reads any more, so that every
  // step waits with the same counted s_waitcnt.  (A run-time choice between vmcnt(N) and vmcnt(0) made the compiler copy the asm
  // loads' registers ahead of the wait on one of the two paths -- the destination of an asm load counts as written at the
  // statement -- and the last K tile was intermittently read before it had landed: found by the knob matrix, audited in the ISA.)
  auto issue = [&](int t_req, auto slot_tag) {
    constexpr int S = decltype(slot_tag)::value;
    const int t = t_req < tiles ? t_req : tiles - 1;
    if (!(ABL == 1 && t_req >= kDepth)) {
#pragma unroll
      for (int i = 0; i < 6; ++i) dma16(src_piece[i] + (size_t)t * (3 * 32) * M, smem + S * kBuf + dst_piece[i]);
    }
    if (!(ABL == 2 && t_req >= kDepth)) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        load16<NT>(rw[S][j][0], weight, w_off[j] + (unsigned)t * (kKT * 4));
        load16_hi<NT>(rw[S][j][1], weight, w_off[j] + (unsigned)t * (kKT * 4));
      }
    }
  };

  f32x4 acc[8][NB];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int a_off[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a_off[i] = img(16 * i + r, q);

  // One K tile.
  auto step = [&](int t, auto slot_tag) {
    constexpr int S = decltype(slot_tag)::value;
    // On entry tiles t and t + 1 are in flight (t + 2 is issued below, after the barrier has freed its slot): everything of
    // tile t must have landed, tile t + 1's 6 + 2 NB operations may stay out.  The registers come back through the operands --
    // an asm load's destination counts as written at the statement, so without this hand-back the compiler may reuse it before
    // the data lands (a too-large count here showed up as a memory fault: a late load landed in a register holding an address).
    if constexpr (NB == 2) {
      u32x4 &w00 = rw[S][0][0], &w01 = rw[S][0][1], &w10 = rw[S][1][0], &w11 = rw[S][1][1];
      if constexpr (ABL != 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w00), "+v"(w01), "+v"(w10), "+v"(w11)::"memory");
      else asm volatile("s_waitcnt vmcnt(10)" : "+v"(w00), "+v"(w01), "+v"(w10), "+v"(w11)::"memory");
    } else {
      static_assert(NB == 2 || NB == 4, "column blocks per wave");
      u32x4 &w00 = rw[S][0][0], &w01 = rw[S][0][1], &w10 = rw[S][1][0], &w11 = rw[S][1][1];
      u32x4 &w20 = rw[S][2][0], &w21 = rw[S][2][1], &w30 = rw[S][3][0], &w31 = rw[S][3][1];
      if constexpr (ABL != 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w00), "+v"(w01), "+v"(w10), "+v"(w11), "+v"(w20), "+v"(w21), "+v"(w30), "+v"(w31)::"memory");
      else asm volatile("s_waitcnt vmcnt(14)" : "+v"(w00), "+v"(w01), "+v"(w10), "+v"(w11), "+v"(w20), "+v"(w21), "+v"(w30), "+v"(w31)::"memory");
    }
    static_assert(kDepth == 3, "the counted waits above are one tile's operations: 6 + 2 NB");
    u32x4 bw[3][NB];  // [plane][column block]
#pragma unroll
    for (int j = 0; j < NB; ++j) split8(rw[S][j][0], rw[S][j][1], bw[0][j], bw[1][j], bw[2][j]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's pieces of tile t are in LDS; every wave is done with tile t - 1 (slot (t - 1) % 3)
    asm volatile("" ::: "memory");
    issue(t + kDepth - 1, std::integral_constant<int, (S + kDepth - 1) % kDepth>{});  // (clamped past the end: see issue)
    const unsigned char* __restrict__ buf = smem + S * kBuf;
    // fragments of row block i + 1 are requested before row block i's MFMAs (two register sets, ping-pong; the scheduling
    // barrier keeps the compiler from sinking the reads to their use, where every block would expose the LDS latency)
    u32x4 af[2][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) af[0][p] = *reinterpret_cast<const u32x4*>(buf + p * kPlane + a_off[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i + 1 < 8) {
#pragma unroll
        for (int p = 0; p < 3; ++p) af[(i + 1) & 1][p] = *reinterpret_cast<const u32x4*>(buf + p * kPlane + a_off[i + 1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // smallest terms first (plane 0 = hi, 1 = mid, 2 = lo): lo hi, hi lo, mid mid, mid hi, hi mid, hi hi
      constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int s = 0; s < 6; ++s)
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i & 1][pa[s]]), __builtin_bit_cast(bf16x8, bw[pb[s]][j]),
                                                              acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  issue(0, S0{});
  issue(1, S1{});
  int t = 0;
  for (; t + 3 <= tiles; t += 3) {
    step(t, S0{});
    step(t + 1, S1{});
    step(t + 2, S2{});
  }
  if (t < tiles) step(t++, S0{});
  if (t < tiles) step(t++, S1{});
  // retire the two filler tiles still in flight before any of their registers or LDS slots can be reused
#pragma unroll
  for (int sl = 0; sl < kDepth; ++sl)
#pragma unroll
    for (int j = 0; j < NB; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(rw[sl][j][0]), "+v"(rw[sl][j][1])::"memory");
  // ---- epilogue: acc[i][j][e] = C[m_base + 16 i + 4 q + e][n_wave + 16 j + r], kept where the position is active
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    unsigned char bit[NB][4];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m_base + 16 * i + 4 * q + e, n = n_wave + 16 * j + r;
        const bool ok = m < M && n < N;
        bit[j][e] = bits[ok ? (size_t)m * N + n : 0];
      }
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m_base + 16 * i + 4 * q + e, n = n_wave + 16 * j + r;
        if (m < M && n < N) out[(size_t)m * N + n] = bit[j][e] ? acc[i][j][e] : 0.0f;
      }
  }
}
}  // namespace

namespace {
int ftv_nb() {  // developer knob: 16-column blocks per wave (tile width 64 * NB): 4 = 128 x 256 tiles, one workgroup per CU
  const char* e = getenv("NNUE_FTM_VAL_NB");
  return (e && atoi(e) == 4) ? 4 : 2;
}
}  // namespace

bool ftv_supported(int B, int F, int P, int L1) {
  // Off by default: measured in the 224x224 step it ties the six-plane tile kernel.  Read per call so
  // that tests can switch it inside one process.
  const char* e = getenv("NNUE_FTM_VAL_DMA");
  const int on = e ? atoi(e) : 0;
  // big maps only: 128-column tiles must fill the chip, K runs in whole tiles of 32, every offset fits 31 bits
  return on && B > 0 && L1 % kKT == 0 && P % 4 == 0 && (long long)((P + 127) / 128) * ((B + kBM - 1) / kBM) >= 384 &&
         (long long)F * L1 * 4 < (1ll << 31) && (long long)B * L1 * 2 < (1ll << 31);
}

int64_t ftv_scratch_bytes(int B, int L1) { return (int64_t)3 * B * L1 * 2; }

void ftv_split_planes(const float* src, int rows, int cols, void* planes, hipStream_t st) {
  const long long count8 = (long long)rows * cols / 8;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((count8 + 255) / 256)), dim3(256), 0, st, src, rows, cols,
                     static_cast<unsigned short*>(planes));
}

int ftv_launch(const uint8_t* bits, const float* d_out, const float* weight, int B, int F, int P, int L1, float* d_conv_out, void* scratch,
               hipStream_t st) {
  unsigned short* planes = static_cast<unsigned short*>(scratch);
  ftv_split_planes(d_out, B, L1, planes, st);
  const int nb = ftv_nb(), bn = 64 * nb;
  const dim3 grid((unsigned)((P + bn - 1) / bn), (unsigned)((B + kBM - 1) / kBM));
  const bool nt = (size_t)F * L1 * 4 > (64u << 20);  // a table larger than the caches is streamed once: non-temporal loads
#ifdef NNUE_ABLATIONS  // timing-only ablations (WRONG results), tools/debug: compiled only with NNUE_BUILD_ABLATIONS=1 (csrc/build.py)
  static const int abl = [] { const char* e = getenv("NNUE_FTM_VAL_ABL"); return e ? atoi(e) : 0; }();
#endif
  static const int prio = [] { const char* e = getenv("NNUE_FTM_VAL_PRIO"); return e ? atoi(e) : 0; }();  // developer knob: see the kernel
#define NNUE_FTV(A, NBV)                                                                                                                      \
  do {                                                                                                                                        \
    if (nt) hipLaunchKernelGGL((ftv_values_kernel<A, true, NBV>), grid, dim3(256), 0, st, planes, weight, bits, d_conv_out, B, P, L1, F, prio); \
    else hipLaunchKernelGGL((ftv_values_kernel<A, false, NBV>), grid, dim3(256), 0, st, planes, weight, bits, d_conv_out, B, P, L1, F, prio);   \
  } while (0)
  if (nb == 4) NNUE_FTV(0, 4);
#ifdef NNUE_ABLATIONS
  else if (abl == 1) NNUE_FTV(1, 2);
  else if (abl == 2) NNUE_FTV(2, 2);
#endif
  else NNUE_FTV(0, 2);
#undef NNUE_FTV
  return NNUE_OK;
}
