// Data parallel for bandwidth-sized tables: the ranks exchange the FACTORS of the table's weight gradient, not the gradient.
//
// d_W = A^T D over the GLOBAL batch is a product of two small factors -- the {0,1} map A [B][P] (one BIT per position on
// the wire) and D = d_ft [B][L1] -- whose product is F x L1 floats (268 MB at the 224x224 configuration).  One all-gather
// of a per-rank chunk
//
//     [ d_ft  B*L1 floats | sink  B floats | small  S floats | map bits  B rows of ceil(P/128)*16 bytes ]
//
// (1.5 MB per rank at that configuration) replaces the 269 MB all-reduce / reduce-scatter of SURVEY 8e; every rank then
// runs the single-rank fused path (nnue_ftm_gram_sqnorm, nnue_sgd_step, nnue_ftm_backward_weight_update) on the global
// factors, so the table's gradient is still never materialised and all replicas apply the bitwise identical update.
// "small" carries every OTHER gradient (threshold, conv weight, table rows the product does not cover, bias, classifier):
// nnue_dp_factor_unpack sums it over the ranks in rank order -- the all-reduce of train.py's single-device step done as a
// deterministic reduction of the gathered pieces, so the step has ONE collective.
//
// Reference semantics reproduced: the gradient of the mean loss over the global batch and clip_grad_norm_ on its global
// norm (train.py:359-366); the reference itself is single-device (train.py:263).
#include "common.h"

namespace {
constexpr int kThreads = 256;

struct ChunkLayout {
  int64_t dft, sink, small, packed, bytes;  // byte offsets inside a chunk, total size
  int row_bytes;                            // packed bytes per map row
};
ChunkLayout chunk_layout(int B, int P, int L1, int64_t small_count) {
  ChunkLayout c{};
  c.row_bytes = (P + 127) / 128 * 16;
  c.dft = 0;
  c.sink = nnue_round_up((int64_t)B * L1 * 4, 16);
  c.small = c.sink + nnue_round_up((int64_t)B * 4, 16);
  c.packed = c.small + nnue_round_up(small_count * 4, 16);
  c.bytes = c.packed + (int64_t)B * c.row_bytes;
  return c;
}

// bits: one thread per packed byte (8 map bytes -> 8 bits; the map holds exactly 0 or 1).  small: float copies.
__global__ __launch_bounds__(kThreads) void factor_pack_kernel(const uint8_t* __restrict__ bits, int B, int P, int row_bytes,
                                                               uint8_t* __restrict__ packed, unsigned bit_blocks,
                                                               const float* __restrict__ grads, int64_t head, int64_t tail_lo, int64_t tail,
                                                               float* __restrict__ small) {
  if (blockIdx.x < bit_blocks) {
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= (int64_t)B * row_bytes) return;
    const int row = (int)(i / row_bytes), byte = (int)(i % row_bytes);
    const int k = byte * 8;
    const uint8_t* src = bits + (size_t)row * P + k;  // P % 4 == 0: both words are aligned and wholly inside or outside the row
    const uint32_t lo = k < P ? *reinterpret_cast<const uint32_t*>(src) : 0u;
    const uint32_t hi = k + 4 < P ? *reinterpret_cast<const uint32_t*>(src + 4) : 0u;
    auto nib = [](uint32_t x) { return (x & 1u) | ((x >> 7) & 2u) | ((x >> 14) & 4u) | ((x >> 21) & 8u); };
    packed[i] = (uint8_t)(nib(lo) | (nib(hi) << 4));
    return;
  }
  const int64_t i = (int64_t)(blockIdx.x - bit_blocks) * kThreads + threadIdx.x;
  if (i < head) small[i] = grads[i];
  else if (i < head + tail) small[i] = grads[tail_lo + (i - head)];
}

// Block families: [map bits -> bytes | d_ft and sink copies | small sums over the ranks]
template <int VEC>  // map bytes written per thread: 16 when P % 16 == 0, else 4
__global__ __launch_bounds__(kThreads) void factor_unpack_kernel(const uint8_t* __restrict__ chunks, ChunkLayout c, int world, int B, int P, int L1,
                                                                 uint8_t* __restrict__ g_bits, float* __restrict__ g_sink,
                                                                 float* __restrict__ g_dft, unsigned bit_blocks, unsigned copy_blocks,
                                                                 float* __restrict__ grads, int64_t head, int64_t tail_lo, int64_t tail) {
  auto spread = [](uint32_t n) { return (n & 1u) | ((n & 2u) << 7) | ((n & 4u) << 14) | ((n & 8u) << 21); };
  if (blockIdx.x < bit_blocks) {
    const int per_row = P / VEC;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= (int64_t)world * B * per_row) return;
    const int64_t grow = i / per_row;  // global row = rank * B + b
    const int v = (int)(i % per_row), rank = (int)(grow / B), b = (int)(grow % B);
    const uint8_t* src = chunks + (size_t)rank * c.bytes + c.packed + (size_t)b * c.row_bytes;
    uint8_t* dst = g_bits + (size_t)grow * P + (size_t)v * VEC;
    if constexpr (VEC == 16) {
      const uint32_t w = *reinterpret_cast<const uint16_t*>(src + v * 2);
      uint4 o;
      o.x = spread(w & 15u); o.y = spread((w >> 4) & 15u); o.z = spread((w >> 8) & 15u); o.w = spread((w >> 12) & 15u);
      *reinterpret_cast<uint4*>(dst) = o;
    } else {
      const uint32_t w = src[v >> 1];
      *reinterpret_cast<uint32_t*>(dst) = spread((v & 1) ? (w >> 4) : (w & 15u));
    }
    return;
  }
  if (blockIdx.x < bit_blocks + copy_blocks) {  // d_ft as float4 (L1 % 4 == 0), then sink
    const int64_t i = (int64_t)(blockIdx.x - bit_blocks) * kThreads + threadIdx.x;
    const int64_t per_rank4 = (int64_t)B * L1 / 4, n4 = per_rank4 * world;
    if (i < n4) {
      const int rank = (int)(i / per_rank4);
      const int64_t j = i % per_rank4;
      reinterpret_cast<float4*>(g_dft)[i] = reinterpret_cast<const float4*>(chunks + (size_t)rank * c.bytes + c.dft)[j];
    } else if (i < n4 + (int64_t)world * B) {
      const int64_t s = i - n4;
      const int rank = (int)(s / B), b = (int)(s % B);
      g_sink[s] = reinterpret_cast<const float*>(chunks + (size_t)rank * c.bytes + c.sink)[b];
    }
    return;
  }
  const int64_t i = (int64_t)(blockIdx.x - bit_blocks - copy_blocks) * kThreads + threadIdx.x;
  if (i >= head + tail) return;
  float acc = 0.0f;
  for (int r = 0; r < world; ++r) acc += reinterpret_cast<const float*>(chunks + (size_t)r * c.bytes + c.small)[i];  // rank order: identical on every rank
  grads[i < head ? i : tail_lo + (i - head)] = acc;
}

bool args_ok(int B, int P, int L1, int64_t head, int64_t tail_lo, int64_t tail) {
  return B > 0 && P > 0 && L1 > 0 && P % 4 == 0 && L1 % 4 == 0 && head >= 0 && tail >= 0 && tail_lo >= head && (int64_t)B * L1 < (1ll << 29) &&
         (int64_t)B * P < (1ll << 31) && head + tail < (1ll << 31);
}
}  // namespace

extern "C" int64_t nnue_dp_factor_chunk_bytes(int B, int P, int L1, int64_t small_count) {
  return (B > 0 && P > 0 && L1 > 0 && small_count >= 0) ? chunk_layout(B, P, L1, small_count).bytes : 0;
}

extern "C" int64_t nnue_dp_factor_offset(int which, int B, int P, int L1, int64_t small_count) {
  if (B <= 0 || P <= 0 || L1 <= 0 || small_count < 0) return -1;
  const ChunkLayout c = chunk_layout(B, P, L1, small_count);
  return which == 0 ? c.dft : which == 1 ? c.sink : which == 2 ? c.small : which == 3 ? c.packed : -1;
}

extern "C" int nnue_dp_factor_pack(const uint8_t* bits, const float* grads, int64_t head_count, int64_t tail_lo, int64_t tail_count, int B,
                                   int P, int L1, void* chunk, nnue_stream_t stream) {
  NNUE_REQUIRE(bits && grads && chunk, NNUE_E_ARG, "nnue_dp_factor_pack: null pointer");
  NNUE_REQUIRE(args_ok(B, P, L1, head_count, tail_lo, tail_count), NNUE_E_ARG,
               "nnue_dp_factor_pack: B=%d P=%d L1=%d head=%lld tail=[%lld,+%lld) out of range (P and L1 multiples of 4)", B, P, L1,
               (long long)head_count, (long long)tail_lo, (long long)tail_count);
  NNUE_REQUIRE(nnue_aligned16(bits) && nnue_aligned16(chunk) && nnue_aligned16(grads), NNUE_E_ARG, "nnue_dp_factor_pack: pointers must be 16-byte aligned");
  const ChunkLayout c = chunk_layout(B, P, L1, head_count + tail_count);
  uint8_t* base = static_cast<uint8_t*>(chunk);
  const unsigned bit_blocks = (unsigned)(((int64_t)B * c.row_bytes + kThreads - 1) / kThreads);
  const unsigned small_blocks = (unsigned)((head_count + tail_count + kThreads - 1) / kThreads);
  hipLaunchKernelGGL(factor_pack_kernel, dim3(bit_blocks + small_blocks), dim3(kThreads), 0, static_cast<hipStream_t>(stream), bits, B, P,
                     c.row_bytes, base + c.packed, bit_blocks, grads, head_count, tail_lo, tail_count, reinterpret_cast<float*>(base + c.small));
  return nnue_launch_status("nnue_dp_factor_pack");
}

extern "C" int nnue_dp_factor_unpack(const void* chunks, int world, int B, int P, int L1, int64_t head_count, int64_t tail_lo,
                                     int64_t tail_count, uint8_t* g_bits, float* g_sink, float* g_dft, float* grads, nnue_stream_t stream) {
  NNUE_REQUIRE(chunks && g_bits && g_sink && g_dft && grads, NNUE_E_ARG, "nnue_dp_factor_unpack: null pointer");
  NNUE_REQUIRE(world >= 1 && world <= 64 && args_ok(B, P, L1, head_count, tail_lo, tail_count) && (int64_t)world * B * P < (1ll << 40) &&
                   (int64_t)world * B < (1 << 24),
               NNUE_E_ARG, "nnue_dp_factor_unpack: world=%d B=%d P=%d L1=%d head=%lld tail=[%lld,+%lld) out of range", world, B, P, L1,
               (long long)head_count, (long long)tail_lo, (long long)tail_count);
  NNUE_REQUIRE(nnue_aligned16(chunks) && nnue_aligned16(g_bits) && nnue_aligned16(g_dft) && nnue_aligned16(grads), NNUE_E_ARG,
               "nnue_dp_factor_unpack: pointers must be 16-byte aligned");
  const ChunkLayout c = chunk_layout(B, P, L1, head_count + tail_count);
  const int vec = P % 16 == 0 ? 16 : 4;
  const unsigned bit_blocks = (unsigned)(((int64_t)world * B * (P / vec) + kThreads - 1) / kThreads);
  const unsigned copy_blocks = (unsigned)(((int64_t)world * B * L1 / 4 + (int64_t)world * B + kThreads - 1) / kThreads);
  const unsigned small_blocks = (unsigned)((head_count + tail_count + kThreads - 1) / kThreads);
  const dim3 grid(bit_blocks + copy_blocks + small_blocks);
  const uint8_t* base = static_cast<const uint8_t*>(chunks);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (vec == 16)
    hipLaunchKernelGGL(factor_unpack_kernel<16>, grid, dim3(kThreads), 0, st, base, c, world, B, P, L1, g_bits, g_sink, g_dft, bit_blocks, copy_blocks,
                       grads, head_count, tail_lo, tail_count);
  else
    hipLaunchKernelGGL(factor_unpack_kernel<4>, grid, dim3(kThreads), 0, st, base, c, world, B, P, L1, g_bits, g_sink, g_dft, bit_blocks, copy_blocks,
                       grads, head_count, tail_lo, tail_count);
  return nnue_launch_status("nnue_dp_factor_unpack");
}
