// FeatureTransformer kernels for gfx950 (MI355X): the sparse gather-accumulate forward,
// the embedding-gradient gather-sum (transposed coefficients, no atomics) and the
// value-gradient gather-dot.  Reference arithmetic: nnue.py:686-710 and its autograd.
//
// Data layout (see include/nnue_hip.h): per-sample act lists (rows, coef, pos, n) with
// fixed capacity, and coefT [F, ldb] = per table row, the coefficient of every sample.
//
// Every kernel has two forms:
//   *_wide   L1 a multiple of 256: one wave owns 256 consecutive columns as one float4
//            per lane, so each gathered row segment is a single coalesced 1 KiB
//            wave-load; list entries / sample ids are wave-uniform (scalar loads,
//            v_readlane), several rows are kept in flight per wave.
//   *_simple any L1: one column per thread.  Used for small / odd widths.
// Roofline: all are memory-bound row gathers; algorithmic bytes are in DESIGN.md.
#include "common.h"

namespace {

__device__ __forceinline__ float lane_read(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ void fma4(float4& acc, float c, const float4& v) {
  acc.x = fmaf(c, v.x, acc.x);
  acc.y = fmaf(c, v.y, acc.y);
  acc.z = fmaf(c, v.z, acc.z);
  acc.w = fmaf(c, v.w, acc.w);
}

// ------------------------------------------------------------------ prepare
// One workgroup per sample: order-preserving compaction of the valid (idx >= 0) entries,
// clamp to [0, F-1], and accumulation of the values into coefT (float atomics; only
// repeated ids of one sample ever collide, and a + b is commutative, so results are
// bitwise reproducible unless one sample repeats an id three or more times).
__global__ __launch_bounds__(256) void ft_prepare_kernel(const int64_t* __restrict__ idx,
                                                         const float* __restrict__ val, int M, int F,
                                                         int* __restrict__ rows, int* __restrict__ pos,
                                                         float* __restrict__ coef, int* __restrict__ n,
                                                         float* __restrict__ coefT, int ldb) {
  __shared__ int wave_count[4];
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const size_t base_in = (size_t)b * M;
  int written = 0;
  for (int i0 = 0; i0 < M; i0 += 256) {
    const int i = i0 + tid;
    const int64_t id = (i < M) ? idx[base_in + i] : -1;
    const bool valid = id >= 0;
    const unsigned long long m = __ballot(valid);
    if (lane == 0) wave_count[wave] = __popcll(m);
    __syncthreads();
    int offset = written;
    int total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = wave_count[w];
      if (w < wave) offset += c;
      total += c;
    }
    if (valid) {
      const int k = offset + __popcll(m & ((1ull << lane) - 1ull));
      const int r = id > (int64_t)(F - 1) ? F - 1 : (int)id;
      const float v = val[base_in + i];
      rows[base_in + k] = r;
      pos[base_in + k] = i;
      coef[base_in + k] = v;
      atomicAdd(&coefT[(size_t)r * ldb + b], v);
    }
    written += total;
    __syncthreads();
  }
  if (tid == 0) n[b] = written;
}

// ------------------------------------------------------------------ forward
// grid (B, ceil(S/4)), S = L1/256; wave -> (sample, 256-column slice).
template <int UNROLL>
__global__ __launch_bounds__(256) void ft_forward_wide(const float* __restrict__ W,
                                                       const float* __restrict__ bias,
                                                       const int* __restrict__ rows,
                                                       const float* __restrict__ coef,
                                                       const int* __restrict__ n, int cap, int L1,
                                                       float* __restrict__ out) {
  const int b = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int col = (blockIdx.y * 4 + wave) * 256 + lane * 4;
  if (col >= L1) return;  // wave-uniform: L1 % 256 == 0
  const int cnt = n[b];
  const int* __restrict__ r = rows + (size_t)b * cap;
  const float* __restrict__ c = coef + (size_t)b * cap;
  const float* __restrict__ Wc = W + col;
  float4 acc = *reinterpret_cast<const float4*>(bias + col);
  int k = 0;
  for (; k + UNROLL <= cnt; k += UNROLL) {
    float4 v[UNROLL];
    float cc[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int rr = r[k + u];
      cc[u] = c[k + u];
      v[u] = *reinterpret_cast<const float4*>(Wc + (size_t)rr * L1);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) fma4(acc, cc[u], v[u]);
  }
  for (; k < cnt; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(Wc + (size_t)r[k] * L1);
    fma4(acc, c[k], v);
  }
  *reinterpret_cast<float4*>(out + (size_t)b * L1 + col) = acc;
}

__global__ __launch_bounds__(256) void ft_forward_simple(const float* __restrict__ W,
                                                         const float* __restrict__ bias,
                                                         const int* __restrict__ rows,
                                                         const float* __restrict__ coef,
                                                         const int* __restrict__ n, int cap, int L1,
                                                         float* __restrict__ out) {
  const int b = blockIdx.x;
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (col >= L1) return;
  const int cnt = n[b];
  const int* __restrict__ r = rows + (size_t)b * cap;
  const float* __restrict__ c = coef + (size_t)b * cap;
  float acc = bias[col];
  for (int k = 0; k < cnt; ++k) acc = fmaf(c[k], W[(size_t)r[k] * L1 + col], acc);
  out[(size_t)b * L1 + col] = acc;
}

// Narrow tables (L1 <= 256, e.g. the reference's default 64-wide config): 1024 threads = (1024 / Lp) list
// slices x Lp columns (Lp = L1 rounded up to a power of two); each slice walks every S-th entry, the
// slices are combined through LDS in slice order (deterministic).
__global__ __launch_bounds__(1024) void ft_forward_narrow(const float* __restrict__ W,
                                                          const float* __restrict__ bias,
                                                          const int* __restrict__ rows,
                                                          const float* __restrict__ coef,
                                                          const int* __restrict__ n, int cap, int L1, int Lp,
                                                          float* __restrict__ out) {
  extern __shared__ float red[];  // [slices][Lp]
  const int b = blockIdx.x;
  const int col = threadIdx.x % Lp, slice = threadIdx.x / Lp, S = 1024 / Lp;
  const int cnt = n[b];
  const int* __restrict__ r = rows + (size_t)b * cap;
  const float* __restrict__ c = coef + (size_t)b * cap;
  float acc = 0.f;
  if (col < L1) {
#pragma unroll 4
    for (int k = slice; k < cnt; k += S) acc = fmaf(c[k], W[(size_t)r[k] * L1 + col], acc);
  }
  red[slice * Lp + col] = acc;
  __syncthreads();
  if (slice == 0 && col < L1) {
    float tot = bias[col];
    for (int s2 = 0; s2 < S; ++s2) tot += red[s2 * Lp + col];
    out[(size_t)b * L1 + col] = tot;
  }
}

// ------------------------------------------------------------------ backward: weight / bias
// grid (F + 1, ceil(S/4)); block row f < F is table row f, f == F is the bias (coefficient 1
// for every sample).  A wave scans the row's coefficients 64 samples at a time, ballots the
// non-zeros and adds the matching d_out rows in ascending sample order.
__global__ __launch_bounds__(256) void ft_backward_weight_wide(const float* __restrict__ d_out,
                                                               const float* __restrict__ coefT, int ldb,
                                                               int B, int F, int L1,
                                                               float* __restrict__ d_weight,
                                                               float* __restrict__ d_bias) {
  const int f = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int col = (blockIdx.y * 4 + wave) * 256 + lane * 4;
  if (col >= L1) return;
  const bool is_bias = (f == F);
  if ((is_bias ? d_bias : d_weight) == nullptr) return;  // output not requested; uniform per block
  float* __restrict__ dst = is_bias ? d_bias : d_weight + (size_t)f * L1;
  const float* __restrict__ crow = coefT + (size_t)(is_bias ? 0 : f) * ldb;
  const float* __restrict__ g = d_out + col;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int bb = b0 + lane;
    const float cv = (bb < B) ? (is_bias ? 1.0f : crow[bb]) : 0.0f;
    unsigned long long m = __ballot(cv != 0.0f);
    while (__popcll(m) >= 4) {
      int j[4];
      float c[4];
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        j[u] = __builtin_ctzll(m);
        m &= m - 1;
        c[u] = lane_read(cv, j[u]);
        v[u] = *reinterpret_cast<const float4*>(g + (size_t)(b0 + j[u]) * L1);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) fma4(acc, c[u], v[u]);
    }
    while (m) {
      const int j = __builtin_ctzll(m);
      m &= m - 1;
      const float4 v = *reinterpret_cast<const float4*>(g + (size_t)(b0 + j) * L1);
      fma4(acc, lane_read(cv, j), v);
    }
  }
  *reinterpret_cast<float4*>(dst + col) = acc;
}

__global__ __launch_bounds__(256) void ft_backward_weight_simple(const float* __restrict__ d_out,
                                                                 const float* __restrict__ coefT, int ldb,
                                                                 int B, int F, int L1,
                                                                 float* __restrict__ d_weight,
                                                                 float* __restrict__ d_bias) {
  const int f = blockIdx.x;
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (col >= L1) return;
  const bool is_bias = (f == F);
  if ((is_bias ? d_bias : d_weight) == nullptr) return;
  float* __restrict__ dst = is_bias ? d_bias : d_weight + (size_t)f * L1;
  const float* __restrict__ crow = coefT + (size_t)(is_bias ? 0 : f) * ldb;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {
    const float c = is_bias ? 1.0f : crow[b];
    if (c != 0.0f) acc = fmaf(c, d_out[(size_t)b * L1 + col], acc);
  }
  dst[col] = acc;
}

// ------------------------------------------------------------------ backward: values
// One workgroup (4 waves) per sample; every wave keeps the whole d_out row in registers
// (S float4 per lane) and takes groups of 16 list entries.  For each entry it reads the
// table row as S coalesced 1 KiB loads and forms a per-lane partial dot; the 16 partials are
// then transposed-and-summed across the wave with a 4-stage butterfly (15 + 2 shuffles per 16
// entries instead of 6 per entry), leaving entry (lane & 15)'s dot product in every lane.
template <int S>
__global__ __launch_bounds__(256) void ft_backward_values_wide(const float* __restrict__ d_out,
                                                               const float* __restrict__ W,
                                                               const int* __restrict__ rows,
                                                               const int* __restrict__ pos,
                                                               const int* __restrict__ n, int cap,
                                                               float* __restrict__ dst, int dst_ld) {
  constexpr int L1 = 256 * S;
  const int b = blockIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int cnt = n[b];
  const int* __restrict__ r = rows + (size_t)b * cap;
  float4 g[S];
#pragma unroll
  for (int s = 0; s < S; ++s)
    g[s] = *reinterpret_cast<const float4*>(d_out + (size_t)b * L1 + s * 256 + lane * 4);
  const float* __restrict__ Wl = W + lane * 4;

  // zero this sample's destination row, then scatter the dot products over it (ordered by the barrier:
  // both writes come from this workgroup)
  for (int i = threadIdx.x; i < dst_ld; i += 256) dst[(size_t)b * dst_ld + i] = 0.0f;
  __syncthreads();

  for (int k0 = wave * 16; k0 < cnt; k0 += 64) {
    float p[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      float acc = 0.f;
      if (k0 + u < cnt) {  // wave-uniform
        const float* __restrict__ wr = Wl + (size_t)r[k0 + u] * L1;
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float4 v = *reinterpret_cast<const float4*>(wr + s * 256);
          acc = fmaf(v.x, g[s].x, acc);
          acc = fmaf(v.y, g[s].y, acc);
          acc = fmaf(v.z, g[s].z, acc);
          acc = fmaf(v.w, g[s].w, acc);
        }
      }
      p[u] = acc;
    }
    // butterfly transpose-reduce over lane bits 3..0
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const bool hi = lane & 8;
      const float keep = hi ? p[u + 8] : p[u];
      const float send = hi ? p[u] : p[u + 8];
      p[u] = keep + __shfl_xor(send, 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool hi = lane & 4;
      const float keep = hi ? p[u + 4] : p[u];
      const float send = hi ? p[u] : p[u + 4];
      p[u] = keep + __shfl_xor(send, 4);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool hi = lane & 2;
      const float keep = hi ? p[u + 2] : p[u];
      const float send = hi ? p[u] : p[u + 2];
      p[u] = keep + __shfl_xor(send, 2);
    }
    {
      const bool hi = lane & 1;
      const float keep = hi ? p[1] : p[0];
      const float send = hi ? p[0] : p[1];
      p[0] = keep + __shfl_xor(send, 1);
    }
    float tot = p[0];
    tot += __shfl_xor(tot, 16);
    tot += __shfl_xor(tot, 32);
    const int k = k0 + lane;
    if (lane < 16 && k < cnt) dst[(size_t)b * dst_ld + pos[(size_t)b * cap + k]] = tot;
  }
}

// any L1: wave per entry, lanes stride over the columns, 6-step wave reduction.
__global__ __launch_bounds__(256) void ft_backward_values_simple(const float* __restrict__ d_out,
                                                                 const float* __restrict__ W,
                                                                 const int* __restrict__ rows,
                                                                 const int* __restrict__ pos,
                                                                 const int* __restrict__ n, int cap, int L1,
                                                                 float* __restrict__ dst, int dst_ld) {
  const int b = blockIdx.x;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int cnt = n[b];
  const float* __restrict__ g = d_out + (size_t)b * L1;
  for (int k = blockIdx.y * 4 + wave; k < cnt; k += 4 * gridDim.y) {
    const float* __restrict__ wr = W + (size_t)rows[(size_t)b * cap + k] * L1;
    float acc = 0.f;
    for (int c = lane; c < L1; c += 64) acc = fmaf(wr[c], g[c], acc);
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s);
    if (lane == 0) dst[(size_t)b * dst_ld + pos[(size_t)b * cap + k]] = acc;
  }
}

// ------------------------------------------------------------------ act list -> reference format
__global__ __launch_bounds__(256) void act_to_padded_kernel(const int* __restrict__ pos,
                                                            const float* __restrict__ coef,
                                                            const int* __restrict__ n, int cap, int M,
                                                            int64_t* __restrict__ idx, float* __restrict__ val) {
  const int b = blockIdx.x;
  const int cnt = n[b];
  for (int k = blockIdx.y * 256 + threadIdx.x; k < M; k += 256 * gridDim.y) {
    const bool on = k < cnt && k < cap;
    idx[(size_t)b * M + k] = on ? (int64_t)pos[(size_t)b * cap + k] : (int64_t)-1;
    val[(size_t)b * M + k] = on ? coef[(size_t)b * cap + k] : 0.0f;
  }
}

bool wide_ok(int L1) { return L1 % 256 == 0; }

}  // namespace

// =============================================================================== C ABI
extern "C" int nnue_ft_prepare(const int64_t* idx, const float* val, int B, int M, int F,
                               int32_t* rows, int32_t* pos, float* coef, int32_t* n,
                               float* coefT, int ldb, nnue_stream_t stream) {
  NNUE_REQUIRE(idx && val && rows && pos && coef && n && coefT, NNUE_E_ARG, "nnue_ft_prepare: null pointer");
  NNUE_REQUIRE(B > 0 && M > 0 && F > 0, NNUE_E_ARG, "nnue_ft_prepare: B=%d M=%d F=%d must be positive", B, M, F);
  NNUE_REQUIRE(ldb >= B && ldb % 64 == 0, NNUE_E_SHAPE, "nnue_ft_prepare: ldb=%d must be a multiple of 64 and >= B=%d", ldb, B);
  hipStream_t s = static_cast<hipStream_t>(stream);
  nnue_zero_floats(coefT, (size_t)F * ldb, s);  // a kernel, not a memset node (common.h)
  hipLaunchKernelGGL(ft_prepare_kernel, dim3(B), dim3(256), 0, s, idx, val, M, F, rows, pos, coef, n, coefT, ldb);
  return nnue_launch_status("nnue_ft_prepare");
}

extern "C" int nnue_ft_forward(const float* weight, const float* bias, const int32_t* rows,
                               const float* coef, const int32_t* n, int cap, int B, int F, int L1,
                               float* out, nnue_stream_t stream) {
  NNUE_REQUIRE(weight && bias && rows && coef && n && out, NNUE_E_ARG, "nnue_ft_forward: null pointer");
  NNUE_REQUIRE(B > 0 && F > 0 && L1 > 0 && cap > 0, NNUE_E_ARG, "nnue_ft_forward: B=%d F=%d L1=%d cap=%d must be positive", B, F, L1, cap);
  NNUE_REQUIRE((int64_t)F * L1 < (1ll << 40), NNUE_E_SHAPE, "nnue_ft_forward: table too large");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (wide_ok(L1)) {
    NNUE_REQUIRE(nnue_aligned16(weight) && nnue_aligned16(bias) && nnue_aligned16(out), NNUE_E_ARG,
                 "nnue_ft_forward: pointers must be 16-byte aligned");
    const int S = L1 / 256;
    hipLaunchKernelGGL(ft_forward_wide<8>, dim3(B, (S + 3) / 4), dim3(64 * (S < 4 ? S : 4)), 0, s, weight, bias,
                       rows, coef, n, cap, L1, out);
  } else if (L1 <= 256) {
    int Lp = 1;
    while (Lp < L1) Lp <<= 1;
    hipLaunchKernelGGL(ft_forward_narrow, dim3(B), dim3(1024), 1024 * sizeof(float), s, weight, bias, rows, coef, n, cap, L1, Lp,
                       out);
  } else {
    hipLaunchKernelGGL(ft_forward_simple, dim3(B, (L1 + 255) / 256), dim3(256), 0, s, weight, bias, rows, coef, n,
                       cap, L1, out);
  }
  return nnue_launch_status("nnue_ft_forward");
}

extern "C" int nnue_ft_backward_weight(const float* d_out, const float* coefT, int ldb, int B, int F, int L1,
                                       float* d_weight, float* d_bias, nnue_stream_t stream) {
  NNUE_REQUIRE(d_out && coefT, NNUE_E_ARG, "nnue_ft_backward_weight: null pointer");
  NNUE_REQUIRE(d_weight || d_bias, NNUE_E_ARG, "nnue_ft_backward_weight: both outputs are null");
  NNUE_REQUIRE(B > 0 && F > 0 && L1 > 0, NNUE_E_ARG, "nnue_ft_backward_weight: B=%d F=%d L1=%d must be positive", B, F, L1);
  NNUE_REQUIRE(ldb >= B && ldb % 64 == 0, NNUE_E_SHAPE, "nnue_ft_backward_weight: ldb=%d must be a multiple of 64 and >= B=%d", ldb, B);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (wide_ok(L1)) {
    NNUE_REQUIRE(nnue_aligned16(d_out) && nnue_aligned16(d_weight) && nnue_aligned16(d_bias), NNUE_E_ARG,
                 "nnue_ft_backward_weight: pointers must be 16-byte aligned");
    const int S = L1 / 256;
    hipLaunchKernelGGL(ft_backward_weight_wide, dim3(F + 1, (S + 3) / 4), dim3(64 * (S < 4 ? S : 4)), 0, s, d_out,
                       coefT, ldb, B, F, L1, d_weight, d_bias);
  } else {
    hipLaunchKernelGGL(ft_backward_weight_simple, dim3(F + 1, (L1 + 255) / 256), dim3(256), 0, s, d_out, coefT, ldb,
                       B, F, L1, d_weight, d_bias);
  }
  return nnue_launch_status("nnue_ft_backward_weight");
}

extern "C" int nnue_ft_backward_values(const float* d_out, const float* weight, const int32_t* rows,
                                       const int32_t* pos, const int32_t* n, int cap, int B, int F, int L1,
                                       float* dst, int dst_ld, nnue_stream_t stream) {
  NNUE_REQUIRE(d_out && weight && rows && pos && n && dst, NNUE_E_ARG, "nnue_ft_backward_values: null pointer");
  NNUE_REQUIRE(B > 0 && F > 0 && L1 > 0 && cap > 0 && dst_ld > 0, NNUE_E_ARG,
               "nnue_ft_backward_values: B=%d F=%d L1=%d cap=%d dst_ld=%d must be positive", B, F, L1, cap, dst_ld);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int S = wide_ok(L1) ? L1 / 256 : 0;
  const bool wide = (S == 1 || S == 2 || S == 4 || S == 8);
  if (!wide) nnue_zero_floats(dst, (size_t)B * dst_ld, s);  // a kernel, not a memset node (common.h)
  if (wide) {  // the wide kernel zero-fills its own rows
    NNUE_REQUIRE(nnue_aligned16(d_out) && nnue_aligned16(weight), NNUE_E_ARG,
                 "nnue_ft_backward_values: pointers must be 16-byte aligned");
    const dim3 grid(B), block(256);
    switch (S) {
      case 1: hipLaunchKernelGGL(ft_backward_values_wide<1>, grid, block, 0, s, d_out, weight, rows, pos, n, cap, dst, dst_ld); break;
      case 2: hipLaunchKernelGGL(ft_backward_values_wide<2>, grid, block, 0, s, d_out, weight, rows, pos, n, cap, dst, dst_ld); break;
      case 4: hipLaunchKernelGGL(ft_backward_values_wide<4>, grid, block, 0, s, d_out, weight, rows, pos, n, cap, dst, dst_ld); break;
      default: hipLaunchKernelGGL(ft_backward_values_wide<8>, grid, block, 0, s, d_out, weight, rows, pos, n, cap, dst, dst_ld); break;
    }
  } else {
    int gy = (cap + 63) / 64;
    if (gy > 64) gy = 64;
    hipLaunchKernelGGL(ft_backward_values_simple, dim3(B, gy), dim3(256), 0, s, d_out, weight, rows, pos, n, cap, L1,
                       dst, dst_ld);
  }
  return nnue_launch_status("nnue_ft_backward_values");
}

extern "C" int nnue_act_to_padded(const int32_t* pos, const float* coef, const int32_t* n, int cap, int B, int M,
                                  int64_t* idx, float* val, nnue_stream_t stream) {
  NNUE_REQUIRE(pos && coef && n && idx && val, NNUE_E_ARG, "nnue_act_to_padded: null pointer");
  NNUE_REQUIRE(B > 0 && M > 0 && cap > 0, NNUE_E_ARG, "nnue_act_to_padded: B=%d M=%d cap=%d must be positive", B, M, cap);
  int gy = (M + 255) / 256;
  if (gy > 64) gy = 64;
  hipLaunchKernelGGL(act_to_padded_kernel, dim3(B, gy), dim3(256), 0, static_cast<hipStream_t>(stream), pos, coef, n,
                     cap, M, idx, val);
  return nnue_launch_status("nnue_act_to_padded");
}
