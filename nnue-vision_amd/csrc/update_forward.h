// The big table's SGD update of step t and the FeatureTransformer forward of step t+1 in ONE pass over the table
// (include inside ftm_kernels.hip's anonymous namespace, after gemm_tile_bf64 and its helpers).
//
// At the 224x224 shape the table (65 536 x 1024 f32, 268 MB) is streamed three times per step: by the forward, by the value
// gradient and -- read and written, together with its momentum -- by the update in the weight-gradient product's epilogue
// (BwwSgdEpi).  The update of step t leaves every new table tile in registers; the forward of step t+1,
// out'[b][n] = sum_p A'[b][p] W_new[p][n], contracts exactly those tiles.  Inside a step group (NnueTrainer.step_many: the next
// batch is already resident and its map A' only needs the conv weights, which the small tensors' update has just written)
// the two are therefore one kernel and the forward's 268 MB read disappears.
//
// Shape of the pass: the forward of a small batch is a split-K product -- slab s of column tile j contracts table rows
// [s klen, (s+1) klen) -- so workgroup (j, s) walks the 128-row table tiles of its slab in ascending order and, per tile,
//   1. forms d_W[128 p][64 n] = A^T d_out on the bf16 unit (gemm_tile_bf64's staging and MFMA order, K = B),
//   2. passes the accumulators through LDS so that a thread owns 8 consecutive table rows x 4 columns, applies
//      clip + weight decay + momentum + step exactly as rmw_tile does (parameters and momentum of the tile were requested
//      one phase earlier), stores momentum and parameters,
//   3. splits its 8 x 4 block of NEW parameters into the three bf16 planes (one 16-byte chunk of 8 k per column and
//      plane: the operand layout of the forward), stages the next step's map tile and accumulates
//      out'[128 b][64 n] += A'[:, tile] W_new[tile, :] into a second set of accumulators that lives across the tiles.
// At the end the forward accumulators are the split-K slab ftm_finish_kernel expects.  Every MFMA sees the operands, the
// accumulator and the order the two separate kernels (ftm_gemm_bf64_kernel<false, BwwSgdEpi>, <true, FwdEpi>) give it, so
// table, momentum and out' are BITWISE what nnue_ftm_backward_weight_update followed by nnue_ftm_forward produce
// (tests/test_gpu_update_forward.py).
struct UpdFwd {
  const uint8_t* __restrict__ bits;       // [B][P] map of the step being applied
  const uint8_t* __restrict__ bits_next;  // [B][P] map of the next step (under the already updated conv weights)
  const float* __restrict__ d_out;        // [B][L1]
  float* __restrict__ weight;             // table rows [0, direct), updated in place
  float* __restrict__ momentum;           // matching momentum rows or NULL
  const float* __restrict__ coef;         // clip coefficient (device scalar)
  const float* __restrict__ lr_dev;       // NULL or the learning rate as a device scalar
  float* __restrict__ slabs;              // [ksplit][B][L1] split-K slabs of the next forward
  int B, Bn, P, L1, direct, klen, tiles_n, xcd_remap;  // B rows of the factors (bits, d_out), Bn rows of the next map / out'
  float lr, mom, wd, scale;
  int abl;  // timing-only ablations (NNUE_ABLATIONS builds): 1 no parameter/momentum loads, 2 no stores, 4 no forward phase, 8 no d_W MFMAs
};

constexpr int kUfLds = 64 * 1024;
// Images (byte offsets in the workgroup's LDS); the phases of a tile alias each other:
//   phase 1: As [128 p][64 k] 16 KB at 0, Bs 3 x [64 n][64 k] 24 KB at 16 KB         (gemm_tile_bf64's images)
//   phase 2: Ct [128][68] f32 34 816 B at 0
//   phase 3: Wp 3 x [64 n][128 k] 48 KB at 0, An [128 b][64 k] 16 KB at 48 KB
constexpr int kUfBs = 16384, kUfAn = 49152, kUfWpPlane = 16384;

// W_new planes: 256-byte rows (128 k), 16-byte chunk index swizzled by (row ^ row >> 2): the sixteen lanes of a ds_write_b128
// group hold rows 4 c + e (c = 0..15) and the sixteen lanes of a fragment ds_read_b128 hold sixteen consecutive rows -- both
// land on sixteen different slots
__device__ __forceinline__ int uf_wp_img(int row, int chunk) { return row * 256 + ((chunk ^ ((row ^ (row >> 2)) & 15)) << 4); }

// kNK: K tiles (of 64 rows of the gradient's factors) of the weight-gradient product -- 2 at the 224x224 batch of 128; 4, 8, 16 when
// the factors are the all-gathered ones of 2, 4, 8 ranks (data parallel by factor exchange: the update contracts the GLOBAL batch,
// the next forward this rank's own next map); kMom: a momentum buffer; kFirst: first step (momentum is written, not read).
// (d_out pre-split once per launch into LDS-ready plane images instead of in every workgroup and tile -- 152 of the ~1100 VALU
// instructions per tile -- was built and measured: the kernel 222-225 vs 225-229 us, which its 2-3 us pre-pass gives back; removed,
// profiles/r03s_uf_img_ab.txt.)  Compile-time, and every load of the loop unconditional (a request past
// the workgroup's last tile is moved out of its buffer window: zeros, no memory traffic), because a runtime branch around a
// memory instruction makes the compiler's wait counts the minimum over both paths: with such branches the waits for the map
// tiles also drained the parameter and momentum loads that had been requested after them, i.e. the HBM latency was exposed
// once per tile.
template <int kNK, bool kMom, bool kFirst>
__global__ __launch_bounds__(256, 2) void ftm_update_forward_kernel(UpdFwd a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[kUfLds];
  constexpr int KT = kBf64K, PB = 64 * KT * 2;
  constexpr int kOut = 0x7ff00000;  // a byte offset outside every buffer window
  unsigned char* __restrict__ As = smem;
  unsigned char* __restrict__ Bs = smem + kUfBs;
  float* __restrict__ Ct = reinterpret_cast<float*>(smem);
  unsigned char* __restrict__ Wp = smem;
  unsigned char* __restrict__ An = smem + kUfAn;
  const int tid0 = threadIdx.x, lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int r = lane & 15, q = lane >> 4;
  int j, s;
  {
    const int bid = blockIdx.x;
    if (a.xcd_remap) {  // workgroups of one XCD (equal blockIdx % 8) take all column tiles of the same slabs: they share the map tiles
      const int xcd = bid & 7, l = bid >> 3;
      j = l % a.tiles_n;
      s = (l / a.tiles_n) * 8 + xcd;
    } else {
      j = bid % a.tiles_n;
      s = bid / a.tiles_n;
    }
  }
  const int n_base = j * 64, p_lo = s * a.klen;
  const int p_hi = p_lo + a.klen < a.direct ? p_lo + a.klen : a.direct;
  const int B = a.B, Bn = a.Bn, P = a.P, L1 = a.L1;
  const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.bits), 0, (unsigned)((size_t)B * P), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.bits_next), 0, (unsigned)((size_t)Bn * P), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d_out), 0, (unsigned)((size_t)B * L1 * 4), 0x00020000);
  // parameters / momentum through buffer descriptors whose window ends at table row `direct`: rows past it read as zero and
  // their stores are dropped by the range check (no branches); non-temporal: touched once per step
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(a.weight, 0, (unsigned)((size_t)a.direct * L1 * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsm = __builtin_amdgcn_make_buffer_rsrc(kMom ? a.momentum : a.weight, 0, kMom ? (unsigned)((size_t)a.direct * L1 * 4) : 0u,
                                                                       0x00020000);
  const int m0 = (wave >> 1) * 64, n0 = (wave & 1) * 32;
  // Per-thread coordinates behind global offsets.  They are re-derived from an opaque copy of the thread index at the top of
  // every tile: loop-invariant offsets (about forty of them) would otherwise be held -- and spilled -- across the loop beside the
  // 64 registers of parameters and momentum in flight.
  int tid, am4, ak8, g8, c4, bn4, bk4;
  auto coords = [&]() {
    tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lz = tid & 63;
    // phase 1 staging (gemm_tile_bf64, weight-gradient form): map blocks of 8 k x 4 m; d_out blocks of 4 k x 4 n
    const int grp = tid >> 4, l = tid & 15, gm = grp % 8, gk = grp / 8;
    const int blk = ((gk * 4 + (l >> 2)) * 32) + gm * 4 + (l & 3);
    am4 = (blk % 32) * 4;
    ak8 = (blk / 32) * 8;
    bn4 = (((lz & 3) | ((lz >> 4) << 2))) * 4;
    bk4 = (((lz >> 2) & 3) | (wave << 2)) * 4;
    // phase 2: thread = table rows 8 g8 .. 8 g8 + 7, columns 4 c4 .. 4 c4 + 3 of the tile
    g8 = tid >> 4;
    c4 = tid & 15;
  };
  coords();
  unsigned rat[8];
  u32x4 rb[4];
  auto fetch1 = [&](int m_base, int k0, bool live) {
    int base = live ? (k0 + ak8) * P + m_base + am4 : kOut;
    asm volatile("" : "+v"(base));  // (keeps the eight offsets as base + i P: one multiply, not eight)
#pragma unroll
    for (int i = 0; i < 8; ++i) rat[i] = __builtin_amdgcn_raw_buffer_load_b32(rsa, base + i * P, 0, 0);
    int bb = live ? ((k0 + bk4) * L1 + n_base + bn4) * 4 : kOut;
    asm volatile("" : "+v"(bb));
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsb, bb + i * (L1 * 4), 0, 0);
  };
  using u32x2 = __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned;
  auto stage1 = [&]() {
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // byte e of the eight words = 8 consecutive k of table row am4 + e
      u32x4 v;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const unsigned sel = 0x0c000c00u | (unsigned)e | ((unsigned)(4 + e) << 16);
        v[t] = __builtin_amdgcn_perm(rat[2 * t + 1], rat[2 * t], sel) * 0x3f80u;
      }
      *reinterpret_cast<u32x4*>(As + bf64_img(am4 + e, ak8 >> 3)) = v;
    }
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
  // ---- phase 3: the next step's map tile, forward form (bytes contiguous along k = p): two 16-byte groups per thread and half
  auto an_row_of = [](int g) { const int x = g >> 2; return (x & ~3) | ((x & 1) << 1) | ((x >> 1) & 1); };
  u32x4 ran[2];  // one half (64 k) at a time
  auto fetch3 = [&](int m_base, int h) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = tid0 + 256 * i;
      ran[i] = __builtin_amdgcn_raw_buffer_load_b128(rsn, an_row_of(g) * P + m_base + 64 * h + (g & 3) * 16, 0, 0);
    }
  };
  auto stage3 = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = tid0 + 256 * i, row = an_row_of(g), c = (g & 3) * 2;
      u32x4 lo, hi;
      unsigned x, y;
      bytes_to_bf16(ran[i][0], x, y); lo[0] = x; lo[1] = y;
      bytes_to_bf16(ran[i][1], x, y); lo[2] = x; lo[3] = y;
      bytes_to_bf16(ran[i][2], x, y); hi[0] = x; hi[1] = y;
      bytes_to_bf16(ran[i][3], x, y); hi[2] = x; hi[3] = y;
      *reinterpret_cast<u32x4*>(An + bf64_img(row, c)) = lo;
      *reinterpret_cast<u32x4*>(An + bf64_img(row, c + 1)) = hi;
    }
  };
  f32x4 accw[4][2], accf[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) accf[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto contract1 = [&]() {
#pragma unroll
    for (int kb = 0; kb < KT / 32; ++kb) {
      const int c = kb * 4 + q;
      bf16x8 av[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const bf16x8*>(As + bf64_img(m0 + 16 * i + r, c));
#pragma unroll
      for (int pnum = 2; pnum >= 0; --pnum) {  // smallest plane first; a plane's fragments are read just before its MFMAs
        bf16x8 bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bv[t] = *reinterpret_cast<const bf16x8*>(Bs + pnum * PB + bf64_img(n0 + 16 * t + r, c));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < 2; ++t) accw[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[t], accw[i][t], 0, 0, 0);
      }
    }
  };
  auto contract3 = [&](int h) {
#pragma unroll
    for (int kb = 0; kb < KT / 32; ++kb) {
      const int c = kb * 4 + q;
      bf16x8 av[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = *reinterpret_cast<const bf16x8*>(An + bf64_img(m0 + 16 * i + r, c));
#pragma unroll
      for (int pnum = 2; pnum >= 0; --pnum) {
        bf16x8 bv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) bv[t] = *reinterpret_cast<const bf16x8*>(Wp + pnum * kUfWpPlane + uf_wp_img(n0 + 16 * t + r, 8 * h + c));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < 2; ++t) accf[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[t], accf[i][t], 0, 0, 0);
      }
    }
  };
  float4 w[8], mo[8];
  int wm0 = 0;  // byte offset of the thread's first parameter of the tile (one multiply per tile)
  auto wm_off = [&](int u) { return wm0 + u * (L1 * 4); };
  auto fetch2 = [&](int m_base, bool live) {
    wm0 = live ? ((m_base + 8 * g8) * L1 + n_base + 4 * c4) * 4 : kOut;
    asm volatile("" : "+v"(wm0));
#ifdef NNUE_ABLATIONS
    if (a.abl & 1) {
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = mo[u] = make_float4(0.5f, 0.25f, 0.125f, 1.f);
      return;
    }
#endif
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsw, wm_off(u), 0, 2);
      w[u] = make_float4(__uint_as_float(x[0]), __uint_as_float(x[1]), __uint_as_float(x[2]), __uint_as_float(x[3]));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if constexpr (kMom && !kFirst) {
        const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rsm, wm_off(u), 0, 2);
        mo[u] = make_float4(__uint_as_float(x[0]), __uint_as_float(x[1]), __uint_as_float(x[2]), __uint_as_float(x[3]));
      } else {
        mo[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  const float gs = a.coef[0] * a.scale;
  const float lr = a.lr_dev ? a.lr_dev[0] : a.lr;

  fetch1(p_lo, 0, p_lo < p_hi);
  fetch2(p_lo, p_lo < p_hi);
  for (int m_base = p_lo; m_base < p_hi; m_base += 128) {
    coords();
    // ---------------- phase 1: d_W tile.  (The K tiles are written out, not looped: the wait for the first tile's operands
    // must leave the parameter / momentum loads, requested after them, in flight.)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) accw[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    stage1();
    __syncthreads();
    if constexpr (kNK == 1) fetch3(m_base, 0);
#pragma unroll
    for (int t = 1; t < kNK; ++t) {
      fetch1(m_base, t * KT, true);
      if (t == 1) fetch3(m_base, 0);
#ifdef NNUE_ABLATIONS
      if (!(a.abl & 8))
#endif
      contract1();
      __syncthreads();
      stage1();
      __syncthreads();
    }
#ifdef NNUE_ABLATIONS
    if (!(a.abl & 8))
#endif
    contract1();
    __syncthreads();
    // ---------------- phase 2: accumulators -> LDS -> 8 rows x 4 columns per thread, SGD
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) Ct[(m0 + 16 * i + 4 * q + e) * 68 + n0 + 16 * t + r] = accw[i][t][e];
    stage3();  // An does not overlap Ct
    fetch3(m_base, 1);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 v = *reinterpret_cast<const float4*>(Ct + (8 * g8 + u) * 68 + 4 * c4);
      float4 g;  // the arithmetic of sgd_apply_kernel / rmw_tile, element by element
      g.x = fmaf(a.wd, w[u].x, v.x * gs); g.y = fmaf(a.wd, w[u].y, v.y * gs);
      g.z = fmaf(a.wd, w[u].z, v.z * gs); g.w = fmaf(a.wd, w[u].w, v.w * gs);
      if constexpr (kMom && !kFirst) {
        g.x = fmaf(a.mom, mo[u].x, g.x); g.y = fmaf(a.mom, mo[u].y, g.y);
        g.z = fmaf(a.mom, mo[u].z, g.z); g.w = fmaf(a.mom, mo[u].w, g.w);
      }
      const float4 wn = make_float4(w[u].x - lr * g.x, w[u].y - lr * g.y, w[u].z - lr * g.z, w[u].w - lr * g.w);
#ifdef NNUE_ABLATIONS
      if (!(a.abl & 2))
#endif
      {
        if constexpr (kMom)
          __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(g.x), __float_as_uint(g.y), __float_as_uint(g.z), __float_as_uint(g.w)}, rsm,
                                                 wm_off(u), 0, 2);
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(wn.x), __float_as_uint(wn.y), __float_as_uint(wn.z), __float_as_uint(wn.w)}, rsw,
                                               wm_off(u), 0, 2);
      }
      // rows past the product's table rows (their stores were dropped) contribute nothing to the forward
      w[u] = m_base + 8 * g8 + u < a.direct ? wn : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();  // every thread has read its part of Ct: Wp may overwrite it
    const int next = m_base + 128;
#ifdef NNUE_ABLATIONS
    if (a.abl & 4) {
      fetch1(next, 0, next < p_hi);
      fetch2(next, next < p_hi);
      continue;
    }
#endif
    // ---------------- phase 3: W_new planes + forward
    {
      u32x4 pl[3][4];  // [plane][column e]: 8 bf16 along k (the thread's eight table rows)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        unsigned h[2][4], m[2][4], l[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float xs[4] = {w[2 * t + u].x, w[2 * t + u].y, w[2 * t + u].z, w[2 * t + u].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = xs[e];
            const unsigned hb = __float_as_uint(x) & 0xffff0000u;
            const float r1 = x - __uint_as_float(hb);
            const unsigned mb_ = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(mb_);
            h[u][e] = hb; m[u][e] = mb_; l[u][e] = __float_as_uint(r2);
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pl[0][e][t] = __builtin_amdgcn_perm(h[1][e], h[0][e], 0x07060302u);
          pl[1][e][t] = __builtin_amdgcn_perm(m[1][e], m[0][e], 0x07060302u);
          pl[2][e][t] = __builtin_amdgcn_perm(l[1][e], l[0][e], 0x07060302u);
        }
      }
#pragma unroll
      for (int pnum = 0; pnum < 3; ++pnum)
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<u32x4*>(Wp + pnum * kUfWpPlane + uf_wp_img(4 * c4 + e, g8)) = pl[pnum][e];
    }
    __syncthreads();
    // requests in the order they are needed: the wait counter retires in order
    fetch1(next, 0, next < p_hi);
    fetch2(next, next < p_hi);
    contract3(0);
    __syncthreads();
    stage3();
    __syncthreads();
    if (m_base + 64 < a.direct) contract3(1);  // uniform; MFMAs only
    __syncthreads();
  }
  // ---------------- the slab of the next forward (FwdEpi with ksplit > 1: plain stores)
  float* __restrict__ dst = a.slabs + (size_t)s * Bn * L1;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = n_base + n0 + 16 * t + r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + 16 * i + 4 * q + e;
        if (m < Bn) dst[(size_t)m * L1 + n] = accf[i][t][e];
      }
  }
}
