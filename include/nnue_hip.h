/*
 * nnue_hip.h -- C ABI of libnnue_hip.so: the MI355X (gfx950) kernels behind the
 * NNUE training hot path of marict/nnue-vision.
 *
 * The reference has no FFI seam for this path: the seam is the Python module
 * surface of nnue.py.  This library is what a ctypes
 * binding under that surface calls; every entry point below names the
 * reference code (file:line under the reference root) whose arithmetic it
 * replaces.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, ints, floats; no torch / C++ types;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the null stream) and never synchronises, allocates or frees:
 *     scratch memory is passed in, so calls can be captured into a hipGraph;
 *   - tensors are dense row-major float32 unless stated; ids are int32 inside
 *     the library and int64 at the reference boundary (nnue_ft_prepare);
 *   - return 0 on success, a negative NNUE_E_* code otherwise; nothing throws.
 *     nnue_hip_last_error() returns a thread-local description of the last
 *     failure.  Arguments are validated BEFORE any launch: a call that returns
 *     an error has launched nothing.
 *   - all float pointers must be 16-byte aligned (torch allocations are).
 *
 * Active-feature list ("act list") layout shared by the calls below, capacity
 * `cap` entries per sample:
 *     rows[b*cap + k]  int32  table row, already clamped to [0, F-1]
 *     coef[b*cap + k]  float  multiplier (feature value; 1.0 for binary features)
 *     pos [b*cap + k]  int32  where the entry's value-gradient goes
 *     n[b]             int32  number of valid entries of sample b (k < n[b])
 * and the transposed coefficient matrix
 *     coefT[f*ldb + b] float  sum of coef over the entries of sample b that map to row f
 * with ldb >= B a multiple of 64.  Entries keep the caller's order.
 */
#ifndef NNUE_HIP_H
#define NNUE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNUE_HIP_ABI_VERSION 30

#define NNUE_OK 0
#define NNUE_E_ARG (-1)     /* null pointer, non-positive size, bad alignment */
#define NNUE_E_SHAPE (-2)   /* sizes inconsistent with each other */
#define NNUE_E_LAUNCH (-3)  /* HIP reported a launch error */
#define NNUE_E_SCRATCH (-4) /* scratch buffer too small */

typedef void* nnue_stream_t; /* hipStream_t */
/* arguments of the classifier's small-gradient tile family when it rides in nnue_ftm_backward's launch (opaque; filled by
 * nnue_classifier_train_rider, read by nnue_ftm_backward[_bucketed]) */
typedef struct nnue_cls_rider { unsigned char opaque[256]; } nnue_cls_rider;

int nnue_hip_abi_version(void);
const char* nnue_hip_last_error(void);

/* ---- front end ---------------------------------------------------------------- */

/* nn.Conv2d(3, fps, 3, stride, padding=1, bias=False)  (nnue.py:486-493, call :640).
 * images [B,3,H,W], weight [fps,3,3,3] (OIHW), conv_out [B,fps,Gh,Gw] with
 * Gh = (H-1)/stride + 1, Gw = (W-1)/stride + 1.  Each output is one fmaf chain over
 * (ci, kh, kw) in that order, so results do not depend on the launch shape. */
int nnue_conv3x3_forward(const float* images, const float* weight, float* conv_out,
                         int B, int H, int W, int fps, int stride, nnue_stream_t stream);

/* Gradient of that conv w.r.t. its input (autograd of nnue.py:640; the training step never needs it -- offered
 * so that NNUE.forward's autograd node differentiates w.r.t. the pixels without a stock op):
 *   d_images[b,ci,y,x] = sum_{c,kh,kw : y = oh*stride+kh-1, x = ow*stride+kw-1} d_conv_out[b,c,oh,ow] * weight[c,ci,kh,kw]
 * one fixed-order fmaf chain per pixel and channel. */
int nnue_conv3x3_backward_input(const float* d_conv_out, const float* weight, int B, int H, int W, int fps,
                                int stride, float* d_images, nnue_stream_t stream);

/* The value half of NNUE._to_sparse_features (nnue.py:601-606, :628-633): the values are the map's own entries at the
 * active ids:  val[b,i] = map[b, idx[b,i]]  (idx >= 0; 0 for the -1 padding);  map [B,P], idx int64 [B,M]. */
int nnue_sparse_values(const float* map, const int64_t* idx, int B, int P, int M, float* val, nnue_stream_t stream);
/* ... and its autograd (CopySlices / IndexBackward of nnue.py:628-633): the values stay attached to the map,
 *   d_map[b,idx[b,i]] = d_val[b,i], every other element 0 (ids of a sample are distinct). */
int nnue_sparse_values_backward(const float* d_val, const int64_t* idx, int B, int P, int M, float* d_map,
                                nnue_stream_t stream);

/* StraightThroughBinary.forward + NNUE._to_sparse_features  (nnue.py:19-25, :590-635)
 * without the data-dependent width: per sample, ascending flat ids p = c*Gh*Gw + h*Gw + w
 * with conv_out > thr[c], written as an act list of capacity P = fps*Gh*Gw
 * (rows = min(p, F-1), pos = p, coef = 1) and as coefT [F, ldb] (rows < F-1: the bit;
 * row F-1: the number of active ids >= F-1 -- the clamp of nnue.py:701).  Ids are bit-exact
 * given conv_out.  Every element of rows/pos/coef beyond n[b] is left untouched;
 * coefT is fully overwritten for b < B. */
int nnue_binarize_features(const float* conv_out, const float* thr,
                           int B, int fps, int Gh, int Gw, int F,
                           int32_t* rows, int32_t* pos, float* coef, int32_t* n,
                           float* coefT, int ldb, nnue_stream_t stream);

/* Same ids in the reference's own format (nnue.py:609-633): idx [B,M] int64 padded with -1,
 * val [B,M] float32 padded with 0, from an act list whose max n[b] the caller has read
 * back (M = max(max n, 1)). */
int nnue_act_to_padded(const int32_t* pos, const float* coef, const int32_t* n, int cap,
                       int B, int M, int64_t* idx, float* val, nnue_stream_t stream);

/* StraightThroughBinary.backward for the threshold + conv weight gradient
 * (nnue.py:28-54; autograd of the conv at nnue.py:640).
 *   d_thr[c]            = -sum_{b,h,w} d_conv_out * k*s*(1-s),  s = sigmoid(k*(x - thr[c])), k = 10
 *   d_weight[c,ci,kh,kw] = sum_{b,h,w} d_conv_out[b,c,h,w] * images[b,ci,h*stride+kh-1,w*stride+kw-1]
 * Deterministic two-stage sum; scratch >= nnue_ste_conv_backward_scratch(...) bytes.
 * stages: 3 = both; 1 = stage 1 only: the per-workgroup partials stay at the start of scratch as
 * partial[(c*28+q)*chunks + k], k < nnue_ste_conv_backward_chunks(...), q < 27 the conv-weight terms, q = 27 the
 * threshold term, and d_thr / d_weight are not written -- finish with stages = 2 (same arguments) or hand the
 * partials to nnue_sgd_step, whose norm launch then carries the second stage. */
int64_t nnue_ste_conv_backward_scratch(int B, int fps, int Gh, int Gw);
int64_t nnue_ste_conv_backward_chunks(int B, int fps, int Gh, int Gw);
int nnue_ste_conv_backward(const float* images, const float* conv_out, const float* thr,
                           const float* d_conv_out, int B, int H, int W, int fps, int stride,
                           float* d_thr, float* d_weight, void* scratch, int64_t scratch_bytes,
                           int stages, nnue_stream_t stream);
/* The same sums (nnue.py:28-54; autograd of the conv at nnue.py:640) from the im2col form of the images that
 * nnue_ftm_conv_binarize_patches leaves: patches [27][B*Gh*Gw] f32, term-major, read coalesced along positions instead of pixels
 * gathered at the conv stride.  conv_out (needed only inside the threshold term's sigmoid) is read when given; with conv_out NULL
 * it is re-formed from the patches and `weight` [fps][27] with the forward's own fmaf chain.  Either way d_thr, d_weight and the
 * stage-1 partials are BITWISE those of nnue_ste_conv_backward.  fps <= 64. */
int nnue_ste_conv_backward_patches(const float* patches, const float* weight, const float* conv_out, const float* thr,
                                   const float* d_conv_out, int B, int fps, int Gh, int Gw,
                                   float* d_thr, float* d_weight, void* scratch, int64_t scratch_bytes,
                                   int stages, nnue_stream_t stream);

/* ---- FeatureTransformer ------------------------------------------------------- */

/* Turns the reference-format inputs of FeatureTransformer.forward (nnue.py:686: idx int64
 * [B,M], -1 = padding, any order, repeats allowed; val float32 [B,M]) into an act list of
 * capacity M (valid entries compacted in order, rows clamped as nnue.py:701, pos = original
 * column) and coefT [F, ldb].  coefT is zeroed by the call.  Repeats accumulate. */
int nnue_ft_prepare(const int64_t* idx, const float* val, int B, int M, int F,
                    int32_t* rows, int32_t* pos, float* coef, int32_t* n,
                    float* coefT, int ldb, nnue_stream_t stream);

/* FeatureTransformer.forward  (nnue.py:686-710):
 *   out[b,:] = bias + sum_{k<n[b]} coef[b,k] * weight[rows[b,k], :]
 * weight [F,L1], out [B,L1]. */
int nnue_ft_forward(const float* weight, const float* bias,
                    const int32_t* rows, const float* coef, const int32_t* n, int cap,
                    int B, int F, int L1, float* out, nnue_stream_t stream);

/* Gradient of the above w.r.t. weight and bias (autograd IndexBackward + index_put_
 * (accumulate) of nnue.py:702-708), as a per-row gather-sum over coefT -- no atomics,
 * bitwise reproducible:
 *   d_weight[f,:] = sum_b coefT[f,b] * d_out[b,:]      d_bias = sum_b d_out[b,:]
 * Either output may be NULL. */
int nnue_ft_backward_weight(const float* d_out, const float* coefT, int ldb,
                            int B, int F, int L1, float* d_weight, float* d_bias,
                            nnue_stream_t stream);

/* Gradient w.r.t. the feature values (autograd of nnue.py:705-707):
 *   dst[b*dst_ld + pos[b,k]] = < d_out[b,:], weight[rows[b,k],:] >   for k < n[b]
 * dst is zero-filled first ([B, dst_ld]); dst_ld = M for the stand-alone op, P when the
 * values are the binary map (the gradient then IS d_conv_out, nnue.py:628-633 + :33). */
int nnue_ft_backward_values(const float* d_out, const float* weight,
                            const int32_t* rows, const int32_t* pos, const int32_t* n, int cap,
                            int B, int F, int L1, float* dst, int dst_ld, nnue_stream_t stream);

/* ---- FeatureTransformer for binary grid features: bit masks, tile lists, LDS-staged tiles ---------
 *
 * Inside NNUE.forward the feature values are exactly {0,1} and the ids ascend (nnue.py:590-635), so
 * the act list collapses to
 *     maskW[b*pw64 + w]  uint64  bit (p & 63) of word p >> 6 = flat position p of sample b is active
 *     maskT[f*bw64 + w]  uint64  bit (b & 63) of word b >> 6 = sample b selects table row f (f < F-1: its
 *                                own position; f = F-1: sink[b] != 0; f = F: every sample -- the bias row)
 *     sink[b]            float   number of active positions >= F-1 (they all clamp to row F-1, nnue.py:701)
 * (pw64, bw64 even) and, per output and per tile of 128 staged rows, a padded list of the rows to add:
 *     tlW [B][tiles_fwd][128] u16 / tcW [B][tiles_fwd] u8       sample b   x table-row tile
 *     tlT [F+1][tiles_bwd][128] u16 / tcT [F+1][tiles_bwd] u8   output row x batch tile
 * Entries are LDS byte offsets (local row or sample index * 256), ascending, padded with 32768 (= an all-zero
 * row); entry e sits in slot ((e>>2)&3)*32 + (e>>4)*4 + (e&3); tc holds the entry count.  The list buffers
 * are passed as uint8_t* (256 bytes per record).
 * A workgroup stages a tile of the table (or of d_out) in LDS once and all its samples (rows) gather from
 * LDS, so memory-side traffic is the table once per sample tile instead of once per sample.  L1 must be 256,
 * 512 or 1024 (nnue_ftb_supported); other widths use the list kernels above. */
int nnue_ftb_supported(int L1);
int nnue_ftb_list_tiles(int B, int F, int P, int* tiles_fwd, int* tiles_bwd);
int64_t nnue_ftb_scratch(int B, int F, int P, int L1); /* bytes for nnue_ftb_forward / _backward_weight */

/* StraightThroughBinary.forward + _to_sparse_features as bit masks and tile lists (nnue.py:19-25,
 * :590-635).  Also writes n[b] = number of active positions.  Bit-exact given conv_out.
 * stages: 1 = per-sample outputs (maskW, sink, n, tlW/tcW), 2 = transposed outputs (maskT, tlT/tcT; needs
 * sink from stage 1), 3 = both.  The transposed outputs are only read by nnue_ftb_backward_weight, so a
 * caller may run stage 2 on a second stream beside the forward. */
int nnue_binarize_bits(const float* conv_out, const float* thr, int B, int fps, int Gh, int Gw, int F,
                       uint64_t* maskW, int pw64, uint64_t* maskT, int bw64, float* sink, int32_t* n,
                       uint8_t* tlW, uint8_t* tcW, uint8_t* tlT, uint8_t* tcT, int stages, nnue_stream_t stream);

/* FeatureTransformer.forward for binary features (nnue.py:686-710):
 *   out[b,:] = bias + sum_{p active, p < min(F-1,P)} weight[p,:] + sink[b] * weight[F-1,:] */
int nnue_ftb_forward(const float* weight, const float* bias, const uint8_t* tlW, const uint8_t* tcW,
                     const float* sink, int B, int F, int P, int L1, float* out,
                     void* scratch, int64_t scratch_bytes, nnue_stream_t stream);

/* Its weight/bias gradient (autograd of nnue.py:702-708), fixed summation order, no atomics.
 * Either output may be NULL. */
int nnue_ftb_backward_weight(const float* d_out, const uint8_t* tlT, const uint8_t* tcT, const float* sink,
                             int B, int F, int P, int L1, float* d_weight, float* d_bias,
                             void* scratch, int64_t scratch_bytes, nnue_stream_t stream);

/* Its value gradient scattered to the map (autograd of nnue.py:705-707 and :628-633; identity STE :33):
 *   d_conv_out[b,p] = active(b,p) ? < d_out[b,:], weight[min(p,F-1),:] > : 0     for every p < P */
int nnue_ftb_backward_values(const float* d_out, const float* weight, const uint64_t* maskW, int pw64,
                             int B, int F, int P, int L1, float* d_conv_out, nnue_stream_t stream);

/* ---- FeatureTransformer for binary grid features as dense products on the f32 MFMA ----------------
 *
 * At the reference's threshold 43 % of the grid features are active (414 of 968 at 32x32), so the products
 *     out      = A W + bias,   d_weight = A^T d_out,   d_value = (d_out W^T) . A
 * with A[b][f] = membership of table row f in sample b are dense enough to run as matrix products
 * (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate; summation order differs from the gather
 * kernels).  A is the binary map itself as a byte {0,1} matrix bits[B][P] (P = fps*Gh*Gw, nnue.py:19-25),
 * widened to float while tiles are staged; no masks or id lists are built.  sink[b] = number of active positions >= F-1 (they clamp to row F-1,
 * nnue.py:701), a rank-one term.  P and L1 must be multiples of 4 (nnue_ftm_supported). */
int nnue_ftm_supported(int F, int P, int L1);
int64_t nnue_ftm_scratch(int B, int F, int P, int L1); /* bytes for nnue_ftm_forward (split-K slabs) */

/* StraightThroughBinary.forward as a byte matrix (nnue.py:19-25): bits[b,p] = conv_out[b,p] > thr[channel of p],
 * n[b] = active positions (nnue.py:603-607), sink[b] as above.  Bit-exact given conv_out. */
int nnue_ftm_binarize(const float* conv_out, const float* thr, int B, int fps, int Gh, int Gw, int F,
                      uint8_t* bits, int32_t* n, float* sink, nnue_stream_t stream);

/* nnue_conv3x3_forward + nnue_ftm_binarize in one launch (self.conv + StraightThroughBinary.forward, nnue.py:640,
 * :646-647, :19-25): conv_out, bits, n and sink are bitwise what the two calls give. */
int nnue_ftm_conv_binarize(const float* images, const float* weight, const float* thr, int B, int H, int W,
                           int fps, int stride, int F, float* conv_out, uint8_t* bits, int32_t* n, float* sink,
                           nnue_stream_t stream);
/* The same launch (nnue.py:640, :646-647, :19-25) also leaving the im2col form of the images for the backward:
 * patches[q][b*Gh*Gw + hw] = the pixel under tap q = ci*9 + kh*3 + kw of position hw (0 where the tap falls off the image),
 * 27*B*Gh*Gw floats.  conv_out may be NULL (not written): with a stride above 3 the taps of neighbouring positions do not
 * overlap, the patches are a fraction of the images (0.18 at 224x224, stride 7) and the training step needs conv_out only
 * where nnue_ste_conv_backward_patches re-forms it.  bits, n, sink (and conv_out when given) are bitwise nnue_ftm_conv_binarize's. */
int nnue_ftm_conv_binarize_patches(const float* images, const float* weight, const float* thr, int B, int H, int W,
                                   int fps, int stride, int F, float* patches, float* conv_out, uint8_t* bits,
                                   int32_t* n, float* sink, nnue_stream_t stream);

/* FeatureTransformer.forward for the binary map (nnue.py:686-710):
 *   out[b,:] = bias + sum_{p active, p < min(F-1,P)} weight[p,:] + sink[b] * weight[F-1,:] */
int nnue_ftm_forward(const uint8_t* bits, const float* sink, const float* weight, const float* bias,
                     int B, int F, int P, int L1, float* out,
                     void* scratch, int64_t scratch_bytes, nnue_stream_t stream);

int nnue_ftm_forward_l1_supported(int B, int F, int P, int L1, int L2); /* shapes nnue_ftm_forward_l1 takes */

/* nnue_ftm_forward that also forms, in its epilogue, each 64-column tile's share of the pairwise block and the
 * classifier's first Linear (nnue.py:660-666, :728-730): part[t][b][j] for t < L1/64, to be consumed by
 * nnue_classifier_train_step with phases bit 8 (part = the start of its scratch).  out (the FeatureTransformer
 * output) is bitwise what nnue_ftm_forward writes.  Shapes: nnue_ftm_forward_l1_supported (declared above). */
int nnue_ftm_forward_l1(const uint8_t* bits, const float* sink, const float* weight, const float* bias,
                        const float* w1, int B, int F, int P, int L1, int L2, float* out, float* part,
                        nnue_stream_t stream);

/* Its weight/bias gradient (autograd of nnue.py:702-708); fixed summation order, no atomics.  Rows the map
 * cannot reach are written as zero.  Either output may be NULL. */
int nnue_ftm_backward_weight(const uint8_t* bits, const float* sink, const float* d_out,
                             int B, int F, int P, int L1, float* d_weight, float* d_bias,
                             nnue_stream_t stream);

/* Its value gradient on the map (autograd of nnue.py:705-707 and :628-633; identity STE :33):
 *   d_conv_out[b,p] = active(b,p) ? < d_out[b,:], weight[min(p,F-1),:] > : 0     for every p < P */
int nnue_ftm_backward_values(const uint8_t* bits, const float* d_out, const float* weight,
                             int B, int F, int P, int L1, float* d_conv_out, nnue_stream_t stream);
/* nnue_ftm_backward_values with a workspace (same result contract: autograd of nnue.py:702-708 and :628-633 -- the gradient
 * reaches the map at active positions only).  For big maps the workspace lets d_out be split ONCE per launch into its three
 * bf16 planes, staged by LDS-DMA, while the table's fragments go straight to registers (csrc/ftv_kernels.hip);
 * nnue_ftm_backward_values_scratch returns the bytes needed (0: this shape runs the workspace-free kernels, scratch may be
 * NULL).  Too little workspace for a shape that needs one: NNUE_E_SCRATCH. */
int64_t nnue_ftm_backward_values_scratch(int B, int F, int P, int L1);
int nnue_ftm_backward_values_ws(const uint8_t* bits, const float* d_out, const float* weight, int B, int F, int P, int L1,
                                float* d_conv_out, void* scratch, int64_t scratch_bytes, nnue_stream_t stream);

int nnue_ftm_backward_cw_supported(int B, int F, int P, int L1, int L2); /* shapes whose nnue_ftm_backward takes d_w1 */
int64_t nnue_ftm_backward_sq_count(int B, int F, int P, int L1); /* floats nnue_ftm_backward's sq_partial receives; 0: none */

/* nnue_ftm_backward_weight + nnue_ftm_backward_values as ONE launch (autograd of nnue.py:702-708, :628-633): the
 * two products and the tail rows are independent, so their workgroups share the chip.  Same results, bit for bit,
 * as the two separate calls.  d_weight, d_bias and d_conv_out are required.
 * Optional rider (d_w1 != NULL, shapes: nnue_ftm_backward_cw_supported, declared above): the weight gradient of the
 * classifier's first Linear, d_w1[L2][L1] = d_z1^T l0 (autograd of nnue.py:728-730 through the pairwise block
 * nnue.py:660-666), from ft[B][L1] (the FeatureTransformer output) and d_z1[B][L2] (left in the classifier's scratch
 * by nnue_classifier_train_step, at nnue_classifier_train_dz1_offset); pair with phases bit 16 there.
 * sq_partial (may be NULL): nnue_ftm_backward_sq_count(B, F, P, L1) floats (declared above) that receive, per
 * weight-gradient tile, the sum of the squares of the elements it wrote -- together the squared norm of
 * d_weight[0 .. min(F-1, P)) -- so that nnue_sgd_step (ext_partial) need not read those rows again for
 * clip_grad_norm_ (train.py:363-364).
 * small != NULL (filled by nnue_classifier_train_rider; merged-launch shapes only): the classifier's small batch-reduced
 * gradients (d_w3, d_w2, the three bias gradients; autograd of nnue.py:728-734) and the mean loss (train.py:250-254) run as
 * one more tile family of this launch -- ~30 workgroups beside a few hundred -- instead of beside the classifier's d_x tiles
 * (nnue_classifier_train_step phases bit 32). */
int nnue_ftm_backward(const uint8_t* bits, const float* sink, const float* d_out, const float* weight,
                      int B, int F, int P, int L1, float* d_weight, float* d_bias, float* d_conv_out,
                      const float* ft, const float* d_z1, int L2, float* d_w1, float* sq_partial,
                      const nnue_cls_rider* small, nnue_stream_t stream);

/* ---- pairwise product + SimpleClassifier -------------------------------------- */

/* Forward of  l0 = cat(x[:, :L1/2] * x[:, L1/2:], x[:, :L1/2])  (nnue.py:660-666, when
 * `pairwise` != 0; l0 = x otherwise) followed by Linear(L1,L2)+act, Linear(L2,L3)+act,
 * Linear(L3,C)  (nnue.py:728-734).  act = ReLU, or min(ReLU, clip) when clip > 0 (build
 * extension, off = reference).  Saves h1 [B,L2], h2 [B,L3] (post-activation) for backward.
 * scratch >= nnue_classifier_scratch(B, L1, L2, L3) bytes (covers forward and backward). */
int64_t nnue_classifier_scratch(int B, int L1, int L2, int L3);
int nnue_classifier_forward(const float* x, int pairwise,
                            const float* w1, const float* b1, const float* w2, const float* b2,
                            const float* w3, const float* b3, float clip,
                            int B, int L1, int L2, int L3, int C,
                            float* h1, float* h2, float* logits,
                            void* scratch, int64_t scratch_bytes, nnue_stream_t stream);

/* Backward of the above (autograd of nnue.py:660-669 and :728-734 in the reference).  d_x [B,L1]
 * is the gradient w.r.t. x (through the pairwise block when `pairwise`); NULL skips it.  Weight/bias gradients are overwritten, not accumulated. */
int nnue_classifier_backward(const float* x, int pairwise,
                             const float* w1, const float* w2, const float* w3, float clip,
                             const float* h1, const float* h2, const float* d_logits,
                             int B, int L1, int L2, int L3, int C,
                             float* d_x, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                             float* d_w3, float* d_b3,
                             void* scratch, int64_t scratch_bytes, nnue_stream_t stream);

/* The three calls above as ONE training step of the pairwise/classifier block with mean cross-entropy
 * (nnue.py:660-669, :728-734 + compute_loss, train.py:250-254, and their autograd): same results as
 * nnue_classifier_forward -> nnue_cross_entropy(grad_scale) -> nnue_classifier_backward, but the narrow
 * layers, the loss and their backward run per sample in a single kernel.  d_x may be NULL.
 * phases: 1 = activations, per-sample losses and d_x (the critical path of a training step), 2 = the mean loss
 * and the six weight/bias gradients (needs phase 1 on the same scratch; nothing downstream waits for it, so a
 * caller may run it on a second stream), 3 = both.  Adding 4 (phases 5 then 6, or 7) moves the first-layer weight
 * product into phase 1's d_x launch -- both only need d_z1, so the two small products share the chip; the phase-2
 * call (same arguments) then only sums its slabs.  Results are identical either way.  Adding 8 says the layer-1
 * pre-activation slabs part[L1/64][B][L2] are already at the START of scratch, written by nnue_ftm_forward_l1 (the
 * FeatureTransformer forward forms them in its epilogue); phase 1 then launches no layer-1 product of its own.
 * Adding 16 (not together with 4) says d_w1 is produced by nnue_ftm_backward's rider from the d_z1 this call leaves
 * in scratch at byte offset nnue_classifier_train_dz1_offset (-1 for non-positive sizes): no first-layer weight
 * product and no slab sum are launched here, d_w1 is not written; with both phases in the one call (19, 27) the
 * small weight/bias gradients and the mean loss share the d_x launch -- unless 32 is added as well (51, 59): then they
 * are left to nnue_ftm_backward's launch too (its `small` argument, filled by nnue_classifier_train_rider with the same
 * tensors and scratch), and the d_x launch holds only d_x tiles.
 * scratch >= nnue_classifier_train_scratch(B, L1, L2, L3, C) bytes. */
int64_t nnue_classifier_train_scratch(int B, int L1, int L2, int L3, int C);
int64_t nnue_classifier_train_dz1_offset(int B, int L1, int L2, int L3, int C, int pairwise);
int nnue_classifier_train_step(const float* x, int pairwise,
                               const float* w1, const float* b1, const float* w2, const float* b2,
                               const float* w3, const float* b3, float clip,
                               const int64_t* labels, float grad_scale,
                               int B, int L1, int L2, int L3, int C,
                               float* h1, float* h2, float* logits, float* sample_loss, float* loss,
                               float* d_x, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                               float* d_w3, float* d_b3,
                               void* scratch, int64_t scratch_bytes, int phases, nnue_stream_t stream);
/* A host-side call (no launch) that fills `out` with the arguments of the small-gradient tile family for a step run with
 * phases bit 32 -- d_w3, d_w2, d_b3, d_b2, d_b1 (autograd of nnue.py:728-734) and the mean loss (train.py:250-254) then run
 * inside nnue_ftm_backward's launch (its `small` argument).  Same tensors and scratch as the train step; buckets = NULL
 * for one layer stack. */
int nnue_classifier_train_rider(int pairwise, int B, int L1, int L2, int L3, int C, const float* h1, const float* h2,
                                const float* sample_loss, float* loss, float* d_b1, float* d_w2, float* d_b2,
                                float* d_w3, float* d_b3, void* scratch, int64_t scratch_bytes,
                                const struct nnue_buckets* buckets, nnue_cls_rider* out);

/* ---- bucketed layer stacks (build extension; BASELINE configs[2]) -------------------------------------------------
 *
 * The reference trains ONE SimpleClassifier (nnue.py:713-738; serialize.py:57 writes num_ls_buckets = 1) while its
 * engine still loads N LayerStacks (engine/src/nnue_engine.cpp:619-635).  K > 1 here = K independent weight sets
 * w1 [K][L2][L1], b1 [K][L2], w2 [K][L3][L2], b2 [K][L3], w3 [K][C][L3], b3 [K][C], one selected per sample:
 *     bucket[b] = min(K-1, n[b] * K / (P + 1)),  n[b] = active features of sample b, P = flat ids of the map
 * -- a by-product of the binarise kernels.  There is no reference for K > 1 (parity unpinned; oracle/nnue_oracle.py
 * holds the CPU restatement); with K == 1 every *_bucketed entry point is exactly its plain counterpart.
 * nnue_bucket_group sorts the samples by bucket (stable) into bucket-homogeneous 16-row tiles so that each MFMA
 * tile of the first layer multiplies by one bucket's weights:
 *     rows[16 t + i]   sample in row i of tile t (-1 = padding)     tile_bucket[t]  its bucket (-1 = unused tile)
 *     seg[k], seg[k+1] row range of bucket k (multiples of 16)      t < nnue_bucket_tile_count(B, K) = ceil(B/16) + K
 * All arrays are int32 device memory; the struct itself is read on the host. */
typedef struct nnue_buckets {
  int32_t K;                  /* layer stacks; <= 1: single stack, pointers ignored */
  const int32_t* bucket;      /* [B] */
  const int32_t* rows;        /* [tiles * 16] */
  const int32_t* tile_bucket; /* [tiles] */
  const int32_t* seg;         /* [K + 1] */
  int32_t tiles;              /* nnue_bucket_tile_count(B, K) */
} nnue_buckets;

int nnue_bucket_tile_count(int B, int K);
/* Selector + grouping in one single-workgroup launch (the K = 1 case of the structure is nnue.py:713-738; header field
 * serialize.py:57).  P == 0: n[b] is taken as the bucket itself (clamped to [0, K-1]) -- the stand-alone classifier
 * call.  K <= 64. */
int nnue_bucket_group(const int32_t* n, int B, int P, int K, int32_t* bucket, int32_t* rows,
                      int32_t* tile_bucket, int32_t* seg, nnue_stream_t stream);

/* nnue_ftm_forward (nnue.py:686-710) with nnue_bucket_group riding in the same launch as one extra workgroup: the grouping
 * only needs the binarise kernel's counts n, like the product itself, so it costs no launch of its own.  Same outputs as
 * the two calls. */
int nnue_ftm_forward_grouping(const uint8_t* bits, const float* sink, const float* weight, const float* bias,
                              int B, int F, int P, int L1, float* out, void* scratch, int64_t scratch_bytes,
                              const int32_t* n, int K, int32_t* bucket, int32_t* rows, int32_t* tile_bucket,
                              int32_t* seg, nnue_stream_t stream);

/* nnue_classifier_forward / _backward / _train_step (nnue.py:660-666, :728-734 and their autograd) with stacked
 * weights and gradients [K][...] and the grouping above.  train_step: phases bits 8 and 16 are refused for K > 1
 * (those products live in the FeatureTransformer launches, which know one stack).  d_w1 of a bucket is summed over
 * slices of that bucket's own rows; scratch sizes from the *_scratch_bucketed functions. */
int64_t nnue_classifier_scratch_bucketed(int B, int L1, int L2, int L3, int K);
int64_t nnue_classifier_train_scratch_bucketed(int B, int L1, int L2, int L3, int C, int K);
int nnue_classifier_forward_bucketed(const float* x, int pairwise,
                                     const float* w1, const float* b1, const float* w2, const float* b2,
                                     const float* w3, const float* b3, float clip,
                                     int B, int L1, int L2, int L3, int C,
                                     float* h1, float* h2, float* logits,
                                     void* scratch, int64_t scratch_bytes, const nnue_buckets* buckets,
                                     nnue_stream_t stream);
/* autograd of the bucketed forward (nnue.py:728-734 per stack): gradients [K][...], d_x [B][L1]. */
int nnue_classifier_backward_bucketed(const float* x, int pairwise,
                                      const float* w1, const float* w2, const float* w3, float clip,
                                      const float* h1, const float* h2, const float* d_logits,
                                      int B, int L1, int L2, int L3, int C,
                                      float* d_x, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                                      float* d_w3, float* d_b3,
                                      void* scratch, int64_t scratch_bytes, const nnue_buckets* buckets,
                                      nnue_stream_t stream);
/* forward + mean cross-entropy + backward of the bucketed block in one call (train.py:250-254, :360-361 on top of
 * nnue.py:728-734 per stack).  phases bit 16 with K > 1 (pairwise block, L1 % 4 == 0): d_w1 is left to
 * nnue_ftm_backward_bucketed; the per-sample kernel then runs once per GROUPED row and leaves that product's operands in
 * grouped row order inside scratch -- d_z1 [tiles*16][L2] at nnue_classifier_train_dz1_grouped_offset and the block's
 * input x [tiles*16][L1] at nnue_classifier_train_x_grouped_offset (byte offsets; padding rows zero; -1 for K <= 1). */
int64_t nnue_classifier_train_dz1_grouped_offset(int B, int L1, int L2, int L3, int C, int K);
int64_t nnue_classifier_train_x_grouped_offset(int B, int L1, int L2, int L3, int C, int K);
int nnue_classifier_train_step_bucketed(const float* x, int pairwise,
                                        const float* w1, const float* b1, const float* w2, const float* b2,
                                        const float* w3, const float* b3, float clip,
                                        const int64_t* labels, float grad_scale,
                                        int B, int L1, int L2, int L3, int C,
                                        float* h1, float* h2, float* logits, float* sample_loss, float* loss,
                                        float* d_x, float* d_w1, float* d_b1, float* d_w2, float* d_b2,
                                        float* d_w3, float* d_b3,
                                        void* scratch, int64_t scratch_bytes, int phases,
                                        const nnue_buckets* buckets, nnue_stream_t stream);

/* The table's weight gradient consumed where it is produced (single rank, SGD): clip_grad_norm_ needs the global norm
 * before any parameter moves (train.py:363-366), and
 *     || A^T D ||_F^2 = sum_{b,b'} (A A^T)_{bb'} (D D^T)_{bb'}      (A = the map [B][direct], D = d_out [B][L1])
 * gives the table's share from two B x B Gram matrices without forming the [direct][L1] gradient.
 * nnue_ftm_gram_sqnorm leaves nnue_ftm_gram_sq_count(B, L1) partial sums (unscaled; sum = that squared norm, formed as
 * sum (G_A D) . D so that D D^T is not needed either) for nnue_sgd_step(ext_partial, ..., coef_out,
 * ext_applied_elsewhere = 1); gram is nnue_ftm_gram_scratch(B, F, P) floats of scratch whose first B*B hold A A^T afterwards
 * (formed on the i8 matrix unit straight from the byte map; K slices leave int32 slabs behind it, added in slice order:
 * no atomics, nothing to clear).  nnue_ftm_backward_tail_rows forms the rows the product does not cover (d_bias, row F-1, zero
 * rows: what nnue_ftm_backward_weight adds to its product).  nnue_ftm_backward_weight_update then runs the product
 * d_W = A^T d_out (autograd of nnue.py:702-708) and, element by element in its epilogue, the optimizer's update
 *     g = coef[0]*grad_scale*d_W + wd*w ;  m = first_step ? g : momentum*m + g ;  w -= lr*m       (train.py:457-464)
 * on table rows [0, direct) -- the same arithmetic nnue_sgd_step applies -- so d_weight is never written or read back
 * (268 MB each way at the 224x224 configuration).  momentum_rows may be NULL when momentum == 0. */
int64_t nnue_ftm_gram_sq_count(int B, int L1);
/* floats of scratch behind nnue_ftm_gram_sqnorm's gram pointer (clip_grad_norm_, train.py:363-366: see above) */
int64_t nnue_ftm_gram_scratch(int B, int F, int P);
int nnue_ftm_gram_sqnorm(const uint8_t* bits, const float* d_out, int B, int F, int P, int L1,
                         float* gram, float* sq_partial, nnue_stream_t stream);
/* nnue_ftm_gram_sqnorm with nnue_ftm_backward_tail_rows' workgroups riding in its first launch (clip_grad_norm_, train.py:363-366 +
 * autograd of nnue.py:691, :701-708): both only read the map / d_out; the same results, one launch fewer. */
int nnue_ftm_gram_sqnorm_tail(const uint8_t* bits, const float* sink, const float* d_out, int B, int F, int P, int L1,
                              float* gram, float* sq_partial, float* d_weight, float* d_bias, nnue_stream_t stream);
/* (autograd of nnue.py:691, :701-708 for the bias row and the clamp-sink row F-1) */
int nnue_ftm_backward_tail_rows(const float* sink, const float* d_out, int B, int F, int P, int L1,
                                float* d_weight, float* d_bias, nnue_stream_t stream);
/* (autograd of nnue.py:702-708 + train.py:363-366, :457-464, see above) */
int nnue_ftm_backward_weight_update(const uint8_t* bits, const float* d_out, int B, int F, int P, int L1,
                                    float* weight, float* momentum_rows, const float* coef,
                                    float lr, float momentum, float weight_decay, float grad_scale,
                                    int first_step, const float* lr_dev, nnue_stream_t stream);

/* nnue_ftm_backward_weight_update of step t and nnue_ftm_forward of step t+1 in ONE pass over the table (inside a group of
 * steps whose batches are already resident: the optimizer of train.py:457-464 followed by FeatureTransformer.forward,
 * nnue.py:686-710, of the next loop iteration, train.py:359-366).  The update leaves every new table tile in registers and
 * the next forward contracts exactly those tiles, so the forward's own read of the table (268 MB at the 224x224
 * configuration) disappears.  bits_next / sink_next: the next batch's map (nnue_ftm_conv_binarize under the conv weights
 * nnue_sgd_step has just updated; a different buffer than bits); bias and weight row F-1 must already hold their updated
 * values (nnue_sgd_step applies them); out_next [B][L1]; scratch as nnue_ftm_forward (nnue_ftm_scratch bytes).  Table,
 * momentum and out_next are BITWISE what the two separate calls produce.  B = rows of bits / d_out (the batch -- or, under the
 * factor exchange, the all-gathered global batch: 64, 128, 256, 512 or 1024 rows), B_next = rows of the next map and of out_next
 * (<= 128).  Only where the forward is a split-K product over a big table (nnue_ftm_update_forward_supported: L1 % 64 == 0,
 * >= 4096 table rows); NNUE_E_SHAPE otherwise. */
int nnue_ftm_update_forward_supported(int B, int B_next, int F, int P, int L1);
/* (train.py:457-464 + nnue.py:686-710 of the next step, see above) */
int nnue_ftm_backward_weight_update_forward(const uint8_t* bits, const float* d_out, int B, int F, int P, int L1,
                                            float* weight, float* momentum_rows, const float* coef,
                                            float lr, float momentum, float weight_decay, float grad_scale,
                                            int first_step, const float* lr_dev,
                                            const uint8_t* bits_next, const float* sink_next, int B_next, const float* bias,
                                            float* out_next, void* scratch, int64_t scratch_bytes, nnue_stream_t stream);

/* Reporting only: 1 when the product of this shape runs on the bf16 matrix unit (exact three-way split of the f32
 * operand), 0 on the f32 MFMA.  which: 0 forward, 1 stand-alone weight gradient, 2 weight-gradient tiles of the merged
 * backward launch, 3 weight gradient with the update in its epilogue; 4 stand-alone value gradient, 5 value-gradient tiles
 * of the merged launch (both operands f32: six bf16 plane products hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi of the
 * two truncation splits -- what is left out is below 2^-23 of a product). */
int nnue_ftm_uses_bf16(int which, int B, int F, int P, int L1);

/* nnue_ftm_backward for bucketed layer stacks (declared with the FeatureTransformer entry points above): d_w1 [K][L2][L1],
 * ft_grouped / d_z1_grouped in grouped row order (grouped_rows = 16 * nnue_bucket_tile_count rows, padding rows zero),
 * seg [K+1] the buckets' row ranges.  The rider's tile family repeats per bucket and contracts only that bucket's rows
 * (autograd of nnue.py:728-730 per stack, through the pairwise block nnue.py:660-666).  K == 1: nnue_ftm_backward. */
int nnue_ftm_backward_bucketed(const uint8_t* bits, const float* sink, const float* d_out, const float* weight,
                               int B, int F, int P, int L1, float* d_weight, float* d_bias, float* d_conv_out,
                               const float* ft_grouped, const float* d_z1_grouped, int L2, float* d_w1,
                               float* sq_partial, int K, const int32_t* seg, int grouped_rows,
                               const nnue_cls_rider* small, nnue_stream_t stream);

/* ---- loss + step tail ---------------------------------------------------------- */

/* F.cross_entropy(logits, labels) (mean over the batch, train.py:250-254) and its gradient
 * d_logits = (softmax - onehot) * grad_scale / B (may be NULL: forward only).
 * sample_loss [B] receives the per-sample losses, loss (device scalar) their mean, summed in
 * sample order (deterministic).  Labels outside [0, C) contribute loss 0 and a zero gradient
 * row -- the caller validates labels; the kernel only stays in bounds. */
int nnue_cross_entropy(const float* logits, const int64_t* labels, int B, int C, float grad_scale,
                       float* sample_loss, float* loss, float* d_logits, nnue_stream_t stream);

/* Prediction bookkeeping of compute_metrics / evaluate_model (evaluate.py:23-59, :62-87): adds this batch to
 * confusion[truth*K + pred] (uint64, K = C, or 2 when C == 1).  pred = first arg-max of the row (numpy's
 * rule); C == 1 uses the reference's binary rule (output > 0.5, target > 0.5).  Accumulates: zero the matrix
 * once before the first batch.  Integer atomics: the result does not depend on execution order. */
int nnue_confusion_accumulate(const float* logits, const int64_t* labels, int B, int C,
                              uint64_t* confusion, nnue_stream_t stream);

/* ---- the compiled engine's integer inference, batched --------------------------------
 *
 * The quantised tensors of a `.nnue` file (layout: serialize.py:33-63, :103-136, :394-491; loader
 * engine/src/nnue_engine.cpp:544-657) in device memory, scalars by value.  The struct itself is read on the
 * host.  l1_w holds the (L2+1) x L1 bytes of the file (the last row is not used by the multiclass path);
 * l2_w is [L3][2*L2] (first L2 columns used). */
typedef struct nnue_engine_model {
  int32_t num_features, l1, l2, l3, classes, grid, oc;
  float conv_scale, threshold, quantized_one, l1_scale, l2_scale, out_scale;
  const int8_t* conv_w;  /* [oc*27], the file's bytes */
  const int32_t* conv_b; /* [oc] */
  const int16_t* ft_w;   /* [num_features][l1] */
  const int32_t* ft_b;   /* [l1] */
  const int8_t* l1_w;
  const int32_t* l1_b;
  const int8_t* l2_w;
  const int32_t* l2_b;
  const int8_t* out_w;   /* [classes][l3] */
  const int32_t* out_b;
} nnue_engine_model;

/* NNUEEvaluator::evaluate_logits (engine/src/nnue_engine.cpp:704-734) for B images at once -- what
 * evaluate_compiled_model obtains from one nnue_inference subprocess per image (evaluate.py:143-176,
 * engine/nnue_inference.cpp:42-60).  images: B flat buffers of 3*H*W floats, indexed HWC exactly as the engine
 * indexes them.  logits [B][classes], density [B] = active features / num_features.  Integer arithmetic
 * throughout: bit-identical to the engine.  Returns NNUE_E_SHAPE where the engine itself would write past its
 * grid buffer (conv map larger than grid x grid).  scratch >= nnue_engine_scratch(m, B) bytes. */
int64_t nnue_engine_scratch(const nnue_engine_model* m, int B);
int nnue_engine_evaluate_logits(const nnue_engine_model* m, const float* images, int B, int H, int W,
                                float* logits, float* density, void* scratch, int64_t scratch_bytes,
                                nnue_stream_t stream);

/* Data parallel for bandwidth-sized tables ("exchange only touched rows"; the reference itself is single-device,
 * train.py:263).  The table's weight gradient of the GLOBAL batch is d_W = A^T D with A the {0,1} map [world*B][P] and
 * D = d_ft [world*B][L1]: the ranks all-gather those FACTORS (the map as one bit per position) instead of reducing the
 * F x L1 product (1.5 MB per rank instead of 269 MB at the 224x224 configuration), and every rank runs the single-rank fused
 * path on them (nnue_ftm_gram_sqnorm, nnue_sgd_step, nnue_ftm_backward_weight_update with B = world*B): the gradient of the
 * mean loss over the global batch and clip_grad_norm_ on its global norm (train.py:359-366), identical on every rank.
 * One rank's chunk of the all-gather (byte offsets from nnue_dp_factor_offset, which = 0 d_ft, 1 sink, 2 small, 3 bits):
 *     d_ft [B][L1] f32 | sink [B] f32 | small [small_count] f32 | map bits [B][ceil(P/128)*16] (bit k of byte j = position 8j+k)
 * d_ft and sink are written in place by their producers (the trainer's buffers are views of its own chunk).
 * nnue_dp_factor_pack fills the bits from the byte map and "small" with every gradient the product does not cover:
 * grads[0, head_count) ++ grads[tail_lo, tail_lo + tail_count) of the flat gradient buffer (threshold and conv weight;
 * table rows >= direct, bias, classifier).  nnue_dp_factor_unpack takes the world gathered chunks (rank-major) and leaves
 * the global byte map g_bits [world*B][P], g_sink [world*B], g_dft [world*B][L1], and in grads the SUM of the ranks' small
 * parts in rank order (the all-reduce of the small gradients as a deterministic reduction: one collective per step).
 * P and L1 multiples of 4; all pointers 16-byte aligned. */
int64_t nnue_dp_factor_chunk_bytes(int B, int P, int L1, int64_t small_count);
int64_t nnue_dp_factor_offset(int which, int B, int P, int L1, int64_t small_count);
int nnue_dp_factor_pack(const uint8_t* bits, const float* grads, int64_t head_count, int64_t tail_lo, int64_t tail_count,
                        int B, int P, int L1, void* chunk, nnue_stream_t stream);
/* (train.py:359-366 over the global batch, see above) */
int nnue_dp_factor_unpack(const void* chunks, int world, int B, int P, int L1, int64_t head_count, int64_t tail_lo,
                          int64_t tail_count, uint8_t* g_bits, float* g_sink, float* g_dft, float* grads, nnue_stream_t stream);

/* The first stage of the clip norm below, alone: nparts block partials of sum g^2 (unscaled, fixed order) -- what a rank contributes
 * when the clip norm of clip_grad_norm_ (train.py:363-364) spans gradient shards held by different ranks: the partials are
 * all-gathered and handed to every rank's nnue_sgd_step as ext_partial over its whole shard. */
int nnue_sqnorm_partials(const float* grads, int64_t count, float* partial, int nparts, nnue_stream_t stream);

/* clip_grad_norm_ + SGD(momentum, weight_decay) on flat buffers (train.py:363-366, :457-464):
 *   g <- g * grad_scale              (1/world after a summed all-reduce)
 *   norm = ||g||_2 ; c = min(1, max_norm/(norm+1e-6)) if max_norm > 0 else 1
 *   g <- c*g + wd*p ; m <- first_step ? g : momentum*m + g ; p <- p - lr*m
 * norm_out (device float, may be NULL) receives the pre-clip norm.  Deterministic
 * two-stage norm; scratch >= nnue_sgd_scratch(count) bytes.
 * ste_partial != NULL: the second stage of a deferred nnue_ste_conv_backward (stages = 1; nnue.py:28-54) runs as extra
 * workgroups of the norm launch: d_thr[c] / d_weight[c*27+q] are summed from ste_partial (same order as stage 2, same
 * bits), written, and enter the norm.  ste_d_thr and ste_d_weight must lie in grads and together form its first
 * elements (a multiple of 4 of them); ste_fps * 28 <= 4096.  All five are NULL / 0 otherwise.
 * ext_partial != NULL: ext_count sums of squares (unscaled) that a producer formed for grads[ext_lo, ext_hi) (multiples of
 * 4, e.g. nnue_ftm_backward's sq_partial for the FeatureTransformer weight rows): the norm launch skips that range and
 * the partials enter the norm in index order, multiplied by grad_scale^2.  ext_count <= 65536.
 * coef_out != NULL: the clip coefficient c is also left in that device float.  ext_applied_elsewhere != 0 (needs
 * ext_partial and coef_out): grads[ext_lo, ext_hi) does not exist -- its producer applies the update itself afterwards
 * (nnue_ftm_backward_weight_update, reading coef_out) -- so this call neither reads that range of grads nor touches that
 * range of params / momentum_buf.
 * lr_dev != NULL (here, in nnue_adam_step and in nnue_ftm_backward_weight_update): the learning rate is read from that device
 * float instead of the `lr` argument -- an optimizer's lr changes between steps (a scheduler stepping param_groups, the
 * counterpart of train.py:457-471's optimizers) without re-recording or re-capturing a step that was captured into a graph. */
int64_t nnue_sgd_scratch(int64_t count);
int nnue_sgd_step(float* params, float* grads, float* momentum_buf, int64_t count,
                  float lr, float momentum, float weight_decay, float max_norm, float grad_scale,
                  int first_step, float* norm_out, void* scratch, int64_t scratch_bytes,
                  const float* ste_partial, int ste_chunks, int ste_fps,
                  float* ste_d_thr, float* ste_d_weight,
                  const float* ext_partial, int ext_count, int64_t ext_lo, int64_t ext_hi,
                  float* coef_out, int ext_applied_elsewhere, const float* lr_dev, nnue_stream_t stream);

/* ---- input pipeline ------------------------------------------------------------------------------
 * One batch of GenericVisionDataset.__getitem__ + collate (data/datasets.py:173-195, :358-372) from a uint8
 * dataset resident in HBM: images_u8 [N,H,W,3], labels_all [N]; indices [B] select the samples.  Writes
 * out [B,3,H,W] float32 = Normalize(ImageNet mean/std, max 255) of the (optionally augmented) pixels, CHW, and
 * labels_out [B].  augment != 0 applies the reference's "light" policy: HorizontalFlip p=0.5,
 * RandomBrightnessContrast(0.1, 0.1) p=0.2 (uint8 LUT semantics), CoarseDropout (one 5% x 5% hole, fill 0)
 * p=0.2.  Random draws are a counter-based hash of (seed, step, dataset index): reproducible, but not
 * albumentations' stream (parity unpinned for augment != 0; the plain path is an exact formula). */
int nnue_load_batch(const uint8_t* images_u8, const int64_t* labels_all, const int64_t* indices,
                    int B, int H, int W, int64_t N, int augment, uint64_t seed, uint64_t step,
                    float* out, int64_t* labels_out, nnue_stream_t stream);

/* clip_grad_norm_ + torch.optim.Adam(lr, weight_decay) on flat buffers -- the optimizer create_optimizer picks
 * when optimizer_type != "sgd" (train.py:363-366, :465-470; torch defaults betas (0.9, 0.999), eps 1e-8, L2
 * weight decay added to the gradient, no amsgrad):
 *   g <- clip * grad_scale * g + wd * p ; m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2
 *   p <- p - lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * t = ++step_counter[0] is kept in device memory so the call takes no per-step host argument (hipGraph replay).
 * scratch >= nnue_sgd_scratch(count) bytes; norm_out may be NULL. */
int nnue_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int32_t* step_counter,
                   int64_t count, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float max_norm, float grad_scale, float* norm_out,
                   void* scratch, int64_t scratch_bytes, const float* lr_dev, nnue_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NNUE_HIP_H */
