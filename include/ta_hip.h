/*
 * ta_hip.h -- C ABI of libta_hip.so: the MI355X (gfx950) kernels behind the iterative FGSM-family
 * hot path of Trustworthy-AI-Group/TransferAttack.
 *
 * The reference has no FFI: the path is Python calling ATen.  Each entry point below replaces the
 * string of ATen ops issued by ONE reference hook (cited as file:line under /root/reference); the
 * host-side mirror of the reference's plug-in API (transferattack_amd/attack.py etc.) binds them with
 * ctypes (transferattack_amd/_hip.py) -- see INTEGRATION.md for the stub a reference maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 NCHW data unless stated otherwise;
 *   - n = images in the batch, e = elements per image (C*H*W); planes = N*C;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all calls are asynchronous
 *     on it, never synchronise, never allocate -> safe inside hipGraph capture;
 *   - return value: 0 on success, otherwise a hipError_t (>0) or TA_EINVAL (-1); the message of the
 *     last failure on the calling thread is returned by ta_last_error();
 *   - results are deterministic: no floating-point atomics anywhere, reduction order is fixed.
 */
#ifndef TA_HIP_H
#define TA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TA_ABI_VERSION 13
#define TA_EINVAL (-1)

int ta_abi_version(void);
const char* ta_last_error(void);
/* Order of the per-image sum of |g| (transferattack/attack.py:128: grad.abs().mean(dim=(1,2,3))).  0 (the default): the
 * kernels' own fixed order.  8 / 16: ATen's vectorised cascade for a CPU of that SIMD width (AVX2 / AVX-512) -- the
 * verification mode in which momentum, delta and whole loops carry the reference's bits; K1 then runs as one workgroup per
 * image, producer-side sums must not be handed in (ws_slots = 0), ta_bsr_bwd adds in ATen's visiting order.  Process-wide,
 * read by every later call (the Python binding sets it from TA_ATEN_SUM_LANES; the library itself reads no environment).
 * Returns TA_EINVAL for any other value. */
int ta_set_sum_order(int lanes);
int ta_get_sum_order(void);
/* Launch timing (bench.py's roofline figure): between ta_timing_begin(capacity) and ta_timing_end, each ta_mi_update
 * call (up to `capacity`) carries a pair of HIP events on the dispatch packets of its own kernels -- start on its first
 * kernel (K1 when the |g| sums are not handed over, else the update kernel), stop on the update kernel.  ta_timing_end
 * waits for them and writes begin->end milliseconds per call, in call order; *count = calls timed.  Not for use inside
 * hipGraph capture.  One timing session per process at a time. */
int ta_timing_begin(int capacity);
int ta_timing_end(float* ms, int capacity, int* count);
/* number of floats of scratch the l1-mean reduction needs for an (n, e) batch (>= n*ceil(e/3072)) */
int64_t ta_l1_workspace_floats(int64_t n, int64_t e);
/* |g| tile sums ("partials"): K1 and every elementwise producer below cut an image of e elements into
 * ta_update_tiles(e) = ceil(e/3072) tiles and write one fp32 sum of |.| per tile, ws[img*S + tile].  The tile kernels
 * (TIM convolution, DIM backward) write one sum per workgroup tile: ws[plane*T + tile], T = ta_conv_tiles(k, h, w) /
 * ta_dim_bwd_tiles(size, resize) / ta_bsr_tiles(h), i.e. C*T consecutive sums per image.  ta_mi_update takes either layout through
 * `ws_slots` = sums per image.  All sums are in a fixed order (no atomics). */
int64_t ta_update_tiles(int64_t e);
int64_t ta_conv_tiles(int k, int h, int w);
int64_t ta_dim_bwd_tiles(int size, int resize);

/* ---- update stack ------------------------------------------------------------------------------
 * Attack.get_momentum  transferattack/attack.py:124-128   m <- m*decay + g / mean_{CHW}|g|
 * Attack.update_delta  transferattack/attack.py:145-153   d <- clamp(d + a*sign(m), -eps, eps);
 *                                                          d <- min(max(d, 0-x), 1-x)  (utils.py:68-69)
 * `v` (nullable) is added to g first (VMI-FGSM: get_momentum(grad+variance, ...), vmifgsm.py:89).
 * `m_in` NULL means the Python int 0 of the first iteration (attack.py:85).
 */
/* K1: per-image partial sums of |g (+v)|; ws[n*S + s], S = ceil(e/3072); fixed summation order */
int ta_abs_sum_partials(const float* g, const float* v, float* ws, int64_t n, int64_t e, void* stream);
/* get_momentum alone (hook-compatible path): K1 + normalise-accumulate; m_out may alias m_in */
int ta_momentum(const float* g, const float* v, const float* m_in, float* m_out, float* ws,
                float decay, int64_t n, int64_t e, void* stream);
/* update_delta alone, L-inf branch; alpha_t (nullable) is a per-element step (gra.py:149), else the
 * scalar alpha (may be negative, cwa.py:69); delta_out may alias delta_in; x_adv (nullable) <- x+d */
int ta_update_delta_linf(const float* delta_in, const float* x, const float* m, float alpha,
                         const float* alpha_t, float eps, float* delta_out, float* x_adv,
                         int64_t numel, void* stream);
/* update_delta, L2 branch (attack.py:148-151): d <- renorm_2(d + alpha*g/(|g|_2+1e-20), eps), box */
int ta_update_delta_l2(const float* delta_in, const float* x, const float* g, float alpha, float eps,
                       float* delta_out, float* ws, int64_t n, int64_t e, void* stream);
/* fused get_momentum + update_delta (the headline kernel): reads g,(v),m,d,x  writes m,d,(x_adv).
 * 24 B/element algorithmic traffic; 20 on the first iteration (m_in==NULL); 16 when m_in==NULL && m_out==NULL (the
 * decay=0 / FGSM / I-FGSM case: the momentum is never stored); +4 when x_adv (= x + d', the next iteration's input,
 * attack.py:88) is written.
 * ws_slots == 0: K1 runs first (ws = scratch of ta_l1_workspace_floats(n, e) floats).
 * ws_slots  > 0: ws already holds ws_slots sums of |g| (|g + v| when v != NULL) per image, written by the kernel that produced g
 * (ta_normalize_bwd, ta_depthwise_conv2d_same, ta_dim_bwd, ta_scale_copies_bwd, ta_admix_bwd, ta_sum_copies_bwd,
 * ta_sum_members, ta_bsr_bwd), so the K1 pass over g is skipped and g is read exactly once. */
int ta_mi_update(const float* g, const float* v, const float* m_in, float* m_out, float* delta,
                 const float* x, float* x_adv, float* ws, int ws_slots, float decay, float alpha,
                 float eps, int64_t n, int64_t e, void* stream);
/* PreprocessingModel WITH a Resize (transferattack/utils.py:50-53: Inception-v3 takes 299 x 299, mean = std = 0.5;
 * utils.py:72-79): y = (bilinear_{in->out}(x) - mean[c]) / std[c] over [n, c, in, in] -> [n, c, out, out], ATen's
 * upsample_bilinear2d arithmetic (align_corners=False; width first, then height), and its backward
 * gx = upsample_bilinear2d_backward(gy / std[c]) in ATen's accumulation order -- one kernel each way.  The backward
 * writes n * c * ta_resize_tiles(in) sums of |gx| to ws (nullable): c * tiles consecutive per image, a layout
 * ta_mi_update takes.  Backward needs out <= ~1.5 * in (at most 4 outputs per input index), else TA_EINVAL. */
int64_t ta_resize_tiles(int side);
int ta_resize_normalize_fwd(const float* x, float* y, const float* mean, const float* stdv, int64_t n, int c,
                            int in_size, int out_size, void* stream);
int ta_resize_normalize_bwd(const float* gy, float* gx, const float* stdv, float* ws, int64_t n, int c, int in_size,
                            int out_size, void* stream);
/* The byte source of the fused update.  The images of this path are PNG-decoded (transferattack/utils.py:136:
 * `image.astype(np.float32) / 255`), so x[i] == float(k) / 255 for a byte k.  ta_u8_source_probe writes k =
 * round(x * 255) for every element to x_u8 and sets *mismatch (device int, zeroed by the call) to 1 if any element is NOT
 * reproduced bit for bit by that division -- once per batch, asynchronous.  ta_mi_update_u8 is ta_mi_update with that
 * pair: the kernel reads *mismatch itself and takes 1 B/element from x_u8 when it is 0 (x rebuilt with the division's
 * bits; 21 (+4) B/element instead of 24 (+4)), the fp32 x otherwise -- same results either way, no host round trip,
 * hipGraph-capturable.  Needs e % 4 == 0 and 16-byte aligned fp32 operands to use the bytes. */
int ta_u8_source_probe(const float* x, uint8_t* x_u8, int* mismatch, int64_t numel, void* stream);
int ta_mi_update_u8(const float* g, const float* v, const float* m_in, float* m_out, float* delta,
                    const float* x, const uint8_t* x_u8, const int* u8_mismatch, float* x_adv, float* ws,
                    int ws_slots, float decay, float alpha, float eps, int64_t n, int64_t e, void* stream);
/* The surrogate's Normalize folded into both ends of an iteration (round 5) -- for the loops in which nothing but
 * PreprocessingModel's Normalize (transferattack/utils.py:72-79) sits between `data + delta` (attack.py:88) and the backbone:
 *   ta_normalize_adv_fwd     y = ((x + d) - mean[c]) / std[c]: the add of attack.py:88 and the Normalize in one pass, x taken
 *                            from the byte source when (x_u8, u8_mismatch) are given and the flag is 0 (both nullable together);
 *                            the fused update then has no x + d' to store: 24 B/element algorithmic, 21 executed.
 *   ta_mi_update_std         ta_mi_update[_u8] whose gradient operand is gy = d(loss)/d(normalised input), the backbone's own
 *                            output; the kernel forms attack.py:118-122's gradient gy / std[c] inline (the division of
 *                            Normalize's backward), so no pass stores it.  ws_slots > 0: ws holds ws_slots sums of
 *                            |gy / std[c]| per image (ta_stem7s2_input_grad leaves them); ws_slots == 0: a sum-only pass
 *                            over gy runs first (ta_abs_sum_partials_std; ws = scratch as for ta_mi_update).  e = c * hw.
 *                            Same rounding points as ta_normalize_bwd + ta_mi_update: same bits.
 *   ta_abs_sum_partials_std  K1 over gy / std[c]: ta_normalize_bwd's sums without its store -- bit for bit when hw % 4 == 0 (every
 *                            image plane this path sees; a plane size that is not a multiple of 4 takes the scalar form, whose
 *                            fixed summation order is another one: equal to fp32 summation error). */
int ta_normalize_adv_fwd(const float* x, const uint8_t* x_u8, const int* u8_mismatch, const float* delta, float* y,
                         const float* mean, const float* stdv, int64_t n, int c, int64_t hw, void* stream);
/* the same pass writing y in NHWC memory ([n][hw][c]: what a channels_last surrogate's first convolution reads -- no layout copy
 * in front of it); three-channel images, hw % 4 == 0, 16-byte aligned operands, TA_EINVAL otherwise.  Same bits per element. */
int ta_normalize_adv_fwd_nhwc(const float* x, const uint8_t* x_u8, const int* u8_mismatch, const float* delta, float* y,
                         const float* mean, const float* stdv, int64_t n, int c, int64_t hw, void* stream);
int ta_mi_update_std(const float* gy, const float* stdv, const float* m_in, float* m_out, float* delta, const float* x,
                     const uint8_t* x_u8, const int* u8_mismatch, float* ws, int ws_slots, float decay, float alpha,
                     float eps, int64_t n, int c, int64_t hw, void* stream);
int ta_abs_sum_partials_std(const float* gy, const float* stdv, float* ws, int64_t n, int c, int64_t hw, void* stream);
/* PreprocessingModel's Normalize (transferattack/utils.py:72-79, torchvision Normalize): y = (x - mean[c]) / std[c]
 * over [n, c, hw]; mean/std: device fp32 [c].  Backward gx = gy / std[c] (the last kernel of the surrogate's
 * backward, i.e. the producer of the gradient) also writes the |gx| tile sums to ws in K1's layout. */
int ta_normalize_fwd(const float* x, float* y, const float* mean, const float* stdv, int64_t n, int c,
                     int64_t hw, void* stream);
/* v (nullable): the tile sums are of |gx + v| instead -- VMI-FGSM normalises grad + variance (vmifgsm.py:89), and
 * ta_mi_update / ta_momentum with that v then take them as they take the plain sums */
int ta_normalize_bwd(const float* gy, float* gx, const float* stdv, const float* v, float* ws, int64_t n, int c, int64_t hw,
                     void* stream);
/* Attack.init_delta random start (attack.py:133-141, linfty): d <- box(U(-eps,eps)); counter-based
 * Philox4x32-10 keyed by (seed, offset); noise (nullable) overrides the draw with caller noise */
int ta_init_delta_uniform(float* delta, const float* x, const float* noise, float eps, uint64_t seed,
                          uint64_t offset, int64_t numel, void* stream);

/* ---- TIM: TIM.get_grad  input_transformation/tim.py:72-74 ------------------------------------------
 * depthwise k x k 'same' zero-padded correlation of every (n,c) plane with ONE k x k kernel `w`
 * (device pointer, k*k fp32, row-major).  Tap order is row-major FMA chain == the reference CPU path.
 * k <= 31; in and out must not alias.  ws (nullable): |out| sums, ta_conv_tiles(k, h, w_) per plane (TIM.get_grad is the
 * last kernel that writes the gradient the update consumes).
 */
int ta_depthwise_conv2d_same(const float* in, float* out, const float* w, float* ws, int k, int64_t planes,
                             int h, int w_, void* stream);
/* ---- DIM: DIM.transform  input_transformation/dim.py:42-68 ----------------------------------------
 * y = bilinear(pad0(bilinear(x, rnd), resize, top, left), size); one geometry for the whole call.
 * fwd: x[planes,size,size] -> y[planes,size,size];  bwd: gy -> gx (exact adjoint, gather form);
 * ws (nullable): |gx| sums, ta_dim_bwd_tiles(size, resize) per plane.
 */
int ta_dim_fwd(const float* x, float* y, int64_t planes, int size, int resize, int rnd, int top,
               int left, void* stream);
int ta_dim_bwd(const float* gy, float* gx, float* ws, int64_t planes, int size, int resize, int rnd, int top,
               int left, void* stream);

/* ---- SIM / Admix: sim.py:36-40, admix.py:40-45 ------------------------------------------------------
 * sim fwd : y[i*n + b] = x[b] / 2^i                              i < num_scale
 * sim bwd : gx[b] = sum_i gy[i*n + b] / 2^i                      (i descending = autograd's order)
 * admix fwd: y[(i*num_admix + j)*n + b] = (x[b] + strength * x[perm[j*n + b]]) / 2^i
 * admix bwd: gx[b] = sum_j (sum_i gy[(i*num_admix + j)*n + b] / 2^i)  j, i descending (detached mix term)
 * perm: device int64 [num_admix*n].  ws (nullable) of the backward kernels: |gx| sums, ta_update_tiles(e) per image.
 */
int ta_scale_copies_fwd(const float* x, float* y, int64_t n, int64_t e, int num_scale, void* stream);
int ta_scale_copies_bwd(const float* gy, float* gx, float* ws, int64_t n, int64_t e, int num_scale, void* stream);
/* gx[b] = sum_i gy[i*n + b], i descending: backward of EMI-FGSM's stack of x + c_i*alpha*g_bar (emifgsm.py:57-58) */
int ta_sum_copies_bwd(const float* gy, float* gx, float* ws, int64_t n, int64_t e, int copies, void* stream);
int ta_admix_fwd(const float* x, const int64_t* perm, float* y, int64_t n, int64_t e, int num_admix,
                 int num_scale, float strength, void* stream);
int ta_admix_bwd(const float* gy, float* gx, float* ws, int64_t n, int64_t e, int num_admix, int num_scale,
                 void* stream);
/* EnsembleModel.forward feeds the same x to every member (utils.py:98-99); autograd then adds the members' input
 * gradients as they arrive, last member first: gx = ((g[m-1] + g[m-2]) + ...) + g[0].  One kernel instead of m-1 ATen
 * adds; `gs`: HOST array of m <= 8 device pointers, each [n, e]; ws (nullable): |gx| sums, ta_update_tiles(e) per image. */
int ta_sum_members(const float* const* gs, int m, float* gx, float* ws, int64_t n, int64_t e, void* stream);

/* ---- SIA block transform: SIA.blocktransform / transform  input_transformation/sia.py:41-100 -------------
 * Every copy of the batch is cut into nb x nb rectangles; each gets one operation: 0 roll rows, 1 roll columns,
 * 2 flip rows, 3 flip columns, 4 rotate 180, 5 multiply by a scalar, 6 add U(-noise_radius, noise_radius) and clip to
 * [0, 1].  `plan` (device int32) holds, per copy: rows[nb+1], cols[nb+1] (cut positions incl. 0 and h / w), then per
 * rectangle (rows outer) op, roll step, scale factor (float bits) -- drawn on the host in the reference's order.
 * `planes` = N*C; y / gy are [copies][planes][h][w].  `noise` (nullable, same shape as y; only the op-6 rectangles
 * are read) replaces the in-kernel Philox (seed, offset) stream.
 * fwd: y[k] = blockwise op(x).   bwd: gx = sum_k (k descending = autograd's order) of the gradient routed back
 * through copy k's permutation / scale / clip mask (0 <= x + noise <= 1); needs x and the same noise as the forward. */
int ta_sia_fwd(const float* x, const int32_t* plan, const float* noise, float* y, int64_t planes, int h, int w,
               int copies, int nb, float noise_radius, uint64_t seed, uint64_t offset, void* stream);
int ta_sia_bwd(const float* gy, const int32_t* plan, const float* x, const float* noise, float* gx,
               int64_t planes, int h, int w, int copies, int nb, float noise_radius, uint64_t seed,
               uint64_t offset, void* stream);

/* ---- BSR block shuffle + rotation: BSR.shuffle / transform  input_transformation/bsr.py:41-67 --------------
 * Every copy of the batch is cut into nb strips along one axis (lengths drawn on the host), the strips are shuffled, each
 * strip is rotated about its centre (torchvision RandomRotation(+-24 deg, BILINEAR): affine grid + grid_sample with zero
 * fill, align_corners=False), cut into nb blocks along the other axis and shuffled again.  `plan` (device int32, per copy
 * 1 + 7*nb + 3*nb*nb words, strips / blocks in OUTPUT order): first axis (0 rows, 1 columns); per strip: src_start,
 * length, out_start and the four entries of theta^T / (w/2, h/2) as float bits; per block: src_start, length, out_start.
 * `planes` = N*C; y / gy are [copies][planes][h][w].
 * fwd: one gather kernel.  bwd: gx = sum over the copies (descending = autograd's order) of the exact adjoint, gather
 * form, no atomics; ws (nullable): |gx| sums, ta_bsr_tiles(h) per plane. */
int64_t ta_bsr_tiles(int h);
int ta_bsr_fwd(const float* x, const int32_t* plan, float* y, int64_t planes, int h, int w, int copies, int nb,
               void* stream);
int ta_bsr_bwd(const float* gy, const int32_t* plan, float* gx, float* ws, int64_t planes, int h, int w, int copies,
               int nb, void* stream);

/* ---- SSM / FGSRA spectrum transform: SSM.transform  input_transformation/ssm.py:41-54 (dct_2d / idct_2d :101-209),
 * FGSRA neighbour sampling gradient/fgsra.py:125-140 ------------------------------------------------------------------
 * out[p] = ( L . (in[p] + add[p]) . R^T ) * mul[p]  for every n x n plane p (n a multiple of 32, <= 256); `add` and
 * `mul` nullable; L, R: row-major n x n fp32 on the device.  With C the DCT-II matrix (C[k][m] = 2 cos(pi (2m+1) k / 2n))
 * and D = C^-1:  y = IDCT2(DCT2(x + noise) * mask) = ta_dct_pair(ta_dct_pair(x, noise, mask; C, C), -, -; D, D), and its
 * backward is the same two launches with D^T and C^T.  fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32, fixed k order). */
int ta_dct_pair(const float* in, const float* add, const float* mul, float* out, const float* lmat, const float* rmat,
                int64_t planes, int n, void* stream);

/* ---- VMI-FGSM: VMIFGSM.get_variance  gradient/vmifgsm.py:42-58 --------------------------------------
 * neighbour: out = x + d + U(-radius, radius)   (Philox (seed, offset) or caller `noise`)
 * accumulate: acc (+)= g  (first!=0 -> acc = g);  finalize: var = acc / count - cur_grad
 */
int ta_vmi_neighbor(const float* x, const float* delta, const float* noise, float* out, float radius,
                    uint64_t seed, uint64_t offset, int64_t numel, void* stream);
int ta_grad_accumulate(float* acc, const float* g, int first, int64_t numel, void* stream);
/* the same chain with the surrogate's Normalize (utils.py:72-79) folded into both ends -- 24 instead of 40 B/element per
 * neighbour: y = (((x + d) + noise) - mean[c]) / std[c] feeds the backbone directly, and the backbone's input gradient gy goes
 * straight into the accumulator, acc (+)= gy / std[c].  Same rounding points as the separate kernels: same bits. */
int ta_vmi_neighbor_normalized(const float* x, const float* delta, const float* noise, float* y, const float* mean,
                               const float* stdv, float radius, uint64_t seed, uint64_t offset, int64_t n, int c, int64_t hw,
                               void* stream);
int ta_normalize_bwd_accumulate(const float* gy, float* acc, const float* stdv, int first, int64_t n, int c, int64_t hw,
                                void* stream);
int ta_variance_finalize(const float* acc, const float* cur_grad, float* var, float count,
                         int64_t numel, void* stream);

/* ---- NI look-ahead: NIFGSM.transform  gradient/nifgsm.py:35-39:  out = x + (alpha*decay) * m ------ */
int ta_axpy(const float* x, const float* m, float coeff, float* out, int64_t numel, void* stream);

/* ---- surrogate glue: the memory-bound passes between a convolutional surrogate's MIOpen convolutions, fused -----------
 * (what torch runs per convolution of a ResNet with folded BatchNorm: bias add, ReLU clamp, residual add; in the backward
 * threshold_backward and the junction add -- transferattack/attack.py:104-122 evaluates that surrogate 10 x per batch).
 * Contiguous fp32 buffers of `numel` elements (numel % 4 == 0, 16-byte aligned); element i has channel (i / inner) % channels:
 * inner = 1 for NHWC, H*W for NCHW (fastest when channels % 4 == 0 resp. H*W % 4 == 0).  Same rounding points as the separate ATen passes.
 *   ta_bias_act        y = y + bias[c], then clamp_min(., 0) if relu                       (in place)
 *   ta_bias_add_relu   y = clamp_min((y + bias[c]) + (other [+ bias_other[c]]), 0)           (in place; bias_other nullable)
 *   ta_relu_mask       out = y <= 0 ? 0 : ga [+ gb]          (gb nullable; out may alias ga) = threshold_backward(ga + gb, y, 0)
 * Pass bits (round 4): `mask` (nullable; numel / 8 bytes, numel % 8 == 0) of the two forward kernels receives one bit per
 * element of the result -- bit i % 8 of byte i / 8, set where !(y <= 0), i.e. where threshold_backward lets the gradient
 * pass -- and ta_relu_mask takes EITHER the activation y OR those bits: the backward of a frozen network reads an
 * activation for nothing but that test, 4 bytes per element to learn one bit. */
int ta_bias_act(float* y, const float* bias, int relu, uint8_t* mask, int64_t numel, int channels, int64_t inner, void* stream);
int ta_bias_add_relu(float* y, const float* bias, const float* other, const float* bias_other, uint8_t* mask, int64_t numel,
                     int channels, int64_t inner, void* stream);
int ta_relu_mask(const float* ga, const float* gb, const float* y, const uint8_t* mask, float* out, int64_t numel, void* stream);
/*   ta_maxpool_bwd_relu  out = threshold_backward(max_pool2d_with_indices_backward(ga [+ gb], idx), y, 0) in ONE gather pass (the
 *                        stem of the ResNets: conv -> ReLU -> max-pool; no zero fill, no atomics, deterministic).  channels_last
 *                        buffers: ga / gb / idx [n, ph, pw, c], y / out [n, h, w, c]; idx = ATen's argmax index h * w_ + w. */
int ta_maxpool_bwd_relu(const float* ga, const float* gb, const int64_t* idx, const float* y, float* out, int64_t n, int channels,
                        int h, int w, int ph, int pw, int k, int s, int p, void* stream);
/*   ta_maxpool3s2_fwd / ta_maxpool3s2_bwd_relu (ABI 13)  the same pair for THE stem pool -- 3 x 3, stride 2, padding 1, an even
 *                        map, channels % 8 == 0 -- with the forward on this side too (replaces F.max_pool2d(..., return_indices=True)
 *                        = at::max_pool2d_with_indices, the pooling behind torchvision's resnet.py maxpool): it leaves
 *                          pooled [n, h/2, w/2, c]   the values max_pool2d gives (ATen's scan order and update rule, NaN included);
 *                          arg    [n, h/2, w/2, c]   one byte: the winning tap kh * 3 + kw (ATen: an int64 h * w_ + w per element);
 *                          mask   n*h*w*c / 8 bytes  the pass bits of the pooled ACTIVATION y (layout as above),
 *                        and the backward takes arg + mask instead of idx + y: 1 1/8 byte where it read 12 per element. */
int ta_maxpool3s2_fwd(const float* y, float* pooled, uint8_t* arg, uint8_t* mask, int64_t n, int channels, int h, int w, void* stream);
int ta_maxpool3s2_bwd_relu(const float* ga, const float* gb, const uint8_t* arg, const uint8_t* mask, float* out, int64_t n,
                           int channels, int h, int w, void* stream);

/* ---- the stem convolution's input gradient (7 x 7, stride 2, padding 3, 3 -> 64 channels: ResNet / ImageNet CNN stems) ----
 * The last convolution of the surrogate's backward, producer of the gradient the update consumes (attack.py:118-122).
 * dx[n,c,2i+py,2j+px] = sum_{a,b<4} sum_k dy[n,i-1+a,j-1+b,k] * w[k,c,5-2a+py,5-2b+px] on the fp32 matrix cores, deterministic.
 *   ta_stem7s2_prepare      w [64,3,7,7] (dense NCHW) -> w2 (16384 floats, device): once per model
 *   ta_stem7s2_input_grad   dy channels_last [n, oh, ow, 64] -> dx NCHW [n, 3, 2*oh, 2*ow] */
int ta_stem7s2_prepare(const float* w, float* w2, void* stream);
/* stdv / ws (nullable together): the kernel also writes n * ta_stem_tiles(oh, ow) sums of |dx / stdv[c]| to ws -- one per
 * workgroup, ta_stem_tiles consecutive per image, fixed order -- the layout ta_mi_update_std takes through `ws_slots` */
int64_t ta_stem_tiles(int oh, int ow);
int ta_stem7s2_input_grad(const float* dy, const float* w2, float* dx, const float* stdv, float* ws, int64_t n, int oh, int ow,
                          void* stream);
/* the same for dy in NCHW memory [n, 64, oh, ow] -- what autograd hands over on the plain module path of an NCHW surrogate
 * (the reference-literal arrangement): only the staging of the dy window differs, same MFMA schedule, same bits as the
 * channels_last form on the same values */
int ta_stem7s2_input_grad_nchw(const float* dy, const float* w2, float* dx, const float* stdv, float* ws, int64_t n, int oh, int ow,
                          void* stream);

/* ---- output: save_images  transferattack/utils.py:63-66 (+ main.py:53 add) ---------------------------
 * u8[n,h,w,c] = trunc((x + d) * 255)   NCHW fp32 -> NHWC uint8 */
int ta_quantize_u8_nhwc(const float* x, const float* delta, uint8_t* out, int64_t n, int c, int h,
                        int w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TA_HIP_H */
