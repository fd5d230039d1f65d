/*
 * TEST INFRASTRUCTURE -- plain-C restatement of the arithmetic of the reference's FGSM-family hot path.
 *
 * The reference is PyTorch code; on its CPU path every step below is an ATen kernel.  This file writes
 * the same arithmetic out in scalar C with explicit rounding points (fmaf where ATen's build contracts,
 * separate mul/add where it does not), so that the bits are defined independently of torch, of the
 * thread count and of the host ISA.  Each function cites the reference call site (file:line under
 * /root/reference) and the ATen op it restates.  Pinned bit-for-bit against torch 2.10 CPU (the op the
 * reference calls) by tests/test_oracle_golden.py; the HIP kernels are then compared with this file.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load the library built from it.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; fmaf is exact with or without hardware FMA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * ATen's CPU float sum over a contiguous row (aten/src/ATen/native/cpu/SumKernel.cpp, cascade_sum ->
 * vectorized_inner_sum -> row_sum -> multi_row_sum), as used by grad.abs().mean(dim=(1,2,3))
 * (transferattack/attack.py:128).  `lanes` = SIMD width of the build that ran the reference
 * (16 = AVX-512, 8 = AVX2); 4 interleaved accumulators per lane; 4 cascade levels.
 * ---------------------------------------------------------------------------------------------- */
static int ceil_log2_i64(int64_t x) {
    int l = 0;
    while (((int64_t)1 << l) < x) ++l;
    return l;
}

float ta_oracle_aten_row_sum(const float* row, int64_t size, int lanes) {
    enum { ILP = 4, LEVELS = 4, MAXCOL = 64 };
    const int cols = lanes * ILP;              /* independent accumulation columns */
    const int64_t steps = size / cols;         /* size_ilp of row_sum after vectorisation */
    float acc[LEVELS][MAXCOL];
    memset(acc, 0, sizeof(acc));
    int level_power = ceil_log2_i64(steps) / LEVELS;
    if (level_power < 4) level_power = 4;
    const int64_t level_step = (int64_t)1 << level_power;
    const int64_t level_mask = level_step - 1;
    int64_t i = 0;
    while (i + level_step <= steps) {
        for (int64_t j = 0; j < level_step; ++j, ++i)
            for (int c = 0; c < cols; ++c) acc[0][c] += row[i * cols + c];
        for (int j = 1; j < LEVELS; ++j) {
            for (int c = 0; c < cols; ++c) {
                acc[j][c] += acc[j - 1][c];
                acc[j - 1][c] = 0.0f;
            }
            const int64_t mask = level_mask << (j * level_power);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < steps; ++i)
        for (int c = 0; c < cols; ++c) acc[0][c] += row[i * cols + c];
    for (int j = 1; j < LEVELS; ++j)
        for (int c = 0; c < cols; ++c) acc[0][c] += acc[j][c];
    /* row_sum: vector elements beyond the last full ILP group go to accumulator 0, then fold the ILP
     * accumulators into accumulator 0 (lane-wise) */
    const int64_t vec_size = size / lanes;
    for (int64_t v = steps * ILP; v < vec_size; ++v)
        for (int l = 0; l < lanes; ++l) acc[0][l] += row[v * lanes + l];
    for (int k = 1; k < ILP; ++k)
        for (int l = 0; l < lanes; ++l) acc[0][l] += acc[0][k * lanes + l];
    /* vectorized_inner_sum: scalar tail first, then the lanes in order */
    float final_acc = 0.0f;
    for (int64_t k = vec_size * lanes; k < size; ++k) final_acc += row[k];
    for (int l = 0; l < lanes; ++l) final_acc += acc[0][l];
    return final_acc;
}

/* Attack.get_momentum, transferattack/attack.py:124-128:  m' = m*decay + g / (sum|g| / E)
 * m_in == NULL stands for the Python int 0 of the first iteration (attack.py:85). */
void ta_oracle_momentum(const float* g, const float* m_in, float* m_out, float decay, int64_t n, int64_t e,
                        int lanes) {
    float* a = (float*)malloc(sizeof(float) * (size_t)e);
    for (int64_t b = 0; b < n; ++b) {
        const float* gb = g + b * e;
        for (int64_t i = 0; i < e; ++i) a[i] = fabsf(gb[i]);
        const float mean = ta_oracle_aten_row_sum(a, e, lanes) / (float)e;   /* sum, then div_ by count */
        for (int64_t i = 0; i < e; ++i) {
            const float prev = m_in ? m_in[b * e + i] * decay : 0.0f * decay;
            m_out[b * e + i] = prev + gb[i] / mean;
        }
    }
    free(a);
}

static float sign_f(float m) { return (float)(m > 0.0f) - (float)(m < 0.0f); }   /* torch.sign: NaN -> 0 */

/* Attack.update_delta (linfty), transferattack/attack.py:145-153 + clamp utils.py:68-69.
 * alpha_t (nullable) = per-element step (gradient/gra.py:149). */
void ta_oracle_update_delta_linf(const float* delta_in, const float* x, const float* m, float alpha,
                                 const float* alpha_t, float eps, float* delta_out, int64_t numel) {
    for (int64_t i = 0; i < numel; ++i) {
        const float a = alpha_t ? alpha_t[i] : alpha;
        float d = delta_in[i] + a * sign_f(m[i]);
        d = fminf(fmaxf(d, -eps), eps);
        d = fmaxf(d, 0.0f - x[i]);
        d = fminf(d, 1.0f - x[i]);
        delta_out[i] = d;
    }
}

/* save_images, transferattack/utils.py:63-66 (after the add of main.py:53): NCHW -> NHWC uint8, truncation */
void ta_oracle_quantize_u8_nhwc(const float* x, const float* delta, uint8_t* out, int64_t n, int c, int h, int w) {
    const int64_t hw = (int64_t)h * w;
    for (int64_t b = 0; b < n; ++b)
        for (int64_t p = 0; p < hw; ++p)
            for (int ch = 0; ch < c; ++ch) {
                const int64_t i = (b * c + ch) * hw + p;
                out[(b * hw + p) * c + ch] = (uint8_t)((x[i] + delta[i]) * 255.0f);
            }
}

/* TIM.get_grad, transferattack/input_transformation/tim.py:72-74: F.conv2d(grad, k, padding='same', groups=C)
 * with one k x k kernel for every plane.  oneDNN's depthwise kernel = row-major FMA chain over the
 * zero-padded input. */
void ta_oracle_depthwise_conv2d_same(const float* in, float* out, const float* w, int k, int64_t planes, int h,
                                     int wd) {
    const int lo = (k - 1) / 2;
    for (int64_t p = 0; p < planes; ++p) {
        const float* ip = in + p * (int64_t)h * wd;
        float* op = out + p * (int64_t)h * wd;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < wd; ++x) {
                float acc = 0.0f;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        const int yy = y + ky - lo, xx = x + kx - lo;
                        const float v = (yy >= 0 && yy < h && xx >= 0 && xx < wd) ? ip[yy * wd + xx] : 0.0f;
                        acc = fmaf(w[ky * k + kx], v, acc);
                    }
                op[y * wd + x] = acc;
            }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Bilinear resampling, align_corners=False -- F.interpolate in DIM.transform,
 * transferattack/input_transformation/dim.py:55,68 (ATen upsample_bilinear2d, UpSampleKernel.cpp /
 * UpSample.h: area_pixel_compute_source_index, compute_source_index_and_lambda).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int i0, i1;
    float l0, l1;
} ta_tap;

void ta_oracle_bilinear_taps(int in_size, int out_size, ta_tap* taps) {
    const float scale = (float)in_size / (float)out_size;
    for (int o = 0; o < out_size; ++o) {
        float src = fmaf(scale, (float)o + 0.5f, -0.5f);
        if (src < 0.0f) src = 0.0f;
        int i0 = (int)src;
        if (i0 > in_size - 1) i0 = in_size - 1;
        const int i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
        float l1 = src - (float)i0;
        l1 = fminf(fmaxf(l1, 0.0f), 1.0f);
        taps[o].i0 = i0;
        taps[o].i1 = i1;
        taps[o].l0 = 1.0f - l1;
        taps[o].l1 = l1;
    }
}

/* forward: [planes, in, in] -> [planes, out, out]; width first then height, FMA as ATen's build contracts */
void ta_oracle_bilinear_fwd(const float* x, float* y, int64_t planes, int in_size, int out_size) {
    ta_tap* t = (ta_tap*)malloc(sizeof(ta_tap) * (size_t)out_size);
    ta_oracle_bilinear_taps(in_size, out_size, t);
    for (int64_t p = 0; p < planes; ++p) {
        const float* xp = x + p * (int64_t)in_size * in_size;
        float* yp = y + p * (int64_t)out_size * out_size;
        for (int oy = 0; oy < out_size; ++oy) {
            const float* r0 = xp + (int64_t)t[oy].i0 * in_size;
            const float* r1 = xp + (int64_t)t[oy].i1 * in_size;
            for (int ox = 0; ox < out_size; ++ox) {
                const float a = fmaf(t[ox].l0, r0[t[ox].i0], t[ox].l1 * r0[t[ox].i1]);
                const float b = fmaf(t[ox].l0, r1[t[ox].i0], t[ox].l1 * r1[t[ox].i1]);
                yp[oy * out_size + ox] = fmaf(t[oy].l0, a, t[oy].l1 * b);
            }
        }
    }
    free(t);
}

/* backward (adjoint): gy [planes,out,out] -> gx [planes,in,in]; scatter in output order, four updates per
 * output pixel, each  acc = fma(lh*lw, g, acc)  (cpu_upsample_linear_backward). mode 0: fma, 1: mul+add */
void ta_oracle_bilinear_bwd(const float* gy, float* gx, int64_t planes, int in_size, int out_size, int mode) {
    ta_tap* t = (ta_tap*)malloc(sizeof(ta_tap) * (size_t)out_size);
    ta_oracle_bilinear_taps(in_size, out_size, t);
    memset(gx, 0, sizeof(float) * (size_t)(planes * in_size * in_size));
    for (int64_t p = 0; p < planes; ++p) {
        float* gp = gx + p * (int64_t)in_size * in_size;
        const float* op = gy + p * (int64_t)out_size * out_size;
        for (int oy = 0; oy < out_size; ++oy)
            for (int ox = 0; ox < out_size; ++ox) {
                const float g = op[oy * out_size + ox];
                const int ys[2] = {t[oy].i0, t[oy].i1}, xs[2] = {t[ox].i0, t[ox].i1};
                const float ly[2] = {t[oy].l0, t[oy].l1}, lx[2] = {t[ox].l0, t[ox].l1};
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) {
                        float* dst = gp + (int64_t)ys[a] * in_size + xs[b];
                        const float wgt = ly[a] * lx[b];
                        *dst = mode == 0 ? fmaf(wgt, g, *dst) : *dst + wgt * g;
                    }
            }
    }
    free(t);
}

/* DIM.transform forward for one geometry (dim.py:55-68): resize -> zero pad -> resize back */
void ta_oracle_dim_fwd(const float* x, float* y, int64_t planes, int size, int resize, int rnd, int top, int left) {
    float* a = (float*)malloc(sizeof(float) * (size_t)(planes * rnd * rnd));
    float* b = (float*)calloc((size_t)(planes * resize * resize), sizeof(float));
    ta_oracle_bilinear_fwd(x, a, planes, size, rnd);
    for (int64_t p = 0; p < planes; ++p)
        for (int r = 0; r < rnd; ++r)
            memcpy(b + (p * resize + top + r) * (int64_t)resize + left, a + (p * rnd + r) * (int64_t)rnd,
                   sizeof(float) * (size_t)rnd);
    ta_oracle_bilinear_fwd(b, y, planes, resize, size);
    free(a);
    free(b);
}

void ta_oracle_dim_bwd(const float* gy, float* gx, int64_t planes, int size, int resize, int rnd, int top, int left,
                       int mode) {
    float* b = (float*)malloc(sizeof(float) * (size_t)(planes * resize * resize));
    float* a = (float*)malloc(sizeof(float) * (size_t)(planes * rnd * rnd));
    ta_oracle_bilinear_bwd(gy, b, planes, resize, size, mode);
    for (int64_t p = 0; p < planes; ++p)
        for (int r = 0; r < rnd; ++r)
            memcpy(a + (p * rnd + r) * (int64_t)rnd, b + (p * resize + top + r) * (int64_t)resize + left,
                   sizeof(float) * (size_t)rnd);
    ta_oracle_bilinear_bwd(a, gx, planes, size, rnd, mode);
    free(a);
    free(b);
}

/* Philox4x32-10 stream used by the in-kernel RNG of the HIP path (not in the reference, which uses the
 * device generator: vmifgsm.py:50, attack.py:134); restated so the stream itself can be checked. */
void ta_oracle_philox_uniform(float* out, int64_t numel, uint64_t seed, uint64_t offset, float r) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    for (int64_t q = 0; q * 4 < numel; ++q) {
        uint32_t c[4] = {(uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
        for (int round = 0; round < 10; ++round) {
            const uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
            const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
            c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
            k0 += W0; k1 += W1;
        }
        for (int j = 0; j < 4 && q * 4 + j < numel; ++j)
            out[q * 4 + j] = (float)(c[j] >> 8) * (1.0f / 16777216.0f) * (2.0f * r) - r;
    }
}

/* ------------------------------------------------------------------------------------------------
 * SIM / Admix copy stacks and their backward (input_transformation/sim.py:36-40, admix.py:40-45).
 * x / 2^i is an exact power-of-two scaling; the backward adds the slices in the order autograd's input buffer
 * receives them: last-created node first (i, then j, descending).
 * ---------------------------------------------------------------------------------------------- */
void ta_oracle_scale_copies_fwd(const float* x, float* y, int64_t ne, int num_scale) {
    float s = 1.0f;
    for (int c = 0; c < num_scale; ++c, s *= 0.5f)
        for (int64_t i = 0; i < ne; ++i) y[c * ne + i] = x[i] * s;
}

void ta_oracle_scale_copies_bwd(const float* gy, float* gx, int64_t ne, int num_scale) {
    for (int64_t i = 0; i < ne; ++i) {
        float s = ldexpf(1.0f, -(num_scale - 1));
        float acc = gy[(int64_t)(num_scale - 1) * ne + i] * s;
        for (int c = num_scale - 2; c >= 0; --c) {
            s *= 2.0f;
            acc += gy[(int64_t)c * ne + i] * s;
        }
        gx[i] = acc;
    }
}

void ta_oracle_admix_fwd(const float* x, const int64_t* perm, float* y, int64_t n, int64_t e, int num_admix,
                         int num_scale, float strength) {
    for (int c = 0; c < num_scale; ++c) {
        const float s = ldexpf(1.0f, -c);
        for (int j = 0; j < num_admix; ++j)
            for (int64_t b = 0; b < n; ++b) {
                const float* xb = x + b * e;
                const float* xp = x + perm[j * n + b] * e;
                float* yo = y + (((int64_t)c * num_admix + j) * n + b) * e;
                for (int64_t i = 0; i < e; ++i) yo[i] = (xb[i] + xp[i] * strength) * s;
            }
    }
}

void ta_oracle_admix_bwd(const float* gy, float* gx, int64_t n, int64_t e, int num_admix, int num_scale) {
    for (int64_t b = 0; b < n; ++b)
        for (int64_t i = 0; i < e; ++i) {
            float total = 0.0f;
            for (int j = num_admix - 1; j >= 0; --j) {
                float s = ldexpf(1.0f, -(num_scale - 1));
                float acc = gy[(((int64_t)(num_scale - 1) * num_admix + j) * n + b) * e + i] * s;
                for (int c = num_scale - 2; c >= 0; --c) {
                    s *= 2.0f;
                    acc += gy[(((int64_t)c * num_admix + j) * n + b) * e + i] * s;
                }
                total = (j == num_admix - 1) ? acc : total + acc;
            }
            gx[b * e + i] = total;
        }
}

/* PreprocessingModel's Normalize and its backward (transferattack/utils.py:72-79): two roundings each way */
void ta_oracle_normalize_fwd(const float* x, float* y, const float* mean, const float* stdv, int64_t n, int c,
                             int64_t hw) {
    for (int64_t b = 0; b < n; ++b)
        for (int ch = 0; ch < c; ++ch)
            for (int64_t i = 0; i < hw; ++i) {
                const int64_t k = (b * c + ch) * hw + i;
                y[k] = (x[k] - mean[ch]) / stdv[ch];
            }
}

void ta_oracle_normalize_bwd(const float* gy, float* gx, const float* stdv, int64_t n, int c, int64_t hw) {
    for (int64_t b = 0; b < n; ++b)
        for (int ch = 0; ch < c; ++ch)
            for (int64_t i = 0; i < hw; ++i) {
                const int64_t k = (b * c + ch) * hw + i;
                gx[k] = gy[k] / stdv[ch];
            }
}

/* VMIFGSM.get_variance finalisation (gradient/vmifgsm.py:58): acc / N - cur_grad */
void ta_oracle_variance_finalize(const float* acc, const float* cur, float* var, float count, int64_t numel) {
    for (int64_t i = 0; i < numel; ++i) var[i] = acc[i] / count - cur[i];
}

/* ------------------------------------------------------------------------------------------------
 * SIA block transform, transferattack/input_transformation/sia.py:41-100, from the int32 plan table the product draws
 * (per copy: rows[nb+1], cols[nb+1], then per rectangle -- rows outer -- op, roll step, scale bits):
 *   0 roll rows (x.roll(step, dims=2))   1 roll columns   2 flip rows   3 flip columns   4 rot90(k=2)
 *   5 torch.rand(1)[0] * x               6 torch.clip(x + noise, 0, 1)
 * x [planes][h][w]; y, noise, gy [copies][planes][h][w].  Backward: autograd's accumulation order over the copies is
 * last copy first (pinned by tests/golden/sia.npz); clip passes the gradient where 0 <= x + noise <= 1.
 * ---------------------------------------------------------------------------------------------- */
static void sia_cell(const int32_t* plan, int nb, int r, int c, int* r_lo, int* bh, int* c_lo, int* bw, int* op,
                     int* step, float* scale) {
    const int32_t* rows = plan;
    const int32_t* cols = plan + nb + 1;
    int bi = 0, bj = 0;
    while (bi + 1 < nb && r >= rows[bi + 1]) ++bi;
    while (bj + 1 < nb && c >= cols[bj + 1]) ++bj;
    const int32_t* blk = plan + 2 * (nb + 1) + 3 * (bi * nb + bj);
    *r_lo = rows[bi]; *bh = rows[bi + 1] - rows[bi];
    *c_lo = cols[bj]; *bw = cols[bj + 1] - cols[bj];
    *op = blk[0]; *step = blk[1];
    memcpy(scale, &blk[2], sizeof(float));
}

void ta_oracle_sia_fwd(const float* x, const int32_t* plan, const float* noise, float* y, int64_t planes, int h, int w,
                       int copies, int nb) {
    const int stride = 2 * (nb + 1) + 3 * nb * nb;
    for (int k = 0; k < copies; ++k)
        for (int64_t p = 0; p < planes; ++p)
            for (int r = 0; r < h; ++r)
                for (int c = 0; c < w; ++c) {
                    int r_lo, bh, c_lo, bw, op, step;
                    float scale;
                    sia_cell(plan + k * stride, nb, r, c, &r_lo, &bh, &c_lo, &bw, &op, &step, &scale);
                    int lr = r - r_lo, lc = c - c_lo;
                    if (op == 0) lr = ((lr - step) % bh + bh) % bh;            /* out[i] = in[(i - step) mod n] */
                    if (op == 1) lc = ((lc - step) % bw + bw) % bw;
                    if (op == 2 || op == 4) lr = bh - 1 - lr;
                    if (op == 3 || op == 4) lc = bw - 1 - lc;
                    float v = x[(p * h + r_lo + lr) * w + c_lo + lc];
                    const int64_t o = ((k * planes + p) * h + r) * (int64_t)w + c;
                    if (op == 5) v = scale * v;
                    if (op == 6) v = fminf(fmaxf(v + noise[o], 0.0f), 1.0f);
                    y[o] = v;
                }
}

void ta_oracle_sia_bwd(const float* gy, const int32_t* plan, const float* x, const float* noise, float* gx,
                       int64_t planes, int h, int w, int copies, int nb) {
    const int stride = 2 * (nb + 1) + 3 * nb * nb;
    for (int64_t p = 0; p < planes; ++p)
        for (int r = 0; r < h; ++r)
            for (int c = 0; c < w; ++c) {
                const int64_t here = (p * h + r) * (int64_t)w + c;
                float acc = 0.0f;
                for (int k = copies - 1; k >= 0; --k) {
                    int r_lo, bh, c_lo, bw, op, step;
                    float scale;
                    sia_cell(plan + k * stride, nb, r, c, &r_lo, &bh, &c_lo, &bw, &op, &step, &scale);
                    int lr = r - r_lo, lc = c - c_lo;
                    if (op == 0) lr = (lr + step) % bh;                        /* where in[i] went */
                    if (op == 1) lc = (lc + step) % bw;
                    if (op == 2 || op == 4) lr = bh - 1 - lr;
                    if (op == 3 || op == 4) lc = bw - 1 - lc;
                    const int64_t base = (k * planes + p) * (int64_t)h * w;
                    float g = gy[base + (int64_t)(r_lo + lr) * w + c_lo + lc];
                    if (op == 5) g = g * scale;
                    if (op == 6) {
                        const float v = x[here] + noise[base + (int64_t)r * w + c];
                        g = (v >= 0.0f && v <= 1.0f) ? g : 0.0f;
                    }
                    acc = (k == copies - 1) ? g : acc + g;
                }
                gx[here] = acc;
            }
}
