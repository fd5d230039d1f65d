/*
 * ta_ck.h -- C ABI of libta_ck.so: convolutions of a ResNet surrogate with the memory-bound pass that follows them folded into
 * the convolution's epilogue (MI355X / gfx950, fp32, NHWC).
 *
 * What it replaces.  The surrogate forward / input-gradient backward of the attack loop (transferattack/attack.py:104-122:
 * `self.model(x)`, `torch.autograd.grad(loss, delta)`) runs, for a ResNet with folded BatchNorm, one MIOpen convolution followed
 * by one streaming pass of libta_hip.so per layer (ta_bias_act, ta_bias_add_relu, ta_relu_mask: include/ta_hip.h).  The entry
 * points below compute convolution + that pass in ONE kernel: instances of composable_kernel's
 * DeviceGroupedConvFwdMultipleABD_Xdl_CShuffle (a library kernel MIOpen itself dispatches for these layers) with the pass as
 * CDE element-wise operation, in the tile configurations of CK's own fp32 instance lists.  The backward sites use the same
 * forward kernels: the input gradient of a stride-1 convolution IS a forward convolution of the output gradient with the
 * flipped, transposed filter (CK's backward-data kernels were measured slower at every site of ResNet-50 on MI355X).  Same rounding points as the two-kernel form: (acc + bias) first, then the shortcut, then the clamp.
 * The convolution's own accumulation order is the CK kernel's (as it is MIOpen's in the two-kernel form).
 *
 * Conventions: device pointers, fp32; activations NHWC ([n, h, w, c] = torch channels_last memory), weights KYXC ([k, y, x, c]);
 * square filters, equal strides / paddings on both axes, dilation 1, one group; `stream` = hipStream_t as void*; asynchronous,
 * no allocation, no zero fill; deterministic (no split-K, no atomics).  Return: 0 = launched, TA_CK_UNSUPPORTED = this configuration does not
 * take the problem (try another index), TA_CK_EINVAL = bad arguments (text in ta_ck_last_error()).
 */
#ifndef TA_CK_H
#define TA_CK_H

#ifdef __cplusplus
extern "C" {
#endif

#define TA_CK_ABI_VERSION 2
#define TA_CK_EINVAL (-1)
#define TA_CK_UNSUPPORTED 1

/* epilogues (`kind`)                                                                           replaces (include/ta_hip.h)   */
#define TA_CK_FWD_BIAS_RELU 1          /* e = clamp_min(acc + d0[k], 0)                           conv + ta_bias_act            */
#define TA_CK_FWD_BIAS_ADD_RELU 2      /* e = clamp_min((acc + d0[k]) + d1, 0)                    conv + ta_bias_add_relu       */
#define TA_CK_FWD_BIAS_ADD_BIAS_RELU 3 /* e = clamp_min((acc + d0[k]) + (d1 + d2[k]), 0)          ... with a projection shortcut */
/* A stride-1 convolution's input gradient as a forward convolution (w'[c][Y-1-y][X-1-x][k] = w[k][y][x][c], padding
 * ksize - 1 - pad): a = output gradient in the role of the input, w = w', e = input gradient; any square filter. */
#define TA_CK_FWD_MASK 4               /* e = d0 <= 0 ? 0 : acc                                   conv backward-data + ta_relu_mask */
#define TA_CK_FWD_ADD_MASK 5           /* e = d1 <= 0 ? 0 : acc + d0                              ... with the junction add     */

int ta_ck_abi_version(void);
const char* ta_ck_last_error(void);
/* number of tile configurations for this epilogue and filter geometry (0: none -- e.g. only 1x1 / stride 1 / no padding
 * filters have the shortcut and backward forms), and the name of configuration `index` */
int ta_ck_instances(int kind, int ksize, int stride, int pad);
const char* ta_ck_instance_name(int kind, int ksize, int stride, int pad, int index);
/* forward kinds:  a = input [n, hi, wi, c], w = weight [k, ksize, ksize, c], e = output [n, ho, wo, k];
 *                 d0 = bias [k]; d1 = shortcut [n, ho, wo, k]; d2 = the shortcut's bias [k]
 * TA_CK_FWD_MASK / _ADD_MASK: the forward geometry of the rewritten problem (c = the gradient's channels, k = the input's);
 *                 d0 = the activation in front of the convolution [n, ho, wo, k] (its ReLU's threshold) for _MASK;
 *                 d0 = the other addend of the residual junction, d1 = that activation for _ADD_MASK
 * unused d pointers are NULL.  ho = (hi + 2*pad - ksize) / stride + 1. */
int ta_ck_conv(int kind, int index, const float* a, const float* w, const float* d0, const float* d1, const float* d2, float* e,
               int n, int c, int hi, int wi, int k, int ksize, int stride, int pad, void* stream);

#ifdef __cplusplus
}
#endif
#endif
