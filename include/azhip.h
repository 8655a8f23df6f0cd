/* azhip.h -- C ABI of libazhip.so: the MI355X (gfx950) hot path of ActiveZero's
 * PSMNet cost volume + 3-D aggregation + soft-argmin, and the warp /
 * reprojection operators.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer to contiguous fp32 (or int32 / uint8
 *     where said) memory owned by the caller; nothing here allocates, frees,
 *     synchronises or throws;
 *   - `stream` is a hipStream_t passed as void* (0 = null stream); kernels are
 *     only enqueued;
 *   - return value: AZ_OK (0) or a negative AZ_E* code; az_strerror() names it;
 *   - scratch memory, where needed, is passed as (workspace, workspace_bytes)
 *     and sized by the matching az_*_workspace() query;
 *   - layouts: "NCHW"/"NCDHW" are the reference's PyTorch layouts; "NDHWC" is
 *     the channels-last layout the 3-D aggregation kernels use internally.
 *
 * Each entry cites the reference interface it replaces (paths relative to the
 * reference repo root).
 */
#ifndef AZHIP_H
#define AZHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_OK 0
#define AZ_EINVAL (-1)       /* bad dimension / flag */
#define AZ_ENULL (-2)        /* null pointer */
#define AZ_ELAUNCH (-3)      /* hipLaunch failed (hipGetLastError) */
#define AZ_EUNSUPPORTED (-4) /* shape outside the compiled kernel set */
#define AZ_EWORKSPACE (-5)   /* workspace too small */

const char *az_strerror(int code);
/* ABI version: bumped when a signature changes.  A host binding compares az_abi_version() with the AZ_ABI_VERSION it
 * was written against and refuses a library that answers anything else (activezero_amd/_lib.py does). */
#define AZ_ABI_VERSION 6
int az_abi_version(void);
/* the value of one A/B switch as the library read it at its first use (name = the environment variable, e.g.
 * "AZ_WGRAD_R16"; DESIGN.md lists them): switches are read once per process into one immutable struct.
 * AZ_EINVAL: no such switch. */
int az_option(const char *name);
/* measurement only: dst[0..n) = src[0..n) as a float4 grid-stride stream (n % 4 == 0, 16-byte aligned) -- the copy rate
 * bench.py reports beside the HBM spec figure */
int az_hbm_copy_probe(float *dst, const float *src, long long n, void *stream);

/* ---- K1/K2: integer scatter warp ------------------------------------------
 * replaces utils/warp_ops.py:22-45 (CUDA-C apply_disparity_pos/_neg) and the
 * cupy launch at utils/warp_ops.py:80-93.
 * dst,src: [N,C,H,W] f32; disp: [N,H,W] i32 (shared by the C channels).
 * sign > 0: all disp >= 0, smallest source j wins a collision;
 * sign < 0: all disp <= 0, largest source j wins.  dst is fully written
 * (holes = 0); the caller need not pre-zero it. */
int az_warp_scatter(float *dst, const float *src, const int32_t *disp, int N, int C, int H,
                    int W, int sign, void *stream);

/* ---- K3: PSMNet concat cost volume ------------------------------------------
 * replaces nets/psmnet/psmnet_3.py:149-163 (nets/psmnet/psmnet.py:151-165).
 * feat_l, feat_r: [B,C,h,w]; cost: [B,2C,d,h,w] (NCDHW).
 *   cost[b,c,i,y,x]   = feat_l[b,c,y,x]      (x >= i) else 0
 *   cost[b,C+c,i,y,x] = feat_r[b,c,y,x-i]    (x >= i) else 0            */
int az_cost_volume_fwd(float *cost, const float *feat_l, const float *feat_r, int B, int C,
                       int d, int h, int w, void *stream);
/* adjoint: grad_l[b,c,y,x] = sum_{i<=x} g[b,c,i,y,x];
 *          grad_r[b,c,y,x] = sum_{i, x+i<w} g[b,C+c,i,y,x+i]              */
int az_cost_volume_bwd(float *grad_l, float *grad_r, const float *grad_cost, int B, int C,
                       int d, int h, int w, void *stream);
/* channels-last twins: feat_*: [B,h,w,C], cost: [B,d,h,w,2C] */
int az_cost_volume_fwd_ndhwc(float *cost, const float *feat_l, const float *feat_r, int B,
                             int C, int d, int h, int w, void *stream);
int az_cost_volume_bwd_ndhwc(float *grad_l, float *grad_r, const float *grad_cost, int B,
                             int C, int d, int h, int w, void *stream);

/* ---- K6: fused soft-argmin head --------------------------------------------
 * replaces nets/psmnet/psmnet_3.py:184-215 (F.interpolate trilinear x4,
 * squeeze, F.softmax over D) + nets/psmnet/psmnet_submodule_3.py:80-89
 * (DisparityRegression).  logits: [B,d,h,w]; disp_out: [B,4h,4w]; D = 4d. */
/* stats_out (may be NULL): [B,4h,4w,2] = per-pixel softmax shift and normaliser for the backward pass */
int az_softargmin_fwd(float *disp_out, float *stats_out, const float *logits, int B, int d, int h,
                      int w, void *stream);
/* grad_logits [B,d,h,w] is OVERWRITTEN with d(sum grad_disp*disp)/d logits (the kernel zero-fills
 * it first); logits are re-read.  stats + disp_fwd: what the forward wrote (both or neither; with
 * neither the softmax statistics are recomputed). */
int az_softargmin_bwd(float *grad_logits, const float *grad_disp, const float *logits,
                      const float *stats, const float *disp_fwd, int B, int d, int h, int w,
                      void *stream);

/* ---- K7: bilinear gather warp ----------------------------------------------
 * replaces utils/reprojection.py:13-35 (apply_disparity: linspace grid +
 * F.grid_sample bilinear / zeros / align_corners=False).
 * img,out: [B,C,H,W]; disp: [B,H,W] f32 (pixels, added to x). */
int az_warp_gather_fwd(float *out, const float *img, const float *disp, int B, int C, int H,
                       int W, void *stream);
/* grad_disp [B,H,W] overwritten; grad_img (may be NULL) [B,C,H,W] must be
 * zero-filled by the caller and is accumulated with float atomics. */
int az_warp_gather_bwd(float *grad_disp, float *grad_img, const float *grad_out,
                       const float *img, const float *disp, int B, int C, int H, int W,
                       void *stream);

/* ---- K8: fused patch reprojection loss -------------------------------------
 * replaces utils/reprojection.py:99-127 (get_reproj_error_patch: Unfold(ps) of
 * both patterns, apply_disparity of the C*ps*ps tap channels with the centre
 * pixel's disparity, masked mse_loss(mean), Fold for visualisation).
 * L,R: [B,C,H,W]; disp: [B,H,W] f32; mask: [B,H,W] uint8 or NULL (= all ones);
 * the warp samples R at x + sign*disp (the reference passes -pred_disp: sign=-1).
 * acc[2] (fp64, device): acc[0] = sum (warped - L)^2, acc[1] = element count
 * (masked pixels * C*ps*ps); loss = acc[0] / acc[1].  ps odd, <= 15. */
int az_patch_reproj_fwd(double *acc, const float *L, const float *R, const float *disp,
                        const uint8_t *mask, int B, int C, int H, int W, int ps, float sign,
                        void *stream);
/* grad_disp[B,H,W] = grad_loss[0] * d loss / d disp (patterns receive no gradient) */
int az_patch_reproj_bwd(float *grad_disp, const float *grad_loss, const double *acc,
                        const float *L, const float *R, const float *disp, const uint8_t *mask,
                        int B, int C, int H, int W, int ps, float sign, void *stream);
/* the reference's third return value before cropping/return: Fold (sum of the
 * ps*ps overlapping warped patches, un-normalised) -> vis [B,C,H,W] */
int az_patch_reproj_vis(float *vis, const float *R, const float *disp, int B, int C, int H,
                        int W, int ps, float sign, void *stream);

/* ---- K9: local contrast normalisation -------------------------------------
 * replaces utils/reprojection.py:175-200.  img: B images of H*W floats spaced
 * img_batch_stride floats apart (so channel 0 of a [B,C,H,W] tensor can be read
 * in place); normed, stdv: [B,1,H,W]. */
int az_lcn(float *normed, float *stdv, const float *img, int B, int H, int W, int ksize,
           float eps, long long img_batch_stride, void *stream);

/* ---- K4/K5: 3x3x3 convolution family on fp32 MFMA (NDHWC) ---------------------
 * replaces the cuDNN/ATen calls behind nn.Conv3d / nn.ConvTranspose3d in
 * nets/psmnet/psmnet_submodule_3.py:44-56 and nets/psmnet/psmnet_3.py:15-58,
 * 87-117 (forward and, through autograd, their input / weight gradients).
 * mode: 0 = stride-1 conv, 1 = stride-2 conv, 2 = stride-2 transposed conv
 *       (kernel 3, padding 1, output_padding 1); channels in {32, 64}.
 * in: [B,Di,Hi,Wi,cin]; out: [B,Do,Ho,Wo,cout] with (Do,Ho,Wo) = (Di,Hi,Wi),
 * ((Di-1)/2+1, ...) or (2Di, 2Hi, 2Wi).
 * src = 1 (mode 0, cin = 64 only): the input is the PSMNet concat cost volume
 * synthesised on the fly from in = left features [B,Hi,Wi,32] and in2 = right
 * features (Di = number of disparity planes); otherwise in2 is ignored. */

/* packed[tap][cin/32][cout/32][4][64][4] from any [.][.][27] weight tensor:
 * element (out-channel n, in-channel k, tap t) is read at
 * w[n*stride_out + k*stride_in + (flip ? 26-t : t)]. */
/* precision: 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32, bit-exact fp32 FMA chain);
 *            1 = "bf16x6": both operands are split exactly into three bf16 parts and the
 *                six significant partial products run on v_mfma_f32_32x32x16_bf16 with fp32
 *                accumulation (error ~1e-7 relative, like fp32; 2.7x the fp32 MFMA rate);
 *            2 = the same bf16x6 arithmetic on v_mfma_f32_16x16x32_bf16 (one accumulator rounding
 *                per 32-deep block) with the weights packed [tap][cin/32][cout/16][3][64][8 bf16]
 *                for the depth-rolling kernel: mode 0, cout = 32, src = 0 only (the V0 layers of
 *                psmnet_3.py:87-117 and their input gradients); anything else returns
 *                AZ_EUNSUPPORTED.  A buffer packed with precision p must be used with precision p.
 * The packed buffer holds az_conv3d_packed_floats(cin, cout, precision) floats. */
long long az_conv3d_packed_floats(int cin, int cout, int precision);
int az_conv3d_pack_weights(float *packed, const float *w, int cin, int cout,
                           long long stride_out, long long stride_in, int flip, int precision,
                           void *stream);
/* number of wavefront tiles a launch uses = rows of the BN partial buffers of the precision 0 / 1 kernels */
long long az_conv3d_num_tiles(int mode, int B, int Di, int Hi, int Wi);
/* rows of the BN partial buffers az_conv3d_fwd_stats fills for these arguments (precision 2, the depth-rolling
 * kernel, writes one entry per depth segment and tile; otherwise = az_conv3d_num_tiles) */
long long az_conv3d_stats_tiles(int mode, int precision, int B, int cin, int cout, int Di, int Hi, int Wi);
/* out = relu?( conv(in)*scale[c] + shift[c] + residual ); scale/shift/residual may be
 * NULL (eval-mode BatchNorm folded into scale/shift, or a plain convolution). */
int az_conv3d_fwd(float *out, const float *in, const float *in2, const float *packed_w,
                  const float *scale, const float *shift, const float *residual, int relu,
                  int mode, int src, int precision, int B, int cin, int cout, int Di, int Hi,
                  int Wi, void *stream);
/* out = conv(in) (raw) and, per channel and tile, partials[c][tile] = (sum, centred
 * sum of squares), counts[tile] = valid voxels: the train-mode BatchNorm statistics. */
int az_conv3d_fwd_stats(float *out, float *partials, float *counts, const float *in,
                        const float *in2, const float *packed_w, int mode, int src,
                        int precision, int B, int cin, int cout, int Di, int Hi, int Wi,
                        void *stream);
/* G[m][n][27] = sum_pos coarse[pos][m] * fine[stride*pos - 1 + k][n]: dW of Conv3d with
 * (coarse, fine) = (grad_out, input), dW of ConvTranspose3d with (input, grad_out). */
long long az_conv3d_wgrad_workspace(int cm, int cn);
int az_conv3d_wgrad(float *grad_w, float *workspace, long long workspace_bytes,
                    const float *coarse, const float *fine, int stride, int precision, int B,
                    int cm, int cn, int Dc, int Hc, int Wc, int Df, int Hf, int Wf, void *stream);

/* ---- "f16x3": the arithmetic of the input- and weight-gradient launches (round 4) ----
 * The reference computes these gradients in fp32 (cuDNN backward kernels behind Conv3d / ConvTranspose3d,
 * psmnet_3.py:15-58, psmnet_submodule_3.py:44-56).  Here both operands are scaled by a power of two taken from
 * the tensor's largest magnitude (device scalars written by az_absmax or by the kernel that produced the tensor)
 * and split into two fp16 parts, hi + lo = x up to 2^-22 |x|; hi*hi, hi*lo and lo*hi run on
 * v_mfma_f32_16x16x32_f16 with fp32 accumulation (block sums from zero, one fp32 add per 32-deep block): per-product
 * error ~2^-22 (bf16x6: 2^-24) at half the matrix instructions.  Results are fp32 and unscaled. */
/* Every "*_amax" argument of this header is an "amax array": AZ_AMAX_FLOATS floats of which every AZ_AMAX_STRIDE-th
 * is a slot, and the LARGEST slot is max |tensor| (the workgroups that write a tensor add their maxima to different
 * slots, in different 256-byte lines: thousands of atomics on one memory channel serialise; readers take the largest
 * slot).  Output amax arguments must be all ZERO before the call unless stated otherwise.
 * An amax is the largest FINITE magnitude: inf / NaN elements are left out by every kernel of this library that writes
 * one, so that a non-finite element becomes inf / NaN in fp16 and spoils exactly the outputs that read it (an inf may
 * come out as NaN) while every other output keeps its value; with amax = 0 the scale is finite and the result is 0.
 *
 * CONTRACT of a caller-supplied amax A for a tensor whose true largest finite magnitude is a (operand x, K products per
 * output, the other operand w with amax exactly known):
 *   - A >= a is legal.  Guaranteed for every output y = sum_k x_k w_k:
 *       |y - y_exact| <= [3 * 2^-22 + (K / 32 + 3) * 2^-24] * sum_k |x_k w_k|
 *                        + 2^-38 * (A * sum_k |w_k| + a_w * sum_k |x_k|)
 *     -- the first term is the two-part split (hi + lo = x up to 2^-22 |x|, the dropped lo * lo) and the fp32 accumulation
 *     of the 32-deep block sums; the second is the fp16 subnormal spacing: elements more than 2^17 below A lose low bits
 *     of `lo`, an ABSOLUTE error of 2^-39 A each.  A loose bound A = 2^L a therefore costs L bits of that 2^17 range and
 *     nothing else (tests/test_gpu_f16x3_contract.py: one element 10^6 / 10^8 times the bulk; A = 2^10 a).
 *   - A < a (a stale value) is a caller error, and a loud one: elements above 2-4 A overflow fp16 and the outputs that
 *     read them come out inf / NaN (the scale leaves a factor 2-4 of headroom: A maps to [2^14, 2^15), fp16 ends at
 *     65504).  AZ_DEBUG_AMAX=1 makes the Python wrappers check every attached amax against a fresh az_absmax
 *     (synchronising) and name the tensor.  A saturating conversion (MODE.FP16_OVFL) is deliberately not used: on gfx950
 *     the same bit makes the fp16 MFMA clamp inf operands to FLT_MAX and drop NaN operands (tools/probes/fp16_ovfl_probe.hip).
 * az_absmax: the amax array of x (16-byte aligned; n > 0), zeroing included. */
#define AZ_AMAX_SLOTS 16
#define AZ_AMAX_STRIDE 64
#define AZ_AMAX_FLOATS (AZ_AMAX_SLOTS * AZ_AMAX_STRIDE)
int az_absmax(float *amax, const float *x, long long n, void *stream);
/* Every f16x3 weight image of a model in ONE launch (round 5: a training step packed each weight twice -- forward image,
 * flipped / swapped input-gradient image -- with a launch each, ~175 per step).  One descriptor per image: `kind` names
 * the layout of the kernel that will read it (the per-tensor entry points az_conv2d_pack_weights_f16 /
 * az_conv2d_roll_pack_f16 / az_conv3d_pack_weights_f16 write the same bytes), src / amax / dst are device pointers, the
 * rest as in those calls (taps = kh * kw or 27; ci_real / co_real < cin / cout: zero-padded channels, 2-D "same" layout
 * only).  descs: nd descriptors in DEVICE memory; block_desc: for each of the nblocks workgroups (256 elements each) the
 * descriptor it works on, blocks of one descriptor consecutive; first_block: per descriptor its first workgroup --
 * both built by the caller (activezero_amd/conv3d.py: PackPlan) once per set of weights. */
#define AZ_PACK_2D_SAME 0
#define AZ_PACK_2D_ROLL 1
#define AZ_PACK_3D_GATHER 2
#define AZ_PACK_3D_ROLL 3
#define AZ_PACK_3D_ROLL2 4 /* 64 output channels on the depth-rolling kernel: two AZ_PACK_3D_ROLL images of 32 channels each, one after the other */
typedef struct AzPackDesc {
    void *dst;           /* packed image (fp16 pairs) */
    const float *src;    /* the weight tensor in PyTorch's layout */
    const float *amax;   /* its amax array */
    long long s_co, s_ci;
    int kind, cin, cout, ci_real, co_real, taps, flip, pad_;
} AzPackDesc;
int az_pack_f16_multi(const AzPackDesc *descs, const int *block_desc, const int *first_block, int nd, int nblocks,
                      void *stream);
/* The other end of a step: az_conv3d_wgrad_f16 / az_conv2d_wgrad(_f16) called with grad_w = NULL only ACCUMULATE into
 * their tap-major workspace [taps][cm][cn], which the caller has zeroed (one memset of an arena per backward pass instead
 * of one per layer); this call unpacks all of a pass's workspaces into PyTorch's [cm_real][cn_real][taps] layout in one
 * launch (descriptor / block tables as above; activezero_amd/overlap.py: Sink.join).  Round 4: 105 memsets and 88 unpack
 * launches per step. */
typedef struct AzUnpackDesc {
    float *dst;
    const float *ws;
    int cm, cn, cm_real, cn_real, taps, pad_;
} AzUnpackDesc;
int az_wgrad_unpack_multi(const AzUnpackDesc *descs, const int *block_desc, const int *first_block, int nd, int nblocks,
                          void *stream);
long long az_conv3d_packed_floats_f16(int cin, int cout);
/* as az_conv3d_pack_weights, for the f16x3 launches below: w * 2^k (k from w_amax[0]) split into two fp16 parts.
 * `mode` = the mode the buffer will be launched with: mode 0 with cout = 32 runs on the depth-rolling kernel
 * ([tap][cin/32][cout/16][2][64][8 fp16]), everything else on the gather kernel ([tap][cin/32][cout/32][2][2][64][8]). */
int az_conv3d_pack_weights_f16(float *packed, const float *w, const float *w_amax, int cin, int cout,
                               long long stride_out, long long stride_in, int flip, int mode, void *stream);
/* which of the layouts that is: AZ_PACK_3D_ROLL / AZ_PACK_3D_ROLL2 / AZ_PACK_3D_GATHER (the `kind` of an az_pack_f16_multi descriptor) */
int az_conv3d_f16_layout(int mode, int cin, int cout);
/* az_conv3d_fwd / az_conv3d_fwd_stats / az_conv3d_stats_tiles on the f16x3 arithmetic (src = 0 only).  in_amax /
 * w_amax: device scalars holding max |in| and max |w| (of the UNPACKED weights).  The input gradient of a layer is
 * this call on the gradient of its raw output with the flipped / swapped packing, no affine map.  Every mode with
 * 32 / 64 channels on either side. */
/* "S2 format" (pre-split operands, round 5): a tensor whose only readers are f16x3 matrix kernels may be stored the way
 * they stage it -- every aligned group of four channels (16 bytes) holds hi(c0) hi(c1) | hi(c2) hi(c3) | lo(c0) lo(c1) |
 * lo(c2) lo(c3), the two fp16 parts of x * 2^k with k from the amax array that travels with it -- in place of the four
 * floats: same bytes, and the consumer stages by copy.  az_bn3d_bwd(split_out = 1) writes dx that way (its amax is then
 * an upper BOUND of max |dx| derived before the first element is written); in_split / split_mask below tell a consumer.
 * az_conv3d_fwd_f16_split_ok / az_conv3d_wgrad_f16_split_ok: which launches take such an operand (1 / bit mask: bit 0 =
 * coarse, bit 1 = fine; a stride-2 launch takes one of the two); the launches return AZ_EUNSUPPORTED otherwise. */
int az_conv3d_fwd_f16_split_ok(int mode, int B, int cin, int cout, int Di, int Hi, int Wi);
int az_conv3d_wgrad_f16_split_ok(int stride, int B, int cm, int cn, int Dc, int Hc, int Wc, int Df, int Hf, int Wf);
int az_conv3d_fwd_f16(float *out, const float *in, const float *packed_w, const float *in_amax,
                      const float *w_amax, int in_split, const float *scale, const float *shift, const float *residual,
                      int relu, int mode, int B, int cin, int cout, int Di, int Hi, int Wi, void *stream);
long long az_conv3d_stats_tiles_f16(int mode, int B, int cin, int cout, int Di, int Hi, int Wi);
int az_conv3d_fwd_stats_f16(float *out, float *partials, float *counts, const float *in, const float *packed_w,
                            const float *in_amax, const float *w_amax, int mode, int B, int cin, int cout,
                            int Di, int Hi, int Wi, void *stream);
/* as az_conv3d_wgrad on the f16x3 arithmetic; coarse_amax / fine_amax: device scalars max |coarse|, max |fine|.
 * Supported: stride 1 with 32 or 64 channels on either side; everything else returns AZ_EUNSUPPORTED. */
int az_conv3d_wgrad_f16(float *grad_w, float *workspace, long long workspace_bytes, const float *coarse,
                        const float *fine, const float *coarse_amax, const float *fine_amax, int split_mask, int stride,
                        int B, int cm, int cn, int Dc, int Hc, int Wc, int Df, int Hf, int Wf, void *stream);

/* az_conv2d_pack_weights / az_conv2d_fwd / az_conv2d_fwd_stats on the f16x3 arithmetic (psmnet_submodule_3.py:13-41,
 * 92-220: every stride-1 "same" Conv2d of the extractor and its input gradient): in_amax / w_amax = device scalars
 * max |in|, max |w| (unpacked); packed buffer = kh*kw*cin*cout floats. */
int az_conv2d_pack_weights_f16(float *packed, const float *w, const float *w_amax, int cin, int cout,
                               int ci_real, int co_real, long long stride_out, long long stride_in, int kh,
                               int kw, int flip, void *stream);
int az_conv2d_fwd_f16(float *out, const float *in, const float *packed_w, const float *in_amax,
                      const float *w_amax, const float *scale, const float *shift, const float *residual,
                      int relu, int B, int H, int W, int cin, int cout, int in_cstride, int out_cstride,
                      int res_cstride, int kh, int kw, int dilation, void *stream);
int az_conv2d_fwd_stats_f16(float *out, float *partials, float *counts, const float *in, const float *packed_w,
                            const float *in_amax, const float *w_amax, int groups, int B, int H, int W,
                            int cin, int cout, int in_cstride, int out_cstride, int kh, int kw, int dilation,
                            void *stream);
/* plain-bf16 forms for the backward pass of the RAFT-Stereo GRU update in the reference's autocast arithmetic
 * (nets/raft/raft_stereo.py:142-172, train.py:303-309; forward: az_conv2d_bf16_fwd): the flipped weight image of the
 * input gradient (call with cin / cout and the two strides swapped) and the 3x3 weight gradient with operands rounded
 * to bf16 once, one MFMA per block, fp32 accumulation */
int az_conv2d_pack_weights_bf16_flipped(float *packed, const float *w, int cin, int cout, long long stride_out,
                                        long long stride_in, int kh, int kw, void *stream);
int az_conv2d_wgrad_bf16(float *grad_w, float *workspace, long long workspace_bytes, const float *grad_out,
                         const float *in, int B, int H, int W, int cm, int cn, int cm_real, int cn_real,
                         int go_cstride, int in_cstride, void *stream);
/* az_conv2d_wgrad on f16x3: go_amax / in_amax = amax arrays of grad_out and in */
int az_conv2d_wgrad_f16(float *grad_w, float *workspace, long long workspace_bytes, const float *grad_out,
                        const float *in, const float *go_amax, const float *in_amax, int B, int H, int W,
                        int cm, int cn, int cm_real, int cn_real, int go_cstride, int in_cstride, int kh,
                        int kw, int dilation, void *stream);
/* az_conv2d_roll_pack / _fwd / _fwd_stats (the batch-walking kernel of the 3x3 layers with 32 / 64 channels) on f16x3;
 * packed buffer = 9*cin*cout floats; partial rows = az_conv2d_roll_stats_rows */
int az_conv2d_roll_pack_f16(float *packed, const float *w, const float *w_amax, int cin, int cout,
                            long long stride_out, long long stride_in, int flip, void *stream);
int az_conv2d_roll_fwd_f16(float *out, const float *in, const float *packed, const float *in_amax,
                           const float *w_amax, const float *scale, const float *shift, const float *residual,
                           int relu, int B, int H, int W, int cin, int cout, void *stream);
int az_conv2d_roll_fwd_stats_f16(float *out, float *partials, float *counts, const float *in, const float *packed,
                                 const float *in_amax, const float *w_amax, int groups, int B, int H, int W,
                                 int cin, int cout, void *stream);

/* 32 -> 1 classifier conv (psmnet_3.py:103-117) with the fused running sum
 * cost_k = classif_k(out_k) + cost_{k-1} (psmnet_3.py:177-179): logits [B,D,H,W] =
 * conv(in [B,D,H,W,32], w [1,32,3,3,3]) + addend (may be NULL).
 * in_scale / in_shift ([32] each, both or neither): the operand is relu(in * scale + shift) -- the train-mode
 * BatchNorm + ReLU of classifN[0..1] in front of this layer, applied while the input is staged instead of by a
 * pass of its own over the tensor (`in` is then the RAW convolution output); the weight gradient takes the
 * same pair. */
int az_conv3d_c1_fwd(float *logits, const float *in, const float *w, const float *addend,
                     const float *in_scale, const float *in_shift, int B, int D, int H, int W, void *stream);
int az_conv3d_c1_dgrad(float *grad_in, const float *grad_logits, const float *w, int B, int D,
                       int H, int W, void *stream);
int az_conv3d_c1_wgrad(float *grad_w, const float *in, const float *grad_logits, const float *in_scale,
                       const float *in_shift, int B, int D, int H, int W, void *stream);

/* ---- BatchNorm3d pieces around K4/K5 (psmnet_submodule_3.py:55) --------------------
 * finalize: merge the conv partials (Chan, fp64) -> mean, invstd, scale = gamma*invstd,
 * shift = beta - mean*scale; running stats updated in place (momentum, unbiased var)
 * unless running_mean/var are NULL; *num_batches_tracked (int64, may be NULL) += 1.
 * scratch (may be NULL; az_bn3d_finalize_scratch(C) floats): with it, layers with >= 4096 partials per channel are
 * merged in two stages (C x 32 blocks, then C blocks) instead of by C blocks alone. */
long long az_bn3d_finalize_scratch(int C);
int az_bn3d_finalize(float *mean, float *invstd, float *scale, float *shift,
                     float *running_mean, float *running_var, const float *partials,
                     const float *counts, const float *gamma, const float *beta,
                     long long ntiles, int C, float eps, float momentum,
                     long long *num_batches_tracked, float *scratch, long long scratch_floats, void *stream);
/* batch statistics of a channels-last tensor x[nvox][C] (C = 32, 64, 128) whose producer is not one
 * of these convolutions (the 2-D extractor's layers, nets/psmnet/psmnet_submodule_3.py:8-22):
 * partials [C][tiles][2], counts [tiles] with tiles = az_bn3d_stats_tiles(nvox, C), for az_bn3d_finalize */
long long az_bn3d_stats_tiles(long long nvox, int C);
int az_bn3d_stats(float *partials, float *counts, const float *x, long long nvox, int C, void *stream);
/* The same for ALL statistic groups of a tensor [groups][nvox][C] in three launches: y = relu?(bn(x) +
 * residual) with batch statistics per group (groups = 2: the left and the right image set of
 * psmnet_3.py:145-146 stacked in one batch; running statistics are updated group after group).
 * mean/invstd/scale/shift are [groups][C] outputs; workspace of az_bn2d_workspace() bytes. */
long long az_bn2d_workspace(int groups, long long nvox, int C);
int az_bn2d_fwd(float *y, float *mean, float *invstd, float *scale, float *shift, float *running_mean,
                float *running_var, const float *x, const float *residual, const float *gamma,
                const float *beta, float *workspace, long long workspace_bytes, int relu, int groups,
                long long nvox, int C, float eps, float momentum, long long *num_batches_tracked /* += groups; may be NULL */,
                const float *partials, const float *counts, long long partial_tiles /* of az_conv2d_fwd_stats; NULL, NULL, 0:
                the statistics pass runs here */,
                float *y_amax /* may be NULL; ZERO before the call: receives max |y| (az_bn3d_apply) */, void *stream);
/* backward: dx [groups][nvox][C]; dgamma/dbeta [C] summed over the groups; dz_out (may be NULL) = the
 * gradient of the residual branch when relu != 0 */
int az_bn2d_bwd(float *dx, float *dz_out, float *dgamma, float *dbeta, float *workspace,
                long long workspace_bytes, const float *dy, const float *y, const float *x,
                const float *mean, const float *invstd, const float *gamma, const float *scale,
                const float *shift, int relu, int groups, long long nvox, int C,
                float *dx_amax /* may be NULL; ZERO before the call: receives max |dx| */, void *stream);
int az_bn3d_eval_affine(float *scale, float *shift, const float *gamma, const float *beta,
                        const float *running_mean, const float *running_var, float eps, int C,
                        void *stream);
/* y = relu?( x*scale[c] + shift[c] + residual ) over nvox voxels of C channels; y_amax (may be NULL): device
 * scalar, ZERO before the call, that receives max |y| (the f16x3 operand scale of the layers that read y) */
int az_bn3d_apply(float *y, const float *x, const float *scale, const float *shift,
                  const float *residual, int relu, long long nvox, int C, float *y_amax, void *stream);
/* backward of y = relu?(bn(x) + residual): dz = dy*[y>0] (or dy), dgamma, dbeta,
 * dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)); dz_out (may be NULL) = dz =
 * gradient of the residual branch.  coef: [C][3] scratch.  scale/shift (both or neither): the
 * forward's affine map; when given for a ReLU layer WITHOUT residual the mask is recomputed as
 * fma(x, scale, shift) > 0 and y is not read (may be NULL).
 * dx_amax (may be NULL; need NOT be zero, the first kernel clears it): receives max |dx| -- the operand scale of the
 * f16x3 input- and weight-gradient kernels that read dx next, taken while dx is written instead of by az_absmax.
 * split_out = 1 (dx_amax required): dx is written in the S2 format above instead of as floats, and dx_amax receives
 * the per-tensor bound  max_c |gamma_c invstd_c| (max |dz_c| + |mean dz_c| + max |xhat_c| |mean dz_c xhat_c|)  >= max |dx|
 * that fixed its scale (the reduce pass takes the two per-channel maxima next to its sums); dz_out stays fp32. */
long long az_bn3d_bwd_workspace(long long nvox, int C);
int az_bn3d_bwd(float *dx, float *dz_out, float *dgamma, float *dbeta, float *coef,
                float *workspace, long long workspace_bytes, const float *dy, const float *y,
                const float *x, const float *mean, const float *invstd, const float *gamma,
                const float *scale, const float *shift, int relu, long long nvox, int C,
                float *dx_amax, int split_out, void *stream);
/* ---- SPP branch upsampling (psmnet_submodule_3.py:198-209: F.upsample(branch, (H, W), mode="bilinear") + torch.cat) ----
 * out rows [B,H,W,out_cstride] (the pointer already offset to the branch's channel slot of the concat buffer) <- bilinear
 * interpolation, align_corners = True, of in rows [B,hs,ws,C] (C % 4 == 0, C <= 64, C / 4 a power of two), with ATen's
 * source positions and expression.  _bwd: grad_in [B,hs,ws,C] <- the adjoint over grad_out rows [B,H,W,gout_cstride] in two
 * separable passes (x then y, through `workspace`); no atomics. */
int az_spp_upsample_fwd(float *out, const float *in, int B, int hs, int ws, int H, int W, int C, int out_cstride,
                        void *stream);
long long az_spp_upsample_bwd_workspace(int B, int ws, int H, int C);  /* bytes: the x-pass result [B,H,ws,C] */
int az_spp_upsample_bwd(float *grad_in, float *workspace, long long workspace_bytes, const float *grad_out, int B,
                        int hs, int ws, int H, int W, int C, int gout_cstride, void *stream);
/* y = relu?(a + b), n floats (n % 4 == 0): the plain residual sums of psmnet_3.py:166-175; y_amax as above */
int az_add_relu(float *y, const float *a, const float *b, int relu, long long n, float *y_amax, void *stream);
/* y = a + b (+ c) (+ d), n floats (n % 4 == 0; c, d may be NULL): the gradient of a tensor with up to
 * four consumers (cost0 of psmnet_3.py:165-175 feeds the first hourglass and three residual sums)
 * in one pass instead of autograd's chain of pairwise adds */
int az_sum4(float *y, const float *a, const float *b, const float *c, const float *d, long long n,
            void *stream);

/* ---- K13: stride-1 "same" 2-D convolution family (bf16x6 MFMA, channels-last) ----------------------
 * replaces the cuDNN/ATen calls behind nn.Conv2d in the 2-D feature extractor,
 * nets/psmnet/psmnet_submodule_3.py:13-41 (convbn / conv), :59-77 (BasicBlock), :92-220
 * (FeatureExtraction) -- forward and, through autograd, input and weight gradients -- and the three
 * F.conv2d calls of the factored cost-volume convolution (nets/psmnet/psmnet_3.py:149-166).
 * Geometries: (kh,kw,dilation) in {(3,3,1), (3,3,2), (1,1,*), (3,5,1)}, stride 1, padding such that
 * the output has the input's H x W.  cin % 16 == 0, cout % 32 == 0 (the packer zero-pads weights
 * of narrower tensors; the activations then carry the padding channels).
 * packed image: [tap][cin/16][cout/32][3 parts][64 lanes][8] bf16 = az_conv2d_packed_floats() floats;
 * element (out-channel n, in-channel k, tap t) is read at w[n*stride_out + k*stride_in + (flip ? T-1-t : t)]
 * for n < co_real, k < ci_real, zero otherwise: forward = (cin*T, T, 0), input gradient = the same call
 * with the channel roles swapped, (T, cin*T, flip 1). */
long long az_conv2d_packed_floats(int cin, int cout, int kh, int kw);
int az_conv2d_pack_weights(float *packed, const float *w, int cin, int cout, int ci_real, int co_real,
                           long long stride_out, long long stride_in, int kh, int kw, int flip,
                           void *stream);
/* out[b,y,x,co] = relu?( conv(in)[..] * scale[co] + shift[co] + residual[b,y,x,co] ); in/out/residual are
 * [B,H,W,*] with pixel strides in/out/res_cstride floats (>= cin / cout / cout: a channel slice of a wider
 * tensor can be read or written in place); scale/shift/residual may be NULL. */
int az_conv2d_fwd(float *out, const float *in, const float *packed_w, const float *scale,
                  const float *shift, const float *residual, int relu, int B, int H, int W, int cin,
                  int cout, int in_cstride, int out_cstride, int res_cstride, int kh, int kw,
                  int dilation, void *stream);
/* grad_w [cm_real][cn_real][kh][kw] (a Conv2d weight's layout) = sum over pixels of
 * grad_out[b,y,x,co] * in[b, y + dil*(i - kh/2), x + dil*(j - kw/2), ci]; cm/cn = channel counts of the
 * operation (multiples of 32, >= the real ones; the tensors' pixel strides must cover them). */
long long az_conv2d_wgrad_workspace(int cm, int cn, int kh, int kw);
int az_conv2d_wgrad(float *grad_w, float *workspace, long long workspace_bytes, const float *grad_out,
                    const float *in, int B, int H, int W, int cm, int cn, int cm_real, int cn_real,
                    int go_cstride, int in_cstride, int kh, int kw, int dilation, void *stream);
/* az_conv2d_fwd without epilogue operands that also emits the BatchNorm partials of its (raw) output -- per channel
 * and 8x16 patch (sum, M2 about the patch mean), layout [groups][cout][tiles][2] + counts [groups][tiles] with
 * tiles = az_conv2d_stats_tiles() -- which az_bn2d_fwd accepts in place of its own statistics pass over the tensor
 * (psmnet_submodule_3.py:13-22 convbn: Conv2d -> BatchNorm2d).  3x3 (dilation 1, 2) and 1x1 layers. */
long long az_conv2d_stats_tiles(int B, int H, int W, int groups);
int az_conv2d_fwd_stats(float *out, float *partials, float *counts, const float *in, const float *packed_w,
                        int groups, int B, int H, int W, int cin, int cout, int in_cstride, int out_cstride,
                        int kh, int kw, int dilation, void *stream);

/* ---- K13r: 3x3 stride-1 dilation-1 Conv2d with cin, cout in {32, 64} as a walk over the batch (az_conv2d_roll.hip;
 * replaces az_conv2d_fwd / az_conv2d_fwd_stats for psmnet_submodule_3.py:92-147 firstconv[1..2], layer1, layer2 and
 * their input gradients).  Dense channels-last tensors [B,H,W,cin] -> [B,H,W,cout]. ---- */
long long az_conv2d_roll_packed_floats(int cin, int cout);
/* weights w[co * stride_out + ci * stride_in + tap] -> packed image; flip: taps reversed (input gradient) */
int az_conv2d_roll_pack(float *packed, const float *w, int cin, int cout, long long stride_out, long long stride_in,
                        int flip, void *stream);
/* out = relu?( conv(in) * scale[c] + shift[c] + residual ); scale / shift / residual may be NULL */
int az_conv2d_roll_fwd(float *out, const float *in, const float *packed, const float *scale, const float *shift,
                       const float *residual, int relu, int B, int H, int W, int cin, int cout, void *stream);
/* rows per statistic group of the partial buffers az_conv2d_roll_fwd_stats fills */
long long az_conv2d_roll_stats_rows(int groups, int B, int H, int W, int cin, int cout);
/* out = conv(in) (raw); partials [groups][cout][rows][2] = (sum, centred sum of squares), counts [groups][rows]:
 * what az_bn2d_fwd accepts in place of its own statistics pass (groups = consecutive B / groups images each) */
int az_conv2d_roll_fwd_stats(float *out, float *partials, float *counts, const float *in, const float *packed,
                             int groups, int B, int H, int W, int cin, int cout, void *stream);
/* plain-bf16 twin of az_conv2d_fwd for 3x3 layers (one MFMA per 16-deep block, operands rounded to bf16,
 * fp32 accumulation and fp32 tensors): the arithmetic of the reference's autocast region around the RAFT-Stereo
 * GRU update (nets/raft/raft_stereo.py:142-172 calling nets/raft/update.py:19-41 ConvGRU).
 * out = act(conv(in) + bias[co] + residual); act: 0 none, 1 ReLU, 2 sigmoid, 3 tanh,
 * 4 = (1 - z) * h + z * tanh(.) with gate_z / gate_h [B,H,W,*] (pixel strides z_cstride / h_cstride).
 * packed image of az_conv2d_pack_weights_bf16: kh*kw*cin*cout/2 floats. */
int az_conv2d_pack_weights_bf16(float *packed, const float *w, int cin, int cout, long long stride_out,
                                long long stride_in, int kh, int kw, void *stream);
int az_conv2d_bf16_fwd(float *out, const float *in, const float *packed_w, const float *bias,
                       const float *residual, const float *gate_z, const float *gate_h, int act, int B, int H,
                       int W, int cin, int cout, int in_cstride, int out_cstride, int res_cstride,
                       int z_cstride, int h_cstride, void *stream);
/* "f16x1" -- the same three entry points with ONE FP16 part per operand: what torch.cuda.amp.autocast computes on CUDA
 * (raft_stereo.py:14; the bf16 form above rounds operands 8x coarser than the reference does).  in_amax / w_amax / go_amax
 * (each may be NULL = no scaling, as autocast): amax arrays for a power-of-two operand scale -- the gradient operands of the
 * backward launches get one (az_gru_bwd1 / az_gru_bwd2 write them), in place of the reference's GradScaler (train.py:303-309).
 * az_conv2d_pack_weights_h1: flip = 1 with (cin, cout, stride_out, stride_in) swapped = the input-gradient image. */
int az_conv2d_pack_weights_h1(float *packed, const float *w, const float *w_amax, int cin, int cout, long long stride_out,
                              long long stride_in, int flip, void *stream);
int az_conv2d_h1_fwd(float *out, const float *in, const float *packed_w, const float *in_amax, const float *w_amax,
                     const float *bias, const float *residual, const float *gate_z, const float *gate_h, int act, int B, int H,
                     int W, int cin, int cout, int in_cstride, int out_cstride, int res_cstride, int z_cstride,
                     int h_cstride, void *stream);
int az_conv2d_wgrad_h1(float *grad_w, float *workspace, long long workspace_bytes, const float *grad_out, const float *in,
                       const float *go_amax, const float *in_amax, int B, int H, int W, int cm, int cn, int cm_real,
                       int cn_real, int go_cstride, int in_cstride, void *stream);
/* Gate arithmetic of the ConvGRU update under autograd (nets/raft/update.py:32-41 in training, train.py:303-309): dense
 * [npix][channels] fp32 rows; hx = [h (hid) | x (inp)], zr = [z | r] (2 hid), rhx = [r * h | x]; see az_gru_gates.hip.
 * forward: az_gru_rh (rhx from zr, hx), az_gru_out (h' = (1 - z) h + z q); backward: az_gru_bwd1 (g = dL/dh' -> dq_pre,
 * dzr[:hid], dh_acc), az_gru_bwd2 (d_rhx -> dzr[hid:], dh_acc +=), az_gru_bwd3 (dh, dx from dh_acc, d_rhx, d_hx). */
/* the GRU's input rows [h | x_0 | x_1 ..] in ONE launch (update.py:33 `hx = torch.cat([h, x], dim=1)` on channels-last rows):
 * dst [npix][sum c_i] <- nsrc <= 4 sources, channel counts multiples of 4; kinds[i] = 0: dense rows [npix][c_i], 1: an NCHW image
 * [npix / hw][c_i][hw] (c_i <= 128).  az_rows_slice_to_image: the inverse for one channel slice [at, at + c) of such rows -> a
 * dense NCHW image: the gradient of an image source. */
int az_rows_concat(float *dst, long long npix, long long hw, int nsrc, const float *const *srcs, const int *channels,
                   const int *kinds, void *stream);
int az_rows_slice_to_image(float *image, const float *rows, long long npix, long long hw, int ctot, int at, int c, void *stream);
int az_gru_rh(float *rhx, const float *zr, const float *hx, long long npix, int hid, int inp, void *stream);
int az_gru_out(float *hn, const float *zr, const float *q, const float *hx, long long npix, int hid, int inp, void *stream);
/* (dq_amax / dzr_amax: amax arrays, both or neither, ZERO before az_gru_bwd1: max |dq_pre| and -- completed by az_gru_bwd2 --
 * max |dzr|, the operand scales of the f16x1 gradient convolutions) */
int az_gru_bwd1(float *dq_pre, float *dzr, float *dh_acc, const float *g, const float *zr, const float *q, const float *hx,
                long long npix, int hid, int inp, float *dq_amax, float *dzr_amax, void *stream);
int az_gru_bwd2(float *dzr, float *dh_acc, const float *d_rhx, const float *zr, const float *hx, long long npix, int hid, int inp,
                float *dzr_amax, void *stream);
int az_gru_bwd3(float *dh, float *dx, const float *dh_acc, const float *d_rhx, const float *d_hx, long long npix, int hid, int inp,
                void *stream);
/* the extractor's first layer (psmnet_submodule_3.py:97-99: 3x3, stride 2, pad 1 on a 3- or 6-channel
 * image) as patch extraction + 1x1 convolution: patches[b,oy,ox, t*C + c] = x[b, 2oy-1+t/3, 2ox-1+t%3, c]
 * (zero outside the image and in channels [9C, Kp)); x: [B,H,W,C]; patches: [B,(H-1)/2+1,(W-1)/2+1,Kp].
 * az_col2im_s2k3 is its adjoint (grad_x fully written). */
int az_im2col_s2k3(float *patches, const float *x, int B, int C, int H, int W, int Kp, void *stream);
int az_col2im_s2k3(float *grad_x, const float *grad_patches, int B, int C, int H, int W, int Kp,
                   void *stream);

/* ---- K14: IR dot-pattern extraction (data side) ------------------------------------------
 * replaces datasets/dataset_utils.py:33-46 get_smoothed_ir_pattern2(img_ir, img, ks, threshold) as called by
 * datasets/messytable.py:406-426: pattern = 1 where the min-max normalised |img_ir - img| exceeds its
 * cv2 INTER_AREA shrink (by ks) / enlarge round trip by more than `threshold`, else 0.
 * img_ir, img, pattern: [B,H,W] f32; workspace of az_ir_pattern_workspace() bytes. */
long long az_ir_pattern_workspace(int B, int H, int W, int ks);
int az_ir_pattern(float *pattern, float *workspace, long long workspace_bytes, const float *img_ir,
                  const float *img, int B, int H, int W, int ks, float threshold, void *stream);

/* ---- K10/K11: RAFT-Stereo 1-D correlation (secondary path) -----------------------
 * replaces nets/raft/corr.py:115-161 (CorrBlock1D: einsum all-pairs correlation /
 * sqrt(C), avg_pool pyramid over the last axis, 2r+1-tap linear lookup through
 * nets/raft/raft_utils.py:68-82 bilinear_sampler, align_corners=True, zero padding).
 * f1: [B,C,H,W1], f2: [B,C,H,W2] (NCHW); corr: [B,H,W1,W2]. */
int az_corr1d_volume(float *corr, const float *f1, const float *f2, int B, int C, int H, int W1,
                     int W2, void *stream);
/* grad_f1 / grad_f2 may be NULL */
int az_corr1d_volume_bwd(float *grad_f1, float *grad_f2, const float *grad_corr, const float *f1,
                         const float *f2, int B, int C, int H, int W1, int W2, void *stream);
/* one pyramid step: dst[rows][Wsrc/2] = mean of neighbouring pairs of src[rows][Wsrc] */
int az_corr1d_pool(float *dst, const float *src, long long rows, int Wsrc, void *stream);
int az_corr1d_pool_bwd(float *grad_src, const float *grad_dst, long long rows, int Wsrc,
                       void *stream);
/* out[b, ch_offset + k, h, w1] = lerp(pyr_level[b,h,w1,:], coords[b,0,h,w1] / 2^level + k - radius),
 * k = 0..2*radius; out is [B,ch_total,H,W1]; coords is [B,2,H,W1] (channel 0 used). */
int az_corr1d_lookup_fwd(float *out, const float *pyr_level, const float *coords, int B, int H,
                         int W1, int W_level, int radius, int level, int ch_offset, int ch_total,
                         void *stream);
/* grad_pyr_level [B,H,W1,W_level] is overwritten (zero-filled first) */
int az_corr1d_lookup_bwd(float *grad_pyr_level, const float *grad_out, const float *coords, int B,
                         int H, int W1, int W_level, int radius, int level, int ch_offset,
                         int ch_total, void *stream);
/* the same scatter ADDED to the buffer's contents (cleared once by the caller): the lookups of one step share a pyramid and a
 * level's gradient is their sum -- one buffer and the kernel's own atomics instead of one cleared buffer per lookup and a
 * tensor addition for each */
int az_corr1d_lookup_bwd_acc(float *grad_pyr_level, const float *grad_out, const float *coords, int B,
                         int H, int W1, int W_level, int radius, int level, int ch_offset,
                         int ch_total, void *stream);

/* ---- K12: disparity loss + error metrics (the step after the path) -------------------------
 * Replaces utils/losses.py:7-15 psmnet_disp (boolean-index compaction + three smooth_l1 means)
 * and utils/cascade_metrics.py:16-62 compute_err_metric (ten masked reductions with .item()).
 * All maps are [B,1,H,W] fp32 flattened to n elements; `mask` is one byte per element, or NULL
 * for the range rule lo < gt < hi (train.py:272).  Accumulators are caller-zeroed fp64. */
/* acc4[0..2] += sum smooth_l1(pred3|pred2|pred1 - gt) over valid pixels, acc4[3] += count */
int az_disp_loss_fwd(double *acc4, const float *pred3, const float *pred2, const float *pred1,
                     const float *gt, const unsigned char *mask, float lo, float hi, long long n,
                     void *stream);
/* g_k = w_k * gloss[0] / acc4[3] * clamp(pred_k - gt, -1, 1) on valid pixels, 0 elsewhere */
int az_disp_loss_bwd(float *g3, float *g2, float *g1, const float *pred3, const float *pred2,
                     const float *pred1, const float *gt, const unsigned char *mask, float lo,
                     float hi, const float *gloss, const double *acc4, float w3, float w2, float w1,
                     long long n, void *stream);
/* acc8 += {sum|dd|, #(|dd|>1), #(|dd|>2), sum clip(|1000 dz|,0,100), #(|dz|>2e-3), #(|dz|>4e-3),
 * #(|dz|>8e-3), count}; depth_pred NULL -> focal_x_baseline[b] / disp_pred (cascade_metrics.py:36) */
int az_disp_metrics(double *acc8, const float *disp_gt, const float *depth_gt,
                    const float *disp_pred, const float *depth_pred, const float *focal_x_baseline,
                    const unsigned char *mask, int B, long long per_batch, void *stream);

/* ---- K3+K4 factored: dres0[0] applied to the concat cost volume without the volume -------------
 * (nets/psmnet/psmnet_3.py:149-166).  F = the left map convolved with the NC x 5 depth-class /
 * staircase-offset variants of the depth-summed 3x3 kernels (bulk + edge maps, below); G: [B,H,W+2,NC*64] = the right map (two zero
 * columns on the left) convolved with the NC x 2 variants of the 3x5 kernels indexed by kw - kd
 * (activezero_amd/costconv.py builds them).  out [B,D,H,W,32] = F[c(d), min(x-d,2)][y,x] + G[c(d), x=W-1][y,x-d]
 * for x - d >= -2, else 0.  The backward writes the matching reductions of grad_out over d. */
int az_costconv_edge_width(int D, int W); /* XE = min(W, D+1): columns on which the delta < 2 maps exist */
int az_costconv_num_classes(int D);       /* NC = min(D, 3) depth classes: first / middle / last plane */
/* F_bulk [B,H,W,NC*32] (x - d >= 2, per depth class), F_edge [B,H,XE,NC*128] (x - d = -2..1), G [B,H,W+2,NC*64] */
/* the merged kernels themselves (costconv.py: 0/1-masked sums of the [32][64][3][3][3] weight over kd, and over kw into the
 * shifted column for K_R): kl [ncls][5][32][32][3][3], kr [ncls][2][32][32][3][5]; ml [ncls][5][3][3], mr [ncls][2][3][3][5];
 * _bwd: the adjoint, grad_weight fully written */
int az_costconv_merge_fwd(float *kl, float *kr, const float *weight, const float *ml, const float *mr, int ncls, void *stream);
int az_costconv_merge_bwd(float *grad_weight, const float *gkl, const float *gkr, const float *ml, const float *mr, int ncls,
                          void *stream);
int az_costconv_assemble_fwd(float *out, const float *F_bulk, const float *F_edge, const float *G, int B,
                             int D, int H, int W, void *stream);
int az_costconv_assemble_bwd(float *dF_bulk, float *dF_edge, float *dG, const float *grad_out, int B, int D,
                             int H, int W, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* AZHIP_H */
