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
/* ABI version: bumped when a signature changes. */
int az_abi_version(void);

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
int az_softargmin_fwd(float *disp_out, const float *logits, int B, int d, int h, int w,
                      void *stream);
/* grad_logits [B,d,h,w] is OVERWRITTEN with d(sum grad_disp*disp)/d logits
 * (the kernel zero-fills it first); logits are re-read, nothing is saved. */
int az_softargmin_bwd(float *grad_logits, const float *grad_disp, const float *logits, int B,
                      int d, int h, int w, void *stream);

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

#ifdef __cplusplus
}
#endif
#endif /* AZHIP_H */
