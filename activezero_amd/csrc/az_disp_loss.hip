// K12 -- the step after the path: the three-head masked smooth-L1 disparity loss
// (reference utils/losses.py:7-15, mask rule train.py:272) and the disparity / depth error
// metrics (utils/cascade_metrics.py:16-62), each as ONE pass over the [B,1,H,W] maps.
// The reference compacts every operand with boolean indexing (a device->host sync per index)
// and reduces the compacted copies one metric at a time; here a pixel is read once, the mask is
// applied in registers and every sum lands in a small fp64 accumulator vector the caller owns.
// HBM-bound: 4 maps read per pixel (16 B + mask byte); tiny next to the path itself, the point
// is the removed syncs and launches.
#include "az_common.h"

#define DL_BLOCK 256

__device__ __forceinline__ bool dl_valid(const unsigned char *mask, float gt, float lo, float hi, long long i) {
    return mask ? (mask[i] != 0) : (gt > lo && gt < hi);
}

// block reduction (wave shuffles, then LDS across the 4 waves) and ONE fp64 atomic per block and
// accumulator: with an atomic per wave the 130 k same-address adds of a 4 x 544 x 960 map
// serialised in L2 and the kernel took 0.79 ms instead of ~15 us
template <int N>
__device__ __forceinline__ void dl_flush(double (&v)[N], double *acc) {
    __shared__ double red[DL_BLOCK / 64][N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        double x = v[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < N) {
        double x = 0.0;
#pragma unroll
        for (int wv = 0; wv < DL_BLOCK / 64; ++wv) x += red[wv][threadIdx.x];
        if (x != 0.0) atomicAdd(&acc[threadIdx.x], x);
    }
}

// acc[0..2] += sum of smooth_l1(pred3|pred2|pred1 - gt) over valid pixels, acc[3] += count
__global__ void __launch_bounds__(DL_BLOCK)
disp_loss_fwd_kernel(double *__restrict__ acc, const float *__restrict__ p3, const float *__restrict__ p2,
                     const float *__restrict__ p1, const float *__restrict__ gt,
                     const unsigned char *__restrict__ mask, float lo, float hi, long long n) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (long long i = (long long)blockIdx.x * DL_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * DL_BLOCK) {
        const float g = gt[i];
        if (!dl_valid(mask, g, lo, hi, i)) continue;
        const float d3 = fabsf(p3[i] - g), d2 = fabsf(p2[i] - g), d1 = fabsf(p1[i] - g);
        v[0] += d3 < 1.f ? 0.5f * d3 * d3 : d3 - 0.5f;  // F.smooth_l1_loss, beta = 1
        v[1] += d2 < 1.f ? 0.5f * d2 * d2 : d2 - 0.5f;
        v[2] += d1 < 1.f ? 0.5f * d1 * d1 : d1 - 0.5f;
        v[3] += 1.0;
    }
    dl_flush<4>(v, acc);
}

// d loss / d pred_k = w_k * gloss / count * clamp(pred_k - gt, -1, 1) on valid pixels, 0 elsewhere
__global__ void __launch_bounds__(DL_BLOCK)
disp_loss_bwd_kernel(float *__restrict__ g3, float *__restrict__ g2, float *__restrict__ g1,
                     const float *__restrict__ p3, const float *__restrict__ p2, const float *__restrict__ p1,
                     const float *__restrict__ gt, const unsigned char *__restrict__ mask, float lo, float hi,
                     const float *__restrict__ gloss, const double *__restrict__ acc, float w3, float w2,
                     float w1, long long n) {
    const float s = gloss[0] / (float)acc[3];
    for (long long i = (long long)blockIdx.x * DL_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * DL_BLOCK) {
        const float g = gt[i];
        const bool ok = dl_valid(mask, g, lo, hi, i);
        g3[i] = ok ? w3 * s * fminf(fmaxf(p3[i] - g, -1.f), 1.f) : 0.f;
        g2[i] = ok ? w2 * s * fminf(fmaxf(p2[i] - g, -1.f), 1.f) : 0.f;
        g1[i] = ok ? w1 * s * fminf(fmaxf(p1[i] - g, -1.f), 1.f) : 0.f;
    }
}

// acc: 0 sum|dd|  1 #(|dd|>1)  2 #(|dd|>2)  3 sum clip(|1000 dz|,0,100)  4 #(|dz|>2e-3)
//      5 #(|dz|>4e-3)  6 #(|dz|>8e-3)  7 count      (dd = disparity error, dz = depth error in m)
__global__ void __launch_bounds__(DL_BLOCK)
disp_metrics_kernel(double *__restrict__ acc, const float *__restrict__ disp_gt,
                    const float *__restrict__ depth_gt, const float *__restrict__ disp_pred,
                    const float *__restrict__ depth_pred, const float *__restrict__ fb,
                    const unsigned char *__restrict__ mask, long long per_batch, long long n) {
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * DL_BLOCK + threadIdx.x; i < n; i += (long long)gridDim.x * DL_BLOCK) {
        if (!mask[i]) continue;
        const float dp = disp_pred[i];
        const float dd = fabsf(disp_gt[i] - dp);
        // cascade_metrics.py:36-37: depth_pred = focal_length * baseline / disp_pred (metres)
        const float zp = depth_pred ? depth_pred[i] : fb[i / per_batch] / dp;
        const float zg = depth_gt[i];
        const float dz = fabsf(zg - zp);
        const float dmm = fminf(fmaxf(fabsf(zg * 1000.f - zp * 1000.f), 0.f), 100.f);
        v[0] += dd;
        v[1] += dd > 1.f ? 1.0 : 0.0;
        v[2] += dd > 2.f ? 1.0 : 0.0;
        v[3] += dmm;
        v[4] += dz > 2e-3f ? 1.0 : 0.0;
        v[5] += dz > 4e-3f ? 1.0 : 0.0;
        v[6] += dz > 8e-3f ? 1.0 : 0.0;
        v[7] += 1.0;
    }
    dl_flush<8>(v, acc);
}

// reductions: at most 1024 blocks (4 per CU), each strides over the map
static unsigned dl_reduce_grid(long long n) {
    const unsigned g = az_grid_for(n, DL_BLOCK);
    return g > 1024u ? 1024u : g;
}

extern "C" int az_disp_loss_fwd(double *acc4, const float *pred3, const float *pred2, const float *pred1,
                                const float *gt, const unsigned char *mask, float lo, float hi,
                                long long n, void *stream) {
    AZ_REQUIRE_PTR(acc4); AZ_REQUIRE_PTR(pred3); AZ_REQUIRE_PTR(pred2); AZ_REQUIRE_PTR(pred1); AZ_REQUIRE_PTR(gt);
    AZ_REQUIRE(n >= 0);
    if (n == 0) return AZ_OK;
    hipLaunchKernelGGL(disp_loss_fwd_kernel, dim3(dl_reduce_grid(n)), dim3(DL_BLOCK), 0, az_stream(stream),
                       acc4, pred3, pred2, pred1, gt, mask, lo, hi, n);
    return az_launch_status();
}

extern "C" int az_disp_loss_bwd(float *g3, float *g2, float *g1, const float *pred3, const float *pred2,
                                const float *pred1, const float *gt, const unsigned char *mask, float lo,
                                float hi, const float *gloss, const double *acc4, float w3, float w2,
                                float w1, long long n, void *stream) {
    AZ_REQUIRE_PTR(g3); AZ_REQUIRE_PTR(g2); AZ_REQUIRE_PTR(g1);
    AZ_REQUIRE_PTR(pred3); AZ_REQUIRE_PTR(pred2); AZ_REQUIRE_PTR(pred1); AZ_REQUIRE_PTR(gt);
    AZ_REQUIRE_PTR(gloss); AZ_REQUIRE_PTR(acc4);
    AZ_REQUIRE(n >= 0);
    if (n == 0) return AZ_OK;
    hipLaunchKernelGGL(disp_loss_bwd_kernel, dim3(az_grid_for(n, DL_BLOCK)), dim3(DL_BLOCK), 0, az_stream(stream),
                       g3, g2, g1, pred3, pred2, pred1, gt, mask, lo, hi, gloss, acc4, w3, w2, w1, n);
    return az_launch_status();
}

extern "C" int az_disp_metrics(double *acc8, const float *disp_gt, const float *depth_gt,
                               const float *disp_pred, const float *depth_pred, const float *focal_x_baseline,
                               const unsigned char *mask, int B, long long per_batch, void *stream) {
    AZ_REQUIRE_PTR(acc8); AZ_REQUIRE_PTR(disp_gt); AZ_REQUIRE_PTR(depth_gt); AZ_REQUIRE_PTR(disp_pred);
    AZ_REQUIRE_PTR(mask);
    if (!depth_pred) AZ_REQUIRE_PTR(focal_x_baseline);
    AZ_REQUIRE(B >= 0 && per_batch >= 0);
    const long long n = (long long)B * per_batch;
    if (n == 0) return AZ_OK;
    hipLaunchKernelGGL(disp_metrics_kernel, dim3(dl_reduce_grid(n)), dim3(DL_BLOCK), 0, az_stream(stream),
                       acc8, disp_gt, depth_gt, disp_pred, depth_pred, focal_x_baseline, mask, per_batch, n);
    return az_launch_status();
}
