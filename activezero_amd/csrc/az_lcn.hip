// K9 -- local contrast normalisation, reference utils/reprojection.py:175-200:
// Unfold(k, zero padding) -> mean / std(unbiased=False) over the k*k window ->
// (x - mean) / (std + eps).  A 16x16 block stages its (16+k-1)^2 zero-padded
// window in LDS; every thread makes the two passes (mean, then centred sum of
// squares -- the two-pass form keeps flat regions exact) over LDS only, so HBM
// sees the image once in and the two outputs once out.
#include "az_common.h"

__global__ void __launch_bounds__(256)
lcn_kernel(float *__restrict__ normed, float *__restrict__ stdv, const float *__restrict__ img,
           int H, int W, int k, float eps, long long in_stride) {
    extern __shared__ float tile[];
    const int r = k / 2, tw = 16 + k - 1;
    const int bx = blockIdx.x * 16, by = blockIdx.y * 16;
    const float *src = img + (size_t)blockIdx.z * in_stride;
    for (int e = threadIdx.x; e < tw * tw; e += 256) {
        const int ly = e / tw, lx = e % tw;
        const int y = by + ly - r, x = bx + lx - r;
        tile[e] = (y >= 0 && y < H && x >= 0 && x < W) ? src[(size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x = bx + lx, y = by + ly;
    if (x >= W || y >= H) return;
    const float n = (float)(k * k);  // divide like torch.mean does: a flat window stays exact
    float sum = 0.f;
    for (int u = 0; u < k; ++u)
        for (int v = 0; v < k; ++v) sum += tile[(ly + u) * tw + lx + v];
    const float mean = sum / n;
    float ss = 0.f;
    for (int u = 0; u < k; ++u)
        for (int v = 0; v < k; ++v) {
            const float d = tile[(ly + u) * tw + lx + v] - mean;
            ss += d * d;
        }
    const float sd = sqrtf(ss / n);
    const size_t o = (size_t)blockIdx.z * H * W + (size_t)y * W + x;
    normed[o] = (tile[(ly + r) * tw + lx + r] - mean) / (sd + eps);
    stdv[o] = sd;
}

extern "C" int az_lcn(float *normed, float *stdv, const float *img, int B, int H, int W,
                      int ksize, float eps, long long img_batch_stride, void *stream) {
    AZ_REQUIRE_PTR(normed); AZ_REQUIRE_PTR(stdv); AZ_REQUIRE_PTR(img);
    AZ_REQUIRE(B > 0 && H > 0 && W > 0 && ksize > 0 && (ksize & 1));
    const int tw = 16 + ksize - 1;
    if ((size_t)tw * tw * sizeof(float) > 64 * 1024 || B > 65535) return AZ_EUNSUPPORTED;
    hipLaunchKernelGGL(lcn_kernel, dim3((W + 15) / 16, (H + 15) / 16, B), dim3(256),
                       (size_t)tw * tw * sizeof(float), az_stream(stream), normed, stdv, img, H, W,
                       ksize, eps, img_batch_stride);
    return az_launch_status();
}
