// K13 helpers -- the extractor's first convolution (reference nets/psmnet/psmnet_submodule_3.py:97-99:
// convbn(3 or 6, 32, 3, stride 2, pad 1)) has K = 27 / 54 per output pixel: far too thin for an MFMA
// slab pipeline of its own.  It runs as patch extraction + a 1x1 convolution on the MFMA kernel
// (az_conv2d.hip): P[b,oy,ox, t*C + c] = x[b, 2oy-1+kh, 2ox-1+kw, c] (t = kh*3+kw, zero outside the image,
// zero in the padding channels [9C, Kp)), HBM-bound.  The adjoint (input gradient, only needed when the
// image itself is a network output: the 6-channel variant fed by the adapter, psmnet.py:144-148) gathers
// the <= 4 patch entries that hold a given input pixel.
#include "az_common.h"

__global__ void __launch_bounds__(256)
im2col_s2k3_kernel(float *__restrict__ P, const float *__restrict__ x, int C, int H, int W, int Ho, int Wo,
                   int Kp, long long total) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int k = (int)(idx % Kp);
        long long r = idx / Kp;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho);
        const int b = (int)(r / Ho);
        float v = 0.f;
        if (k < 9 * C) {
            const int t = k / C, c = k - t * C;
            const int iy = 2 * oy - 1 + t / 3, ix = 2 * ox - 1 + t % 3;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[(((size_t)b * H + iy) * W + ix) * C + c];
        }
        P[idx] = v;
    }
}

__global__ void __launch_bounds__(256)
col2im_s2k3_kernel(float *__restrict__ gx, const float *__restrict__ gP, int C, int H, int W, int Ho, int Wo,
                   int Kp, long long total) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % C);
        long long r = idx / C;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int b = (int)(r / H);
        float s = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ny = iy + 1 - kh;  // = 2 * oy
            if (ny < 0 || (ny & 1) || (ny >> 1) >= Ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int nx = ix + 1 - kw;
                if (nx < 0 || (nx & 1) || (nx >> 1) >= Wo) continue;
                s += gP[(((size_t)b * Ho + (ny >> 1)) * Wo + (nx >> 1)) * Kp + (kh * 3 + kw) * C + c];
            }
        }
        gx[idx] = s;
    }
}

extern "C" int az_im2col_s2k3(float *patches, const float *x, int B, int C, int H, int W, int Kp, void *stream) {
    AZ_REQUIRE_PTR(patches); AZ_REQUIRE_PTR(x);
    AZ_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Kp >= 9 * C);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)B * Ho * Wo * Kp;
    hipLaunchKernelGGL(im2col_s2k3_kernel, dim3(az_grid_for(total, 256)), dim3(256), 0, az_stream(stream), patches,
                       x, C, H, W, Ho, Wo, Kp, total);
    return az_launch_status();
}

extern "C" int az_col2im_s2k3(float *grad_x, const float *grad_patches, int B, int C, int H, int W, int Kp,
                              void *stream) {
    AZ_REQUIRE_PTR(grad_x); AZ_REQUIRE_PTR(grad_patches);
    AZ_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && Kp >= 9 * C);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)B * H * W * C;
    hipLaunchKernelGGL(col2im_s2k3_kernel, dim3(az_grid_for(total, 256)), dim3(256), 0, az_stream(stream), grad_x,
                       grad_patches, C, H, W, Ho, Wo, Kp, total);
    return az_launch_status();
}
