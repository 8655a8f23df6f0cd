// K7 -- bilinear gather warp (apply_disparity), reference
// utils/reprojection.py:13-35: x_base/y_base = linspace(0,1,W|H), grid =
// 2*(base + disp/W) - 1, F.grid_sample(bilinear, zeros, align_corners=False).
//
// The sampled pixel coordinate is ix = ((g+1)*W - 1)/2 with g the normalised
// grid value, i.e. ix = j*W/(W-1) + disp - 0.5, iy = i*H/(H-1) - 0.5 (the quirk
// of feeding a [0,1] linspace to an align_corners=False sampler).  The kernel
// reproduces the reference's fp32 operation order (CPU linspace formula,
// disp/W, +, 2*x-1, unnormalise) so the coordinates round the same way.
// One thread per (b, y, x) loops over the C channels; loads and stores are
// coalesced along x.  Bytes: 4*H*W*(2C+1).
#include "az_common.h"

// torch.linspace(0, 1, n)[i] as the CPU kernel computes it in fp32
__device__ __forceinline__ float linspace01(int i, int n) {
    const float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? (step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

struct Taps {
    int x0, y0;          // north-west tap; the others are +1
    float wx1, wy1;      // fractional parts
};

__device__ __forceinline__ Taps make_taps(int i, int j, float disp, int H, int W) {
#pragma clang fp contract(off)
    const float gx = linspace01(j, W) + disp / (float)W;
    const float gy = linspace01(i, H);
    const float nx = 2.0f * gx - 1.0f, ny = 2.0f * gy - 1.0f;
    const float ix = ((nx + 1.0f) * (float)W - 1.0f) / 2.0f;
    const float iy = ((ny + 1.0f) * (float)H - 1.0f) / 2.0f;
    const float fx = floorf(ix), fy = floorf(iy);
    Taps t;
    // clamp before the int conversion: far out-of-range coordinates only need
    // to stay out of range.
    t.x0 = (int)fminf(fmaxf(fx, -2.0f), (float)W + 1.0f);
    t.y0 = (int)fminf(fmaxf(fy, -2.0f), (float)H + 1.0f);
    t.wx1 = ix - fx;
    t.wy1 = iy - fy;
    return t;
}

__global__ void __launch_bounds__(256)
warp_gather_fwd_kernel(float *__restrict__ out, const float *__restrict__ img,
                       const float *__restrict__ disp, int C, int H, int W, long long total) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int j = idx % W;
        const long long r = idx / W;
        const int i = r % H;
        const long long b = r / H;
        const Taps t = make_taps(i, j, disp[idx], H, W);
        const bool vx0 = t.x0 >= 0 && t.x0 < W, vx1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
        const bool vy0 = t.y0 >= 0 && t.y0 < H, vy1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
        const float wnw = (1.f - t.wx1) * (1.f - t.wy1), wne = t.wx1 * (1.f - t.wy1);
        const float wsw = (1.f - t.wx1) * t.wy1, wse = t.wx1 * t.wy1;
        const size_t plane = (size_t)H * W;
        const float *p = img + (size_t)b * C * plane;
        float *o = out + (size_t)b * C * plane + (size_t)i * W + j;
        const long long o00 = (long long)t.y0 * W + t.x0;
        for (int c = 0; c < C; ++c, p += plane, o += plane) {
            float acc = 0.f;
            if (vy0 && vx0) acc += p[o00] * wnw;
            if (vy0 && vx1) acc += p[o00 + 1] * wne;
            if (vy1 && vx0) acc += p[o00 + W] * wsw;
            if (vy1 && vx1) acc += p[o00 + W + 1] * wse;
            *o = acc;
        }
    }
}

__global__ void __launch_bounds__(256)
warp_gather_bwd_kernel(float *__restrict__ gdisp, float *__restrict__ gimg,
                       const float *__restrict__ gout, const float *__restrict__ img,
                       const float *__restrict__ disp, int C, int H, int W, long long total) {
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int j = idx % W;
        const long long r = idx / W;
        const int i = r % H;
        const long long b = r / H;
        const Taps t = make_taps(i, j, disp[idx], H, W);
        const bool vx0 = t.x0 >= 0 && t.x0 < W, vx1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
        const bool vy0 = t.y0 >= 0 && t.y0 < H, vy1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
        const float wy0 = 1.f - t.wy1, wx0 = 1.f - t.wx1;
        const size_t plane = (size_t)H * W;
        const float *p = img + (size_t)b * C * plane;
        float *gi = gimg ? gimg + (size_t)b * C * plane : nullptr;
        const float *go = gout + (size_t)b * C * plane + (size_t)i * W + j;
        const long long o00 = (long long)t.y0 * W + t.x0;
        float gix = 0.f;
        for (int c = 0; c < C; ++c, p += plane, go += plane) {
            const float g = *go;
            const float nw = (vy0 && vx0) ? p[o00] : 0.f, ne = (vy0 && vx1) ? p[o00 + 1] : 0.f;
            const float sw = (vy1 && vx0) ? p[o00 + W] : 0.f, se = (vy1 && vx1) ? p[o00 + W + 1] : 0.f;
            gix += g * ((ne - nw) * wy0 + (se - sw) * t.wy1);
            if (gi) {
                if (vy0 && vx0) atomicAdd(gi + o00, g * wx0 * wy0);
                if (vy0 && vx1) atomicAdd(gi + o00 + 1, g * t.wx1 * wy0);
                if (vy1 && vx0) atomicAdd(gi + o00 + W, g * wx0 * t.wy1);
                if (vy1 && vx1) atomicAdd(gi + o00 + W + 1, g * t.wx1 * t.wy1);
                gi += plane;
            }
        }
        // d ix / d disp = (W/2) * 2 * (1/W) = 1
        gdisp[idx] = gix;
    }
}

extern "C" int az_warp_gather_fwd(float *out, const float *img, const float *disp, int B, int C,
                                  int H, int W, void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(img); AZ_REQUIRE_PTR(disp);
    AZ_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1);
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(warp_gather_fwd_kernel, dim3(az_grid_for(total, 256)), dim3(256), 0,
                       az_stream(stream), out, img, disp, C, H, W, total);
    return az_launch_status();
}

extern "C" int az_warp_gather_bwd(float *grad_disp, float *grad_img, const float *grad_out,
                                  const float *img, const float *disp, int B, int C, int H,
                                  int W, void *stream) {
    AZ_REQUIRE_PTR(grad_disp); AZ_REQUIRE_PTR(grad_out); AZ_REQUIRE_PTR(img);
    AZ_REQUIRE_PTR(disp);
    AZ_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1);
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(warp_gather_bwd_kernel, dim3(az_grid_for(total, 256)), dim3(256), 0,
                       az_stream(stream), grad_disp, grad_img, grad_out, img, disp, C, H, W,
                       total);
    return az_launch_status();
}
