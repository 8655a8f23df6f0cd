// K3 -- PSMNet concat cost volume, reference nets/psmnet/psmnet_3.py:149-163.
//
// NCDHW (the reference's layout, kept for the drop-in API and parity tests):
//   one block per (b, c2, y) feature row: the row (w floats of L or R) is
//   staged once in LDS and written d times, shifted, one wavefront per
//   disparity plane -> every store is a contiguous row, every feature byte is
//   read from HBM once.  Traffic = 4*(2C*h*w + 2C*d*h*w) bytes per sample.
// NDHWC (channels-last, what the 3-D aggregation kernels consume):
//   a voxel is 2C contiguous floats; one float4 lane per 4 channels.
// Backward = the adjoint reductions over the disparity axis.
#include "az_common.h"

__global__ void __launch_bounds__(256)
cost_volume_fwd_ncdhw(float *__restrict__ cost, const float *__restrict__ fl,
                      const float *__restrict__ fr, int C, int d, int h, int w) {
    extern __shared__ float row[];
    // blockIdx.x = (b * 2C + c2) * h + y
    const int y = blockIdx.x % h;
    const int bc = blockIdx.x / h;
    const int c2 = bc % (2 * C), b = bc / (2 * C);
    const bool right = c2 >= C;
    const float *src = (right ? fr : fl) + (((size_t)b * C + (right ? c2 - C : c2)) * h + y) * w;
    for (int x = threadIdx.x; x < w; x += blockDim.x) row[x] = src[x];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *out = cost + ((size_t)bc * d * h + y) * w;  // plane i adds i*h*w
    const size_t plane = (size_t)h * w;
    if ((w & 3) == 0) {
        const int w4 = w >> 2;
        for (int i = wave; i < d; i += 4) {
            float4 *o4 = reinterpret_cast<float4 *>(out + i * plane);
            for (int q = lane; q < w4; q += 64) {
                const int x = q * 4;
                float4 v;
                if (right) {
                    v.x = (x + 0 >= i) ? row[x + 0 - i] : 0.f;
                    v.y = (x + 1 >= i) ? row[x + 1 - i] : 0.f;
                    v.z = (x + 2 >= i) ? row[x + 2 - i] : 0.f;
                    v.w = (x + 3 >= i) ? row[x + 3 - i] : 0.f;
                } else {
                    v.x = (x + 0 >= i) ? row[x + 0] : 0.f;
                    v.y = (x + 1 >= i) ? row[x + 1] : 0.f;
                    v.z = (x + 2 >= i) ? row[x + 2] : 0.f;
                    v.w = (x + 3 >= i) ? row[x + 3] : 0.f;
                }
                o4[q] = v;
            }
        }
    } else {
        for (int i = wave; i < d; i += 4)
            for (int x = lane; x < w; x += 64)
                out[i * plane + x] = (x >= i) ? row[right ? x - i : x] : 0.f;
    }
}

__global__ void __launch_bounds__(256)
cost_volume_bwd_ncdhw(float *__restrict__ gl, float *__restrict__ gr,
                      const float *__restrict__ g, int C, int d, int h, int w, long long total) {
    // one thread per output element of grad_l / grad_r: index over [B,2C,h,w]
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = idx % w;
        const long long r = idx / w;
        const int y = r % h;
        const long long bc = r / h;
        const int c2 = bc % (2 * C);
        const long long b = bc / (2 * C);
        const float *gp = g + ((size_t)bc * d * h + y) * w;
        const size_t plane = (size_t)h * w;
        float acc = 0.f;
        if (c2 < C) {
            const int lim = min(d - 1, x);
            for (int i = 0; i <= lim; ++i) acc += gp[i * plane + x];
            gl[(((size_t)b * C + c2) * h + y) * w + x] = acc;
        } else {
            const int lim = min(d - 1, w - 1 - x);
            for (int i = 0; i <= lim; ++i) acc += gp[i * plane + x + i];
            gr[(((size_t)b * C + (c2 - C)) * h + y) * w + x] = acc;
        }
    }
}

// ---- channels-last ----------------------------------------------------------
__global__ void __launch_bounds__(256)
cost_volume_fwd_ndhwc(float4 *__restrict__ cost, const float4 *__restrict__ fl,
                      const float4 *__restrict__ fr, int C4, int d, int h, int w, long long rows) {
    // one (b, i, y) output row per block iteration: w voxels of 2*C4 float4, written as one
    // contiguous stream; no per-element index decode
    const int per = 2 * C4;  // float4 per voxel
    for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
        const int y = (int)(r % h);
        const long long t = r / h;
        const int i = (int)(t % d);
        const long long b = t / d;
        const float4 *lrow = fl + ((size_t)b * h + y) * w * C4;
        const float4 *rrow = fr + ((size_t)b * h + y) * w * C4;
        float4 *orow = cost + (size_t)r * w * per;
        for (int e = threadIdx.x; e < w * per; e += 256) {
            const int x = e / per, c = e - x * per;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (x >= i) v = (c < C4) ? lrow[x * C4 + c] : rrow[(x - i) * C4 + (c - C4)];
            orow[e] = v;
        }
    }
}

__global__ void __launch_bounds__(256)
cost_volume_bwd_ndhwc(float4 *__restrict__ gl, float4 *__restrict__ gr,
                      const float4 *__restrict__ g, int C4, int d, int h, int w,
                      long long total4) {
    // index over [B, h, w, 2*C4]; c < C4 -> grad_l, else grad_r
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total4;
         idx += (long long)gridDim.x * blockDim.x) {
        const int c = idx % (2 * C4);
        long long r = idx / (2 * C4);
        const int x = r % w; r /= w;
        const int y = r % h;
        const long long b = r / h;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const size_t vstride = (size_t)2 * C4;
        if (c < C4) {
            const int lim = min(d - 1, x);
            for (int i = 0; i <= lim; ++i) {
                const float4 t = g[((((size_t)b * d + i) * h + y) * w + x) * vstride + c];
                acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            }
            gl[(((size_t)b * h + y) * w + x) * C4 + c] = acc;
        } else {
            const int lim = min(d - 1, w - 1 - x);
            for (int i = 0; i <= lim; ++i) {
                const float4 t = g[((((size_t)b * d + i) * h + y) * w + x + i) * vstride + c];
                acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            }
            gr[(((size_t)b * h + y) * w + x) * C4 + (c - C4)] = acc;
        }
    }
}

static int check_dims(int B, int C, int d, int h, int w) {
    if (!(B > 0 && C > 0 && d > 0 && h > 0 && w > 0)) return AZ_EINVAL;
    return AZ_OK;
}

extern "C" int az_cost_volume_fwd(float *cost, const float *fl, const float *fr, int B, int C,
                                  int d, int h, int w, void *stream) {
    AZ_REQUIRE_PTR(cost); AZ_REQUIRE_PTR(fl); AZ_REQUIRE_PTR(fr);
    if (int e = check_dims(B, C, d, h, w)) return e;
    if ((size_t)w * sizeof(float) > 64 * 1024) return AZ_EUNSUPPORTED;
    const long long blocks = (long long)B * 2 * C * h;
    if (blocks > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    hipLaunchKernelGGL(cost_volume_fwd_ncdhw, dim3((unsigned)blocks), dim3(256),
                       (size_t)w * sizeof(float), az_stream(stream), cost, fl, fr, C, d, h, w);
    return az_launch_status();
}

extern "C" int az_cost_volume_bwd(float *gl, float *gr, const float *g, int B, int C, int d,
                                  int h, int w, void *stream) {
    AZ_REQUIRE_PTR(gl); AZ_REQUIRE_PTR(gr); AZ_REQUIRE_PTR(g);
    if (int e = check_dims(B, C, d, h, w)) return e;
    const long long total = (long long)B * 2 * C * h * w;
    hipLaunchKernelGGL(cost_volume_bwd_ncdhw, dim3(az_grid_for(total, 256)), dim3(256), 0,
                       az_stream(stream), gl, gr, g, C, d, h, w, total);
    return az_launch_status();
}

extern "C" int az_cost_volume_fwd_ndhwc(float *cost, const float *fl, const float *fr, int B,
                                        int C, int d, int h, int w, void *stream) {
    AZ_REQUIRE_PTR(cost); AZ_REQUIRE_PTR(fl); AZ_REQUIRE_PTR(fr);
    if (int e = check_dims(B, C, d, h, w)) return e;
    if (C % 4) return AZ_EUNSUPPORTED;
    const long long rows = (long long)B * d * h;
    hipLaunchKernelGGL(cost_volume_fwd_ndhwc, dim3((unsigned)(rows < 8192 ? rows : 8192)), dim3(256), 0,
                       az_stream(stream), reinterpret_cast<float4 *>(cost),
                       reinterpret_cast<const float4 *>(fl),
                       reinterpret_cast<const float4 *>(fr), C / 4, d, h, w, rows);
    return az_launch_status();
}

extern "C" int az_cost_volume_bwd_ndhwc(float *gl, float *gr, const float *g, int B, int C,
                                        int d, int h, int w, void *stream) {
    AZ_REQUIRE_PTR(gl); AZ_REQUIRE_PTR(gr); AZ_REQUIRE_PTR(g);
    if (int e = check_dims(B, C, d, h, w)) return e;
    if (C % 4) return AZ_EUNSUPPORTED;
    const long long total4 = (long long)B * h * w * (2 * C / 4);
    hipLaunchKernelGGL(cost_volume_bwd_ndhwc, dim3(az_grid_for(total4, 256)), dim3(256), 0,
                       az_stream(stream), reinterpret_cast<float4 *>(gl),
                       reinterpret_cast<float4 *>(gr), reinterpret_cast<const float4 *>(g),
                       C / 4, d, h, w, total4);
    return az_launch_status();
}
