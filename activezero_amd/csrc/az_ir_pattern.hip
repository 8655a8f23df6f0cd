// K14 -- IR dot-pattern extraction on the GPU: reference datasets/dataset_utils.py:33-46
// get_smoothed_ir_pattern2(img_ir, img, ks = 11, threshold = 0.005), called per item by
// datasets/messytable.py:406-426 (__getpattern__, "p2" and the simulated "temporal" pattern):
//     diff  = |img_ir - img|, min-max normalised over the image
//     avg   = cv2.resize(cv2.resize(diff, (w//ks, h//ks), INTER_AREA), (w, h), INTER_AREA)
//     ir    = (diff - avg > threshold) ? 1 : 0
// cv2 (opencv-python 4.5.x, requirements.txt:68-69) is a third-party dependency that is absent here; its
// INTER_AREA is restated from the published algorithm (imgproc/resize.cpp): TRUE area averaging when
// shrinking (every destination cell is the area-weighted mean of the source pixels it covers: fractional
// weights at both cell borders), and -- when enlarging -- the "area mode" variant of bilinear
// interpolation: source index floor(dx * s), weight fx = (dx + 1) - (sx + 1) / s clipped to [0, 1) by its
// fractional part (0 for every destination pixel that lies inside one source cell).
// Three small kernels (HBM-bound, 540x960 images): min/max reduction, area shrink, enlarge + threshold.
#include "az_common.h"

__global__ void irp_init_kernel(unsigned *mm, int B) {  // (min, max) slots of every image
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < B) { mm[2 * i] = 0xffffffffu; mm[2 * i + 1] = 0u; }
}

__global__ void __launch_bounds__(256)
irp_minmax_kernel(unsigned *__restrict__ mm, const float *__restrict__ a, const float *__restrict__ b, int hw) {
    const float *pa = a + (size_t)blockIdx.y * hw, *pb = b + (size_t)blockIdx.y * hw;
    float mn = 3.4e38f, mx = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < hw; i += gridDim.x * 256) {
        const float d = fabsf(pa[i] - pb[i]);
        mn = fminf(mn, d);
        mx = fmaxf(mx, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    // non-negative floats order like their bit patterns
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[2 * blockIdx.y], __float_as_uint(mn));
        atomicMax(&mm[2 * blockIdx.y + 1], __float_as_uint(mx));
    }
}

// area weights of destination cell `d` along one axis (OpenCV computeResizeAreaTab): source range
// [s0, s0 + n), weight of source index s0 + k in w(k)
struct AreaSpan { int s0, n; float w_first, w_mid, w_last; bool has_first, has_last; int mid0, mid1; };
__device__ __forceinline__ AreaSpan area_span(int d, int ssize, double scale) {
    AreaSpan r;
    const double fs1 = d * scale, fs2 = fs1 + scale;
    const double cell = fmin(scale, ssize - fs1);
    int s1 = (int)ceil(fs1), s2 = (int)floor(fs2);
    s2 = min(s2, ssize - 1);
    s1 = min(s1, s2);
    r.has_first = (s1 - fs1 > 1e-3);
    r.w_first = (float)((s1 - fs1) / cell);
    r.mid0 = s1; r.mid1 = s2;
    r.w_mid = (float)(1.0 / cell);
    r.has_last = (fs2 - s2 > 1e-3);
    r.w_last = (float)(fmin(fmin(fs2 - s2, 1.0), cell) / cell);
    r.s0 = r.has_first ? s1 - 1 : s1;
    r.n = (r.has_last ? s2 + 1 : s2) - r.s0;
    return r;
}
__device__ __forceinline__ float area_w(const AreaSpan &r, int s) {
    if (r.has_first && s == r.mid0 - 1) return r.w_first;
    if (s >= r.mid0 && s < r.mid1) return r.w_mid;
    if (r.has_last && s == r.mid1) return r.w_last;
    return 0.f;
}

__global__ void __launch_bounds__(256)
irp_shrink_kernel(float *__restrict__ avg, const unsigned *__restrict__ mm, const float *__restrict__ a,
                  const float *__restrict__ b, int H, int W, int hs, int ws) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= hs * ws) return;
    const int cy = idx / ws, cx = idx - cy * ws, img = blockIdx.y;
    const float mn = __uint_as_float(mm[2 * img]), mx = __uint_as_float(mm[2 * img + 1]);
    const float inv = 1.f / (mx - mn);
    const AreaSpan sy = area_span(cy, H, (double)H / hs), sx = area_span(cx, W, (double)W / ws);
    const float *pa = a + (size_t)img * H * W, *pb = b + (size_t)img * H * W;
    // separable, x first (as resizeArea_ does): row sums weighted along x, then along y
    float acc = 0.f;
    for (int y = sy.s0; y < sy.s0 + sy.n; ++y) {
        float rowsum = 0.f;
        for (int x = sx.s0; x < sx.s0 + sx.n; ++x) {
            const float d = (fabsf(pa[y * W + x] - pb[y * W + x]) - mn) * inv;
            rowsum += d * area_w(sx, x);
        }
        acc += rowsum * area_w(sy, y);
    }
    avg[(size_t)img * hs * ws + idx] = acc;
}

// "area mode" enlarging: source index and weight of the upper neighbour
__device__ __forceinline__ void area_up(int d, int ssize, int dsize, int &s, float &f) {
    const double scale = (double)ssize / dsize, inv_scale = (double)dsize / ssize;
    s = (int)floor(d * scale);
    float fx = (float)((d + 1) - (s + 1) * inv_scale);
    fx = fx <= 0.f ? 0.f : fx - floorf(fx);
    if (s < 0) { fx = 0.f; s = 0; }
    if (s >= ssize - 1) { fx = 0.f; s = ssize - 1; }
    f = fx;
}

__global__ void __launch_bounds__(256)
irp_pattern_kernel(float *__restrict__ out, const float *__restrict__ avg, const unsigned *__restrict__ mm,
                   const float *__restrict__ a, const float *__restrict__ b, int H, int W, int hs, int ws,
                   float threshold) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= H * W) return;
    const int y = idx / W, x = idx - y * W, img = blockIdx.y;
    const float mn = __uint_as_float(mm[2 * img]), mx = __uint_as_float(mm[2 * img + 1]);
    const float d = (fabsf(a[(size_t)img * H * W + idx] - b[(size_t)img * H * W + idx]) - mn) / (mx - mn);
    int sy, sx;
    float fy, fx;
    area_up(y, hs, H, sy, fy);
    area_up(x, ws, W, sx, fx);
    const float *p = avg + (size_t)img * hs * ws;
    const int sy1 = min(sy + 1, hs - 1), sx1 = min(sx + 1, ws - 1);
    // horizontal pass first, then vertical (OpenCV's resizeGeneric_ order)
    const float r0 = p[sy * ws + sx] * (1.f - fx) + p[sy * ws + sx1] * fx;
    const float r1 = p[sy1 * ws + sx] * (1.f - fx) + p[sy1 * ws + sx1] * fx;
    const float v = r0 * (1.f - fy) + r1 * fy;
    out[(size_t)img * H * W + idx] = (d - v > threshold) ? 1.f : 0.f;
}

extern "C" long long az_ir_pattern_workspace(int B, int H, int W, int ks) {
    if (B <= 0 || H <= 0 || W <= 0 || ks <= 0 || H / ks <= 0 || W / ks <= 0) return AZ_EINVAL;
    return ((long long)B * 2 + (long long)B * (H / ks) * (W / ks)) * 4 + 64;
}

extern "C" int az_ir_pattern(float *pattern, float *workspace, long long workspace_bytes, const float *img_ir,
                             const float *img, int B, int H, int W, int ks, float threshold, void *stream) {
    AZ_REQUIRE_PTR(pattern); AZ_REQUIRE_PTR(workspace); AZ_REQUIRE_PTR(img_ir); AZ_REQUIRE_PTR(img);
    const long long need = az_ir_pattern_workspace(B, H, W, ks);
    if (need < 0) return AZ_EINVAL;
    if (workspace_bytes < need) return AZ_EWORKSPACE;
    if ((long long)H * W > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    hipStream_t s = az_stream(stream);
    unsigned *mm = reinterpret_cast<unsigned *>(workspace);
    float *avg = workspace + 16 + 2 * B;  // (16-float gap keeps the two regions on separate lines)
    hipLaunchKernelGGL(irp_init_kernel, dim3((B + 63) / 64), dim3(64), 0, s, mm, B);
    const int hs = H / ks, ws = W / ks, hw = H * W;
    hipLaunchKernelGGL(irp_minmax_kernel, dim3(min((hw + 255) / 256, 256), B), dim3(256), 0, s, mm, img_ir, img, hw);
    hipLaunchKernelGGL(irp_shrink_kernel, dim3((hs * ws + 255) / 256, B), dim3(256), 0, s, avg, mm, img_ir, img, H, W,
                       hs, ws);
    hipLaunchKernelGGL(irp_pattern_kernel, dim3((hw + 255) / 256, B), dim3(256), 0, s, pattern, avg, mm, img_ir, img,
                       H, W, hs, ws, threshold);
    return az_launch_status();
}
