// SPP branch upsampling of the feature extractor (psmnet_submodule_3.py:198-209: F.upsample(branch, (H, W), mode="bilinear")
// -- align_corners semantics of the reference's torch 1.10 call as restated in nets/psmnet/psmnet_submodule_3.py of this
// package -- followed by torch.cat): the four pooled 32-channel maps (2x3 ... 17x30 pixels) are interpolated to the
// 136x240 grid STRAIGHT INTO their channel slots of the 320-channel concat buffer.  Rounds 1-4 wrote the interpolation as
// two dense products Wy @ x @ Wx^T on rocBLAS (20 small GEMM launches per step, the last vendor-library kernels of the
// path, plus a layout copy per branch); it is a fixed 2 x 2-tap stencil, not a GEMM.
//   forward : one thread per (output pixel, 4 channels): four 16-byte reads of the tiny source (cache resident), one write;
//             ATen's expression and grouping:  h0 (w0 x00 + w1 x01) + h1 (w0 x10 + w1 x11),  src = dst * (in-1)/(out-1).
//   backward: the adjoint as two separable gathers (x, then y) with the weights recomputed from the forward's expressions;
//             no atomics: deterministic.
#include "az_common.h"

struct SppArgs {
    float *out;         // forward: rows [B,H,W,out_cs] (+ channel offset);  backward: gin rows [B,hs,ws,C]
    const float *in;    // forward: rows [B,hs,ws,C];                        backward: gout rows [B,H,W,out_cs] (+ offset)
    int B, hs, ws, H, W, C4, out_cs;
    float sy, sx;       // (in - 1) / (out - 1), 0 for out == 1
};

__device__ __forceinline__ void spp_src(float scale, int dst, int n_in, int &i0, int &ip, float &l0, float &l1) {
    const float src = scale * (float)dst;
    i0 = min((int)src, n_in - 1);
    ip = i0 < n_in - 1 ? 1 : 0;
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

__global__ void __launch_bounds__(256)
spp_upsample_fwd_kernel(const SppArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.B * a.H * a.W * a.C4;
    if (idx >= total) return;
    const int c4 = (int)(idx % a.C4);
    long long p = idx / a.C4;
    const int x = (int)(p % a.W); p /= a.W;
    const int y = (int)(p % a.H);
    const int b = (int)(p / a.H);
    int h1, h1p, w1, w1p;
    float h0l, h1l, w0l, w1l;
    spp_src(a.sy, y, a.hs, h1, h1p, h0l, h1l);
    spp_src(a.sx, x, a.ws, w1, w1p, w0l, w1l);
    const float4 *src = reinterpret_cast<const float4 *>(a.in) + ((size_t)b * a.hs * a.ws) * a.C4 + c4;
    const float4 x00 = src[((size_t)h1 * a.ws + w1) * a.C4], x01 = src[((size_t)h1 * a.ws + w1 + w1p) * a.C4];
    const float4 x10 = src[((size_t)(h1 + h1p) * a.ws + w1) * a.C4], x11 = src[((size_t)(h1 + h1p) * a.ws + w1 + w1p) * a.C4];
    float4 o;
    o.x = h0l * (w0l * x00.x + w1l * x01.x) + h1l * (w0l * x10.x + w1l * x11.x);
    o.y = h0l * (w0l * x00.y + w1l * x01.y) + h1l * (w0l * x10.y + w1l * x11.y);
    o.z = h0l * (w0l * x00.z + w1l * x01.z) + h1l * (w0l * x10.z + w1l * x11.z);
    o.w = h0l * (w0l * x00.w + w1l * x01.w) + h1l * (w0l * x10.w + w1l * x11.w);
    *reinterpret_cast<float4 *>(a.out + (((size_t)b * a.H + y) * a.W + x) * a.out_cs + 4 * c4) = o;
}

// weight of output index `dst` on source index `i` (both taps may land on i at the last source index)
__device__ __forceinline__ float spp_weight(float scale, int dst, int n_in, int i) {
    int i0, ip;
    float l0, l1;
    spp_src(scale, dst, n_in, i0, ip, l0, l1);
    return (i0 == i ? l0 : 0.f) + (i0 + ip == i ? l1 : 0.f);
}
// output indices that can touch source index i: src in (i - 1, i + 1)
__device__ __forceinline__ void spp_window(float scale, int i, int n_out, int &lo, int &hi) {
    lo = scale > 0.f ? max(0, (int)floorf((float)(i - 1) / scale) - 1) : 0;
    hi = scale > 0.f ? min(n_out - 1, (int)ceilf((float)(i + 1) / scale) + 1) : n_out - 1;
}

// The adjoint in two separable passes (a one-pass gather per source pixel made the 2x3 map's six workgroups walk 21 760
// output pixels each: 0.53 ms):  tmp[b, y, j, c] = sum_x wx(x, j) g[b, y, x, c]  -- one workgroup per (b, y, j) --, then
// gin[b, i, j, c] = sum_y wy(y, i) tmp[b, y, j, c].  No atomics: deterministic.  Thread = (window lane, channel quad).
template <int PASS>
__global__ void __launch_bounds__(256)
spp_upsample_bwd_kernel(const SppArgs a, float *__restrict__ tmp) {
    int r = blockIdx.x;
    const int j = r % a.ws; r /= a.ws;
    const int n1 = PASS == 0 ? a.H : a.hs;   // pass 0: (b, y, j); pass 1: (b, i, j)
    const int yi = r % n1;
    const int b = r / n1;
    const int c4 = threadIdx.x % a.C4, pl = threadIdx.x / a.C4, npl = 256 / a.C4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PASS == 0) {
        int lo, hi;
        spp_window(a.sx, j, a.W, lo, hi);
        const float *g = a.in + ((size_t)b * a.H + yi) * a.W * a.out_cs + 4 * c4;
        for (int x = lo + pl; x <= hi; x += npl) {
            const float w = spp_weight(a.sx, x, a.ws, j);
            const float4 v = *reinterpret_cast<const float4 *>(g + (size_t)x * a.out_cs);
            acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
        }
    } else {
        int lo, hi;
        spp_window(a.sy, yi, a.H, lo, hi);
        const float4 *t = reinterpret_cast<const float4 *>(tmp) + ((size_t)b * a.H * a.ws + j) * a.C4 + c4;
        for (int y = lo + pl; y <= hi; y += npl) {
            const float w = spp_weight(a.sy, y, a.hs, yi);
            const float4 v = t[(size_t)y * a.ws * a.C4];
            acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
        }
    }
    __shared__ float4 red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = npl >> 1; s > 0; s >>= 1) {  // (npl is a power of two: C4 in {1, 2, 4, 8, 16})
        if (pl < s) {
            const float4 o = red[threadIdx.x + s * a.C4];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
            red[threadIdx.x] = acc;
        }
        __syncthreads();
    }
    if (pl == 0) {
        float4 *dst = PASS == 0 ? reinterpret_cast<float4 *>(tmp) + (((size_t)b * a.H + yi) * a.ws + j) * a.C4 + c4
                                : reinterpret_cast<float4 *>(a.out) + (((size_t)b * a.hs + yi) * a.ws + j) * a.C4 + c4;
        *dst = acc;
    }
}

static int spp_args(SppArgs &a, int B, int hs, int ws, int H, int W, int C, int out_cs) {
    if (B <= 0 || hs <= 0 || ws <= 0 || H <= 0 || W <= 0) return AZ_EINVAL;
    if (C <= 0 || C % 4 || out_cs < C || out_cs % 4) return AZ_EINVAL;
    const int c4 = C / 4;
    if (c4 > 16 || (c4 & (c4 - 1))) return AZ_EUNSUPPORTED;
    a.B = B; a.hs = hs; a.ws = ws; a.H = H; a.W = W; a.C4 = c4; a.out_cs = out_cs;
    // fp32 source positions exactly as ATen computes them (area_pixel_compute_scale, align_corners = True)
    a.sy = H > 1 ? (float)(hs - 1) / (float)(H - 1) : 0.f;
    a.sx = W > 1 ? (float)(ws - 1) / (float)(W - 1) : 0.f;
    return AZ_OK;
}

extern "C" int az_spp_upsample_fwd(float *out, const float *in, int B, int hs, int ws, int H, int W, int C, int out_cstride,
                                   void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(in);
    SppArgs a{};
    if (int e = spp_args(a, B, hs, ws, H, W, C, out_cstride)) return e;
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(in)) & 15) return AZ_EINVAL;
    a.out = out; a.in = in;
    const long long total = (long long)B * H * W * a.C4;
    hipLaunchKernelGGL(spp_upsample_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, az_stream(stream), a);
    return az_launch_status();
}

extern "C" long long az_spp_upsample_bwd_workspace(int B, int ws, int H, int C) {
    if (B <= 0 || ws <= 0 || H <= 0 || C <= 0) return AZ_EINVAL;
    return (long long)B * H * ws * C * (long long)sizeof(float);
}

extern "C" int az_spp_upsample_bwd(float *grad_in, float *workspace, long long workspace_bytes, const float *grad_out, int B,
                                   int hs, int ws, int H, int W, int C, int gout_cstride, void *stream) {
    AZ_REQUIRE_PTR(grad_in); AZ_REQUIRE_PTR(grad_out); AZ_REQUIRE_PTR(workspace);
    SppArgs a{};
    if (int e = spp_args(a, B, hs, ws, H, W, C, gout_cstride)) return e;
    if ((reinterpret_cast<uintptr_t>(grad_in) | reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(workspace)) & 15) return AZ_EINVAL;
    if (workspace_bytes < az_spp_upsample_bwd_workspace(B, ws, H, C)) return AZ_EWORKSPACE;
    a.out = grad_in; a.in = grad_out;
    hipLaunchKernelGGL(spp_upsample_bwd_kernel<0>, dim3((unsigned)(B * H * ws)), dim3(256), 0, az_stream(stream), a, workspace);
    hipLaunchKernelGGL(spp_upsample_bwd_kernel<1>, dim3((unsigned)(B * hs * ws)), dim3(256), 0, az_stream(stream), a, workspace);
    return az_launch_status();
}
