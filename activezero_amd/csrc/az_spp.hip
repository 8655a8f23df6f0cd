// SPP branch upsampling of the feature extractor (psmnet_submodule_3.py:198-209: F.upsample(branch, (H, W), mode="bilinear")
// -- align_corners semantics of the reference's torch 1.10 call as restated in nets/psmnet/psmnet_submodule_3.py of this
// package -- followed by torch.cat): the four pooled 32-channel maps (2x3 ... 17x30 pixels) are interpolated to the
// 136x240 grid STRAIGHT INTO their channel slots of the 320-channel concat buffer.  Rounds 1-4 wrote the interpolation as
// two dense products Wy @ x @ Wx^T on rocBLAS (20 small GEMM launches per step, the last vendor-library kernels of the
// path, plus a layout copy per branch); it is a fixed 2 x 2-tap stencil, not a GEMM.
//   forward : one thread per (output pixel, 4 channels): four 16-byte reads of the tiny source (cache resident), one write;
//             ATen's expression and grouping:  h0 (w0 x00 + w1 x01) + h1 (w0 x10 + w1 x11),  src = dst * (in-1)/(out-1).
//   backward: one workgroup per SOURCE pixel gathers its window of the output gradient (no atomics: deterministic), the
//             weights recomputed with the forward's expressions.
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

__global__ void __launch_bounds__(256)
spp_upsample_bwd_kernel(const SppArgs a) {
    // block -> (b, source row i, source column j); thread -> (window pixel lane, channel quad); C4 <= 16, 256 % C4 == 0
    int r = blockIdx.x;
    const int j = r % a.ws; r /= a.ws;
    const int i = r % a.hs;
    const int b = r / a.hs;
    const int c4 = threadIdx.x % a.C4, pl = threadIdx.x / a.C4, npl = 256 / a.C4;
    // output rows / columns that can touch source (i, j): src in (i - 1, i + 1)
    const int y_lo = a.sy > 0.f ? max(0, (int)floorf((float)(i - 1) / a.sy) - 1) : 0;
    const int y_hi = a.sy > 0.f ? min(a.H - 1, (int)ceilf((float)(i + 1) / a.sy) + 1) : a.H - 1;
    const int x_lo = a.sx > 0.f ? max(0, (int)floorf((float)(j - 1) / a.sx) - 1) : 0;
    const int x_hi = a.sx > 0.f ? min(a.W - 1, (int)ceilf((float)(j + 1) / a.sx) + 1) : a.W - 1;
    const int nx = x_hi - x_lo + 1, npx = (y_hi - y_lo + 1) * nx;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *g = a.in + (size_t)b * a.H * a.W * a.out_cs + 4 * c4;
    for (int p = pl; p < npx; p += npl) {
        const int y = y_lo + p / nx, x = x_lo + p % nx;
        // (the two weights kept apart: hipcc otherwise pairs their arithmetic into v_pk_mul_f32 with a high-half src1
        //  selection, the operand form that is banned beside MFMA waves -- tools/isa_lint.py, profiles/r03_pkfma_corun.md)
        float wy = spp_weight(a.sy, y, a.hs, i);
        asm volatile("" : "+v"(wy));
        const float w = wy * spp_weight(a.sx, x, a.ws, j);
        if (w != 0.f) {
            const float4 v = *reinterpret_cast<const float4 *>(g + ((size_t)y * a.W + x) * a.out_cs);
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
    if (pl == 0) reinterpret_cast<float4 *>(a.out)[(((size_t)b * a.hs + i) * a.ws + j) * a.C4 + c4] = acc;
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

extern "C" int az_spp_upsample_bwd(float *grad_in, const float *grad_out, int B, int hs, int ws, int H, int W, int C,
                                   int gout_cstride, void *stream) {
    AZ_REQUIRE_PTR(grad_in); AZ_REQUIRE_PTR(grad_out);
    SppArgs a{};
    if (int e = spp_args(a, B, hs, ws, H, W, C, gout_cstride)) return e;
    if ((reinterpret_cast<uintptr_t>(grad_in) | reinterpret_cast<uintptr_t>(grad_out)) & 15) return AZ_EINVAL;
    a.out = grad_in; a.in = grad_out;
    hipLaunchKernelGGL(spp_upsample_bwd_kernel, dim3((unsigned)(B * hs * ws)), dim3(256), 0, az_stream(stream), a);
    return az_launch_status();
}
