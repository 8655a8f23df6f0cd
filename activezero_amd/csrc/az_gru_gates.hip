// Gate arithmetic of the RAFT-Stereo ConvGRU update in TRAINING (reference nets/raft/update.py:32-41 under autograd;
// SURVEY.md 8 f-3) as five channels-last streaming kernels between the four convolutions of a step -- the torch operators
// they replace (sigmoid / tanh / mul / add / cat / contiguous and their autograd nodes: about thirty elementwise launches over
// 67-200 MB tensors per update, 22 updates per step) were 70 % of the RAFT workload's step.
//
//   forward   hx = [h | x]                                      (assembled by the caller)
//             zr = sigmoid(conv_zr(hx) + b + [cz | cr])          az_conv2d_bf16_fwd, act 2
//             rhx = [r * h | x]                                  az_gru_rh
//             q = tanh(conv_q(rhx) + b + cq)                     az_conv2d_bf16_fwd, act 3
//             h' = (1 - z) h + z q                               az_gru_out
//   backward  g = dL/dh'
//             dq_pre = g z (1 - q^2),  dzr[:hid] = g (q - h) z (1 - z),  dh_acc = g (1 - z)        az_gru_bwd1
//             d_rhx = conv_q^T(dq_pre)
//             dzr[hid:] = d_rhx[:hid] h r (1 - r),  dh_acc += d_rhx[:hid] r                        az_gru_bwd2
//             d_hx = conv_zr^T(dzr)
//             dh = dh_acc + d_hx[:hid],  dx = d_rhx[hid:] + d_hx[hid:]                             az_gru_bwd3
// All tensors are dense [npix][channels] fp32 rows; hid and inp are multiples of 4.
#include "az_common.h"

#define GG_GRID(n4) az_grid_for((n4), 256)

__global__ void __launch_bounds__(256)
gru_rh_kernel(float4 *__restrict__ rhx, const float4 *__restrict__ zr, const float4 *__restrict__ hx, long long npix, int hid4, int inp4) {
    const int ct4 = hid4 + inp4;
    const long long total = npix * ct4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
        const long long p = i / ct4;
        const int c = (int)(i - p * ct4);
        float4 v = hx[i];
        if (c < hid4) {
            const float4 r = zr[p * (2 * hid4) + hid4 + c];
            v.x *= r.x; v.y *= r.y; v.z *= r.z; v.w *= r.w;
        }
        rhx[i] = v;
    }
}

__global__ void __launch_bounds__(256)
gru_out_kernel(float4 *__restrict__ hn, const float4 *__restrict__ zr, const float4 *__restrict__ q, const float4 *__restrict__ hx,
               long long npix, int hid4, int inp4) {
    const long long total = npix * hid4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
        const long long p = i / hid4;
        const int c = (int)(i - p * hid4);
        const float4 z = zr[p * (2 * hid4) + c], h = hx[p * (hid4 + inp4) + c], qq = q[i];
        hn[i] = make_float4((1.f - z.x) * h.x + z.x * qq.x, (1.f - z.y) * h.y + z.y * qq.y,
                            (1.f - z.z) * h.z + z.z * qq.z, (1.f - z.w) * h.w + z.w * qq.w);
    }
}

__global__ void __launch_bounds__(256)
gru_bwd1_kernel(float4 *__restrict__ dq_pre, float4 *__restrict__ dzr, float4 *__restrict__ dh_acc, const float4 *__restrict__ g,
                const float4 *__restrict__ zr, const float4 *__restrict__ q, const float4 *__restrict__ hx, long long npix, int hid4, int inp4,
                unsigned *__restrict__ dq_amax, unsigned *__restrict__ dzr_amax) {
    // (dq_amax / dzr_amax, both or neither, zero before the call: max |dq_pre| and -- together with az_gru_bwd2 -- max |dzr|, the
    //  power-of-two operand scales of the f16x1 gradient convolutions that read them next)
    unsigned am_q = 0, am_z = 0;
    const long long total = npix * hid4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
        const long long p = i / hid4;
        const int c = (int)(i - p * hid4);
        const float4 z = zr[p * (2 * hid4) + c], h = hx[p * (hid4 + inp4) + c], qq = q[i], gg = g[i];
        const float4 dq = make_float4(gg.x * z.x * (1.f - qq.x * qq.x), gg.y * z.y * (1.f - qq.y * qq.y),
                                      gg.z * z.z * (1.f - qq.z * qq.z), gg.w * z.w * (1.f - qq.w * qq.w));
        const float4 dz = make_float4(gg.x * (qq.x - h.x) * z.x * (1.f - z.x), gg.y * (qq.y - h.y) * z.y * (1.f - z.y),
                                      gg.z * (qq.z - h.z) * z.z * (1.f - z.z), gg.w * (qq.w - h.w) * z.w * (1.f - z.w));
        dq_pre[i] = dq;
        dzr[p * (2 * hid4) + c] = dz;
        dh_acc[i] = make_float4(gg.x * (1.f - z.x), gg.y * (1.f - z.y), gg.z * (1.f - z.z), gg.w * (1.f - z.w));
        az_amax_acc(am_q, dq);
        az_amax_acc(am_z, dz);
    }
    if (dq_amax) {
        az_amax_flush(dq_amax, am_q);
        __syncthreads();  // (az_amax_flush's staging words are shared by its calls)
        az_amax_flush(dzr_amax, am_z);
    }
}

__global__ void __launch_bounds__(256)
gru_bwd2_kernel(float4 *__restrict__ dzr, float4 *__restrict__ dh_acc, const float4 *__restrict__ d_rhx, const float4 *__restrict__ zr,
                const float4 *__restrict__ hx, long long npix, int hid4, int inp4, unsigned *__restrict__ dzr_amax) {
    unsigned am_z = 0;
    const long long total = npix * hid4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
        const long long p = i / hid4;
        const int c = (int)(i - p * hid4);
        const float4 r = zr[p * (2 * hid4) + hid4 + c], h = hx[p * (hid4 + inp4) + c], d = d_rhx[p * (hid4 + inp4) + c];
        const float4 dr = make_float4(d.x * h.x * r.x * (1.f - r.x), d.y * h.y * r.y * (1.f - r.y),
                                      d.z * h.z * r.z * (1.f - r.z), d.w * h.w * r.w * (1.f - r.w));
        dzr[p * (2 * hid4) + hid4 + c] = dr;
        az_amax_acc(am_z, dr);
        float4 a = dh_acc[i];
        a.x += d.x * r.x; a.y += d.y * r.y; a.z += d.z * r.z; a.w += d.w * r.w;
        dh_acc[i] = a;
    }
    if (dzr_amax) az_amax_flush(dzr_amax, am_z);
}

__global__ void __launch_bounds__(256)
gru_bwd3_kernel(float4 *__restrict__ dh, float4 *__restrict__ dx, const float4 *__restrict__ dh_acc, const float4 *__restrict__ d_rhx,
                const float4 *__restrict__ d_hx, long long npix, int hid4, int inp4) {
    const int ct4 = hid4 + inp4;
    const long long total = npix * ct4;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
        const long long p = i / ct4;
        const int c = (int)(i - p * ct4);
        const float4 a = d_hx[i];
        if (c < hid4) {
            const float4 b = dh_acc[p * hid4 + c];
            dh[p * hid4 + c] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        } else {
            const float4 b = d_rhx[i];
            dx[p * inp4 + (c - hid4)] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
    }
}

static bool gg_ok(long long npix, int hid, int inp) { return npix > 0 && hid > 0 && inp >= 0 && hid % 4 == 0 && inp % 4 == 0; }

extern "C" int az_gru_rh(float *rhx, const float *zr, const float *hx, long long npix, int hid, int inp, void *stream) {
    AZ_REQUIRE_PTR(rhx); AZ_REQUIRE_PTR(zr); AZ_REQUIRE_PTR(hx);
    if (!gg_ok(npix, hid, inp)) return AZ_EINVAL;
    hipLaunchKernelGGL(gru_rh_kernel, dim3(GG_GRID(npix * (hid + inp) / 4)), dim3(256), 0, az_stream(stream), (float4 *)rhx,
                       (const float4 *)zr, (const float4 *)hx, npix, hid / 4, inp / 4);
    return az_launch_status();
}

extern "C" int az_gru_out(float *hn, const float *zr, const float *q, const float *hx, long long npix, int hid, int inp, void *stream) {
    AZ_REQUIRE_PTR(hn); AZ_REQUIRE_PTR(zr); AZ_REQUIRE_PTR(q); AZ_REQUIRE_PTR(hx);
    if (!gg_ok(npix, hid, inp)) return AZ_EINVAL;
    hipLaunchKernelGGL(gru_out_kernel, dim3(GG_GRID(npix * hid / 4)), dim3(256), 0, az_stream(stream), (float4 *)hn, (const float4 *)zr,
                       (const float4 *)q, (const float4 *)hx, npix, hid / 4, inp / 4);
    return az_launch_status();
}

extern "C" int az_gru_bwd1(float *dq_pre, float *dzr, float *dh_acc, const float *g, const float *zr, const float *q, const float *hx,
                           long long npix, int hid, int inp, float *dq_amax, float *dzr_amax, void *stream) {
    if ((dq_amax == nullptr) != (dzr_amax == nullptr)) return AZ_EINVAL;
    AZ_REQUIRE_PTR(dq_pre); AZ_REQUIRE_PTR(dzr); AZ_REQUIRE_PTR(dh_acc); AZ_REQUIRE_PTR(g); AZ_REQUIRE_PTR(zr); AZ_REQUIRE_PTR(q); AZ_REQUIRE_PTR(hx);
    if (!gg_ok(npix, hid, inp)) return AZ_EINVAL;
    hipLaunchKernelGGL(gru_bwd1_kernel, dim3(GG_GRID(npix * hid / 4)), dim3(256), 0, az_stream(stream), (float4 *)dq_pre, (float4 *)dzr,
                       (float4 *)dh_acc, (const float4 *)g, (const float4 *)zr, (const float4 *)q, (const float4 *)hx, npix, hid / 4, inp / 4,
                       reinterpret_cast<unsigned *>(dq_amax), reinterpret_cast<unsigned *>(dzr_amax));
    return az_launch_status();
}

extern "C" int az_gru_bwd2(float *dzr, float *dh_acc, const float *d_rhx, const float *zr, const float *hx, long long npix, int hid, int inp,
                           float *dzr_amax, void *stream) {
    AZ_REQUIRE_PTR(dzr); AZ_REQUIRE_PTR(dh_acc); AZ_REQUIRE_PTR(d_rhx); AZ_REQUIRE_PTR(zr); AZ_REQUIRE_PTR(hx);
    if (!gg_ok(npix, hid, inp)) return AZ_EINVAL;
    hipLaunchKernelGGL(gru_bwd2_kernel, dim3(GG_GRID(npix * hid / 4)), dim3(256), 0, az_stream(stream), (float4 *)dzr, (float4 *)dh_acc,
                       (const float4 *)d_rhx, (const float4 *)zr, (const float4 *)hx, npix, hid / 4, inp / 4, reinterpret_cast<unsigned *>(dzr_amax));
    return az_launch_status();
}

extern "C" int az_gru_bwd3(float *dh, float *dx, const float *dh_acc, const float *d_rhx, const float *d_hx, long long npix, int hid, int inp,
                           void *stream) {
    AZ_REQUIRE_PTR(dh); AZ_REQUIRE_PTR(dx); AZ_REQUIRE_PTR(dh_acc); AZ_REQUIRE_PTR(d_rhx); AZ_REQUIRE_PTR(d_hx);
    if (!gg_ok(npix, hid, inp) || inp == 0) return AZ_EINVAL;
    hipLaunchKernelGGL(gru_bwd3_kernel, dim3(GG_GRID(npix * (hid + inp) / 4)), dim3(256), 0, az_stream(stream), (float4 *)dh, (float4 *)dx,
                       (const float4 *)dh_acc, (const float4 *)d_rhx, (const float4 *)d_hx, npix, hid / 4, inp / 4);
    return az_launch_status();
}
