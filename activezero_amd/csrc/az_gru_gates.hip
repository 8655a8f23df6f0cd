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

// ---- [h | x_0 | x_1 ..] rows in ONE launch (round 5) --------------------------------------------------------------------
// hx[p][..] = the concatenation over the sources of their channels at pixel p.  A source is either dense rows [npix][c]
// (kind 0: the previous update's state, a cached channels-last copy of the context) or an NCHW image [B][c][HW] (kind 1: the
// correlation lookup).  torch did this with one strided copy per source -- 64 + 52 + 109 us per update on the RAFT workload
// (transposing elementwise kernels), 22 updates per step.  A workgroup takes 64 pixels: row sources are copied 16 bytes per
// lane (a tile's rows are ONE contiguous block of the source), image sources go through a [c][64 + 1] LDS tile (reads
// coalesced along the pixels, writes along the channels).
struct RowsCatArgs {
    const float *src[4];
    int c[4], kind[4], at[4];  // channels, layout, first destination channel
    int nsrc, ctot;
    long long npix, hw;
};
#define RC_PIX 64
#define RC_MAXC 128  // channels of an image source (LDS tile)
__global__ void __launch_bounds__(256)
rows_concat_kernel(float *__restrict__ dst, const RowsCatArgs a) {
    __shared__ float tile[RC_MAXC * (RC_PIX + 1)];
    const long long p0 = (long long)blockIdx.x * RC_PIX;
    const int np = (int)min((long long)RC_PIX, a.npix - p0);
    for (int s = 0; s < a.nsrc; ++s) {
        const int c = a.c[s], at = a.at[s];
        if (a.kind[s] == 0) {
            const int c4 = c >> 2;
            const float4 *src = reinterpret_cast<const float4 *>(a.src[s]) + p0 * c4;
            for (int i = threadIdx.x; i < np * c4; i += 256) {
                const int p = i / c4, q = i - p * c4;
                *reinterpret_cast<float4 *>(dst + (p0 + p) * a.ctot + at + 4 * q) = src[i];
            }
        } else {
            __syncthreads();  // (the tile of the source before)
            for (int i = threadIdx.x; i < c * RC_PIX; i += 256) {
                const int ch = i / RC_PIX, p = i - ch * RC_PIX;
                const long long g = p0 + p;
                if (p < np) {
                    const long long b = g / a.hw, r = g - b * a.hw;
                    tile[ch * (RC_PIX + 1) + p] = a.src[s][(b * c + ch) * a.hw + r];
                }
            }
            __syncthreads();
            const int c4 = c >> 2;
            for (int i = threadIdx.x; i < np * c4; i += 256) {
                const int p = i / c4, q = i - p * c4;
                const float4 v = make_float4(tile[(4 * q + 0) * (RC_PIX + 1) + p], tile[(4 * q + 1) * (RC_PIX + 1) + p],
                                             tile[(4 * q + 2) * (RC_PIX + 1) + p], tile[(4 * q + 3) * (RC_PIX + 1) + p]);
                *reinterpret_cast<float4 *>(dst + (p0 + p) * a.ctot + at + 4 * q) = v;
            }
        }
    }
}

// dst rows [npix][ctot] <- up to four sources (channel counts multiples of 4, image sources at most 128 channels); kinds: 0 dense
// rows [npix][c], 1 NCHW image [npix / hw][c][hw]
extern "C" int az_rows_concat(float *dst, long long npix, long long hw, int nsrc, const float *const *srcs, const int *channels,
                              const int *kinds, void *stream) {
    AZ_REQUIRE_PTR(dst); AZ_REQUIRE_PTR(srcs); AZ_REQUIRE_PTR(channels); AZ_REQUIRE_PTR(kinds);
    if (nsrc < 1 || nsrc > 4 || npix <= 0 || hw <= 0 || npix % hw) return AZ_EINVAL;
    RowsCatArgs a{};
    int at = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (srcs[s] == nullptr) return AZ_ENULL;
        if (channels[s] <= 0 || channels[s] % 4 || (kinds[s] != 0 && kinds[s] != 1) || (kinds[s] == 1 && channels[s] > RC_MAXC)) return AZ_EUNSUPPORTED;
        a.src[s] = srcs[s]; a.c[s] = channels[s]; a.kind[s] = kinds[s]; a.at[s] = at;
        at += channels[s];
    }
    a.nsrc = nsrc; a.ctot = at; a.npix = npix; a.hw = hw;
    const long long blocks = (npix + RC_PIX - 1) / RC_PIX;
    if (blocks > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    hipLaunchKernelGGL(rows_concat_kernel, dim3((unsigned)blocks), dim3(256), 0, az_stream(stream), dst, a);
    return az_launch_status();
}

// the inverse for ONE channel slice: image [B][c][hw] (NCHW, dense) <- rows[p][at .. at + c) of [npix][ctot] rows: the gradient
// of an image source, which torch took with a transposing .contiguous() of a strided view
__global__ void __launch_bounds__(256)
rows_slice_to_image_kernel(float *__restrict__ img, const float *__restrict__ rows, long long npix, long long hw, int ctot, int at, int c) {
    __shared__ float tile[RC_MAXC * (RC_PIX + 1)];
    const long long p0 = (long long)blockIdx.x * RC_PIX;
    const int np = (int)min((long long)RC_PIX, npix - p0);
    const int c4 = c >> 2;
    for (int i = threadIdx.x; i < np * c4; i += 256) {
        const int p = i / c4, q = i - p * c4;
        const float4 v = *reinterpret_cast<const float4 *>(rows + (p0 + p) * ctot + at + 4 * q);
        tile[(4 * q + 0) * (RC_PIX + 1) + p] = v.x; tile[(4 * q + 1) * (RC_PIX + 1) + p] = v.y;
        tile[(4 * q + 2) * (RC_PIX + 1) + p] = v.z; tile[(4 * q + 3) * (RC_PIX + 1) + p] = v.w;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c * RC_PIX; i += 256) {
        const int ch = i / RC_PIX, p = i - ch * RC_PIX;
        const long long g = p0 + p;
        if (p < np) {
            const long long b = g / hw, r = g - b * hw;
            img[(b * c + ch) * hw + r] = tile[ch * (RC_PIX + 1) + p];
        }
    }
}

extern "C" int az_rows_slice_to_image(float *image, const float *rows, long long npix, long long hw, int ctot, int at, int c, void *stream) {
    AZ_REQUIRE_PTR(image); AZ_REQUIRE_PTR(rows);
    if (npix <= 0 || hw <= 0 || npix % hw || ctot <= 0 || ctot % 4 || at < 0 || at % 4 || c <= 0 || c % 4 || at + c > ctot) return AZ_EINVAL;
    if (c > RC_MAXC) return AZ_EUNSUPPORTED;
    const long long blocks = (npix + RC_PIX - 1) / RC_PIX;
    if (blocks > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    hipLaunchKernelGGL(rows_slice_to_image_kernel, dim3((unsigned)blocks), dim3(256), 0, az_stream(stream), image, rows, npix, hw, ctot, at, c);
    return az_launch_status();
}

