// K10/K11 -- RAFT-Stereo 1-D correlation volume, pyramid and lookup (secondary path,
// BASELINE config 5).  Reference: nets/raft/corr.py:115-161 (CorrBlock1D) and
// nets/raft/raft_utils.py:68-82 (bilinear_sampler).
//
//   volume : corr[b,h,w1,w2] = sum_c f1[b,c,h,w1] * f2[b,c,h,w2] / sqrt(C)
//            (corr.py:153-161: einsum("aijk,aijh->ajkh") / sqrt(D))
//   pyramid: level i+1 = avg_pool over pairs of the LAST axis (corr.py:126-130)
//   lookup : out[b, 9*lvl + k, h, w1] = linear interpolation of pyr_lvl[b,h,w1,:] at
//            x = coords[b,0,h,w1] / 2^lvl + (k - r), zero outside, the align_corners=True
//            convention of bilinear_sampler (corr.py:132-151)
//
// The three dense contractions (volume, d/d f1, d/d f2) are one strided batched GEMM,
//   D[m][n] = scale * sum_k A(k,m) * B(k,n), every operand addressed through (k-stride, m|n-stride), so NCHW
// feature maps and the [w1][w2] volume are read in place:
//   volume : m=w1, n=w2, k=c   A=f1, B=f2
//   d f1   : m=w1, n=c,  k=w2  A=G,  B=f2
//   d f2   : m=w2, n=c,  k=w1  A=G,  B=f1
// on the bf16x6 arithmetic of the rest of the library (exact 3-way bf16 split of both operands, six
// v_mfma_f32_32x32x16_bf16 per 16-deep block summed from zero, az_common.h): bgemm_x6_kernel.  Workgroup = 4 waves,
// 64x64 output tile (a 32x32 accumulator per wave), K chunks of 32.  A thread fetches 8 consecutive k of one row of
// each operand -- eight loads that are each coalesced across the lanes when the row index is the unit-stride one
// (feature maps), two 16-byte loads when k is (the volume's last axis) --, splits them and writes one 16-byte piece
// per part: LDS holds [part][row][32 k] bf16 with an 80-byte row pitch (16 lanes of a fragment read cover the 64
// banks once).  Chunk s+1 travels global -> registers under the MFMAs of chunk s; one 30 KB buffer, so that five
// workgroups share a CU: the kernel is bound by HBM latency, which occupancy hides and a one-chunk prefetch does not.  (Rounds 1-2 ran these GEMMs on v_mfma_f32_32x32x2_f32 with scalar LDS
// staging: 0.45 ms for the volume at [4,256,136,240], 11 % of its HBM bound; bgemm_tn_kernel is kept for
// AZ_CORR_FP32=1.)
#include <stdlib.h>

#include "az_common.h"
#include "az_options.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float *A, *B;
    float *D;
    long long sAk, sAm, sBk, sBn, sDm, sDn;   // element strides
    long long bA0, bA1, bB0, bB1, bD0, bD1;   // batch strides for (b, h)
    int M, N, K, H;
    float scale;
};

#define G_TM 64
#define G_TN 64
#define G_TK 32

__global__ void __launch_bounds__(256)
bgemm_tn_kernel(const GemmArgs g) {
    __shared__ float sA[G_TK][G_TM + 1];
    __shared__ float sB[G_TK][G_TN + 1];
    const int batch = blockIdx.z;
    const int b = batch / g.H, h = batch - b * g.H;
    const float *A = g.A + b * g.bA0 + h * g.bA1;
    const float *B = g.B + b * g.bB0 + h * g.bB1;
    float *D = g.D + b * g.bD0 + h * g.bD1;
    const int m0 = blockIdx.y * G_TM, n0 = blockIdx.x * G_TN;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane & 31, half = lane >> 5;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    // staging order: make the unit-stride index the fastest-varying one across threads
    const bool a_m_fast = g.sAm <= g.sAk, b_n_fast = g.sBn <= g.sBk;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < g.K; k0 += G_TK) {
        __syncthreads();
        for (int e = threadIdx.x; e < G_TK * G_TM; e += 256) {
            const int kk = a_m_fast ? e / G_TM : e % G_TK;
            const int mm = a_m_fast ? e % G_TM : e / G_TK;
            const int k = k0 + kk, m = m0 + mm;
            sA[kk][mm] = (k < g.K && m < g.M) ? A[k * g.sAk + m * g.sAm] : 0.f;
        }
        for (int e = threadIdx.x; e < G_TK * G_TN; e += 256) {
            const int kk = b_n_fast ? e / G_TN : e % G_TK;
            const int nn = b_n_fast ? e % G_TN : e / G_TK;
            const int k = k0 + kk, n = n0 + nn;
            sB[kk][nn] = (k < g.K && n < g.N) ? B[k * g.sBk + n * g.sBn] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < G_TK / 2; ++q)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[2 * q + half][wm + row],
                                                       sB[2 * q + half][wn + row], acc, 0, 0, 0);
    }
    const int n = n0 + wn + row;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < g.M && n < g.N) D[m * g.sDm + n * g.sDn] = acc[r] * g.scale;
    }
}

// ---- bf16x6 version -------------------------------------------------------------------------------------
#define GX_PITCH 80                      // bytes per (row, part): 32 k x bf16 + 16 pad
#define GX_PART (G_TM * GX_PITCH)        // one part of one operand tile
#define GX_OPER (3 * GX_PART)            // one operand tile: 15 360 B
#define GX_BUF (2 * GX_OPER)             // A + B: 30 720 B per buffer
template <bool A_KFAST, bool B_KFAST>
__global__ void __launch_bounds__(256, 4)
bgemm_x6_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[GX_BUF];
    const int batch = blockIdx.z;
    const int b = batch / g.H, h = batch - b * g.H;
    const float *A = g.A + b * g.bA0 + h * g.bA1;
    const float *B = g.B + b * g.bB0 + h * g.bB1;
    float *D = g.D + b * g.bD0 + h * g.bD1;
    const int m0 = blockIdx.y * G_TM, n0 = blockIdx.x * G_TN;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane & 31, half = lane >> 5;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int sr = threadIdx.x & 63, kg = threadIdx.x >> 6;  // staging role: tile row, group of 8 k
    // 16-byte loads along k need aligned rows (true for the volume axis of the shapes the model uses)
    const bool a_vec = A_KFAST && (g.sAm % 4 == 0) && (g.bA0 % 4 == 0) && (g.bA1 % 4 == 0) && (g.K % 4 == 0);
    const bool b_vec = B_KFAST && (g.sBn % 4 == 0) && (g.bB0 % 4 == 0) && (g.bB1 % 4 == 0) && (g.K % 4 == 0);
    // operands through buffer resources over this (b, h) slab: the per-lane byte offset of the thread's row is computed
    // once, the k offset travels in the scalar offset / immediate, validity is an out-of-range offset (no 64-bit
    // address arithmetic and no branches per element: the kernel is VALU-issue bound in its staging)
    const unsigned OOB = 0xffffff00u;
    auto slab_bytes = [&](long long sk, long long sr_stride, int R) {
        const long long last = (long long)(g.K - 1) * sk + (long long)(R - 1) * sr_stride + 1;
        return (unsigned)(last * 4 < 0xfffffff0LL ? last * 4 : 0xfffffff0LL);
    };
    const auto rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A), 0, slab_bytes(g.sAk, g.sAm, g.M), 0x00020000);
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, slab_bytes(g.sBk, g.sBn, g.N), 0x00020000);
    const unsigned ra_off = (m0 + sr < g.M) ? (unsigned)((long long)(m0 + sr) * g.sAm * 4) : OOB;
    const unsigned rb_off = (n0 + sr < g.N) ? (unsigned)((long long)(n0 + sr) * g.sBn * 4) : OOB;
    float xa[8], xb[8];
    auto fetch = [&](float (&x)[8], decltype(rs_a) rs, unsigned row_off, long long sk, bool kfast, bool vec, int k0) {
        const int kk = k0 + kg * 8;  // wave-uniform
        if (kfast && vec) {
            typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
            const u32x4_ u = __builtin_amdgcn_raw_buffer_load_b128(rs, kk < g.K ? row_off : OOB, kk * 4, 0);
            const u32x4_ v = __builtin_amdgcn_raw_buffer_load_b128(rs, kk + 4 < g.K ? row_off : OOB, kk * 4 + 16, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { x[j] = __uint_as_float(u[j]); x[4 + j] = __uint_as_float(v[j]); }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                x[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, kk + j < g.K ? row_off : OOB,
                                                                            (int)((long long)(kk + j) * sk * 4), 0));
        }
    };
    auto commit = [&](const float (&x)[8], unsigned char *oper) {
        uint2 h0, m0_, l0, h1, m1, l1;
        az_split3_bf16x4(make_float4(x[0], x[1], x[2], x[3]), h0, m0_, l0);
        az_split3_bf16x4(make_float4(x[4], x[5], x[6], x[7]), h1, m1, l1);
        unsigned char *dst = oper + sr * GX_PITCH + kg * 16;
        *reinterpret_cast<uint4 *>(dst) = make_uint4(h0.x, h0.y, h1.x, h1.y);
        *reinterpret_cast<uint4 *>(dst + GX_PART) = make_uint4(m0_.x, m0_.y, m1.x, m1.y);
        *reinterpret_cast<uint4 *>(dst + 2 * GX_PART) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    };
    az_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    fetch(xa, rs_a, ra_off, g.sAk, A_KFAST, a_vec, 0);
    fetch(xb, rs_b, rb_off, g.sBk, B_KFAST, b_vec, 0);
    for (int k0 = 0; k0 < g.K; k0 += G_TK) {
        commit(xa, lds);
        commit(xb, lds + GX_OPER);
        __syncthreads();
        if (k0 + G_TK < g.K) {  // the next chunk travels while this one is multiplied
            fetch(xa, rs_a, ra_off, g.sAk, A_KFAST, a_vec, k0 + G_TK);
            fetch(xb, rs_b, rb_off, g.sBk, B_KFAST, b_vec, k0 + G_TK);
        }
        const unsigned char *ta = lds + (wm + row) * GX_PITCH + half * 16;
        const unsigned char *tb = lds + GX_OPER + (wn + row) * GX_PITCH + half * 16;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float4 aq[3], bq[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                aq[p] = *reinterpret_cast<const float4 *>(ta + p * GX_PART + q * 32);
                bq[p] = *reinterpret_cast<const float4 *>(tb + p * GX_PART + q * 32);
            }
            az_mfma6_now(acc, aq, bq);
        }
        __syncthreads();  // (single buffer: five workgroups per CU hide the HBM latency that one chunk of prefetch
                          //  distance cannot; the double-buffered version with two per CU ran 0.33 ms, latency-bound)
    }
    // C layout of the 32x32 MFMA: lane = column n, register r = row (r & 3) + 8 (r >> 2) + 4 half
    const int n = n0 + wn + row;
    if (n < g.N) {
        if (g.sDm == 1) {  // rows contiguous in memory: four at a time
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int m = m0 + wm + 8 * r4 + 4 * half;
                float *dst = D + (long long)n * g.sDn + m;
                if (m + 3 < g.M && (reinterpret_cast<size_t>(dst) & 15) == 0) {
                    *reinterpret_cast<float4 *>(dst) = make_float4(acc[4 * r4] * g.scale, acc[4 * r4 + 1] * g.scale,
                                                                   acc[4 * r4 + 2] * g.scale, acc[4 * r4 + 3] * g.scale);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (m + i < g.M) dst[i] = acc[4 * r4 + i] * g.scale;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < g.M) D[(long long)m * g.sDm + (long long)n * g.sDn] = acc[r] * g.scale;
            }
        }
    }
}

static int launch_gemm(const GemmArgs &g, int batches, hipStream_t s) {
    if (batches <= 0 || batches > 65535) return AZ_EUNSUPPORTED;
    dim3 grid((g.N + G_TN - 1) / G_TN, (g.M + G_TM - 1) / G_TM, batches);
    const int fp32 = az_options().corr_fp32;
    if (fp32) {
        hipLaunchKernelGGL(bgemm_tn_kernel, grid, dim3(256), 0, s, g);
        return az_launch_status();
    }
    const bool ak = g.sAk == 1 && g.sAm != 1, bk = g.sBk == 1 && g.sBn != 1;
    if (ak && bk) hipLaunchKernelGGL((bgemm_x6_kernel<true, true>), grid, dim3(256), 0, s, g);
    else if (ak) hipLaunchKernelGGL((bgemm_x6_kernel<true, false>), grid, dim3(256), 0, s, g);
    else if (bk) hipLaunchKernelGGL((bgemm_x6_kernel<false, true>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((bgemm_x6_kernel<false, false>), grid, dim3(256), 0, s, g);
    return az_launch_status();
}

// dst[r][j] = 0.5 * (src[r][2j] + src[r][2j+1]), j < Wsrc / 2
__global__ void __launch_bounds__(256)
pool_pairs_kernel(float *__restrict__ dst, const float *__restrict__ src, long long rows, int Wsrc) {
    const int Wd = Wsrc / 2;
    const long long total = rows * Wd;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
        const long long r = i / Wd;
        const int j = (int)(i - r * Wd);
        const float *p = src + r * Wsrc + 2 * j;
        dst[i] = (p[0] + p[1]) * 0.5f;
    }
}

// gsrc[r][x] = 0.5 * gdst[r][x / 2] for x < 2*(Wsrc/2), else 0
__global__ void __launch_bounds__(256)
pool_pairs_bwd_kernel(float *__restrict__ gsrc, const float *__restrict__ gdst, long long rows, int Wsrc) {
    const int Wd = Wsrc / 2;
    const long long total = rows * Wsrc;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
        const long long r = i / Wsrc;
        const int x = (int)(i - r * Wsrc);
        gsrc[i] = (x < 2 * Wd) ? 0.5f * gdst[r * Wd + (x >> 1)] : 0.f;
    }
}

__device__ __forceinline__ float lookup_x(float coord, float inv_scale, int k, int r, int W) {
#pragma clang fp contract(off)
    const float x = coord * inv_scale + (float)(k - r);  // dx + coords / 2^lvl (exact power of 2)
    const float gx = 2.0f * x / (float)(W - 1) - 1.0f;   // bilinear_sampler
    return ((gx + 1.0f) / 2.0f) * (float)(W - 1);        // grid_sample, align_corners=True
}

// one thread per (b, h, w1): 2r+1 taps of one pyramid level
__global__ void __launch_bounds__(256)
lookup_fwd_kernel(float *__restrict__ out, const float *__restrict__ pyr,
                  const float *__restrict__ coords, int H, int W1, int Wl, int radius,
                  float inv_scale, int ch_off, int ch_total, long long total) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
        const int w1 = (int)(i % W1);
        const long long t = i / W1;
        const int h = (int)(t % H);
        const long long b = t / H;
        const float c = coords[(b * 2 * H + h) * W1 + w1];  // channel 0 of [B,2,H,W1]
        const float *rowp = pyr + i * Wl;
        for (int k = 0; k <= 2 * radius; ++k) {
            const float ix = lookup_x(c, inv_scale, k, radius, Wl);
            const float fx = floorf(ix);
            const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)Wl + 1.f);
            const float w1f = ix - fx;
            float v = 0.f;
            if (x0 >= 0 && x0 < Wl) v += rowp[x0] * (1.f - w1f);
            if (x0 + 1 >= 0 && x0 + 1 < Wl) v += rowp[x0 + 1] * w1f;
            out[((b * ch_total + ch_off + k) * H + h) * W1 + w1] = v;
        }
    }
}

// gpyr (pre-zeroed): the thread owns row (b,h,w1) of the level -> plain accumulation
__global__ void __launch_bounds__(256)
lookup_bwd_kernel(float *__restrict__ gpyr, const float *__restrict__ gout,
                  const float *__restrict__ coords, int H, int W1, int Wl, int radius,
                  float inv_scale, int ch_off, int ch_total, long long total) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += gridDim.x * 256LL) {
        const int w1 = (int)(i % W1);
        const long long t = i / W1;
        const int h = (int)(t % H);
        const long long b = t / H;
        const float c = coords[(b * 2 * H + h) * W1 + w1];
        float *rowp = gpyr + i * Wl;
        for (int k = 0; k <= 2 * radius; ++k) {
            const float ix = lookup_x(c, inv_scale, k, radius, Wl);
            const float fx = floorf(ix);
            const int x0 = (int)fminf(fmaxf(fx, -2.f), (float)Wl + 1.f);
            const float w1f = ix - fx;
            const float g = gout[((b * ch_total + ch_off + k) * H + h) * W1 + w1];
            if (x0 >= 0 && x0 < Wl) rowp[x0] += g * (1.f - w1f);
            if (x0 + 1 >= 0 && x0 + 1 < Wl) rowp[x0 + 1] += g * w1f;
        }
    }
}

static int corr_check(int B, int C, int H, int W1, int W2) {
    if (!(B > 0 && C > 0 && H > 0 && W1 > 0 && W2 > 0)) return AZ_EINVAL;
    if ((long long)B * H > 65535) return AZ_EUNSUPPORTED;
    return AZ_OK;
}

extern "C" int az_corr1d_volume(float *corr, const float *f1, const float *f2, int B, int C,
                                int H, int W1, int W2, void *stream) {
    AZ_REQUIRE_PTR(corr); AZ_REQUIRE_PTR(f1); AZ_REQUIRE_PTR(f2);
    if (int e = corr_check(B, C, H, W1, W2)) return e;
    GemmArgs g{};
    g.A = f1; g.B = f2; g.D = corr;
    g.M = W1; g.N = W2; g.K = C; g.H = H;
    g.sAk = (long long)H * W1; g.sAm = 1; g.bA0 = (long long)C * H * W1; g.bA1 = W1;
    g.sBk = (long long)H * W2; g.sBn = 1; g.bB0 = (long long)C * H * W2; g.bB1 = W2;
    g.sDm = W2; g.sDn = 1; g.bD0 = (long long)H * W1 * W2; g.bD1 = (long long)W1 * W2;
    g.scale = 1.0f / sqrtf((float)C);
    return launch_gemm(g, B * H, az_stream(stream));
}

extern "C" int az_corr1d_volume_bwd(float *grad_f1, float *grad_f2, const float *grad_corr,
                                    const float *f1, const float *f2, int B, int C, int H,
                                    int W1, int W2, void *stream) {
    AZ_REQUIRE_PTR(grad_corr); AZ_REQUIRE_PTR(f1); AZ_REQUIRE_PTR(f2);
    if (int e = corr_check(B, C, H, W1, W2)) return e;
    const float scale = 1.0f / sqrtf((float)C);
    if (grad_f1) {  // gf1[c][w1] = scale * sum_w2 G[w1][w2] f2[c][w2]
        GemmArgs g{};
        g.A = grad_corr; g.B = f2; g.D = grad_f1;
        g.M = W1; g.N = C; g.K = W2; g.H = H; g.scale = scale;
        g.sAk = 1; g.sAm = W2; g.bA0 = (long long)H * W1 * W2; g.bA1 = (long long)W1 * W2;
        g.sBk = 1; g.sBn = (long long)H * W2; g.bB0 = (long long)C * H * W2; g.bB1 = W2;
        g.sDm = 1; g.sDn = (long long)H * W1; g.bD0 = (long long)C * H * W1; g.bD1 = W1;
        if (int e = launch_gemm(g, B * H, az_stream(stream))) return e;
    }
    if (grad_f2) {  // gf2[c][w2] = scale * sum_w1 G[w1][w2] f1[c][w1]
        GemmArgs g{};
        g.A = grad_corr; g.B = f1; g.D = grad_f2;
        g.M = W2; g.N = C; g.K = W1; g.H = H; g.scale = scale;
        g.sAk = W2; g.sAm = 1; g.bA0 = (long long)H * W1 * W2; g.bA1 = (long long)W1 * W2;
        g.sBk = 1; g.sBn = (long long)H * W1; g.bB0 = (long long)C * H * W1; g.bB1 = W1;
        g.sDm = 1; g.sDn = (long long)H * W2; g.bD0 = (long long)C * H * W2; g.bD1 = W2;
        if (int e = launch_gemm(g, B * H, az_stream(stream))) return e;
    }
    return AZ_OK;
}

extern "C" int az_corr1d_pool(float *dst, const float *src, long long rows, int Wsrc,
                              void *stream) {
    AZ_REQUIRE_PTR(dst); AZ_REQUIRE_PTR(src);
    AZ_REQUIRE(rows > 0 && Wsrc >= 2);
    hipLaunchKernelGGL(pool_pairs_kernel, dim3(az_grid_for(rows * (Wsrc / 2), 256)), dim3(256), 0,
                       az_stream(stream), dst, src, rows, Wsrc);
    return az_launch_status();
}

extern "C" int az_corr1d_pool_bwd(float *grad_src, const float *grad_dst, long long rows,
                                  int Wsrc, void *stream) {
    AZ_REQUIRE_PTR(grad_src); AZ_REQUIRE_PTR(grad_dst);
    AZ_REQUIRE(rows > 0 && Wsrc >= 2);
    hipLaunchKernelGGL(pool_pairs_bwd_kernel, dim3(az_grid_for(rows * Wsrc, 256)), dim3(256), 0,
                       az_stream(stream), grad_src, grad_dst, rows, Wsrc);
    return az_launch_status();
}

extern "C" int az_corr1d_lookup_fwd(float *out, const float *pyr_level, const float *coords,
                                    int B, int H, int W1, int W_level, int radius, int level,
                                    int ch_offset, int ch_total, void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(pyr_level); AZ_REQUIRE_PTR(coords);
    AZ_REQUIRE(B > 0 && H > 0 && W1 > 0 && W_level > 1 && radius >= 0 && level >= 0 && level < 16);
    AZ_REQUIRE(ch_offset >= 0 && ch_offset + 2 * radius + 1 <= ch_total);
    const long long total = (long long)B * H * W1;
    hipLaunchKernelGGL(lookup_fwd_kernel, dim3(az_grid_for(total, 256)), dim3(256), 0,
                       az_stream(stream), out, pyr_level, coords, H, W1, W_level, radius,
                       1.0f / (float)(1 << level), ch_offset, ch_total, total);
    return az_launch_status();
}

extern "C" int az_corr1d_lookup_bwd(float *grad_pyr_level, const float *grad_out,
                                    const float *coords, int B, int H, int W1, int W_level,
                                    int radius, int level, int ch_offset, int ch_total,
                                    void *stream) {
    AZ_REQUIRE_PTR(grad_pyr_level); AZ_REQUIRE_PTR(grad_out); AZ_REQUIRE_PTR(coords);
    AZ_REQUIRE(B > 0 && H > 0 && W1 > 0 && W_level > 1 && radius >= 0 && level >= 0 && level < 16);
    AZ_REQUIRE(ch_offset >= 0 && ch_offset + 2 * radius + 1 <= ch_total);
    const long long total = (long long)B * H * W1;
    if (hipMemsetAsync(grad_pyr_level, 0, (size_t)total * W_level * sizeof(float),
                       az_stream(stream)) != hipSuccess)
        return AZ_ELAUNCH;
    hipLaunchKernelGGL(lookup_bwd_kernel, dim3(az_grid_for(total, 256)), dim3(256), 0,
                       az_stream(stream), grad_pyr_level, grad_out, coords, H, W1, W_level, radius,
                       1.0f / (float)(1 << level), ch_offset, ch_total, total);
    return az_launch_status();
}

// the same scatter ADDED to what grad_pyr_level holds: the 22 lookups of a RAFT-Stereo step (raft_stereo.py:138-172) read ONE
// pyramid, and the gradient of a level is the sum over them -- accumulated here by the kernel's own atomics into one buffer
// (cleared once by the caller) instead of 22 cleared buffers and 21 tensor additions per level
extern "C" int az_corr1d_lookup_bwd_acc(float *grad_pyr_level, const float *grad_out, const float *coords, int B, int H, int W1,
                                        int W_level, int radius, int level, int ch_offset, int ch_total, void *stream) {
    AZ_REQUIRE_PTR(grad_pyr_level); AZ_REQUIRE_PTR(grad_out); AZ_REQUIRE_PTR(coords);
    AZ_REQUIRE(B > 0 && H > 0 && W1 > 0 && W_level > 1 && radius >= 0 && level >= 0 && level < 16);
    AZ_REQUIRE(ch_offset >= 0 && ch_offset + 2 * radius + 1 <= ch_total);
    const long long total = (long long)B * H * W1;
    hipLaunchKernelGGL(lookup_bwd_kernel, dim3(az_grid_for(total, 256)), dim3(256), 0,
                       az_stream(stream), grad_pyr_level, grad_out, coords, H, W1, W_level, radius,
                       1.0f / (float)(1 << level), ch_offset, ch_total, total);
    return az_launch_status();
}
