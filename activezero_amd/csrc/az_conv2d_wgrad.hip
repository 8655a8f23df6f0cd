// K13 (weight gradients) -- stride-1 "same" 2-D convolutions of the feature extractor and of the factored
// cost-volume convolution (reference nets/psmnet/psmnet_submodule_3.py:13-41, 92-220; autograd of
// F.conv2d there), bf16x6 arithmetic (exact 3-way bf16 split of both operands, six MFMAs per product,
// fp32 accumulation: az_conv3d.hip).
//
//   G[co][ci][kh][kw] = sum_{b,y,x} dy[b,y,x,co] * x[b, y + DIL*(kh - cy), x + DIL*(kw - cx), ci]
//
// GEMM view: M = 32 output channels, N = 32 input channels, K = pixel positions (16 per MFMA), one 32x32
// accumulator per tap.  A WORKGROUP of MT x NT waves owns MT*32 x NT*32 (co, ci) channels and walks a
// strided list of (image, row segment, 16-pixel chunk) items; per output row it stages the dy chunk
// (all MT*32 channels) and ONE new x row (NT*32 channels) -- the rows of the KH taps live in a rolling
// LDS ring, slot = row mod (DIL*(KH-1)+1) -- so the exact bf16 split, the VALU cost of this arithmetic,
// is paid once per MT*NT waves.  Tiles are stored [32-ch tile][part][position][32 ch] (64-byte rows: a
// ds_read_b64_tr_b16 of four rows covers the 64 banks once) and read with the hardware transpose read,
// which hands every lane 8 consecutive POSITIONS of its channel: K = positions.
// Each wave adds its T x 32 x 32 block into the tap-major workspace with float atomics once, when its
// list is exhausted; az_conv2d_wgrad's unpack kernel transposes [tap][co][ci] -> [co][ci][tap].
#include <stdlib.h>

#include "az_roll_common.h"
#include "az_options.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

#define W2_WCH 16  // output positions per chunk = one K16 block

struct W2Args {
    const float *coarse;  // dy: [B,H,W,*], pixel stride cs_c floats, channels [0, CM) of it
    const float *fine;    // x:  [B,H,W,*], pixel stride cs_f floats, channels [0, CN) of it
    float *ws;            // [taps_total][CM][CN]
    int B, H, W, CM, CN, cs_c, cs_f;
    int hseg_rows, nhseg, nwchunk;
    long long nitems;
    int blocks_per_combo;
    int fine_dy;  // extra row offset of the fine rows (single-row launches of a multi-row kernel)
    int tap0;     // first workspace tap of this launch
    const float *coarse_amax, *fine_amax;  // AR 1 (f16x3): amax arrays of dy and x
    int plain_bf16;                        // 1: AR 2 (one bf16 part), 2: AR 3 (one fp16 part, amax arrays optional)
};

// AR: 0 = bf16x6, 1 = f16x3 (az_roll_common.h; the LDS images keep their three-part strides), 2 = plain bf16 operands
// (round-to-nearest, ONE MFMA per tap, fp32 accumulation): the arithmetic of the reference's autocast region around the
// RAFT-Stereo GRU update (nets/raft/raft_stereo.py:142-172, train.py:303-309), for that block's weight gradients
template <int MT, int NT, int KH, int KW, int DIL, int AR = 0>
__global__ void __launch_bounds__(64 * MT * NT, 2)
conv2d_wgrad_kernel(const W2Args a) {
    constexpr int T = KH * KW;
    // AR 3 = "f16x1": ONE fp16 part per operand (what the reference's autocast computes on CUDA), each optionally scaled by a
    // power of two from an amax array (null = scale 1)
    constexpr int NP = (AR == 2 || AR == 3) ? 1 : AR ? 2 : 3;
    float c_scale = 1.f, f_scale = 1.f, o_scale = 1.f;
    if (AR == 1) {
        const int kc = az_f16_scale_exp(az_amax_read(a.coarse_amax)), kf = az_f16_scale_exp(az_amax_read(a.fine_amax));
        c_scale = az_pow2(kc); f_scale = az_pow2(kf);
        o_scale = ldexpf(1.f, -(kc + kf));
    }
    if (AR == 3) {
        const int kc = a.coarse_amax ? az_f16_scale_exp(az_amax_read(a.coarse_amax)) : 0;
        const int kf = a.fine_amax ? az_f16_scale_exp(az_amax_read(a.fine_amax)) : 0;
        c_scale = az_pow2(kc); f_scale = az_pow2(kf);
        o_scale = ldexpf(1.f, -(kc + kf));
    }
    constexpr int CY = (KH - 1) / 2, CX = (KW - 1) / 2;
    constexpr int WCH = W2_WCH;
    constexpr int FW = WCH + (KW - 1) * DIL;       // fine positions per staged row
    constexpr int R = DIL * (KH - 1) + 1;          // rows in the rolling ring
    constexpr int NTHR = 64 * MT * NT;
    constexpr int PA = MT * 8, PF = NT * 8;        // float4 pieces per position
    constexpr int NQA = WCH * PA, NLA = (NQA + NTHR - 1) / NTHR;
    constexpr int NQF = FW * PF, NLF = (NQF + NTHR - 1) / NTHR;
    constexpr int ATILE = 3 * WCH * 32, FTILE = 3 * FW * 32;  // bf16 elements per 32-channel tile image
    __shared__ __attribute__((aligned(16))) unsigned short sa[MT * ATILE];
    __shared__ __attribute__((aligned(16))) unsigned short sf[R * NT * FTILE];  // [slot][tile][part][pos][32]

    const int tid = threadIdx.x, lane = tid & 63, row = lane & 31, half = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt = wv % MT, nt = wv / MT;
    const int ncn = a.CN / (32 * NT);  // ci groups
    // blocks b and b+8 share an XCD: the combos that walk the SAME work list get ids 8 apart
    const int ncombo = (a.CM / (32 * MT)) * ncn;
    const int grp = blockIdx.x / (8 * ncombo), rem = blockIdx.x % (8 * ncombo);
    const int combo = rem >> 3;
    const int widx = grp * 8 + (rem & 7);
    const int cog = combo / ncn, cig = combo - cog * ncn;
    const int co0 = cog * MT * 32, ci0 = cig * NT * 32;

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // transpose-read geometry (az_conv3d_wgrad.hip): 16-lane group g reads columns 16*(g&1).. of rows
    // 8*(g>>1) + {0..3} / {4..7}; lane t = 4q+p of the group supplies row q, columns 4p..4p+3
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int tr_col = 16 * ((lane >> 4) & 1) + 4 * tp;
    const int tr_row = 8 * (lane >> 5) + tq;

    auto split_store = [&](unsigned short *dst_part0, int part_stride, const float4 &v, float scale_) {
        uint2 hi, mid, lo;
        if (AR == 2) {
            *reinterpret_cast<uint2 *>(dst_part0) = make_uint2(az_pk_bf16(v.x, v.y), az_pk_bf16(v.z, v.w));
            return;
        }
        if (AR == 3) {
            *reinterpret_cast<uint2 *>(dst_part0) = make_uint2(az_pk_f16(v.x * scale_, v.y * scale_), az_pk_f16(v.z * scale_, v.w * scale_));
            return;
        }
        if (AR) {
            az_split2_f16x4(make_float4(v.x * scale_, v.y * scale_, v.z * scale_, v.w * scale_), hi, mid);
            lo = mid;
        } else {
            az_split3_bf16x4(v, hi, mid, lo);
        }
        *reinterpret_cast<uint2 *>(dst_part0) = hi;
        *reinterpret_cast<uint2 *>(dst_part0 + part_stride) = mid;
        if (!AR) *reinterpret_cast<uint2 *>(dst_part0 + 2 * part_stride) = lo;
    };
    auto frag = [&](const unsigned short *img, int r0, int r1) -> bf16x8 {
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + r0 * 32 + tr_col));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + r1 * 32 + tr_col));
        s16x8 v;
        v[0] = lo4[0]; v[1] = lo4[1]; v[2] = lo4[2]; v[3] = lo4[3];
        v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
        return __builtin_bit_cast(bf16x8, v);
    };
    auto slot_of = [&](int fh) -> int { return (fh + 4 * R) % R; };  // fh >= -R always here

    for (long long item = widx; item < a.nitems; item += a.blocks_per_combo) {
        long long r = item;
        const int wc = (int)(r % a.nwchunk); r /= a.nwchunk;
        const int hs = (int)(r % a.nhseg);
        const int b = (int)(r / a.nhseg);
        const int cw0 = wc * WCH, fw0 = cw0 - CX * DIL;
        const int h_beg = hs * a.hseg_rows, h_end = min((hs + 1) * a.hseg_rows, a.H);
        const float *cbase = a.coarse + (size_t)b * a.H * a.W * a.cs_c + co0;
        const float *fbase = a.fine + (size_t)b * a.H * a.W * a.cs_f + ci0;

        float4 pa[NLA], pf[NLF];
        unsigned okbits = 0;
        // the fine row that ENTERS the ring at output row y is y + DIL*(KH-1-CY) + fine_dy
        auto issue = [&](int y) {
            okbits = 0;
#pragma unroll
            for (int it = 0; it < NLA; ++it) {
                const int q = tid + NTHR * it, pos = q / PA, part = q - pos * PA;
                const int cw = cw0 + pos;
                okbits |= ((q < NQA) && cw < a.W) ? (1u << it) : 0u;
                pa[it] = *reinterpret_cast<const float4 *>(
                    cbase + (unsigned)(y * a.W + min(cw, a.W - 1)) * (unsigned)a.cs_c + part * 4);
            }
            const int fh = y + DIL * (KH - 1 - CY) + a.fine_dy;
            const int fhc = min(max(fh, 0), a.H - 1);
#pragma unroll
            for (int it = 0; it < NLF; ++it) {
                const int q = tid + NTHR * it, lw = min(q / PF, FW - 1), part = q - (q / PF) * PF;
                const int fw = fw0 + lw, fwc = min(max(fw, 0), a.W - 1);
                okbits |= ((q < NQF) && fh == fhc && fw == fwc) ? (1u << (8 + it)) : 0u;
                pf[it] = *reinterpret_cast<const float4 *>(
                    fbase + (unsigned)(fhc * a.W + fwc) * (unsigned)a.cs_f + part * 4);
            }
        };
        auto commit = [&](int y) {
#pragma unroll
            for (int it = 0; it < NLA; ++it) {
                const int q = tid + NTHR * it, pos = q / PA, part = q - pos * PA;
                if (!((okbits >> it) & 1u)) pa[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < NQA) split_store(&sa[(part >> 3) * ATILE + pos * 32 + (part & 7) * 4], WCH * 32, pa[it], c_scale);
            }
            const int slot = slot_of(y + DIL * (KH - 1 - CY) + a.fine_dy);
#pragma unroll
            for (int it = 0; it < NLF; ++it) {
                const int q = tid + NTHR * it, lw = q / PF, part = q - lw * PF;
                if (!((okbits >> (8 + it)) & 1u)) pf[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < NQF)
                    split_store(&sf[(slot * NT + (part >> 3)) * FTILE + lw * 32 + (part & 7) * 4], FW * 32, pf[it], f_scale);
            }
        };

        __syncthreads();
        // prologue: the R-1 older rows of the first window
        for (int q = tid; q < (R - 1) * NQF; q += NTHR) {
            const int rr = q / NQF, q2 = q - rr * NQF;
            const int lw = q2 / PF, part = q2 - lw * PF;
            const int fh = h_beg - DIL * CY + a.fine_dy + rr, fw = fw0 + lw;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (fh >= 0 && fh < a.H && fw >= 0 && fw < a.W)
                v = *reinterpret_cast<const float4 *>(fbase + (unsigned)(fh * a.W + fw) * (unsigned)a.cs_f + part * 4);
            split_store(&sf[(slot_of(fh) * NT + (part >> 3)) * FTILE + lw * 32 + (part & 7) * 4], FW * 32, v, f_scale);
        }
        issue(h_beg);
        for (int y = h_beg; y < h_end; ++y) {
            __syncthreads();  // MFMAs of the previous row have read their operands
            commit(y);
            __syncthreads();
            if (y + 1 < h_end) issue(y + 1);  // in flight under this row's MFMAs
            bf16x8 af[3];
#pragma unroll
            for (int p = 0; p < NP; ++p) af[p] = frag(sa + mt * ATILE + p * WCH * 32, tr_row, tr_row + 4);
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                const unsigned short *frow =
                    sf + (slot_of(y + DIL * (kh - CY) + a.fine_dy) * NT + nt) * FTILE;
#pragma unroll
                for (int kw = 0; kw < KW; ++kw) {
                    bf16x8 bfr[3];
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        bfr[p] = frag(frow + p * FW * 32, tr_row + kw * DIL, tr_row + 4 + kw * DIL);
                    f32x16 c = acc[kh * KW + kw];
                    if constexpr (AR == 2) {
                        acc[kh * KW + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[0], c, 0, 0, 0);
                        continue;
                    }
                    if constexpr (AR == 3) {
                        acc[kh * KW + kw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(az_f16x8, af[0]), __builtin_bit_cast(az_f16x8, bfr[0]), c, 0, 0, 0);
                        continue;
                    }
                    if constexpr (AR == 1) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(az_f16x8, af[1]), __builtin_bit_cast(az_f16x8, bfr[0]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(az_f16x8, af[0]), __builtin_bit_cast(az_f16x8, bfr[1]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(az_f16x8, af[0]), __builtin_bit_cast(az_f16x8, bfr[0]), c, 0, 0, 0);
                        acc[kh * KW + kw] = c;
                        continue;
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bfr[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[0], c, 0, 0, 0);
                    acc[kh * KW + kw] = c;
                }
            }
        }
    }
    // D[i][j]: i = coarse channel (row map of the 32x32 MFMA), j = fine channel (lane & 31)
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int m = co0 + mt * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * half;
            atomicAdd(&a.ws[((size_t)(a.tap0 + t) * a.CM + m) * a.CN + ci0 + nt * 32 + row], (AR == 1 || AR == 3) ? acc[t][rg] * o_scale : acc[t][rg]);
        }
    }
}

// ws [taps][cm][cn] -> grad_w[(co*cn_real + ci)*taps + t] for co < cm_real, ci < cn_real (channel padding of
// the operation dropped); `flip`/`swap` are not needed: every layer's gradient is taken in its own layout
__global__ void __launch_bounds__(256)
wgrad2d_unpack_kernel(float *__restrict__ dst, const float *__restrict__ ws, int cm, int cn, int cm_real,
                      int cn_real, int taps) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cm_real * cn_real * taps) return;
    const int t = idx % taps, mn = idx / taps;
    const int ci = mn % cn_real, co = mn / cn_real;
    dst[idx] = ws[((size_t)t * cm + co) * cn + ci];
}

template <int MT, int NT, int KH, int KW, int DIL>
static int launch_w2(W2Args a, hipStream_t s) {
    const bool f16 = a.coarse_amax != nullptr && a.fine_amax != nullptr;
    const bool plain = a.plain_bf16 != 0;
    const int ncombo = (a.CM / (32 * MT)) * (a.CN / (32 * NT));
    a.nwchunk = (a.W + W2_WCH - 1) / W2_WCH;
    // Static work lists (as az_conv3d_wgrad.hip): block w of a combo takes items w, w + Wb, ...; the kernel
    // ends with its most loaded block, so pick (row segments, blocks per combo) with the item count a
    // near multiple of the block count while ncombo * Wb * MT*NT stays close to the resident waves.
    const int slots = 256 * 8 / (MT * NT);  // resident workgroups at 2 waves/SIMD
    const int wmax = max(8, (slots / ncombo) & ~7);
    const long long base_items = (long long)a.B * a.nwchunk;
    double best = -1.0;
    int best_w = 8, best_rows = a.H;
    constexpr int R = DIL * (KH - 1) + 1;
    for (int nseg = 1; nseg <= min(a.H, 32); ++nseg) {
        const int rows = (a.H + nseg - 1) / nseg;
        const int segs = (a.H + rows - 1) / rows;
        const long long items = base_items * segs;
        for (int w = wmax; w >= max(8, wmax / 2); w -= 8) {
            const long long per = (items + w - 1) / w;
            const double balance = (double)items / (double)(per * w);
            const double occ = (double)w / (double)wmax;
            const double amort = (double)rows / (double)(rows + R - 1 + 2);  // ring prologue per item
            const double score = balance * (0.5 + 0.5 * occ) * amort;
            if (score > best) { best = score; best_w = w; best_rows = rows; }
        }
    }
    a.blocks_per_combo = best_w;
    a.hseg_rows = best_rows;
    a.nhseg = (a.H + a.hseg_rows - 1) / a.hseg_rows;
    a.nitems = base_items * a.nhseg;
    if (a.nitems < a.blocks_per_combo) a.blocks_per_combo = (int)((a.nitems + 7) & ~7LL);
    if (plain) {
        if constexpr (KH == 3 && KW == 3 && DIL == 1) {  // (the GRU's convolutions: the only users)
            if (a.plain_bf16 == 2)
                hipLaunchKernelGGL((conv2d_wgrad_kernel<MT, NT, KH, KW, DIL, 3>), dim3(a.blocks_per_combo * ncombo),
                                   dim3(64 * MT * NT), 0, s, a);
            else
                hipLaunchKernelGGL((conv2d_wgrad_kernel<MT, NT, KH, KW, DIL, 2>), dim3(a.blocks_per_combo * ncombo),
                                   dim3(64 * MT * NT), 0, s, a);
        } else
            return AZ_EUNSUPPORTED;
    } else if (f16)
        hipLaunchKernelGGL((conv2d_wgrad_kernel<MT, NT, KH, KW, DIL, 1>), dim3(a.blocks_per_combo * ncombo),
                           dim3(64 * MT * NT), 0, s, a);
    else
        hipLaunchKernelGGL((conv2d_wgrad_kernel<MT, NT, KH, KW, DIL, 0>), dim3(a.blocks_per_combo * ncombo),
                           dim3(64 * MT * NT), 0, s, a);
    return az_launch_status();
}

template <int KH, int KW, int DIL>
static int dispatch_tiles(const W2Args &a, hipStream_t s) {
    const bool m2 = (a.CM % 64) == 0, n2 = (a.CN % 64) == 0;
    if (m2 && n2) return launch_w2<2, 2, KH, KW, DIL>(a, s);
    if (m2) return launch_w2<2, 1, KH, KW, DIL>(a, s);
    if (n2) return launch_w2<1, 2, KH, KW, DIL>(a, s);
    return launch_w2<1, 1, KH, KW, DIL>(a, s);
}

int az_conv2d_wgrad_r16_launch(float *ws, const float *coarse, const float *fine, int B, int H, int W, int cm, int cn,
                               int cs_c, int cs_f, hipStream_t s, const float *coarse_amax, const float *fine_amax);

extern "C" long long az_conv2d_wgrad_workspace(int cm, int cn, int kh, int kw) {
    if (cm <= 0 || cn <= 0 || cm % 32 || cn % 32 || kh <= 0 || kw <= 0) return AZ_EINVAL;
    return (long long)kh * kw * cm * cn * (long long)sizeof(float);
}

/* grad_w[cm_real][cn_real][kh][kw] = sum over pixels of grad_out[.., co] * in[.. + tap offset, ci]  (the
 * torch layout of a Conv2d weight).  cm / cn: channel counts of the OPERATION (multiples of 32; tensors whose
 * pixel stride exceeds their real channel count are read as zero-padded by the caller's layout). */
static int conv2d_wgrad_impl(float *grad_w, float *workspace, long long workspace_bytes, const float *grad_out,
                             const float *in, const float *go_amax, const float *in_amax, int B, int H, int W, int cm,
                             int cn, int cm_real, int cn_real, int go_cstride, int in_cstride, int kh, int kw,
                             int dilation, void *stream, int plain_bf16 = 0) {
    // grad_w == NULL: accumulate-only into an ALREADY ZERO workspace, which keeps the tap-major result (az_conv3d_wgrad_f16)
    AZ_REQUIRE_PTR(workspace); AZ_REQUIRE_PTR(grad_out); AZ_REQUIRE_PTR(in);
    AZ_REQUIRE(B > 0 && H > 0 && W > 0);
    const long long need = az_conv2d_wgrad_workspace(cm, cn, kh, kw);
    if (need < 0) return AZ_EUNSUPPORTED;
    if (workspace_bytes < need) return AZ_EWORKSPACE;
    AZ_REQUIRE(cm_real > 0 && cm_real <= cm && cn_real > 0 && cn_real <= cn);
    AZ_REQUIRE(go_cstride >= cm && in_cstride >= cn && go_cstride % 4 == 0 && in_cstride % 4 == 0);
    if ((long long)H * W * (go_cstride > in_cstride ? go_cstride : in_cstride) > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    hipStream_t s = az_stream(stream);
    if (grad_w && hipMemsetAsync(workspace, 0, (size_t)need, s) != hipSuccess) return AZ_ELAUNCH;
    W2Args a{};
    a.coarse = grad_out; a.fine = in; a.ws = workspace; a.coarse_amax = go_amax; a.fine_amax = in_amax;
    a.plain_bf16 = plain_bf16;
    a.B = B; a.H = H; a.W = W; a.CM = cm; a.CN = cn; a.cs_c = go_cstride; a.cs_f = in_cstride;
    int rc = AZ_EUNSUPPORTED;
    const int r16 = az_options().conv2d_wgrad_r16;
    if (kh == 3 && kw == 3 && dilation == 1 && r16 && !plain_bf16 && (cm == 32 || cm == 64) && (cn == 32 || cn == 64))
        rc = az_conv2d_wgrad_r16_launch(workspace, grad_out, in, B, H, W, cm, cn, go_cstride, in_cstride, s, go_amax, in_amax);  // az_conv2d_wgrad16.hip
    else if (kh == 3 && kw == 3 && dilation == 1) rc = dispatch_tiles<3, 3, 1>(a, s);
    else if (kh == 3 && kw == 3 && dilation == 2) rc = dispatch_tiles<3, 3, 2>(a, s);
    else if (kh == 1 && kw == 1) rc = dispatch_tiles<1, 1, 1>(a, s);
    else if (kh == 3 && kw == 5 && dilation == 1) {
        // 15 taps do not fit one wave's accumulators: one launch per kernel row (five taps each)
        for (int r = 0; r < 3; ++r) {
            a.fine_dy = r - 1; a.tap0 = r * 5;
            rc = dispatch_tiles<1, 5, 1>(a, s);
            if (rc != AZ_OK) return rc;
        }
    }
    if (rc != AZ_OK) return rc;
    if (!grad_w) return az_launch_status();
    const int total = cm_real * cn_real * kh * kw;
    hipLaunchKernelGGL(wgrad2d_unpack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, grad_w, workspace, cm,
                       cn, cm_real, cn_real, kh * kw);
    return az_launch_status();
}

extern "C" int az_conv2d_wgrad(float *grad_w, float *workspace, long long workspace_bytes, const float *grad_out,
                               const float *in, int B, int H, int W, int cm, int cn, int cm_real, int cn_real,
                               int go_cstride, int in_cstride, int kh, int kw, int dilation, void *stream) {
    return conv2d_wgrad_impl(grad_w, workspace, workspace_bytes, grad_out, in, nullptr, nullptr, B, H, W, cm, cn, cm_real,
                             cn_real, go_cstride, in_cstride, kh, kw, dilation, stream);
}

/* az_conv2d_wgrad on the f16x3 arithmetic: go_amax / in_amax = amax arrays of grad_out and in */
extern "C" int az_conv2d_wgrad_f16(float *grad_w, float *workspace, long long workspace_bytes, const float *grad_out,
                                   const float *in, const float *go_amax, const float *in_amax, int B, int H, int W,
                                   int cm, int cn, int cm_real, int cn_real, int go_cstride, int in_cstride, int kh,
                                   int kw, int dilation, void *stream) {
    AZ_REQUIRE_PTR(go_amax); AZ_REQUIRE_PTR(in_amax);
#ifdef WGRAD_SKIP  // timing-only build (tools/abl_step_sensitivity.sh): accumulate-only launches are left out, the zeroed workspace IS the gradient
    if (!grad_w) return AZ_OK;
#endif
    return conv2d_wgrad_impl(grad_w, workspace, workspace_bytes, grad_out, in, go_amax, in_amax, B, H, W, cm, cn, cm_real,
                             cn_real, go_cstride, in_cstride, kh, kw, dilation, stream);
}

/* az_conv2d_wgrad_bf16 with ONE FP16 part per operand ("f16x1", az_conv2d_h1_fwd): go_amax / in_amax may each be NULL */
extern "C" int az_conv2d_wgrad_h1(float *grad_w, float *workspace, long long workspace_bytes, const float *grad_out,
                                  const float *in, const float *go_amax, const float *in_amax, int B, int H, int W, int cm, int cn,
                                  int cm_real, int cn_real, int go_cstride, int in_cstride, void *stream) {
    return conv2d_wgrad_impl(grad_w, workspace, workspace_bytes, grad_out, in, go_amax, in_amax, B, H, W, cm, cn, cm_real,
                             cn_real, go_cstride, in_cstride, 3, 3, 1, stream, 2);
}

/* az_conv2d_wgrad with plain bf16 operands (round-to-nearest, one MFMA per 16-deep block, fp32 accumulation, fp32 in / out):
 * the weight gradient of the RAFT-Stereo GRU's convolutions in the reference's autocast arithmetic; 3x3, dilation 1 */
extern "C" int az_conv2d_wgrad_bf16(float *grad_w, float *workspace, long long workspace_bytes, const float *grad_out,
                                    const float *in, int B, int H, int W, int cm, int cn, int cm_real, int cn_real,
                                    int go_cstride, int in_cstride, void *stream) {
    return conv2d_wgrad_impl(grad_w, workspace, workspace_bytes, grad_out, in, nullptr, nullptr, B, H, W, cm, cn, cm_real,
                             cn_real, go_cstride, in_cstride, 3, 3, 1, stream, 1);
}
