// Weight gradients of the 3x3x3 conv / transposed-conv layers on fp32 MFMA.
//
// Every layer's weight gradient has the form
//   G[m][n][k] = sum_pos coarse[pos][m] * fine[S*pos - 1 + k][n],   k = (kd,kh,kw)
// with (coarse, fine) = (grad_out, input) for Conv3d (G = dW[co][ci][k]) and
// (coarse, fine) = (input, grad_out) for ConvTranspose3d stride 2 (G = dW[ci][co][k],
// reference psmnet_3.py:34-58) -- both are PyTorch's native weight layouts.
//
// (Staging is software-pipelined: the next row's operands are fetched branch-free into
// registers while the current row's 144 MFMAs run; the three fine rows form a rolling
// window in LDS, slot = row mod 3, so each fine row is fetched once per kd.)
// GEMM view: M = 32 coarse channels, N = 32 fine channels, K = positions, one
// 32x32 accumulator per tap.  A wavefront (= workgroup) keeps the 9 taps of ONE kd
// (144 accumulator registers) and walks a strided list of (b, plane, row-segment,
// w-chunk) work items; per coarse row it stages the coarse chunk and the three fine
// rows in its private LDS region and feeds v_mfma_f32_32x32x2_f32 with one dword of
// each per lane (lane = channel, the two K slices = two neighbouring positions).
// Only when its list is exhausted does it add the 9x32x32 block into the
// tap-major workspace with float atomics (128-B segments), so atomics are ~1e-4 of
// the flops.  az_conv3d_wgrad_unpack transposes [k][m][n] -> [m][n][k].
#include <stdlib.h>

#include "az_roll_common.h"
#include "az_options.h"
#include "az_launch_math.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// coarse positions per staged chunk (even).  PSMNet's widths are 240 / 120 / 60: 30 and 20
// divides them exactly (no MFMA on padding positions); the stride-2 kernels keep 16 (a wider
// chunk spills at the 2-waves/SIMD register budget).
#define WG_WCH1 30
#define WG_WCH2 16

struct WgArgs {
    const float *coarse, *fine;
    float *ws;  // [27][CM][CN]
    int B, Dc, Hc, Wc, Df, Hf, Wf;
    int hseg_rows, nhseg, nwchunk;
    long long nitems;
    int waves_per_combo;
    int order;  // 0: (b, cd, hs, wc) linear; 1: XCD-chunked list with the depth index fastest
    const float *coarse_amax, *fine_amax;  // f16x3 kernels: device scalars max |coarse|, max |fine|
};

// item -> (row-chunk, row segment, coarse depth, batch).  Waves w, w+8, ... run on one XCD (its
// L2); order 1 gives every XCD a contiguous eighth of the list and makes the depth index the
// fastest one, so the waves that are resident together on an XCD work on neighbouring depths of
// the SAME rows: fine plane S*cd-1+kd is then read by (cd,kd), (cd+1,kd-1), (cd+2,kd-2) within
// one L2 instead of three times from HBM (PMC: 3.8 GB fetched per 32->32 V0 launch for 1.6 GB
// of operands with the linear order).
__device__ __forceinline__ void wg_decode(const WgArgs &a, long long item, int &wc, int &hs, int &cd, int &b) {
    if (a.order == 0) {
        long long r = item;
        wc = (int)(r % a.nwchunk); r /= a.nwchunk;
        hs = (int)(r % a.nhseg); r /= a.nhseg;
        cd = (int)(r % a.Dc);
        b = (int)(r / a.Dc);
        return;
    }
    const long long q8 = a.nitems >> 3, r8 = a.nitems & 7;
    const int x = (int)(item & 7);
    long long r = (x < r8 ? x * (q8 + 1) : r8 * (q8 + 1) + (x - r8) * q8) + (item >> 3);
    cd = (int)(r % a.Dc); r /= a.Dc;
    wc = (int)(r % a.nwchunk); r /= a.nwchunk;
    hs = (int)(r % a.nhseg);
    b = (int)(r / a.nhseg);
}

// (stride 2 stages twice the fine rows: it gets the 1-wave/SIMD register budget instead of spilling)
template <int CM, int CN, int S>
__global__ void __launch_bounds__(64, S == 1 ? 2 : 1)
conv3d_wgrad_kernel(const WgArgs a) {
    constexpr int WCH = (S == 1) ? WG_WCH1 : WG_WCH2;  // coarse positions per chunk
    constexpr int FW = S * (WCH - 1) + 3;    // fine positions per staged row
    constexpr int MT = CM / 32, NT = CN / 32, NCOMBO = 3 * MT * NT;
    constexpr int NEW = S;                   // fine rows that enter the 3-row window per step
    constexpr int NQA = WCH * 8, NLA = (NQA + 63) / 64;  // float4 pieces per lane: coarse row chunk
    constexpr int NQF = NEW * FW * 8, NLF = (NQF + 63) / 64;  // ... new fine rows
    __shared__ __attribute__((aligned(16))) float sa[WCH * 32];
    __shared__ __attribute__((aligned(16))) float sf[3 * FW * 32];  // row fh lives in slot fh mod 3

    const int lane = threadIdx.x, row = lane & 31, half = lane >> 5;
    // blocks b and b+8 share an XCD: the NCOMBO waves that walk the SAME work list (same
    // coarse rows, neighbouring fine planes) are given ids 8 apart so they share one L2.
    const int grp = blockIdx.x / (8 * NCOMBO), rem = blockIdx.x % (8 * NCOMBO);
    int combo = rem >> 3;
    const int widx = grp * 8 + (rem & 7);
    const int nt = combo % NT; combo /= NT;
    const int mt = combo % MT;
    const int kd = combo / MT;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long item = widx; item < a.nitems; item += a.waves_per_combo) {
        int wc, hs, cd, b;
        wg_decode(a, item, wc, hs, cd, b);
        const int fd = S * cd - 1 + kd;
        if (fd < 0 || fd >= a.Df) continue;  // wave-uniform
        const int cw0 = wc * WCH, fw0 = S * cw0 - 1;
        const int h_beg = hs * a.hseg_rows, h_end = min((hs + 1) * a.hseg_rows, a.Hc);
        const float *cbase = a.coarse + (((size_t)b * a.Dc + cd) * a.Hc) * a.Wc * CM + mt * 32;
        const float *fbase = a.fine + (((size_t)b * a.Df + fd) * a.Hf) * a.Wf * CN + nt * 32;

        float4 pa[NLA], pf[NLF];
        unsigned okbits = 0;  // bit it: pa[it] is real data; bit 8+it: pf[it] (else zero padding)
        // fetch (branch-free) what step `ch` adds: its coarse row chunk and its NEW newest fine rows.
        // Loads only, from clamped (always valid) addresses; the padding is applied at commit time
        // so that nothing waits on these registers before this row's MFMAs have been issued.
        auto issue = [&](int ch) {
            okbits = 0;
#pragma unroll
            for (int it = 0; it < NLA; ++it) {
                const int q = lane + 64 * it, pos = q >> 3, part = q & 7;
                const int cw = cw0 + pos;
                okbits |= ((q < NQA) && cw < a.Wc) ? (1u << it) : 0u;
                pa[it] = *reinterpret_cast<const float4 *>(
                    cbase + ((size_t)ch * a.Wc + min(cw, a.Wc - 1)) * CM + part * 4);
            }
#pragma unroll
            for (int it = 0; it < NLF; ++it) {
                const int q = lane + 64 * it, part = q & 7, p = q >> 3;
                const int rr = min(p / FW, NEW - 1), lw = p - (p / FW) * FW;
                const int fh = S * ch + 2 - NEW + rr, fw = fw0 + lw;
                const int fhc = min(max(fh, 0), a.Hf - 1), fwc = min(max(fw, 0), a.Wf - 1);
                okbits |= ((q < NQF) && fh == fhc && fw == fwc) ? (1u << (8 + it)) : 0u;
                pf[it] = *reinterpret_cast<const float4 *>(fbase + ((size_t)fhc * a.Wf + fwc) * CN + part * 4);
            }
        };
        auto commit = [&](int ch) {
#pragma unroll
            for (int it = 0; it < NLA; ++it) {
                const int q = lane + 64 * it;
                if (!((okbits >> it) & 1u)) pa[it] = zero4;
                if (q < NQA) *reinterpret_cast<float4 *>(&sa[(q >> 3) * 32 + (q & 7) * 4]) = pa[it];
            }
#pragma unroll
            for (int it = 0; it < NLF; ++it) {
                const int q = lane + 64 * it, part = q & 7, p = q >> 3;
                const int rr = p / FW, lw = p - rr * FW;
                const int fh = S * ch + 2 - NEW + rr;
                const int slot = (fh + 3) % 3;
                if (!((okbits >> (8 + it)) & 1u)) pf[it] = zero4;
                if (q < NQF) *reinterpret_cast<float4 *>(&sf[(slot * FW + lw) * 32 + part * 4]) = pf[it];
            }
        };

        __syncthreads();
        // prologue: the rows of the first window that `issue` does not bring in
        for (int q = lane; q < (3 - NEW) * FW * 8; q += 64) {
            const int part = q & 7, p = q >> 3;
            const int rr = p / FW, lw = p - rr * FW;
            const int fh = S * h_beg - 1 + rr, fw = fw0 + lw;
            float4 v = zero4;
            if (fh >= 0 && fh < a.Hf && fw >= 0 && fw < a.Wf)
                v = *reinterpret_cast<const float4 *>(fbase + ((size_t)fh * a.Wf + fw) * CN + part * 4);
            *reinterpret_cast<float4 *>(&sf[(((fh + 3) % 3) * FW + lw) * 32 + part * 4]) = v;
        }
        issue(h_beg);
        for (int ch = h_beg; ch < h_end; ++ch) {
            __syncthreads();  // MFMAs of the previous row have read their operands
            commit(ch);
            __syncthreads();
            if (ch + 1 < h_end) issue(ch + 1);  // in flight under this row's 144 MFMAs
            const int s0 = (S * ch - 1 + 3) % 3;  // slot of kh = 0; kh = 1, 2 follow cyclically
            const float *f0 = &sf[s0 * FW * 32 + row];
            const float *f1 = &sf[((s0 + 1) % 3) * FW * 32 + row];
            const float *f2 = &sf[((s0 + 2) % 3) * FW * 32 + row];
#pragma unroll 5
            for (int q = 0; q < WCH / 2; ++q) {
                const int pos = 2 * q + half;
                const float av = sa[pos * 32 + row];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int o = (S * pos + kw) * 32;
                    acc[0 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, f0[o], acc[0 + kw], 0, 0, 0);
                    acc[3 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, f1[o], acc[3 + kw], 0, 0, 0);
                    acc[6 + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, f2[o], acc[6 + kw], 0, 0, 0);
                }
            }
        }
    }
    // D[i][j]: i = coarse channel (row map), j = fine channel (lane & 31)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tap = kd * 9 + t;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int m = mt * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * half;
            atomicAdd(&a.ws[((size_t)tap * CM + m) * CN + nt * 32 + row], acc[t][rg]);
        }
    }
}


// ---- bf16x6 variant ------------------------------------------------------------------------
// Same decomposition, but the products run on v_mfma_f32_32x32x16_bf16 with both operands split
// exactly into three bf16 parts (hi, mid, lo; six significant partial products, fp32
// accumulation -- see az_conv3d.hip).  K = 16 POSITIONS per MFMA, so a lane needs 8 consecutive
// positions of its channel: the staged [position][channel] bf16 tiles are read with
// ds_read_b64_tr_b16 (the hardware transpose read: per 16-lane group, lane 4q+p gives the
// address of row q / columns 4p..4p+3 and lane i receives column i of the 4 rows).  Rows are
// addressed individually, which also covers the stride-2 position map and the kw tap shift.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

#define X6_WCH 16  // coarse positions per chunk = one K16 block

#ifndef WGX6_BPIPE
#define WGX6_BPIPE 0
#endif
#ifndef WGX6_S2_OCC
#define WGX6_S2_OCC 1
#endif
#define WGX6_S2_WAVES(S) ((S) == 1 ? 2 : WGX6_S2_OCC)
// AR: 0 = bf16x6, 1 = f16x3 (two scaled fp16 parts, three MFMAs per tap; az_roll_common.h) -- the LDS images keep their
// three-part strides, the third part is then unused
template <int CM, int CN, int S, int AR = 0>
__global__ void __launch_bounds__(64, AR ? 2 : WGX6_S2_WAVES(S))
conv3d_wgrad_x6_kernel(const WgArgs a) {
    constexpr int WCH = X6_WCH;
    constexpr int NP = AR ? 2 : 3;
    float c_scale = 1.f, f_scale = 1.f, o_scale = 1.f;
    if (AR) {
        const int kc = az_f16_scale_exp(az_amax_read(a.coarse_amax));
        const int kf = az_f16_scale_exp(az_amax_read(a.fine_amax));
        c_scale = az_pow2(kc); f_scale = az_pow2(kf);
        o_scale = ldexpf(1.f, -(kc + kf));
    }
    constexpr int FW = S * (WCH - 1) + 3;  // fine positions per staged row
    constexpr int MT = CM / 32, NT = CN / 32, NCOMBO = 3 * MT * NT;
    constexpr int NEW = S;
    constexpr int NQA = WCH * 8, NLA = (NQA + 63) / 64;
    constexpr int NQF = NEW * FW * 8, NLF = (NQF + 63) / 64;
    // bf16 images, 32 channels (64 B) per position: [part][pos][32]
    __shared__ __attribute__((aligned(16))) unsigned short sa[3 * WCH * 32];
    __shared__ __attribute__((aligned(16))) unsigned short sf[3 * 3 * FW * 32];  // [slot][part][pos][32]

    const int lane = threadIdx.x, row = lane & 31, half = lane >> 5;
    const int grp = blockIdx.x / (8 * NCOMBO), rem = blockIdx.x % (8 * NCOMBO);
    int combo = rem >> 3;
    const int widx = grp * 8 + (rem & 7);
    const int nt = combo % NT; combo /= NT;
    const int mt = combo % MT;
    const int kd = combo / MT;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // transpose-read geometry of this lane: 16-lane group g = lane>>4 reads columns
    // 16*(g&1).. of rows 8*(g>>1) + {0..3} and {4..7}; inside the group lane t = 4q+p
    // supplies the address of row q, columns 4p..4p+3.
    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int tr_col = 16 * ((lane >> 4) & 1) + 4 * tp;
    const int tr_row = 8 * (lane >> 5) + tq;  // + 4 for the second read

    auto split_store = [&](unsigned short *dst_part0, int part_stride, const float4 &v, float scale_) {
        // exact 3-way bf16 split (or the scaled two-part fp16 split); dst_part0 points at 4 channels of part 0
        uint2 hi, mid, lo;
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
    // fragment of 8 consecutive K rows (positions) for this lane's column, rows given by rowfn(k)
    auto frag = [&](const unsigned short *img, int r0, int r1) -> bf16x8 {
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + r0 * 32 + tr_col));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + r1 * 32 + tr_col));
        s16x8 v;
        v[0] = lo4[0]; v[1] = lo4[1]; v[2] = lo4[2]; v[3] = lo4[3];
        v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
        return __builtin_bit_cast(bf16x8, v);
    };

    for (long long item = widx; item < a.nitems; item += a.waves_per_combo) {
        int wc, hs, cd, b;
        wg_decode(a, item, wc, hs, cd, b);
        const int fd = S * cd - 1 + kd;
        if (fd < 0 || fd >= a.Df) continue;  // wave-uniform
        const int cw0 = wc * WCH, fw0 = S * cw0 - 1;
        const int h_beg = hs * a.hseg_rows, h_end = min((hs + 1) * a.hseg_rows, a.Hc);
        const float *cbase = a.coarse + (((size_t)b * a.Dc + cd) * a.Hc) * a.Wc * CM + mt * 32;
        const float *fbase = a.fine + (((size_t)b * a.Df + fd) * a.Hf) * a.Wf * CN + nt * 32;

        float4 pa[NLA], pf[NLF];
        unsigned okbits = 0;  // bit it: pa[it] is real data; bit 8+it: pf[it] (else zero padding)
        // loads only: clamped (always valid) addresses, padding applied at commit time -- touching
        // the loaded registers here puts the whole memory latency in front of this row's MFMAs
        auto issue = [&](int ch) {
            int lane = threadIdx.x;
            if (AR) asm volatile("" : "+v"(lane));  // (recompute the piece offsets per call: at the register limit)
            okbits = 0;
#pragma unroll
            for (int it = 0; it < NLA; ++it) {
                const int q = lane + 64 * it, pos = q >> 3, part = q & 7;
                const int cw = cw0 + pos;
                okbits |= ((q < NQA) && cw < a.Wc) ? (1u << it) : 0u;
                pa[it] = *reinterpret_cast<const float4 *>(
                    cbase + ((size_t)ch * a.Wc + min(cw, a.Wc - 1)) * CM + part * 4);
            }
#pragma unroll
            for (int it = 0; it < NLF; ++it) {
                const int q = lane + 64 * it, part = q & 7, p = q >> 3;
                const int rr = min(p / FW, NEW - 1), lw = p - (p / FW) * FW;
                const int fh = S * ch + 2 - NEW + rr, fw = fw0 + lw;
                const int fhc = min(max(fh, 0), a.Hf - 1), fwc = min(max(fw, 0), a.Wf - 1);
                okbits |= ((q < NQF) && fh == fhc && fw == fwc) ? (1u << (8 + it)) : 0u;
                pf[it] = *reinterpret_cast<const float4 *>(fbase + ((size_t)fhc * a.Wf + fwc) * CN + part * 4);
            }
        };
        auto commit = [&](int ch) {
            int lane = threadIdx.x;
            if (AR) asm volatile("" : "+v"(lane));
#pragma unroll
            for (int it = 0; it < NLA; ++it) {
                const int q = lane + 64 * it;
                if (!((okbits >> it) & 1u)) pa[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < NQA) split_store(&sa[(q >> 3) * 32 + (q & 7) * 4], WCH * 32, pa[it], c_scale);
            }
#pragma unroll
            for (int it = 0; it < NLF; ++it) {
                const int q = lane + 64 * it, part = q & 7, p = q >> 3;
                const int rr = p / FW, lw = p - rr * FW;
                const int fh = S * ch + 2 - NEW + rr;
                const int slot = (fh + 3) % 3;
                if (!((okbits >> (8 + it)) & 1u)) pf[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < NQF) split_store(&sf[(slot * 3 * FW + lw) * 32 + part * 4], FW * 32, pf[it], f_scale);
            }
        };

        __syncthreads();
        // prologue: rows of the first window that `issue` does not bring in
        for (int q = lane; q < (3 - NEW) * FW * 8; q += 64) {
            const int part = q & 7, p = q >> 3;
            const int rr = p / FW, lw = p - rr * FW;
            const int fh = S * h_beg - 1 + rr, fw = fw0 + lw;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (fh >= 0 && fh < a.Hf && fw >= 0 && fw < a.Wf)
                v = *reinterpret_cast<const float4 *>(fbase + ((size_t)fh * a.Wf + fw) * CN + part * 4);
            split_store(&sf[(((fh + 3) % 3) * 3 * FW + lw) * 32 + part * 4], FW * 32, v, f_scale);
        }
        issue(h_beg);
        for (int ch = h_beg; ch < h_end; ++ch) {
            __syncthreads();
            commit(ch);
            __syncthreads();
            if (ch + 1 < h_end) issue(ch + 1);
            const int s0 = (S * ch - 1 + 3) % 3;  // slot of kh = 0
            // A fragments (coarse): parts hi/mid/lo, K rows = positions 8*(lane>>5) + 0..7
            bf16x8 af[3];
#pragma unroll
            for (int p = 0; p < NP; ++p) af[p] = frag(sa + p * WCH * 32, tr_row, tr_row + 4);
#if WGX6_BPIPE
            // fine-row fragments one tap ahead of their MFMAs (ping-pong registers, order pinned)
            auto load_bf = [&](bf16x8 (&bfr)[3], int t) {
                const unsigned short *frow = sf + ((s0 + t / 3) % 3) * 3 * FW * 32;
                const int kw = t % 3;
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bfr[p] = frag(frow + p * FW * 32, S * tr_row + kw, S * (tr_row + 4) + kw);
            };
            auto tap_mfma = [&](int t, const bf16x8 (&bfr)[3]) {
                f32x16 c = acc[t];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bfr[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[0], c, 0, 0, 0);
                acc[t] = c;
            };
            bf16x8 b0[3], b1[3];
            load_bf(b0, 0);
#pragma unroll
            for (int t = 0; t < 9; t += 2) {
                if (t + 1 < 9) load_bf(b1, t + 1);
                __builtin_amdgcn_sched_barrier(0);
                tap_mfma(t, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < 9) {
                    if (t + 2 < 9) load_bf(b0, t + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    tap_mfma(t + 1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#else
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const unsigned short *frow = sf + ((s0 + kh) % 3) * 3 * FW * 32;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    bf16x8 bfr[3];
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        bfr[p] = frag(frow + p * FW * 32, S * tr_row + kw, S * (tr_row + 4) + kw);
                    f32x16 c = acc[kh * 3 + kw];
                    if constexpr (AR) {
                        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, af[1]), __builtin_bit_cast(h8, bfr[0]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, af[0]), __builtin_bit_cast(h8, bfr[1]), c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, af[0]), __builtin_bit_cast(h8, bfr[0]), c, 0, 0, 0);
                        acc[kh * 3 + kw] = c;
                        continue;
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bfr[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bfr[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bfr[0], c, 0, 0, 0);
                    acc[kh * 3 + kw] = c;
                }
            }
#endif
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tap = kd * 9 + t;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int m = mt * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * half;
            atomicAdd(&a.ws[((size_t)tap * CM + m) * CN + nt * 32 + row], AR ? acc[t][rg] * o_scale : acc[t][rg]);
        }
    }
}

// ---- bf16x6, stride 1: the same decomposition walked by FINE rows -------------------------------------------
// conv3d_wgrad_x6_kernel walks coarse rows: per row it reads the coarse fragments once and the fine fragments of
// all nine (kh, kw) taps -- 60 transposing LDS reads per 54 MFMAs, a third of the kernel's time in the ablation of
// profiles/r02_clock_pipe_ablation.md section 7.  Fine row r meets coarse rows r+1, r, r-1 under kh = 0, 1, 2: walking
// the FINE rows, one fragment triple of (row r, shift kw) feeds the three kh taps, and the three coarse rows sit in
// a rolling LDS window: 36 reads per 54 MFMAs.  Staging per step is unchanged (one coarse row chunk, one fine row).
template <int CM, int CN>
__global__ void __launch_bounds__(64, 2)
conv3d_wgrad_x6_fw_kernel(const WgArgs a) {
    constexpr int WCH = X6_WCH, FW = WCH + 2;
    constexpr int MT = CM / 32, NT = CN / 32, NCOMBO = 3 * MT * NT;
    constexpr int NQA = WCH * 8, NLA = (NQA + 63) / 64;
    constexpr int NQF = FW * 8, NLF = (NQF + 63) / 64;
    __shared__ __attribute__((aligned(16))) unsigned short sa[3 * 3 * WCH * 32];  // [row mod 3][part][pos][32]
    __shared__ __attribute__((aligned(16))) unsigned short sf[3 * FW * 32];       // [part][pos][32]

    const int lane = threadIdx.x, row = lane & 31, half = lane >> 5;
    const int grp = blockIdx.x / (8 * NCOMBO), rem = blockIdx.x % (8 * NCOMBO);
    int combo = rem >> 3;
    const int widx = grp * 8 + (rem & 7);
    const int nt = combo % NT; combo /= NT;
    const int mt = combo % MT;
    const int kd = combo / MT;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int tq = (lane & 15) >> 2, tp = lane & 3;
    const int tr_col = 16 * ((lane >> 4) & 1) + 4 * tp;
    const int tr_row = 8 * (lane >> 5) + tq;
    auto split_store = [&](unsigned short *dst_part0, int part_stride, const float4 &v) {
        uint2 hi, mid, lo;
        az_split3_bf16x4(v, hi, mid, lo);
        *reinterpret_cast<uint2 *>(dst_part0) = hi;
        *reinterpret_cast<uint2 *>(dst_part0 + part_stride) = mid;
        *reinterpret_cast<uint2 *>(dst_part0 + 2 * part_stride) = lo;
    };
    auto frag = [&](const unsigned short *img, int r0, int r1) -> bf16x8 {
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + r0 * 32 + tr_col));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(img + r1 * 32 + tr_col));
        s16x8 v;
        v[0] = lo4[0]; v[1] = lo4[1]; v[2] = lo4[2]; v[3] = lo4[3];
        v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
        return __builtin_bit_cast(bf16x8, v);
    };

    for (long long item = widx; item < a.nitems; item += a.waves_per_combo) {
        int wc, hs, cd, b;
        wg_decode(a, item, wc, hs, cd, b);
        const int fd = cd - 1 + kd;
        if (fd < 0 || fd >= a.Df) continue;  // wave-uniform
        const int cw0 = wc * WCH, fw0 = cw0 - 1;
        const int h_beg = hs * a.hseg_rows, h_end = min((hs + 1) * a.hseg_rows, a.Hc);
        const float *cbase = a.coarse + (((size_t)b * a.Dc + cd) * a.Hc) * a.Wc * CM + mt * 32;
        const float *fbase = a.fine + (((size_t)b * a.Df + fd) * a.Hf) * a.Wf * CN + nt * 32;

        // step r = fine row r; it needs coarse rows r-1 .. r+1 of this segment in the window: the step brings
        // fine row r and coarse row r+1 (rows outside the segment / the plane are zero)
        float4 pa[NLA], pf[NLF];
        unsigned okbits = 0;
        auto issue = [&](int r) {
            okbits = 0;
            const int ch = r + 1, chc = min(max(ch, 0), a.Hc - 1);
            const bool ch_ok = ch >= h_beg && ch < h_end;
#pragma unroll
            for (int it = 0; it < NLA; ++it) {
                const int q = lane + 64 * it, pos = q >> 3, part = q & 7;
                const int cw = cw0 + pos;
                okbits |= ((q < NQA) && cw < a.Wc && ch_ok) ? (1u << it) : 0u;
                pa[it] = *reinterpret_cast<const float4 *>(cbase + ((size_t)chc * a.Wc + min(cw, a.Wc - 1)) * CM + part * 4);
            }
            const int fhc = min(max(r, 0), a.Hf - 1);
#pragma unroll
            for (int it = 0; it < NLF; ++it) {
                const int q = lane + 64 * it, part = q & 7, lw = min(q >> 3, FW - 1);
                const int fw = fw0 + lw, fwc = min(max(fw, 0), a.Wf - 1);
                okbits |= ((q < NQF) && r == fhc && fw == fwc) ? (1u << (8 + it)) : 0u;
                pf[it] = *reinterpret_cast<const float4 *>(fbase + ((size_t)fhc * a.Wf + fwc) * CN + part * 4);
            }
        };
        auto commit = [&](int r) {
            const int slot = (r + 1 + 3) % 3;
#pragma unroll
            for (int it = 0; it < NLA; ++it) {
                const int q = lane + 64 * it;
                if (!((okbits >> it) & 1u)) pa[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < NQA) split_store(&sa[(slot * 3 * WCH + (q >> 3)) * 32 + (q & 7) * 4], WCH * 32, pa[it]);
            }
#pragma unroll
            for (int it = 0; it < NLF; ++it) {
                const int q = lane + 64 * it;
                if (!((okbits >> (8 + it)) & 1u)) pf[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < NQF) split_store(&sf[(q >> 3) * 32 + (q & 7) * 4], FW * 32, pf[it]);
            }
        };

        __syncthreads();
        // the window starts empty: rows h_beg-2, h_beg-1 (slots of r-1, r at the first step) are zero
        for (int q = lane; q < 2 * 3 * WCH * 8; q += 64) {
            const int sl = (q / (3 * WCH * 8) == 0) ? (h_beg - 2 + 3) % 3 : (h_beg - 1 + 3) % 3;
            *reinterpret_cast<uint2 *>(&sa[sl * 3 * WCH * 32 + (q % (3 * WCH * 8)) * 4]) = make_uint2(0u, 0u);
        }
        issue(h_beg - 1);
        for (int r = h_beg - 1; r <= h_end; ++r) {
            __syncthreads();
            commit(r);
            __syncthreads();
            if (r + 1 <= h_end) issue(r + 1);
            if (r < 0 || r >= a.Hf) continue;  // a zero fine row adds nothing (wave-uniform)
            // coarse fragments of rows r+1 (kh 0), r (kh 1), r-1 (kh 2): parts hi/mid/lo
            bf16x8 af[3][3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const unsigned short *arow = sa + ((r + 1 - kh + 3) % 3) * 3 * WCH * 32;
#pragma unroll
                for (int p = 0; p < 3; ++p) af[kh][p] = frag(arow + p * WCH * 32, tr_row, tr_row + 4);
            }
            const bool ok0 = r + 1 >= h_beg && r + 1 < h_end, ok1 = r >= h_beg && r < h_end, ok2 = r - 1 >= h_beg && r - 1 < h_end;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                bf16x8 bfr[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) bfr[p] = frag(sf + p * FW * 32, tr_row + kw, tr_row + 4 + kw);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    if (!(kh == 0 ? ok0 : kh == 1 ? ok1 : ok2)) continue;  // coarse row outside the segment (wave-uniform)
                    f32x16 c = acc[kh * 3 + kw];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kh][2], bfr[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kh][0], bfr[2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kh][1], bfr[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kh][1], bfr[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kh][0], bfr[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kh][0], bfr[0], c, 0, 0, 0);
                    acc[kh * 3 + kw] = c;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int tap = kd * 9 + t;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int m = mt * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * half;
            atomicAdd(&a.ws[((size_t)tap * CM + m) * CN + nt * 32 + row], acc[t][rg]);
        }
    }
}

__global__ void __launch_bounds__(256)
wgrad_unpack_kernel(float *__restrict__ dst, const float *__restrict__ ws, int cm, int cn) {
    const int idx = blockIdx.x * 256 + threadIdx.x;  // over [m][n][27]
    if (idx >= cm * cn * 27) return;
    const int tap = idx % 27, mn = idx / 27;
    dst[idx] = ws[(size_t)tap * cm * cn + mn];
}

template <int CM, int CN, int S, int PREC>
static int launch_wgrad(WgArgs a, hipStream_t s) {
    constexpr int WCH = (PREC == 1 || PREC == 3) ? X6_WCH : (S == 1) ? WG_WCH1 : WG_WCH2;
    constexpr int NCOMBO = 3 * (CM / 32) * (CN / 32);
    a.nwchunk = (a.Wc + WCH - 1) / WCH;
    // Static work lists: wave w of a combo takes items w, w + W, w + 2W, ...  The kernel ends
    // with its most loaded wave, so pick (row segments, waves per combo W) such that the item
    // count is (nearly) a multiple of W while W * NCOMBO stays close to the ~2048 resident
    // waves: score = balance * occupancy, >= 4 items per wave to amortise the per-item prologue.
    const int slots = az_options().wgrad_slots;  // resident waves the launch may take (AZ_WGRAD_SLOTS: room left for the other stream)
    const int wmax = max(8, (slots / NCOMBO) & ~7);
    const long long base_items = (long long)a.B * a.Dc * a.nwchunk;
    double best = -1.0;
    int best_w = 8, best_rows = a.Hc;
    for (int nseg = 1; nseg <= min(a.Hc, 24); ++nseg) {
        const int rows = (a.Hc + nseg - 1) / nseg;
        const int segs = (a.Hc + rows - 1) / rows;
        const long long items = base_items * segs;
        for (int w = wmax; w >= max(8, wmax / 2); w -= 8) {
            const long long per = (items + w - 1) / w;
            if (per < 4 && nseg < min(a.Hc, 24)) continue;
            const double balance = (double)items / (double)(per * w);
            const double occ = (double)w / (double)wmax;
            const double score = balance * (0.5 + 0.5 * occ) - 0.002 * segs;
            if (score > best) { best = score; best_w = w; best_rows = rows; }
        }
    }
    a.waves_per_combo = best_w;
    a.hseg_rows = best_rows;
    a.nhseg = (a.Hc + a.hseg_rows - 1) / a.hseg_rows;
    a.nitems = base_items * a.nhseg;
    if (a.nitems < a.waves_per_combo) a.waves_per_combo = (int)((a.nitems + 7) & ~7LL);
    a.order = az_options().wgrad_order;
    const int fine_walk = az_options().wgrad_fw;  // AZ_WGRAD_FW=0: the coarse-row walk for the stride-1 layers too (A/B)
    if (PREC == 3)
        hipLaunchKernelGGL((conv3d_wgrad_x6_kernel<CM, CN, S, 1>), dim3(a.waves_per_combo * NCOMBO), dim3(64), 0, s, a);
    else if (PREC == 1 && S == 1 && fine_walk)
        hipLaunchKernelGGL((conv3d_wgrad_x6_fw_kernel<CM, CN>), dim3(a.waves_per_combo * NCOMBO), dim3(64), 0, s, a);
    else if (PREC == 1)
        hipLaunchKernelGGL((conv3d_wgrad_x6_kernel<CM, CN, S>), dim3(a.waves_per_combo * NCOMBO),
                           dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL((conv3d_wgrad_kernel<CM, CN, S>), dim3(a.waves_per_combo * NCOMBO),
                           dim3(64), 0, s, a);
    return az_launch_status();
}

// all 27 taps per wave on 16x16x32 tiles, four waves sharing one staged set (az_conv3d_wgrad16.hip)
int az_conv3d_wgrad_r16_launch(float *ws, const float *coarse, const float *fine, int B, int cm, int cn, int D, int H, int W, hipStream_t s,
                               const float *coarse_amax = nullptr, const float *fine_amax = nullptr, int split_mask = 0);

// stride 2, f16x3: eight waves sharing one staged set (az_conv3d_wgrad16s2.hip)
int az_conv3d_wgrad_s2r16_launch(float *ws, const float *coarse, const float *fine, int B, int cm, int cn, int Dc, int Hc, int Wc,
                                 int Df, int Hf, int Wf, hipStream_t s, const float *coarse_amax, const float *fine_amax, int split_mask);

extern "C" long long az_conv3d_wgrad_workspace(int cm, int cn) {
    if (cm <= 0 || cn <= 0 || cm % 32 || cn % 32) return AZ_EINVAL;
    return 27LL * cm * cn * (long long)sizeof(float);
}

extern "C" int az_conv3d_wgrad(float *grad_w, float *workspace, long long workspace_bytes,
                               const float *coarse, const float *fine, int stride, int precision,
                               int B, int cm, int cn, int Dc, int Hc, int Wc, int Df, int Hf,
                               int Wf, void *stream) {
    AZ_REQUIRE_PTR(grad_w); AZ_REQUIRE_PTR(workspace); AZ_REQUIRE_PTR(coarse); AZ_REQUIRE_PTR(fine);
    AZ_REQUIRE(B > 0 && Dc > 0 && Hc > 0 && Wc > 0 && Df > 0 && Hf > 0 && Wf > 0);
    AZ_REQUIRE(stride == 1 || stride == 2);
    AZ_REQUIRE(precision == 0 || precision == 1);
    const long long need = az_conv3d_wgrad_workspace(cm, cn);
    if (need < 0) return AZ_EUNSUPPORTED;
    if (workspace_bytes < need) return AZ_EWORKSPACE;
    hipStream_t s = az_stream(stream);
    if (hipMemsetAsync(workspace, 0, (size_t)need, s) != hipSuccess) return AZ_ELAUNCH;
    WgArgs a{};
    a.coarse = coarse; a.fine = fine; a.ws = workspace;
    a.B = B; a.Dc = Dc; a.Hc = Hc; a.Wc = Wc; a.Df = Df; a.Hf = Hf; a.Wf = Wf;
    int rc = AZ_EUNSUPPORTED;
    {
        const int r16 = az_options().wgrad_r16;  // 0: the one-kd-per-wave kernels for the V0 layers too, 1: only the V0 layers on the new kernel (A/B)
        // (r16 == 1: the V0 32 x 32 layers only; 2: the 64-channel stride-1 layers too, as 32 x 32 tiles in the grid)
        if (r16 && precision == 1 && stride == 1 && (cm == 32 || cm == 64) && (cn == 32 || cn == 64) && (r16 >= 2 || (cm == 32 && cn == 32)) &&
            Dc == Df && Hc == Hf && Wc == Wf) {
            rc = az_conv3d_wgrad_r16_launch(workspace, coarse, fine, B, cm, cn, Dc, Hc, Wc, s);
            if (rc == AZ_OK) {
                const int total = cm * cn * 27;
                hipLaunchKernelGGL(wgrad_unpack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, grad_w, workspace, cm, cn);
                return az_launch_status();
            }
            // AZ_EUNSUPPORTED: a batch element beyond the kernel's 32-bit buffer offsets (>= 4 GiB) -- the flat-address
            // kernels below take it
            if (rc != AZ_EUNSUPPORTED) return rc;
        }
    }
#define WG_CASE(M, N)                                                                        \
    if (cm == M && cn == N)                                                                  \
        rc = precision == 0                                                                  \
                 ? ((stride == 1) ? launch_wgrad<M, N, 1, 0>(a, s) : launch_wgrad<M, N, 2, 0>(a, s)) \
                 : ((stride == 1) ? launch_wgrad<M, N, 1, 1>(a, s) : launch_wgrad<M, N, 2, 1>(a, s));
    WG_CASE(32, 32) WG_CASE(32, 64) WG_CASE(64, 32) WG_CASE(64, 64)
#undef WG_CASE
    if (rc != AZ_OK) return rc;
    const int total = cm * cn * 27;
    hipLaunchKernelGGL(wgrad_unpack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, grad_w,
                       workspace, cm, cn);
    return az_launch_status();
}

// f16x3 weight gradient (include/azhip.h): the stride-1 layers with 32 / 64 channels on either side
// which operands of an f16x3 weight-gradient launch of this shape may be pre-split tensors: the kernels that stage by copy
// (az_conv3d_wgrad16.hip: both; az_conv3d_wgrad16s2.hip: one of the two), under the conditions their launchers check
extern "C" int az_conv3d_wgrad_f16_split_ok(int stride, int B, int cm, int cn, int Dc, int Hc, int Wc, int Df, int Hf, int Wf) {
    if (B <= 0 || !((cm == 32 || cm == 64) && (cn == 32 || cn == 64))) return 0;
    if (stride == 1 && Dc == Df && Hc == Hf && Wc == Wf)
        return az_fits_buffer_offset((long long)Dc * Hc * Wc * (cm > cn ? cm : cn) * 4) ? 3 : 0;
    if (stride == 2 && az_options().wgrad_s2r16 && cm == 64 && az_fits_buffer_offset((long long)Df * Hf * Wf * cn * 4) &&
        az_fits_buffer_offset((long long)Dc * Hc * Wc * cm * 4))
        return 3;  // (either operand, not both at once: az_conv3d_wgrad_f16 returns AZ_EUNSUPPORTED for mask 3)
    return 0;
}

extern "C" int az_conv3d_wgrad_f16(float *grad_w, float *workspace, long long workspace_bytes, const float *coarse,
                                   const float *fine, const float *coarse_amax, const float *fine_amax, int split_mask,
                                   int stride, int B, int cm, int cn, int Dc, int Hc, int Wc, int Df, int Hf, int Wf,
                                   void *stream) {
    // grad_w == NULL: accumulate-only -- the workspace is ALREADY ZERO (one arena memset per backward pass instead of one per
    // layer) and keeps the tap-major result [27][cm][cn]; az_wgrad_unpack_multi turns a whole pass's workspaces into
    // PyTorch-layout gradients in one launch
    AZ_REQUIRE_PTR(workspace); AZ_REQUIRE_PTR(coarse); AZ_REQUIRE_PTR(fine);
    AZ_REQUIRE_PTR(coarse_amax); AZ_REQUIRE_PTR(fine_amax);
#ifdef WGRAD_SKIP  // timing-only build (tools/abl_step_sensitivity.sh): accumulate-only launches are left out, the zeroed workspace IS the gradient
    if (!grad_w) return AZ_OK;
#endif
    AZ_REQUIRE(B > 0 && Dc > 0 && Hc > 0 && Wc > 0 && Df > 0 && Hf > 0 && Wf > 0);
    AZ_REQUIRE(stride == 1 || stride == 2);
    AZ_REQUIRE(split_mask >= 0 && split_mask <= 3);
    const long long need = az_conv3d_wgrad_workspace(cm, cn);
    if (need < 0) return AZ_EUNSUPPORTED;
    if (workspace_bytes < need) return AZ_EWORKSPACE;
    if (!((cm == 32 || cm == 64) && (cn == 32 || cn == 64))) return AZ_EUNSUPPORTED;
    if (split_mask & ~az_conv3d_wgrad_f16_split_ok(stride, B, cm, cn, Dc, Hc, Wc, Df, Hf, Wf)) return AZ_EUNSUPPORTED;
    hipStream_t s = az_stream(stream);
    if (grad_w && hipMemsetAsync(workspace, 0, (size_t)need, s) != hipSuccess) return AZ_ELAUNCH;
    int rc = AZ_EUNSUPPORTED;
    // (the flush stays tap-major + one unpack launch: with the atomics going straight into PyTorch's [m][n][27] layout -- 27-float
    //  stride between lanes -- the V0 kernel took 1.52 instead of 1.06 ms and the stride-2 one 1.03 instead of 0.35: one cache
    //  line per lane and atomic instead of four 64-byte segments per instruction)
    if (stride == 1 && Dc == Df && Hc == Hf && Wc == Wf)  // all 27 taps per wave on 16x16x32 tiles (az_conv3d_wgrad16.hip)
        rc = az_conv3d_wgrad_r16_launch(workspace, coarse, fine, B, cm, cn, Dc, Hc, Wc, s, coarse_amax, fine_amax, split_mask);
    else if (stride == 2 && az_options().wgrad_s2r16)
        rc = az_conv3d_wgrad_s2r16_launch(workspace, coarse, fine, B, cm, cn, Dc, Hc, Wc, Df, Hf, Wf, s, coarse_amax, fine_amax, split_mask);
    if (rc == AZ_EUNSUPPORTED && split_mask) return rc;  // (the one-kd-per-wave kernels read fp32 tensors only: ask az_conv3d_wgrad_f16_split_ok first)
    if (rc == AZ_EUNSUPPORTED) {  // shapes / sizes those kernels do not take (e.g. a batch element beyond 32-bit offsets): one kd per wave
        WgArgs a{};
        a.coarse = coarse; a.fine = fine; a.ws = workspace; a.coarse_amax = coarse_amax; a.fine_amax = fine_amax;
        a.B = B; a.Dc = Dc; a.Hc = Hc; a.Wc = Wc; a.Df = Df; a.Hf = Hf; a.Wf = Wf;
#define WG16_CASE(M, N) if (cm == M && cn == N) rc = (stride == 1) ? launch_wgrad<M, N, 1, 3>(a, s) : launch_wgrad<M, N, 2, 3>(a, s);
        WG16_CASE(32, 32) WG16_CASE(32, 64) WG16_CASE(64, 32) WG16_CASE(64, 64)
#undef WG16_CASE
    }
    if (rc != AZ_OK) return rc;
    if (!grad_w) return az_launch_status();
    const int total = cm * cn * 27;
    hipLaunchKernelGGL(wgrad_unpack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, grad_w, workspace, cm, cn);
    return az_launch_status();
}
