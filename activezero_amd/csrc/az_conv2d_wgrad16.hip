// K13w' -- weight gradient of the 3x3, stride-1, dilation-1 Conv2d layers with 32 / 64 channels on either side
// (reference nets/psmnet/psmnet_submodule_3.py:92-147: firstconv[1..2], layer1, layer2; autograd of F.conv2d), the
// 2-D form of az_conv3d_wgrad16.hip:
//
//   G[co][ci][kh][kw] = sum over (b, y, x) of dy[b,y,x,co] * x[b, y-1+kh, x-1+kw, ci]
//
//   * a workgroup = 4 waves owns ONE 32 x 32 (co, ci) tile -- wave (mi, ni) its 16 x 16 block for all 9 taps
//     (36 accumulator registers) -- and shares every staged byte; 64-channel layers are 2 x 2 such tiles in the grid;
//   * K = 32 positions per v_mfma_f32_16x16x32_bf16 = two adjacent dy rows x 16 positions; a step is 9 taps x 6 MFMAs
//     per wave and needs the dy pair (double-buffered) and the 4-row window of x rows (a ring of six), of which only
//     the pair and two x rows are new: 68 positions staged per 54 MFMAs x 4 waves (az_conv2d_wgrad.hip: 34 per 54
//     MFMAs of ONE wave for the 32 -> 32 layers, between two barriers);
//   * the next step's rows are split and written while this step is multiplied (one barrier per step); loads and
//     validity (zero padding in y / x) through buffer instructions with out-of-range offsets: a step is one basic block;
//   * workgroups walk (image, 16-position chunk, row segment) columns and flush their 9 x 32 x 32 block once, with
//     float atomics, into the tap-major workspace of az_conv2d_wgrad.hip ([tap][CM][CN]).
// Arithmetic per K block: the six-MFMA chain of the other weight-gradient kernels (smallest terms first, fp32
// accumulate).  LDS 33 KB, so three to four workgroups share a CU (the 3-D kernel: 77 KB, two).
#include <stdlib.h>

#include "az_roll_common.h"
#include "az_options.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

#define V16_POS 16                       // positions of a dy row per step
#define V16_FW (V16_POS + 2)             // x positions per staged row
#define V16_ROWB 64                      // bytes of one (position, 32 channels) bf16 row
#define V16_CBUF (3 * 32 * V16_ROWB)     // one dy buffer: [part][k = 2 rows x 16][32 ch]      6 144 B
#define V16_FROW (3 * V16_FW * V16_ROWB) // one x row: [part][18 positions][32 ch]              3 456 B
#define V16_RING 6
#define V16_LDS (2 * V16_CBUF + V16_RING * V16_FROW)  // 33 024 B
#define V16_NQ ((2 * V16_POS + 2 * V16_FW) * 8)       // float4 pieces per step: 544
#define V16_NLD ((V16_NQ + 255) / 256)                // 3 per thread
#define V16_OOB 0xffffff00u

struct Wg2dArgs {
    const float *coarse, *fine;  // dy [B,H,W,cs_c], x [B,H,W,cs_f]
    float *ws;                   // [9][CM][CN]
    int B, H, W, CM, CN, cs_c, cs_f;
    int nwchunk, nrseg, seg_rows;  // 16-position chunks per row, row segments, rows per segment (even)
    long long ncols;               // B * nwchunk * nrseg columns of work per (co, ci) tile
    int wgs;                       // workgroups per tile
    const float *coarse_amax, *fine_amax;  // AR 1 (f16x3): amax arrays of dy and x
};

// AR: 0 = bf16x6, 1 = f16x3 (az_roll_common.h; the LDS images keep their three-part strides)
template <int AR>
__global__ void __launch_bounds__(256, 3)
conv2d_wgrad_r16_kernel(const Wg2dArgs a) {
    constexpr int NP = AR ? 2 : 3;
    float c_scale = 1.f, f_scale = 1.f, o_scale = 1.f;
    if (AR) {
        const int kc = az_f16_scale_exp(az_amax_read(a.coarse_amax)), kf = az_f16_scale_exp(az_amax_read(a.fine_amax));
        c_scale = az_pow2(kc); f_scale = az_pow2(kf);
        o_scale = ldexpf(1.f, -(kc + kf));
    }
    __shared__ __attribute__((aligned(16))) unsigned char lds[V16_LDS + 64];  // + a sink for the lanes of a partial piece
    unsigned char *const cbuf = lds;                  // [2][V16_CBUF]
    unsigned char *const fring = lds + 2 * V16_CBUF;  // [slot][V16_FROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wv >> 1, ni = wv & 1;
    // (co, ci) tile of this workgroup; the workgroups of a tile are blockIdx.x / ntiles
    const int ntn = a.CN >> 5, ntiles = (a.CM >> 5) * ntn;
    const int tile = blockIdx.x % ntiles, wg = blockIdx.x / ntiles;
    const int co0 = (tile / ntn) * 32, ci0 = (tile % ntn) * 32;

    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposing-read geometry and the row-half swap against the 2-way conflicts: az_conv3d_wgrad16.hip
    const int oct = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const unsigned a_lane = (unsigned)(8 * oct + tq) * V16_ROWB + (((unsigned)(16 * mi + 4 * tp) * 2) ^ ((unsigned)(oct & 1) << 5));
    const int rp = oct >> 1;
    unsigned b_off[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const unsigned rowi = (unsigned)(8 * (oct & 1) + tq + kw + 4 * h2);
            b_off[kw][h2] = rowi * V16_ROWB + (((unsigned)(16 * ni + 4 * tp) * 2) ^ (((rowi >> 3) & 1u) << 5));
        }
    auto frag2 = [&](const unsigned char *lo, const unsigned char *hi) -> az_bf16x8 {
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lo));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(hi));
        s16x8 v;
        v[0] = lo4[0]; v[1] = lo4[1]; v[2] = lo4[2]; v[3] = lo4[3];
        v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
        return __builtin_bit_cast(az_bf16x8, v);
    };

    const unsigned img_c = (unsigned)a.H * a.W * a.cs_c * 4u, img_f = (unsigned)a.H * a.W * a.cs_f * 4u;
    for (long long col = wg; col < a.ncols; col += a.wgs) {
        long long r_ = col;
        const int rs = (int)(r_ % a.nrseg); r_ /= a.nrseg;
        const int wc = (int)(r_ % a.nwchunk);
        const int b = (int)(r_ / a.nwchunk);
        const int cw0 = wc * V16_POS;
        const int h0 = rs * a.seg_rows, h1 = min(h0 + a.seg_rows, a.H);  // dy rows [h0, h1), h0 even
        const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.coarse) + (size_t)b * (img_c / 4) + co0, 0, img_c, 0x00020000);
        const auto rs_f = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.fine) + (size_t)b * (img_f / 4) + ci0, 0, img_f, 0x00020000);

        // piece q = tid + 256 it of a step's set: q < 256: dy (row pair, 16 positions, 8 float4 per position);
        // q >= 256: x, new row j = (q - 256) / 144, position .. / 8 (18 x 8 = 144 per row)
        u32x4 pre[V16_NLD];
        auto issue = [&](int crow0, bool with_coarse, int frow0) {  // dy rows crow0, crow0+1; x rows frow0, frow0+1
#pragma unroll
            for (int it = 0; it < V16_NLD; ++it) {
                const int q = tid + 256 * it;
                unsigned off = V16_OOB;
                if (it == 0) {
                    const int k = q >> 3, row = crow0 + (k >> 4), cw = cw0 + (k & 15);
                    if (with_coarse && row < h1 && cw < a.W)
                        off = (unsigned)(row * a.W + cw) * (unsigned)(a.cs_c * 4) + (unsigned)(q & 7) * 16u;
                    pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, off, 0, 0);
                } else {
                    const int f = q - 256;
                    const int j = f / 144, pp = (f - j * 144) >> 3;
                    const int fr = frow0 + j, fw = cw0 - 1 + pp;
                    if (q < V16_NQ && (unsigned)fr < (unsigned)a.H && (unsigned)fw < (unsigned)a.W)
                        off = (unsigned)(fr * a.W + fw) * (unsigned)(a.cs_f * 4) + (unsigned)(q & 7) * 16u;
                    pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_f, off, 0, 0);
                }
            }
        };
        auto commit_piece = [&](int it, int cbuf_idx, int frow0) {
            const int q = tid + 256 * it;
            uint2 hi, mid, lo;
            if (AR) {
                float4 v = __builtin_bit_cast(float4, pre[it]);
                const float sc_ = it == 0 ? c_scale : f_scale;
                v.x *= sc_; v.y *= sc_; v.z *= sc_; v.w *= sc_;
                az_split2_f16x4(v, hi, mid);
                lo = mid;
            } else {
                az_split3_bf16x4(__builtin_bit_cast(float4, pre[it]), hi, mid, lo);
            }
            unsigned char *dst;
            unsigned part_stride;
            if (it == 0) {
                dst = cbuf + cbuf_idx * V16_CBUF + (q >> 3) * V16_ROWB + (((q & 7) * 8) ^ ((((q >> 3) >> 3) & 1) << 5));
                part_stride = 32 * V16_ROWB;
            } else {
                const int f = q - 256;
                const int j = f / 144, pp = (f - j * 144) >> 3;
                const int slot = (frow0 + j + 2 * V16_RING) % V16_RING;
                dst = fring + slot * V16_FROW + pp * V16_ROWB + (((q & 7) * 8) ^ (((pp >> 3) & 1) << 5));
                part_stride = V16_FW * V16_ROWB;
            }
            if (q >= V16_NQ) { dst = lds + V16_LDS + (tid & 7) * 8; part_stride = 0; }  // (no branch: a step stays one block)
            *reinterpret_cast<uint2 *>(dst) = hi;
            *reinterpret_cast<uint2 *>(dst + part_stride) = mid;
            if (!AR) *reinterpret_cast<uint2 *>(dst + 2 * part_stride) = lo;
        };

        // ---- prologue: the window of the first step (x rows h0-1 .. h0+2, dy rows h0, h0+1), then the request for the next ----
        __syncthreads();  // the previous column's last step no longer reads
        issue(h0, false, h0 - 1);
#pragma unroll
        for (int it = 0; it < V16_NLD; ++it) commit_piece(it, 0, h0 - 1);
        issue(h0, true, h0 + 1);
#pragma unroll
        for (int it = 0; it < V16_NLD; ++it) commit_piece(it, 0, h0 + 1);
        issue(h0 + 2, true, h0 + 3);
        __syncthreads();

        const int nsteps = (h1 - h0 + 1) / 2;
        for (int s = 0; s < nsteps; ++s) {
            const int ch = h0 + 2 * s;
            const unsigned char *ca = cbuf + (s & 1) * V16_CBUF + a_lane;
            int slot[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) slot[r] = (ch - 1 + r + 2 * V16_RING) % V16_RING;
            unsigned fb[3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) fb[kh] = (unsigned)(rp ? slot[kh + 1] : slot[kh]) * V16_FROW;

            az_bf16x8 af[3];
#pragma unroll
            for (int p = 0; p < NP; ++p) af[p] = frag2(ca + p * 32 * V16_ROWB, ca + p * 32 * V16_ROWB + 4 * V16_ROWB);
            az_bf16x8 bf[2][3];
            auto load_b = [&](az_bf16x8 (&bq)[3], int t) {
                const int kh = t / 3, kw = t % 3;
                const unsigned char *fp = fring + fb[kh];
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    bq[p] = frag2(fp + b_off[kw][0] + p * V16_FW * V16_ROWB, fp + b_off[kw][1] + p * V16_FW * V16_ROWB);
            };
            load_b(bf[0], 0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < 9) load_b(bf[(t + 1) & 1], t + 1);
                __builtin_amdgcn_sched_barrier(0);
                f32x4 c = acc[t];
                const az_bf16x8(&bq)[3] = bf[t & 1];
                if constexpr (AR) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(az_f16x8, af[1]), __builtin_bit_cast(az_f16x8, bq[0]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(az_f16x8, af[0]), __builtin_bit_cast(az_f16x8, bq[1]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(az_f16x8, af[0]), __builtin_bit_cast(az_f16x8, bq[0]), c, 0, 0, 0);
                } else {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bq[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bq[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bq[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bq[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bq[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bq[0], c, 0, 0, 0);
                }
                acc[t] = c;
                // the set of the next step (requested a step ago): one piece after each of the taps 0, 2, 4; then the
                // request for the step after that
                if (t <= 4 && !(t & 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    commit_piece(t / 2, (s + 1) & 1, ch + 3);
                }
                if (t == 5) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue(ch + 4, true, ch + 5);
                }
            }
            __syncthreads();  // next step's rows written by all four waves; this step's no longer read
        }
    }
    // D[i][j]: i = output channel co0 + 16 mi + 4 (lane >> 4) + r, j = input channel ci0 + 16 ni + (lane & 15)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = co0 + 16 * mi + 4 * (lane >> 4) + r;
            atomicAdd(&a.ws[((size_t)t * a.CM + m) * a.CN + ci0 + 16 * ni + (lane & 15)], AR ? acc[t][r] * o_scale : acc[t][r]);
        }
}

// ---- round 5: the 64 -> 64 layers (layer2: 32 of the 39 launches per step) as ONE 64 x 64 tile per workgroup -------------------
// The kernel above gives a 64 x 64 layer four workgroups (2 x 2 tiles of 32 x 32) that each stage their own slices of dy and x --
// every staged byte is staged twice -- and a wave one 16 x 16 block of all nine taps: 40 transposing LDS reads per 27 MFMAs,
// 1 920 LDS cycles per CU and step for 1 296 matrix cycles per SIMD (the arithmetic of az_conv3d_wgrad16.hip's round-5 note), at a
// matrix pipe a third busy and 2.3 GHz -- a kernel where cycles saved ARE time saved, unlike the V0 one.
// Here a workgroup of EIGHT waves owns the whole 64 x 64 tile: dy and x are staged once (two 32-channel images each, the LDS
// layout and row-half swap of the kernel above per image), and a wave owns one 16-channel block of x (nb = wave & 3) against ALL
// FOUR 16-channel blocks of dy for five (waves 0-3: taps 0, 2, 4, 6, 8) or four (waves 4-7: taps 1, 3, 5, 7; a fifth, duplicate
// pair is computed and not flushed) taps: a fine fragment read feeds 4 x 3 MFMAs, the four coarse fragments of a step are read
// once for all of the wave's taps -- 16 + 5 x 4 = 36 reads per 60 MFMAs (0.6 per MFMA against 1.48).  80 accumulator registers.
// f16x3 only (the default arithmetic of the training step); everything else stays on the kernel above.
// MEASURED (tools/wgrad2d_probe.py, B = 8, 136 x 240, alone, incl. the 5 us memset and the 5 us unpack): 103.2 us against 102.6 us
// for the kernel above -- no gain, and the timing-only builds (X16_ABL) say why: without the atomic flush 67.5 us, without the
// global loads 97.3, without both 54.0.  A 19 GFLOP launch split over 256 workgroups ends in 256 x 9 x 64 x 64 = 9.4 M float
// atomics, which the memory side retires in ~35 us whatever produced them (the kernel above: 768 workgroups x 9 x 32 x 32 = 7.1 M):
// BOTH kernels are flush-bound at this size, not LDS- or MFMA-bound.  In the step these launches sit on the side stream with
// ~25 ms of slack, so the flush costs chip contention, not wall time; kept as the default for its lighter LDS / VALU footprint
// beside the main stream's kernels (same-box step A/B 86.28 vs 86.43 ms), AZ_CONV2D_WGRAD_W64=0 restores the 2 x 2 tiles.
#define X16_CIMG (2 * 32 * V16_ROWB)            // one 32-channel dy image: [part][k = 32][32 ch]            4 096 B
#define X16_CBUF (2 * X16_CIMG)                 // both channel halves                                        8 192 B
#define X16_FIMG (2 * V16_FW * V16_ROWB)        // one 32-channel x row image: [part][18 positions][32 ch]   2 304 B
#define X16_FROW (2 * X16_FIMG)                 //                                                            4 608 B
#define X16_LDS (2 * X16_CBUF + V16_RING * X16_FROW)  // 44 032 B
#define X16_NQ ((2 * V16_POS + 2 * V16_FW) * 16)      // float4 pieces per step: 1 088
#define X16_NLD 3                                      // per thread (512 threads)
#ifndef X16_ABL
#define X16_ABL 0  // timing-only builds: 1 no atomic flush, 2 no global loads, 4 one column per workgroup only (no second prologue)
#endif

__global__ void __launch_bounds__(512, 1)
conv2d_wgrad_w64_kernel(const Wg2dArgs a) {
    const int kc = az_f16_scale_exp(az_amax_read(a.coarse_amax)), kf = az_f16_scale_exp(az_amax_read(a.fine_amax));
    const float c_scale = az_pow2(kc), f_scale = az_pow2(kf), o_scale = ldexpf(1.f, -(kc + kf));
    __shared__ __attribute__((aligned(16))) unsigned char lds[X16_LDS + 64];  // + a sink for the lanes of the partial piece
    unsigned char *const cbuf = lds;                  // [2][X16_CBUF]
    unsigned char *const fring = lds + 2 * X16_CBUF;  // [slot][X16_FROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = wv & 3, t0 = wv >> 2;  // this wave's x-channel block and first tap (taps t0, t0 + 2, ..)

    f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[i][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposing-read geometry of the kernel above inside one 32-channel image
    const int oct = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int rp = oct >> 1;
    // dy blocks mb = 2 half + h16: image `half`, channels 16 h16 ..: the two h16 bases differ in bit 5 of the in-row offset
    unsigned a_lane[2];
#pragma unroll
    for (int h16 = 0; h16 < 2; ++h16)
        a_lane[h16] = (unsigned)(8 * oct + tq) * V16_ROWB + (((unsigned)(16 * h16 + 4 * tp) * 2) ^ ((unsigned)(oct & 1) << 5));
    // x block nb, tap i of this wave (tap = t0 + 2 i, clamped to 8): per-lane byte offset inside a ring slot for the two row
    // groups of a fragment, and kh in two bits of a scalar -- a step stays straight-line code although wv is a run-time value
    unsigned boff[5][2], khpack = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int t = min(t0 + 2 * i, 8);
        const int kh = t / 3, kw = t % 3;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const unsigned rowi = (unsigned)(8 * (oct & 1) + tq + kw + 4 * h2);
            boff[i][h2] = (unsigned)(nb >> 1) * X16_FIMG + rowi * V16_ROWB + (((unsigned)(16 * (nb & 1) + 4 * tp) * 2) ^ (((rowi >> 3) & 1u) << 5));
        }
        khpack |= (unsigned)kh << (2 * i);
    }
    auto frag2 = [&](const unsigned char *lo, const unsigned char *hi) -> az_f16x8 {
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lo));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(hi));
        s16x8 v;
        v[0] = lo4[0]; v[1] = lo4[1]; v[2] = lo4[2]; v[3] = lo4[3];
        v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
        return __builtin_bit_cast(az_f16x8, v);
    };

    const unsigned img_c = (unsigned)a.H * a.W * a.cs_c * 4u, img_f = (unsigned)a.H * a.W * a.cs_f * 4u;
    for (long long col = blockIdx.x; col < a.ncols; col += a.wgs) {
        long long r_ = col;
        const int rs = (int)(r_ % a.nrseg); r_ /= a.nrseg;
        const int wc = (int)(r_ % a.nwchunk);
        const int b = (int)(r_ / a.nwchunk);
        const int cw0 = wc * V16_POS;
        const int h0 = rs * a.seg_rows, h1 = min(h0 + a.seg_rows, a.H);  // dy rows [h0, h1), h0 even
        const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.coarse) + (size_t)b * (img_c / 4), 0, img_c, 0x00020000);
        const auto rs_f = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.fine) + (size_t)b * (img_f / 4), 0, img_f, 0x00020000);

        // piece q = tid + 512 it of a step's set: it = 0: dy (row pair, 16 positions, 16 float4 per position); it = 1, 2:
        // x piece f = q - 512: new row f / 288, position (f % 288) / 16 (18 x 16 = 288 per row), float4 f & 15
        u32x4 pre[X16_NLD];
        auto issue = [&](int crow0, bool with_coarse, int frow0) {
#pragma unroll
            for (int it = 0; it < X16_NLD; ++it) {
                const int q = tid + 512 * it;
                unsigned off = V16_OOB;
                if (it == 0) {
                    const int k = q >> 4, row = crow0 + (k >> 4), cw = cw0 + (k & 15);
                    if (with_coarse && row < h1 && cw < a.W)
                        off = (unsigned)(row * a.W + cw) * (unsigned)(a.cs_c * 4) + (unsigned)(q & 15) * 16u;
                    if (X16_ABL & 2) off = V16_OOB;
                    pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, off, 0, 0);
                } else {
                    const int f = q - 512;
                    const int j = f / 288, pp = (f - j * 288) >> 4;
                    const int fr = frow0 + j, fw = cw0 - 1 + pp;
                    if (q < X16_NQ && (unsigned)fr < (unsigned)a.H && (unsigned)fw < (unsigned)a.W)
                        off = (unsigned)(fr * a.W + fw) * (unsigned)(a.cs_f * 4) + (unsigned)(q & 15) * 16u;
                    if (X16_ABL & 2) off = V16_OOB;
                    pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_f, off, 0, 0);
                }
            }
        };
        auto commit_piece = [&](int it, int cbuf_idx, int frow0) {
            const int q = tid + 512 * it;
            uint2 hi, lo;
            az_stage_f16x4<false>(pre[it], it == 0 ? c_scale : f_scale, hi, lo);
            const int f4 = q & 15, half = f4 >> 3, j8 = f4 & 7;  // 32-channel image, 8-byte piece inside its row
            unsigned char *dst;
            unsigned part_stride;
            if (it == 0) {
                const int k = q >> 4;
                dst = cbuf + cbuf_idx * X16_CBUF + half * X16_CIMG + k * V16_ROWB + ((j8 * 8) ^ (((k >> 3) & 1) << 5));
                part_stride = 32 * V16_ROWB;
            } else {
                const int f = q - 512;
                const int j = f / 288, pp = (f - j * 288) >> 4;
                const int slot = (frow0 + j + 2 * V16_RING) % V16_RING;
                dst = fring + slot * X16_FROW + half * X16_FIMG + pp * V16_ROWB + ((j8 * 8) ^ (((pp >> 3) & 1) << 5));
                part_stride = V16_FW * V16_ROWB;
            }
            if (q >= X16_NQ) { dst = lds + X16_LDS + (tid & 7) * 8; part_stride = 0; }  // (no branch: a step stays one block)
            *reinterpret_cast<uint2 *>(dst) = hi;
            *reinterpret_cast<uint2 *>(dst + part_stride) = lo;
        };

        // ---- prologue: the window of the first step (x rows h0-1 .. h0+2, dy rows h0, h0+1), then the request for the next ----
        __syncthreads();  // the previous column's last step no longer reads
        issue(h0, false, h0 - 1);
#pragma unroll
        for (int it = 0; it < X16_NLD; ++it) commit_piece(it, 0, h0 - 1);
        issue(h0, true, h0 + 1);
#pragma unroll
        for (int it = 0; it < X16_NLD; ++it) commit_piece(it, 0, h0 + 1);
        issue(h0 + 2, true, h0 + 3);
        __syncthreads();

        const int nsteps = (h1 - h0 + 1) / 2;
        for (int s = 0; s < nsteps; ++s) {
            const int ch = h0 + 2 * s;
            const unsigned char *cb = cbuf + (s & 1) * X16_CBUF;
            const int slot0 = (ch - 1 + 2 * V16_RING) % V16_RING;  // ring slot of x row ch - 1 (wave-uniform)
            // the four dy fragments (both parts) of this step, for all of the wave's taps
            az_f16x8 af[4][2];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const unsigned char *cp = cb + (mb >> 1) * X16_CIMG + p * 32 * V16_ROWB + a_lane[mb & 1];
                    af[mb][p] = frag2(cp, cp + 4 * V16_ROWB);
                }
            az_f16x8 bf[2][2];
            auto load_b = [&](az_f16x8 (&bq)[2], int i) {
                // x row ch - 1 + kh + (row of the dy pair this lane's k belongs to)
                int r = slot0 + (int)((khpack >> (2 * i)) & 3u) + rp;
                r = r >= V16_RING ? r - V16_RING : r;
                const unsigned char *fp = fring + (unsigned)r * X16_FROW;
#pragma unroll
                for (int p = 0; p < 2; ++p) bq[p] = frag2(fp + boff[i][0] + p * V16_FW * V16_ROWB, fp + boff[i][1] + p * V16_FW * V16_ROWB);
            };
            load_b(bf[0], 0);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                __builtin_amdgcn_sched_barrier(0);
                if (i + 1 < 5) load_b(bf[(i + 1) & 1], i + 1);
                __builtin_amdgcn_sched_barrier(0);
                const az_f16x8(&bq)[2] = bf[i & 1];
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    f32x4 c = acc[i][mb];  // lo*hi, hi*lo, hi*hi chained into the running accumulator (smallest first)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mb][1], bq[0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mb][0], bq[1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mb][0], bq[0], c, 0, 0, 0);
                    acc[i][mb] = c;
                }
                // the set of the next step (requested a step ago): one piece after each of the pairs 0, 1, 2; then the request
                // for the step after that
                if (i <= 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    commit_piece(i, (s + 1) & 1, ch + 3);
                }
                if (i == 3) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue(ch + 4, true, ch + 5);
                }
            }
            __syncthreads();  // next step's rows written by all eight waves; this step's no longer read
        }
    }
    // D[i][j]: i = output channel 16 mb + 4 (lane >> 4) + r, j = input channel 16 nb + (lane & 15); waves 4-7: four taps
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int t = t0 + 2 * i;
        if (t > 8) break;  // (wave-uniform)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * mb + 4 * (lane >> 4) + r;
                if (!(X16_ABL & 1) || acc[i][mb][r] == 123.456f)
                    atomicAdd(&a.ws[((size_t)t * 64 + m) * 64 + 16 * nb + (lane & 15)], acc[i][mb][r] * o_scale);
            }
    }
}

// 3x3, dilation 1, CM and CN in {32, 64}; workspace [9][CM][CN] already zeroed by the caller (az_conv2d_wgrad)
int az_conv2d_wgrad_r16_launch(float *ws, const float *coarse, const float *fine, int B, int H, int W, int cm, int cn,
                               int cs_c, int cs_f, hipStream_t s, const float *coarse_amax, const float *fine_amax) {
    if (!((cm == 32 || cm == 64) && (cn == 32 || cn == 64))) return AZ_EUNSUPPORTED;
    if ((long long)H * W * (cs_c > cs_f ? cs_c : cs_f) * 4 >= 0xffffff00LL) return AZ_EUNSUPPORTED;
    Wg2dArgs a{};
    a.coarse = coarse; a.fine = fine; a.ws = ws; a.coarse_amax = coarse_amax; a.fine_amax = fine_amax;
    a.B = B; a.H = H; a.W = W; a.CM = cm; a.CN = cn; a.cs_c = cs_c; a.cs_f = cs_f;
    a.nwchunk = (W + V16_POS - 1) / V16_POS;
    const int ntiles = (cm / 32) * (cn / 32);
    const int slots = 256 * 3 / ntiles;  // resident workgroups per tile at three per CU
    // row segments (even row counts): enough columns to fill the slots, long enough to amortise the window prologue
    int best_seg = 1, best_w = 1;
    double best = -1.0;
    for (int nseg = 1; nseg <= 16; ++nseg) {
        int rows = (H + nseg - 1) / nseg;
        rows += rows & 1;
        const int segs = (H + rows - 1) / rows;
        const long long cols = (long long)B * a.nwchunk * segs;
        const int w = (int)(cols < slots ? cols : slots);
        const long long per = (cols + w - 1) / w;
        const double balance = (double)cols / (double)(per * w);
        const double occ = (double)w / (double)slots;
        const double amort = (double)rows / (double)(rows + 4);
        const double score = balance * (0.5 + 0.5 * occ) * amort;
        if (score > best) { best = score; best_seg = segs; best_w = w; a.seg_rows = rows; }
    }
    a.nrseg = best_seg;
    a.ncols = (long long)B * a.nwchunk * a.nrseg;
    a.wgs = best_w;
    if (coarse_amax && fine_amax && cm == 64 && cn == 64 && az_options().conv2d_wgrad_w64) {
        // one 64 x 64 tile per workgroup of eight waves, one workgroup per CU (AZ_CONV2D_WGRAD_W64=0: 2 x 2 tiles on the kernel above)
        int best_seg2 = 1, best_w2 = 1;
        double best2 = -1.0;
        for (int nseg = 1; nseg <= 16; ++nseg) {
            int rows = (H + nseg - 1) / nseg;
            rows += rows & 1;
            const int segs = (H + rows - 1) / rows;
            const long long cols = (long long)B * a.nwchunk * segs;
            const int w = (int)(cols < 256 ? cols : 256);
            const long long per = (cols + w - 1) / w;
            const double score = ((double)cols / (double)(per * w)) * (0.5 + 0.5 * (double)w / 256.0) * ((double)rows / (double)(rows + 4));
            if (score > best2) { best2 = score; best_seg2 = segs; best_w2 = w; a.seg_rows = rows; }
        }
        a.nrseg = best_seg2;
        a.ncols = (long long)B * a.nwchunk * a.nrseg;
        a.wgs = best_w2;
        hipLaunchKernelGGL(conv2d_wgrad_w64_kernel, dim3((unsigned)a.wgs), dim3(512), 0, s, a);
        return az_launch_status();
    }
    if (coarse_amax && fine_amax) hipLaunchKernelGGL(conv2d_wgrad_r16_kernel<1>, dim3((unsigned)(a.wgs * ntiles)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(conv2d_wgrad_r16_kernel<0>, dim3((unsigned)(a.wgs * ntiles)), dim3(256), 0, s, a);
    return az_launch_status();
}
