// Weight gradient of the STRIDE-2 3x3x3 layers (f16x3): hourglass conv1 / conv3 (Conv3d stride 2) and conv5 / conv6
// (ConvTranspose3d stride 2), psmnet_3.py:34-58.  Same sum as az_conv3d_wgrad.hip,
//
//   G[m][n][kd,kh,kw] = sum over coarse positions (b, d, h, w) of  coarse[b,d,h,w][m] * fine[b, 2d-1+kd, 2h-1+kh, 2w-1+kw][n]
//
// with (coarse, fine) = (grad_out, input) for the convolutions and (input, grad_out) for the transposed ones; the coarse
// tensor has 64 channels in every such layer of the network, the fine one 32 (V1 / V0) or 64 (V2 / V1).
//
// az_conv3d_wgrad.hip gives a wave one kd and stages between two barriers of a one-wave workgroup: 0.96-0.98 ms for the
// 86.6 GFLOP of the V1 / V0 layers in either arithmetic (profiles/r04_s2_family_alone.txt) -- the kernel waits for its
// staging, not for the matrix pipe.  A stride-2 layer has the fine tensor of a V0 layer behind a quarter of the flops
// (every fine voxel meets 27/8 taps on average instead of 27), so the staging has to be shared as widely as possible and
// hidden completely.  This kernel is az_conv3d_wgrad16.hip's design turned to that:
//   * a workgroup of EIGHT waves owns a 64 x 32 (coarse channel, fine channel) tile of all 27 taps on
//     v_mfma_f32_16x16x32_f16: a wave takes SEVEN taps (a quarter of the 27, the last quarter one dummy) of a 64 x 16 block
//     (112 accumulator registers), so a fine fragment read from LDS feeds four M blocks: twelve MFMAs per four transposing
//     reads.  The first version gave a wave a 16 x 16 block of all 27 taps, the layout of az_conv3d_wgrad16.hip: every fine
//     fragment was read four times over and the kernel was bound by LDS read bandwidth (112 reads x 512 B per wave and
//     step against 81 MFMAs: 3 450 LDS cycles per step and CU against 2 600 matrix cycles per SIMD; 0.35 ms); every staged
//     byte is shared by all eight waves;
//   * K = 32 positions per MFMA = 4 coarse rows x 8 positions (V1: 68 x 120 and V2: 34 x 60 tile with at most half a
//     chunk of padding); a step needs the 9-row window of three fine planes, 17 fine positions wide, of which 8 rows are
//     new.  A fine row is kept in LDS as two images -- odd and even positions -- so that tap kw reads 8 CONSECUTIVE rows of
//     one image (kw = 0: odd image, kw = 1: even image, kw = 2: odd image one further), exactly like a stride-1 tap;
//   * each plane keeps a ring of 17 fine rows (9 being read + 8 being written), the coarse chunk is double-buffered: one
//     barrier per step; loads, zero padding and therefore vmcnt bookkeeping go through buffer instructions with
//     out-of-range offsets (a step is one basic block); the set of step s+2 is requested in two halves while step s is
//     multiplied, the set of step s+1 is split and written two pieces per tap;
//   * timing-only ablations (S2_ABL, tools/abl_wgrad_s2.sh; conv1's weight gradient, 0.37 ms on that box): no split / LDS
//     writes 0.21, no MFMAs 0.24, no fine-fragment reads 0.38, every column on the same addresses (L2 hits) 0.34, no
//     flush 0.35 -- the matrix work and the staging VALU work of a step add up instead of overlapping (HBM: 1.0-1.1 GB read for
//     1.0 GB of operands).  Tried without effect: the request for step s+2 right behind each split pair instead of in two
//     halves; the two waves of a SIMD staging behind different halves of the taps (two copies of the walk);
//   * one workgroup per CU (127 KB of LDS), persistent over a list of (batch, coarse depth, 8-position chunk) columns,
//     one atomic flush at the end into the tap-major workspace of az_conv3d_wgrad.hip.
// Bank conflicts of the transposing reads: the two octets of a 32-lane half read fine rows two apart; the 32-byte channel
// halves of a 64-byte LDS row are swapped on every other PAIR of fine rows (writes and reads XOR the in-row offset with
// (((row + 1) >> 1) & 1) << 5), so the two octets always hit different halves of the bank period whatever the ring slots.
#include <stdlib.h>

#include "az_roll_common.h"
#include "az_options.h"
#include "az_launch_math.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

#define S2_ROWB 64                                  // bytes of one (position, 32 channels) fp16 row
#define S2_FPOS AZ_S2W_FPOS                                  // fine positions staged per row: odd image 9, even image 8
#define S2_FPART (S2_FPOS * S2_ROWB)                // one part of a fine row                                1 088 B
#define S2_FROW (2 * S2_FPART)                      // [part][odd 9 | even 8][32 ch]                         2 176 B
#define S2_RING AZ_S2W_RING
#define S2_CIMG (32 * S2_ROWB)                      // one (part, 32-channel half) image of a coarse chunk   2 048 B
#define S2_CBUF (4 * S2_CIMG)                       // [part][half][k = 4 rows x 8][32 ch]                   8 192 B
#define S2_FBASE (2 * S2_CBUF)
#define S2_LDS (S2_FBASE + 3 * S2_RING * S2_FROW)   // 127 360 B
#define S2_FROWQ (S2_FPOS * 8)                      // float4 pieces of one fine row: 136
#define S2_NFQ (3 * 8 * S2_FROWQ)                   // fine pieces per step: 3 264
#define S2_NLD 8                                    // pieces per thread and step: 1 coarse + 7 fine (the last one partial)
#define S2_OOB 0xffffff00u
#ifndef S2_ABL
#define S2_ABL 0  // timing-only ablations (tools/abl_wgrad_s2.sh): 1 no global loads, 2 no split / LDS writes, 4 no fine-fragment reads, 8 no MFMAs, 16 no flush, 32 every column loads the same addresses (L2 hits)
#endif

struct Wg16s2Args {
    const float *coarse, *fine;
    float *ws;  // [27][64][CN]
    int B, Dc, Hc, Wc, Df, Hf, Wf, CN;
    int nwchunk;
    long long ncols;  // B * Dc * nwchunk columns of work
    int wgs;          // persistent workgroups per fine-channel tile
    const float *coarse_amax, *fine_amax;
};

// PSM: bit 0 = the coarse operand, bit 1 = the fine operand is a pre-split tensor (az_roll_common.h): a stride-2 layer's
// coarse operand is the gradient of its raw output, a transposed layer's fine one.
template <int PSM = 0>
__global__ void __launch_bounds__(512, 2)
conv3d_wgrad_s2r16_kernel(const Wg16s2Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[S2_LDS + 64];  // + a sink for the lanes of the partial piece
    unsigned char *const cbuf = lds;               // [2][S2_CBUF]
    unsigned char *const fring = lds + S2_FBASE;   // [plane kd][slot][S2_FROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tq4 = wv >> 1, ni = wv & 1;  // a wave: taps 7 tq4 .. 7 tq4 + 6, all 64 coarse x 16 fine channels
    const int tap0 = 7 * tq4;
    // workgroups b and b + 8 share an XCD: each XCD walks a contiguous run of columns (neighbouring depths of one chunk read
    // the same odd fine planes: one L2 serves both)
    const int tile = blockIdx.x / a.wgs, wgl = blockIdx.x - tile * a.wgs;
    const int wg0 = (a.wgs & 7) ? wgl : az_xcd_map(wgl, a.wgs);
    const int n0 = tile * 32;

    f32x4 acc[7][4];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[t][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kc = az_f16_scale_exp(az_amax_read(a.coarse_amax));
    const int kf = az_f16_scale_exp(az_amax_read(a.fine_amax));
    const float c_scale = az_pow2(kc), f_scale = az_pow2(kf), o_scale = ldexpf(1.f, -(kc + kf));

    // transposing-read geometry (ds_read_b64_tr_b16 on a [k][32 ch] image, az_conv3d_wgrad16.hip): a 16-lane group = one K
    // octet = one coarse row of the step; lane 4q + p supplies the address of k-row q (position q, then q + 4), channels
    // 4p..4p+3, and receives channel (lane & 15)
    const int oct = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    // M block mb of the coarse chunk: image half mb >> 1, 16-channel group mb & 1 (= the other 32-byte half: ^ 32)
    const unsigned a_lane = (unsigned)(8 * oct + tq) * S2_ROWB + (((unsigned)(4 * tp) * 2) ^ ((unsigned)(oct & 1) << 5));
    // fine: tap kw starts at LDS row 0 (odd image), 9 (even image), 1 (odd image, one further) of a staged row
    const unsigned b_lane = (unsigned)tq * S2_ROWB + (((unsigned)(16 * ni + 4 * tp) * 2) ^ ((unsigned)(oct & 1) << 5));

    auto frag2 = [&](const unsigned char *lo, const unsigned char *hi) -> az_f16x8 {  // k rows 0..3 at lo, 4..7 at hi
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lo));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(hi));
        s16x8 v;
        v[0] = lo4[0]; v[1] = lo4[1]; v[2] = lo4[2]; v[3] = lo4[3];
        v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
        return __builtin_bit_cast(az_f16x8, v);
    };

    const unsigned vb_c = 64u * 4u, vb_f = (unsigned)a.CN * 4u;  // bytes per voxel
    const unsigned plane_c = (unsigned)a.Hc * a.Wc * vb_c, plane_f = (unsigned)a.Hf * a.Wf * vb_f;
    const unsigned vol_c = (unsigned)a.Dc * plane_c, vol_f = (unsigned)a.Df * plane_f;
    const int nsteps = (a.Hc + 3) >> 2;

    // Staging pieces of a thread, fixed for the whole kernel.  it = 0: coarse, k = tid >> 4 (row k >> 3, position k & 7),
    // float4 tid & 15 of the 64 channels.  it = 1..7: fine piece f = tid + 512 (it - 1): plane kd = f / 1088, new row
    // j = .. / 136, position pp = .. / 8 (fine x = 2 cw0 - 1 + pp), float4 f & 7 of the tile's 32 channels.  What a step
    // needs of a piece is kept in ONE register each for the global and the LDS side (the compiler otherwise hoists the whole
    // decomposition of all eight pieces out of the step loop and spills): rel_g = offset from voxel (plane 2cd-1, row frow0,
    // x 2cw0-1), rel_l = offset from slot 0 of plane 0; the rows j packed three bits each.
    const unsigned relc_g = (unsigned)(((tid >> 4) >> 3) * a.Wc + ((tid >> 4) & 7)) * vb_c + (unsigned)(tid & 15) * 16u;
    const unsigned relc_l = (unsigned)((tid & 15) >> 3) * S2_CIMG + (unsigned)(tid >> 4) * S2_ROWB +
                            ((unsigned)((tid & 7) * 8) ^ ((((unsigned)(tid >> 4) >> 3) & 1u) << 5));
    unsigned rel_g[7], rel_l[7], jpack = 0, pppack0 = 0, pppack1 = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int f = tid + 512 * i;
        int kd, jj, pp, lrow;
        az_s2w_fine_piece(f, kd, jj, pp, lrow);  // (az_launch_math.h: swept on the CPU)
        rel_g[i] = (unsigned)kd * plane_f + (unsigned)(jj * a.Wf + pp) * vb_f + (unsigned)(f & 7) * 16u;
        rel_l[i] = (unsigned)(S2_FBASE + kd * (S2_RING * S2_FROW) + lrow * S2_ROWB + (f & 7) * 8);
        jpack |= (unsigned)jj << (3 * i);
        if (i < 4) pppack0 |= (unsigned)pp << (8 * i); else pppack1 |= (unsigned)pp << (8 * (i - 4));
        if (f >= S2_NFQ) { rel_g[i] = S2_OOB; rel_l[i] = S2_LDS + (unsigned)(tid & 7) * 8; }  // the partial last piece: sink
    }
    const bool live7 = tid + 512 * 6 < S2_NFQ;

    for (long long col = wg0; col < a.ncols; col += a.wgs) {
        int cd, wc, b;
        az_s2w_col_decode(col, a.Dc, a.nwchunk, cd, wc, b);
        const int cw0 = wc * 8;
        const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.coarse) + (size_t)b * (vol_c / 4), 0, vol_c, 0x00020000);
        const auto rs_f = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.fine) + (size_t)b * (vol_f / 4) + n0, 0, vol_f, 0x00020000);
        // validity of a piece that does not change along the column (plane and x in range): one bit per piece
        unsigned colok = (cw0 + ((tid >> 4) & 7) < a.Wc) ? 1u : 0u;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const int pp = (int)(((i < 4 ? pppack0 : pppack1) >> (8 * (i & 3))) & 31u);
            const int kd = (tid + 512 * i) / (8 * S2_FROWQ);
            const bool ok = (unsigned)(2 * cd - 1 + kd) < (unsigned)a.Df && (unsigned)(2 * cw0 - 1 + pp) < (unsigned)a.Wf && (i < 6 || live7);
            colok |= (ok ? 1u : 0u) << (i + 1);
        }
        // (unsigned arithmetic: the base may lie "before" the tensor, base + rel of a valid piece never does)
        const unsigned gbase_f = (S2_ABL & 32) ? plane_f + 15u * vb_f : (unsigned)(2 * cd - 1) * plane_f + (unsigned)(2 * cw0 - 1) * vb_f;
        const unsigned gbase_c = (S2_ABL & 32) ? plane_c + 8u * vb_c : (unsigned)cd * plane_c + (unsigned)cw0 * vb_c;

        u32x4 pre[S2_NLD];
        const bool s_abl_first = col == wg0;
        auto issue = [&](int crow0, int frow0, int it0, int it1) {  // coarse rows crow0..+3, fine rows frow0..+7 (frow0 >= 0)
#pragma unroll
            for (int it = it0; it < it1; ++it) {
                unsigned off = S2_OOB;
                if (S2_ABL & 1) { if (s_abl_first) pre[it] = u32x4{1u, 2u, 3u, 4u}; continue; }
                if (it == 0) {
                    if ((colok & 1u) && crow0 + (int)((tid >> 4) >> 3) < a.Hc)
                        off = gbase_c + (unsigned)crow0 * (unsigned)a.Wc * vb_c + relc_g;
                    pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, off, 0, 0);
                } else {
                    const int jj = (int)((jpack >> (3 * (it - 1))) & 7u);
                    if (((colok >> it) & 1u) && frow0 + jj < a.Hf)
                        off = gbase_f + (unsigned)frow0 * (unsigned)a.Wf * vb_f + rel_g[it - 1];
                    pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_f, off, 0, 0);
                }
            }
        };
        auto commit_piece = [&](int it, int cbuf_idx, int frow0) {
            if (S2_ABL & 2) return;
            uint2 hi, lo;
            if (it == 0) az_stage_f16x4<(PSM & 1) != 0>(pre[it], c_scale, hi, lo);
            else az_stage_f16x4<(PSM & 2) != 0>(pre[it], f_scale, hi, lo);
            unsigned d0, d1;
            if (it == 0) {
                d0 = (unsigned)cbuf_idx * S2_CBUF + relc_l;
                d1 = d0 + 2 * S2_CIMG;
            } else {
                const int fr = frow0 + (int)((jpack >> (3 * (it - 1))) & 7u);
                int slot = az_s2w_ring_slot(frow0) + (fr - frow0);
                slot = slot >= S2_RING ? slot - S2_RING : slot;
                d0 = (rel_l[it - 1] + (unsigned)slot * S2_FROW) ^ ((unsigned)(((fr + 1) >> 1) & 1) << 5);
                d1 = d0 + S2_FPART;
                if (it == 7) { d0 = live7 ? d0 : rel_l[6]; d1 = live7 ? d1 : rel_l[6]; }  // (no branch: a step stays one block)
            }
            *reinterpret_cast<uint2 *>(lds + d0) = hi;
            *reinterpret_cast<uint2 *>(lds + d1) = lo;
        };

        // ---- prologue: fine row -1 (zeros, slot 0), the set of step 0, the request for step 1 ------------------------------
        __syncthreads();  // the previous column's last step no longer reads
        if (tid < 3 * (S2_FROW / 16)) {
            const int pl = tid / (S2_FROW / 16), o = tid - pl * (S2_FROW / 16);
            *reinterpret_cast<uint4 *>(fring + pl * (S2_RING * S2_FROW) + o * 16) = uint4{0u, 0u, 0u, 0u};
        }
        issue(0, 0, 0, S2_NLD);
#pragma unroll
        for (int it = 0; it < S2_NLD; ++it) commit_piece(it, 0, 0);
        issue(4, 8, 0, S2_NLD);
        __syncthreads();

        for (int s = 0; s < nsteps; ++s) {
            const int sb = (8 * s) % S2_RING;  // ring slot of this lane's fine row 8s + 2 oct - 1 + kh: (8s + 2 oct + kh) mod 17
            az_f16x8 af[4][2];  // [M block][part]
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const unsigned char *cm = cbuf + (s & 1) * S2_CBUF + (mb >> 1) * S2_CIMG + (a_lane ^ ((mb & 1) ? 32u : 0u)) + p * 2 * S2_CIMG;
                    af[mb][p] = frag2(cm, cm + 4 * S2_ROWB);
                }
            az_f16x8 bf[2][2];
            auto load_b = [&](az_f16x8 (&bq)[2], int t) {
                int tap = tap0 + t;            // wave-uniform; the 28th tap repeats the 27th (computed, never flushed)
                tap = tap > 26 ? 26 : tap;
                const int kd = tap / 9, r9 = tap - 9 * kd, kh = r9 / 3, kw = r9 - 3 * kh;
                int v = sb + kh + 2 * oct;
                v = v >= S2_RING ? v - S2_RING : v;
                unsigned off = (unsigned)v * S2_FROW + b_lane + (unsigned)(kd * (S2_RING * S2_FROW) + az_s2w_tap_row(kw) * S2_ROWB);
                off ^= kh == 2 ? 32u : 0u;  // the row pair of kh = 2 is the next one: other half order
#pragma unroll
                for (int p = 0; p < 2; ++p) bq[p] = frag2(fring + off + p * S2_FPART, fring + off + 4 * S2_ROWB + p * S2_FPART);
            };
            load_b(bf[0], 0);
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < 7 && !((S2_ABL & 4) && s > 0)) load_b(bf[(t + 1) & 1], t + 1);
                __builtin_amdgcn_sched_barrier(0);
                const az_f16x8(&bq)[2] = bf[t & 1];
                // four independent chains, each lo*hi, hi*lo, hi*hi into the running accumulator (smallest first)
                if (!(S2_ABL & 8)) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[t][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mb][1], bq[0], acc[t][mb], 0, 0, 0);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[t][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mb][0], bq[1], acc[t][mb], 0, 0, 0);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[t][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mb][0], bq[0], acc[t][mb], 0, 0, 0);
                } else {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) acc[t][mb][0] += __builtin_bit_cast(float, (int)af[mb][0][0] ^ (int)bq[0][1]);  // keep the operands live
                }
                // the set of step s+1 (requested a step ago): two pieces behind each of four taps, and the request for the same
                // two pieces of step s+2 as soon as their registers are free
                if (t < 4) {
                    const int pc = 2 * t;
                    __builtin_amdgcn_sched_barrier(0);
                    commit_piece(pc, (s + 1) & 1, 8 * (s + 1));
                    commit_piece(pc + 1, (s + 1) & 1, 8 * (s + 1));
                    __builtin_amdgcn_sched_barrier(0);
                    issue(4 * (s + 2), 8 * (s + 2), pc, pc + 2);
                }
            }
            __syncthreads();  // next step's rows written by all eight waves; this step's no longer read
        }
    }
    // D[i][j]: i = coarse channel 16 mb + 4 (lane >> 4) + r, j = fine channel 16 ni + (lane & 15)
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        const int tap = tap0 + t;
        if (tap > 26) break;  // (wave-uniform)
        if ((S2_ABL & 16) && a.B > 0) break;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * mb + 4 * (lane >> 4) + r;
                atomicAdd(&a.ws[((size_t)tap * 64 + m) * a.CN + n0 + 16 * ni + (lane & 15)], acc[t][mb][r] * o_scale);
            }
    }
}

// one persistent workgroup per CU over the fine-channel tiles; AZ_EUNSUPPORTED: shapes the kernel does not take (the caller
// falls back to az_conv3d_wgrad.hip's one-kd-per-wave kernel)
int az_conv3d_wgrad_s2r16_launch(float *ws, const float *coarse, const float *fine, int B, int cm, int cn, int Dc, int Hc, int Wc,
                                 int Df, int Hf, int Wf, hipStream_t s, const float *coarse_amax, const float *fine_amax, int split_mask) {
    if (cm != 64 || !(cn == 32 || cn == 64) || !coarse_amax || !fine_amax) return AZ_EUNSUPPORTED;
    if (!az_fits_buffer_offset((long long)Df * Hf * Wf * cn * 4) || !az_fits_buffer_offset((long long)Dc * Hc * Wc * cm * 4)) return AZ_EUNSUPPORTED;
    Wg16s2Args a{};
    a.coarse = coarse; a.fine = fine; a.ws = ws; a.coarse_amax = coarse_amax; a.fine_amax = fine_amax;
    a.B = B; a.Dc = Dc; a.Hc = Hc; a.Wc = Wc; a.Df = Df; a.Hf = Hf; a.Wf = Wf; a.CN = cn;
    const int ntiles = cn / 32;
    a.nwchunk = (Wc + 7) / 8;
    a.ncols = (long long)B * Dc * a.nwchunk;
    a.wgs = az_wgrad16_workgroups(a.ncols, 256 / ntiles, ntiles, 0);
    const dim3 grid((unsigned)(a.wgs * ntiles));
    if (split_mask == 0) hipLaunchKernelGGL(conv3d_wgrad_s2r16_kernel<0>, grid, dim3(512), 0, s, a);
    else if (split_mask == 1) hipLaunchKernelGGL(conv3d_wgrad_s2r16_kernel<1>, grid, dim3(512), 0, s, a);
    else if (split_mask == 2) hipLaunchKernelGGL(conv3d_wgrad_s2r16_kernel<2>, grid, dim3(512), 0, s, a);
    else return AZ_EUNSUPPORTED;  // (both operands pre-split: no producer writes a pre-split forward activation yet)
    return az_launch_status();
}
