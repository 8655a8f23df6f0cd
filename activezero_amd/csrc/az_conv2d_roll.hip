// K13r -- the 3x3, stride-1, dilation-1 Conv2d layers of the feature extractor with 32 or 64 input AND output
// channels (reference nets/psmnet/psmnet_submodule_3.py:92-147: firstconv[1..2], layer1, layer2) and their input
// gradients, on the machinery of the depth-rolling 3-D kernel (az_conv3d_roll.hip) with the BATCH as the walk axis:
//
//   * a workgroup owns an 8x16 (y, x) patch and WALKS the images of its statistic group: image p+1 is fetched, split
//     into its bf16 triplet and written to the other half of a double-buffered LDS slab while image p is multiplied;
//     the per-workgroup costs that bound az_conv2d.hip on these small layers -- first slab with the full memory
//     latency in front of it, weight ring refill, descriptor set-up, the drain of the epilogue stores -- are paid
//     once per (patch, group) instead of once per (patch, image), and a stage (image x 32-channel chunk: 216 MFMAs per
//     wave) is ONE basic block with exact vmcnt bookkeeping (validity through buffer-instruction bounds);
//   * v_mfma_f32_16x16x32_bf16, four waves = four 4x8-voxel quarters of the patch, each for ALL output channels
//     (2 M tiles x NT N tiles of 16 channels: NT = 2 for 32 output channels, 4 for 64 -- 16 / 32 accumulator
//     registers): an A fragment read from LDS feeds every N tile (12 / 24 MFMAs) -- with one N tile per wave, as in the
//     3-D kernel where the three kd share the fragment, a 2-D stage would need 3 KB of LDS reads per 96 matrix cycles
//     on every SIMD: all of the LDS bandwidth -- and the slab is staged once per patch whatever the channel count
//     (AZ_CONV2D_ROLL_NT4=0: 64 channels as two workgroups per patch, for A/B runs: 104 vs 102 us alone, 0.106 /
//     0.123 vs 0.098 / 0.111 ms forward / input gradient in the step);
//   * BatchNorm partials (EPI 1): per-lane running sums about a per-lane shift, one row per (wave quarter, group
//     segment), in az_bn2d_fwd's layout [group][cout][rows][2] / [group][rows].
// Arithmetic: az_common.h's bf16x6 product; accumulation order per output: 32-channel chunk, kh, kw.
#include <stdlib.h>
#include <type_traits>

#include "az_roll_common.h"
#include "az_options.h"
#include "az_launch_math.h"

struct C2RArgs {
    const float *in;    // [G*N, H, W, CIN] dense channels-last
    const float *wp;    // packed [cout/32][tap 9][CIN/32][n16 2][part 3][lane 64][8 bf16]
    float *out;         // [G*N, H, W, out_cs]
    const float *scale, *shift, *res;  // optional epilogue operands (res: pixel stride res_cs floats)
    float *part, *cnt;  // EPI 1: [G][cout][rows][2], [G][rows]
    int G, N, H, W;     // statistic groups, images per group
    int cout, out_cs, res_cs;
    int tiles_yb, tiles_x;  // 8-row patch rows, 16-column patch columns
    int nseg, seg_len;      // image segments per group
    int rows;               // partial rows per group = nseg * tiles_yb * tiles_x * 4
    int relu;
    const float *in_amax, *w_amax;  // AR 1 (f16x3): amax arrays of the input and of the (unpacked) weights
};

// EPI: 0 = y = relu?(acc * scale + shift), 2 = the same + residual, 1 = raw output + BatchNorm partials
// AR: 0 = bf16x6, 1 = f16x3 (az_roll_common.h; the slab keeps its 192-byte pixels, the third part unused)
template <int CIN, int EPI, int NT, int AR = 0>
__global__ void __launch_bounds__(256, 2)
conv2d_roll_kernel(const C2RArgs a) {
    constexpr int NCH = CIN / 32;
    constexpr int NP = AR ? 2 : 3;
    constexpr int TAPF4 = NCH * 2 * NP * 64;  // float4 per tap of one channel group: [tap][cc][n16][part][lane]
    float in_scale = 1.f, osc = 1.f;
    if (AR) {
        const int ki = az_f16_scale_exp(az_amax_read(a.in_amax)), kw_ = az_f16_scale_exp(az_amax_read(a.w_amax));
        in_scale = az_pow2(ki);
        osc = ldexpf(1.f, -(ki + kw_));
    }
    __shared__ __attribute__((aligned(16))) unsigned char slab[2 * R_SLAB_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv >> 1, wc = wv & 1;  // this wave's quarter: rows 4 wr .., columns 8 wc ..

    // ---- block -> (channel group, statistic group, segment, patch): contiguous chunk of the linear order per XCD ----
    int lin = az_xcd_map(blockIdx.x, gridDim.x);
    const int ngrp = a.cout / (16 * NT);  // channel groups in the grid (NT = 4: one workgroup does all 64 channels)
    const int cg = lin % ngrp; lin /= ngrp;
    const int tix = lin % a.tiles_x; lin /= a.tiles_x;
    const int tiy = lin % a.tiles_yb; lin /= a.tiles_yb;
    const int seg = lin % a.nseg;
    const int g = lin / a.nseg;
    const int d0 = seg * a.seg_len, d1 = min(d0 + a.seg_len, a.N);  // images [d0, d1) of group g
    const int ty0 = tiy * R_TY, tx0 = tix * R_TX;
    const int ih0 = ty0 - 1, iw0 = tx0 - 1;
    const int ch0 = cg * 16 * NT;

    const unsigned in_bytes = (unsigned)a.N * a.H * a.W * CIN * 4u, out_bytes = (unsigned)a.N * a.H * a.W * a.out_cs * 4u;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in) + (size_t)g * (in_bytes / 4), 0, in_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)g * (out_bytes / 4), 0, out_bytes, 0x00020000);
    const unsigned res_bytes = (unsigned)a.N * a.H * a.W * a.res_cs * 4u;
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(EPI == 2 ? a.res + (size_t)g * (res_bytes / 4) : a.in), 0, EPI == 2 ? res_bytes : 0u, 0x00020000);
    const auto rs_part = __builtin_amdgcn_make_buffer_rsrc(EPI == 1 ? a.part : a.out, 0,
                                                           EPI == 1 ? (unsigned)((size_t)a.G * a.cout * a.rows * 8) : 0u, 0x00020000);
    const auto rs_cnt = __builtin_amdgcn_make_buffer_rsrc(EPI == 1 ? a.cnt : a.out, 0,
                                                          EPI == 1 ? (unsigned)((size_t)a.G * a.rows * 4) : 0u, 0x00020000);

    f32x4 acc[2][NT];  // [M tile: columns 4 m .. of the quarter][N tile: channels 16 n ..]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging: one image chunk = 10 x 18 pixels x 32 channels fp32 -> bf16 triplets in LDS (az_conv3d_roll.hip) ----
    u32x4 pre[R_NLD];
    auto issue = [&](int p, int cc) {
        int sy = 0, sx = tid >> 3;
        if (sx >= R_SX) { sx -= R_SX; ++sy; }
#pragma unroll
        for (int it = 0; it < R_NLD; ++it) {
            const int ih = ih0 + sy, iw = iw0 + sx;
            const bool ok = (tid + 256 * it < R_NQ) && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W &&
                            (unsigned)p < (unsigned)a.N;
            const unsigned off = (unsigned)((p * a.H + ih) * a.W + iw) * (CIN * 4) + cc * 128 + (tid & 7) * 16;
            pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? off : R_OOB, 0, 0);
            sx += 14; ++sy;
            if (sx >= R_SX) { sx -= R_SX; ++sy; }
        }
    };
    auto commit_piece = [&](int it, unsigned char *dstbuf) {
        const bool live = (tid + 256 * it < R_NQ);
        const int ite = (it > 0 && !live) ? it - 1 : it;
        const int vox = (tid >> 3) + 32 * ite;
        const int sy = vox / R_SX, sx = vox - sy * R_SX;
        const int j = tid & 7;
        u32x4 raw = pre[it];
        if (it > 0 && 256 * it + 255 >= R_NQ) {
            raw[0] = live ? raw[0] : pre[it - 1][0]; raw[1] = live ? raw[1] : pre[it - 1][1];
            raw[2] = live ? raw[2] : pre[it - 1][2]; raw[3] = live ? raw[3] : pre[it - 1][3];
        }
        unsigned char *dst = dstbuf + (sy * R_SX + sx) * R_VB + ((((j >> 1) ^ ((sy & 1) << 1))) << 4) + (j & 1) * 8;
        if (AR) {
            float4 v = __builtin_bit_cast(float4, raw);
            v.x *= in_scale; v.y *= in_scale; v.z *= in_scale; v.w *= in_scale;
            uint2 hi, lo;
            az_split2_f16x4(v, hi, lo);
            *reinterpret_cast<uint2 *>(dst) = hi;
            *reinterpret_cast<uint2 *>(dst + 64) = lo;
        } else {
            uint2 hi, mid, lo;
            az_split3_bf16x4(__builtin_bit_cast(float4, raw), hi, mid, lo);
            *reinterpret_cast<uint2 *>(dst) = hi;
            *reinterpret_cast<uint2 *>(dst + 64) = mid;
            *reinterpret_cast<uint2 *>(dst + 128) = lo;
        }
    };

    // ---- operands ----
    const int trow = (lane >> 2) & 3, tcol = lane & 3, oct = lane >> 4;
    unsigned abase[2];
    abase[0] = ((4 * wr + trow) * R_SX + 8 * wc + tcol) * R_VB + ((oct ^ ((trow & 1) << 1)) << 4);
    abase[1] = ((4 * wr + trow) * R_SX + 8 * wc + tcol) * R_VB + ((oct ^ (((trow + 1) & 1) << 1)) << 4);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp) + (size_t)cg * (NT / 2) * (9 * TAPF4 * 4), 0,
                                                        (unsigned)(NT / 2) * 9u * TAPF4 * 16u, 0x00020000);
    const unsigned wlane = (unsigned)lane * 16u;
    auto load_b = [&](float4 (&bq)[3], int n, int tap_f4) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
            bq[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                rs_w, wlane, (n >> 1) * (9 * TAPF4 * 16) + tap_f4 * 16 + (n & 1) * (NP * 1024) + p * 1024, 0));
    };
    // per-channel epilogue constants (four consecutive channels per lane and N tile after the quad transpose)
    const int cq = lane & 12;
    auto epi_consts = [&](int n, float4 &sc, float4 &sf) {  // (L2-resident; not held across the walk: registers)
        sc = make_float4(1.f, 1.f, 1.f, 1.f); sf = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI != 1) {
            if (a.scale) sc = *reinterpret_cast<const float4 *>(a.scale + ch0 + 16 * n + cq);
            if (a.shift) sf = *reinterpret_cast<const float4 *>(a.shift + ch0 + 16 * n + cq);
        }
    };
    const float floor_ = a.relu ? 0.f : -__builtin_inff();

    // BatchNorm partials: see az_conv3d_roll.hip -- element e = 4 m + r of a lane (its x offset in the quarter) is valid
    // iff e < st_nv; two channels per lane (N tiles 0, 1)
    const int oh_l = ty0 + 4 * wr + (lane >> 4);
    int st_nv = 0;
    if (EPI == 1) st_nv = (oh_l < a.H) ? min(max(a.W - (tx0 + 8 * wc), 0), 8) : 0;
    float st_k[NT], st_s1[NT], st_s2[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) { st_k[n] = 0.f; st_s1[n] = 0.f; st_s2[n] = 0.f; }
    int st_planes = 0;
    bool st_first = true;

    auto finish = [&](int o, bool ok) {
        const bool row_ok = ok && oh_l < a.H;
        const unsigned pix_row = (unsigned)((o * a.H + oh_l) * a.W);
        if (AR) {  // undo the operand scales on the finished sums
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] *= osc;
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float4 sc, sf;
            epi_consts(n, sc, sf);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int ow = tx0 + 8 * wc + 4 * m + (lane & 3);
                const f32x4 v = r16_quad_transpose(acc[m][n], lane);
                const bool vok = row_ok && ow < a.W;
                const unsigned off = vok ? (pix_row + (unsigned)ow) * (unsigned)(a.out_cs * 4) + (unsigned)(ch0 + 16 * n + cq) * 4u : R_OOB;
                float4 y = make_float4(v[0] * sc.x + sf.x, v[1] * sc.y + sf.y, v[2] * sc.z + sf.z, v[3] * sc.w + sf.w);
                if (EPI == 2) {
                    const unsigned roff = vok ? (pix_row + (unsigned)ow) * (unsigned)(a.res_cs * 4) + (unsigned)(ch0 + 16 * n + cq) * 4u : R_OOB;
                    const float4 rr = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, roff, 0, 0));
                    y.x += rr.x; y.y += rr.y; y.z += rr.z; y.w += rr.w;
                }
                if (EPI != 1) { y.x = fmaxf(y.x, floor_); y.y = fmaxf(y.y, floor_); y.z = fmaxf(y.z, floor_); y.w = fmaxf(y.w, floor_); }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), rs_out, off, 0, 0);
            }
        }
        if (EPI == 1) {
            const int nv_p = ok ? st_nv : 0;
            st_planes += ok ? 1 : 0;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                st_k[n] = (ok && st_first) ? acc[0][n][0] : st_k[n];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dlt = (m * 4 + r < nv_p) ? acc[m][n][r] - st_k[n] : 0.f;
                        st_s1[n] += dlt;
                        st_s2[n] = fmaf(dlt, dlt, st_s2[n]);
                    }
            }
            st_first = st_first && !ok;
        }
    };
    auto flush_stats = [&]() {
        const unsigned row_id = (unsigned)(((seg * a.tiles_yb + tiy) * a.tiles_x + tix) * 4 + wv);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float cnt = (float)(st_nv * st_planes);
            float mean = cnt > 0.f ? st_k[n] + st_s1[n] / cnt : 0.f;
            float m2 = cnt > 0.f ? st_s2[n] - st_s1[n] * st_s1[n] / cnt : 0.f;
#pragma unroll
            for (int off = 16; off < 64; off <<= 1) {
                const float n_o = __shfl_xor(cnt, off), mean_o = __shfl_xor(mean, off), m2_o = __shfl_xor(m2, off);
                const float nn = cnt + n_o;
                const float dlt = mean_o - mean;
                const float w_o = nn > 0.f ? n_o / nn : 0.f;
                m2 = m2 + m2_o + dlt * dlt * cnt * w_o;
                mean = mean + dlt * w_o;
                cnt = nn;
            }
            const unsigned ch = (unsigned)(ch0 + 16 * n + (lane & 15));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, make_float2(cnt * mean, fmaxf(m2, 0.f))), rs_part,
                                                  lane < 16 ? (unsigned)((((size_t)g * a.cout + ch) * a.rows + row_id) * 8) : R_OOB, 0, 0);
            if (n == 0)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cnt), rs_cnt,
                                                      (lane == 0 && cg == 0) ? (unsigned)(((size_t)g * a.rows + row_id) * 4) : R_OOB, 0, 0);
        }
    };

    // ---- one stage: image p, chunk CC, slab in buffer `buf`; straight-line ----
    float4 wk[2][NT][3];
    f32x4 tq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    auto stage_s = [&](auto cc_tag, int p, int buf) {
        constexpr int CC = decltype(cc_tag)::value;
        constexpr bool LAST = (CC == NCH - 1);
        constexpr int CCN = (CC + 1) % NCH;
        const unsigned char *sl = slab + buf * R_SLAB_BYTES;
        unsigned char *sn = slab + (buf ^ 1) * R_SLAB_BYTES;
        constexpr int wcur = CC * (2 * NP * 64), wnxt = CCN * (2 * NP * 64);
        const int pn = LAST ? p + 1 : p;
        if (CC == 0) {  // the image before is complete: its epilogue opens this stage, its stores have the stage to land
            finish(p - 1, p - 1 >= d0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float4 av[2][3];
        auto load_a = [&](float4 (&aq)[3], int m, int kh, int kw) {
            const unsigned char *ap = sl + abase[kh & 1] + (kh * R_SX + 4 * m + kw) * R_VB;
#pragma unroll
            for (int q = 0; q < NP; ++q) aq[q] = *reinterpret_cast<const float4 *>(ap + 64 * q);
        };
        load_a(av[0], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int kh = j / 3, kw = j % 3;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NT; ++n)
                load_b(wk[(j + 1) & 1][n], n, (j + 1 < 9 ? wcur + (j + 1) * TAPF4 : wnxt));
            if (j == 0) {
                __builtin_amdgcn_sched_barrier(0);
                issue(pn, CCN);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int t = j * 2 + m;
                __builtin_amdgcn_sched_barrier(0);
                if (m + 1 < 2) load_a(av[(t + 1) & 1], m + 1, kh, kw);
                else if (j + 1 < 9) load_a(av[(t + 1) & 1], 0, (j + 1) / 3, (j + 1) % 3);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const int st = t * NT + n;
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 &prev = n > 0 ? acc[m][n - 1] : (m > 0 ? acc[m - 1][NT - 1] : acc[1][NT - 1]);
                    if constexpr (AR) r16_step3(tq[st & 1], av[t & 1], wk[j & 1][n], prev, tq[(st + 1) & 1]);
                    else r16_step(tq[st & 1], av[t & 1], wk[j & 1][n], prev, tq[(st + 1) & 1]);
                    // the next image's six pieces, one per tap from the fourth on, behind the tap's last step
                    if (m == 1 && n == NT - 1 && j >= 3) {
                        __builtin_amdgcn_sched_barrier(0);
                        commit_piece(j - 3, sn);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[1][NT - 1] += tq[1];  // 18 NT steps (even): the last one wrote tq[1]
        tq[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < NP; ++q) wk[0][n][q] = wk[1][n][q];
        __syncthreads();
    };

    // ---- the walk over the images of the segment ----
    issue(d0, 0);
#pragma unroll
    for (int n = 0; n < NT; ++n) load_b(wk[0][n], n, 0);
#pragma unroll
    for (int it = 0; it < R_NLD; ++it) commit_piece(it, slab);
    __syncthreads();
    int buf = 0;
    for (int p = d0; p < d1; ++p) {
        if (NCH == 1) {
            stage_s(std::integral_constant<int, 0>{}, p, buf); buf ^= 1;
        } else {
            stage_s(std::integral_constant<int, 0>{}, p, buf); buf ^= 1;
            stage_s(std::integral_constant<int, NCH - 1>{}, p, buf); buf ^= 1;
        }
    }
    finish(d1 - 1, true);
    if (EPI == 1) flush_stats();
}

// ---- f16x3, 64 output channels: the wave layout of az_conv3d_roll.hip's f16x3 stage -------------------------------------------
// conv2d_roll_kernel<CIN, EPI, 4, 1> gives a wave a quarter of the patch (two 4x4 tiles) for ALL 64 channels: every tap costs it
// four weight-fragment pairs (8 KB through the L1) for 24 MFMAs, one accumulate-add group per three MFMAs (r16_step3) -- per stage
// and CU 576 KB of weight fragments through a 64 B/clk L1 beside 6.9 k matrix cycles, and two VALU instructions per MFMA.  Here
// a wave owns HALF of the channels (two 16-channel blocks) and HALF of the patch (four tiles), exactly the f16x3 stage of the 3-D
// kernel with its three kd slots replaced by the two channel blocks: chains of nine MFMAs over the kw taps (r16_chain9, swapped
// operand roles: a lane holds four channels of one pixel, no transpose), the weights of a kh row (2 blocks x 3 kw x 2 parts) in
// registers for all four tiles, the pixel fragments of a tile pair (2 x 3 kw x 2 parts) serving both blocks: half the weight
// bytes and a third of the accumulate-adds per MFMA.  Same slab, staging, walk over the images and partial-row layout.
template <int CIN, int EPI>
__global__ void __launch_bounds__(256, 2)
conv2d_roll64_kernel(const C2RArgs a) {
    constexpr int NCH = CIN / 32;
    constexpr int TAPF4 = NCH * 2 * 2 * 64;  // float4 per tap of one 32-channel group: [tap][cc][n16][part][lane]
    const int ki = az_f16_scale_exp(az_amax_read(a.in_amax)), kw_ = az_f16_scale_exp(az_amax_read(a.w_amax));
    const float in_scale = az_pow2(ki), osc = ldexpf(1.f, -(ki + kw_));
    __shared__ __attribute__((aligned(16))) unsigned char slab[2 * R_SLAB_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv & 1, wm = wv >> 1;  // 32-channel half, 4-row half of the patch

    int lin = az_xcd_map(blockIdx.x, gridDim.x);
    const int tix = lin % a.tiles_x; lin /= a.tiles_x;
    const int tiy = lin % a.tiles_yb; lin /= a.tiles_yb;
    const int seg = lin % a.nseg;
    const int g = lin / a.nseg;
    const int d0 = seg * a.seg_len, d1 = min(d0 + a.seg_len, a.N);
    const int ty0 = tiy * R_TY, tx0 = tix * R_TX;
    const int ih0 = ty0 - 1, iw0 = tx0 - 1;

    const unsigned in_bytes = (unsigned)a.N * a.H * a.W * CIN * 4u, out_bytes = (unsigned)a.N * a.H * a.W * a.out_cs * 4u;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in) + (size_t)g * (in_bytes / 4), 0, in_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)g * (out_bytes / 4), 0, out_bytes, 0x00020000);
    const unsigned res_bytes = (unsigned)a.N * a.H * a.W * a.res_cs * 4u;
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(EPI == 2 ? a.res + (size_t)g * (res_bytes / 4) : a.in), 0, EPI == 2 ? res_bytes : 0u, 0x00020000);
    const auto rs_part = __builtin_amdgcn_make_buffer_rsrc(EPI == 1 ? a.part : a.out, 0,
                                                           EPI == 1 ? (unsigned)((size_t)a.G * a.cout * a.rows * 8) : 0u, 0x00020000);
    const auto rs_cnt = __builtin_amdgcn_make_buffer_rsrc(EPI == 1 ? a.cnt : a.out, 0,
                                                          EPI == 1 ? (unsigned)((size_t)a.G * a.rows * 4) : 0u, 0x00020000);

    f32x4 acc[2][4];  // [16-channel block of this wave's 32][4x4 tile: columns 4 m ..]
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging (as conv2d_roll_kernel; pieces in two halves) ----
    u32x4 pre[R_NLD];
    auto issue = [&](int p, int cc, int it0, int it1) __attribute__((always_inline)) {
        int t_ = threadIdx.x;
        asm volatile("" : "+v"(t_));  // (recompute the piece offsets at every call: az_conv3d_roll.hip)
        int sy = 0, sx = t_ >> 3;
        if (sx >= R_SX) { sx -= R_SX; ++sy; }
#pragma unroll
        for (int it = 0; it < R_NLD; ++it) {
            const int ih = ih0 + sy, iw = iw0 + sx;
            const bool ok = (t_ + 256 * it < R_NQ) && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W &&
                            (unsigned)p < (unsigned)a.N;
            const unsigned off = (unsigned)((p * a.H + ih) * a.W + iw) * (CIN * 4) + cc * 128 + (t_ & 7) * 16;
            if (it >= it0 && it < it1) pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? off : R_OOB, 0, 0);
            sx += 14; ++sy;
            if (sx >= R_SX) { sx -= R_SX; ++sy; }
        }
    };
    auto commit_piece = [&](int it, unsigned char *dstbuf) __attribute__((always_inline)) {
        int t_ = threadIdx.x;
        asm volatile("" : "+v"(t_));
        const bool live = (t_ + 256 * it < R_NQ);
        const int ite = (it > 0 && !live) ? it - 1 : it;
        const int vox = (t_ >> 3) + 32 * ite;
        const int sy = vox / R_SX, sx = vox - sy * R_SX;
        const int j = t_ & 7;
        u32x4 raw = pre[it];
        if (it > 0 && 256 * it + 255 >= R_NQ) {
            raw[0] = live ? raw[0] : pre[it - 1][0]; raw[1] = live ? raw[1] : pre[it - 1][1];
            raw[2] = live ? raw[2] : pre[it - 1][2]; raw[3] = live ? raw[3] : pre[it - 1][3];
        }
        unsigned char *dst = dstbuf + (sy * R_SX + sx) * R_VB + ((((j >> 1) ^ ((sy & 1) << 1))) << 4) + (j & 1) * 8;
        float4 v = __builtin_bit_cast(float4, raw);
        v.x *= in_scale; v.y *= in_scale; v.z *= in_scale; v.w *= in_scale;
        uint2 hi, lo;
        az_split2_f16x4(v, hi, lo);
        *reinterpret_cast<uint2 *>(dst) = hi;
        *reinterpret_cast<uint2 *>(dst + 64) = lo;
    };

    // pixel fragment: lane -> pixel (row (lane >> 2) & 3, x lane & 3) of a 4x4 tile, channel octet lane >> 4
    const int trow = (lane >> 2) & 3, tcol = lane & 3, oct = lane >> 4;
    unsigned abase[2];
    abase[0] = ((4 * wm + trow) * R_SX + tcol) * R_VB + ((oct ^ ((trow & 1) << 1)) << 4);
    abase[1] = ((4 * wm + trow) * R_SX + tcol) * R_VB + ((oct ^ (((trow + 1) & 1) << 1)) << 4);
    // weights of this wave's 32-channel group: [tap][cc][n16][part][lane] float4
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp) + (size_t)wn * (9 * TAPF4 * 4), 0, 9u * TAPF4 * 16u, 0x00020000);
    const unsigned wlane = (unsigned)lane * 16u;
    float4 wh[2][3][2];  // [16-channel block][kw][part] of the current kh row
    auto load_bh = [&](float4 (&bq)[3][2], int n, int tap0_f4) __attribute__((always_inline)) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                bq[kw][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                    rs_w, wlane, (tap0_f4 + kw * TAPF4) * 16 + n * 2048 + q * 1024, 0));
    };

    // epilogue: this lane's pixel (row oh_l, columns tx0 + 4 m + tcol) and four channels 32 wn + 16 n + 4 (lane >> 4) + r
    const int oh_l = ty0 + 4 * wm + trow;
    const int cl = 32 * wn + 4 * (lane >> 4);
    const float floor_ = a.relu ? 0.f : -__builtin_inff();
    float hk[2][4], hs1[2][4], hs2[2][4];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) { hk[n][r] = 0.f; hs1[n][r] = 0.f; hs2[n][r] = 0.f; }
    int h_n = 0;
    bool h_first = true;
    auto finish = [&](int o, bool ok) __attribute__((always_inline)) {
        const bool row_ok = ok && oh_l < a.H;
        const unsigned pix_row = (unsigned)((o * a.H + oh_l) * a.W);
        if (EPI == 1) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) hk[n][r] = (ok && h_first) ? acc[n][0][r] * osc : hk[n][r];
            h_first = h_first && !ok;
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
            if (EPI != 1) {
                if (a.scale) sc = *reinterpret_cast<const float4 *>(a.scale + cl + 16 * n);
                if (a.shift) sf = *reinterpret_cast<const float4 *>(a.shift + cl + 16 * n);
            }
            sc.x *= osc; sc.y *= osc; sc.z *= osc; sc.w *= osc;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int ow = tx0 + 4 * m + tcol;
                const bool vok = row_ok && ow < a.W;
                const unsigned off = vok ? (pix_row + (unsigned)ow) * (unsigned)(a.out_cs * 4) + (unsigned)(cl + 16 * n) * 4u : R_OOB;
                const f32x4 c = acc[n][m];
                float4 y = make_float4(fmaf(c[0], sc.x, sf.x), fmaf(c[1], sc.y, sf.y), fmaf(c[2], sc.z, sf.z), fmaf(c[3], sc.w, sf.w));
                if (EPI == 2) {
                    const unsigned roff = vok ? (pix_row + (unsigned)ow) * (unsigned)(a.res_cs * 4) + (unsigned)(cl + 16 * n) * 4u : R_OOB;
                    const float4 rr = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, roff, 0, 0));
                    y.x += rr.x; y.y += rr.y; y.z += rr.z; y.w += rr.w;
                }
                if (EPI != 1) { y.x = fmaxf(y.x, floor_); y.y = fmaxf(y.y, floor_); y.z = fmaxf(y.z, floor_); y.w = fmaxf(y.w, floor_); }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), rs_out, off, 0, 0);
                if (EPI == 1) {
                    if (n == 0) h_n += vok ? 1 : 0;
                    const float d0_ = vok ? y.x - hk[n][0] : 0.f, d1_ = vok ? y.y - hk[n][1] : 0.f, d2_ = vok ? y.z - hk[n][2] : 0.f, d3_ = vok ? y.w - hk[n][3] : 0.f;
                    hs1[n][0] += d0_; hs1[n][1] += d1_; hs1[n][2] += d2_; hs1[n][3] += d3_;
                    hs2[n][0] = fmaf(d0_, d0_, hs2[n][0]); hs2[n][1] = fmaf(d1_, d1_, hs2[n][1]);
                    hs2[n][2] = fmaf(d2_, d2_, hs2[n][2]); hs2[n][3] = fmaf(d3_, d3_, hs2[n][3]);
                }
            }
        }
    };
    // partial rows: four per patch in conv2d_roll_kernel's layout; here row wm carries this half's sums, row 2 + wm is empty
    auto flush_stats = [&]() __attribute__((always_inline)) {
        const unsigned row_base = (unsigned)(((seg * a.tiles_yb + tiy) * a.tiles_x + tix) * 4);
        float ntot = 0.f;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float cnt = (float)h_n;
                float mean = cnt > 0.f ? hk[n][r] + hs1[n][r] / cnt : 0.f;
                float m2 = cnt > 0.f ? hs2[n][r] - hs1[n][r] * hs1[n][r] / cnt : 0.f;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {  // the 16 pixel lanes of this channel quad
                    const float n_o = __shfl_xor(cnt, off), mean_o = __shfl_xor(mean, off), m2_o = __shfl_xor(m2, off);
                    const float nn = cnt + n_o;
                    const float dlt = mean_o - mean;
                    const float w_o = nn > 0.f ? n_o / nn : 0.f;
                    m2 = m2 + m2_o + dlt * dlt * cnt * w_o;
                    mean = mean + dlt * w_o;
                    cnt = nn;
                }
                ntot = cnt;
                const unsigned ch = (unsigned)(cl + 16 * n + r);
                const size_t rowp = ((size_t)g * a.cout + ch) * a.rows + row_base;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, make_float2(cnt * mean, fmaxf(m2, 0.f))), rs_part,
                                                      (lane & 15) == 0 ? (unsigned)((rowp + wm) * 8) : R_OOB, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, make_float2(0.f, 0.f)), rs_part,
                                                      (lane & 15) == 0 ? (unsigned)((rowp + 2 + wm) * 8) : R_OOB, 0, 0);
            }
        const unsigned crow = (unsigned)((size_t)g * a.rows + row_base);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ntot), rs_cnt, (lane == 0 && wn == 0) ? (crow + wm) * 4u : R_OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(0u, rs_cnt, (lane == 0 && wn == 0) ? (crow + 2 + wm) * 4u : R_OOB, 0, 0);
    };

    // ---- one stage: image p, chunk CC; 24 chains of nine MFMAs, ordered kh, tile pair, channel block, tile ----
    f32x4 tq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    auto stage = [&](auto cc_tag, int p, int buf) __attribute__((always_inline)) {
        constexpr int CC = decltype(cc_tag)::value;
        constexpr bool LAST = (CC == NCH - 1);
        constexpr int CCN = (CC + 1) % NCH;
        const unsigned char *sl = slab + buf * R_SLAB_BYTES;
        unsigned char *sn = slab + (buf ^ 1) * R_SLAB_BYTES;
        constexpr int wcur = CC * (2 * 2 * 64), wnxt = CCN * (2 * 2 * 64);
        const int pn = LAST ? p + 1 : p;
        if (CC == 0) {
            finish(p - 1, p - 1 >= d0);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float4 ah[2][3][2];
        auto load_ah = [&](int m, int kh) __attribute__((always_inline)) {
            const unsigned char *ap = sl + abase[kh & 1] + (kh * R_SX + 4 * m) * R_VB;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int q = 0; q < 2; ++q) ah[m & 1][kw][q] = *reinterpret_cast<const float4 *>(ap + kw * R_VB + 64 * q);
        };
        load_ah(0, 0);
        load_ah(1, 0);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        const int m = 2 * pr + mm;
                        const int q = ((kh * 2 + pr) * 2 + n) * 2 + mm;
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4 &prev = mm > 0 ? acc[n][m - 1] : n > 0 ? acc[n - 1][m + 1] : pr > 0 ? acc[1][1] : acc[1][3];
                        r16_chain9(tq[q & 1], ah[mm], wh[n], prev, tq[(q + 1) & 1]);
                        if (n == 1 && !(kh == 2 && pr == 1)) {  // this tile's fragments: the next pair's / next row's
                            __builtin_amdgcn_sched_barrier(0);
                            load_ah(pr == 0 ? m + 2 : mm, pr == 0 ? kh : kh + 1);
                        }
                        if (pr == 1 && mm == 1) {  // last use of this block's kh-row weights: the next row's, or the next stage's first
                            __builtin_amdgcn_sched_barrier(0);
                            load_bh(wh[n], n, kh < 2 ? wcur + (kh + 1) * 3 * TAPF4 : wnxt);
                        }
                        if (q == 1 || q == 11) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue(pn, CCN, q == 1 ? 0 : 3, q == 1 ? 3 : R_NLD);
                        }
                        if ((q >= 6 && q <= 10 && !(q & 1)) || (q >= 16 && q <= 20 && !(q & 1))) {
                            __builtin_amdgcn_sched_barrier(0);
                            commit_piece(q < 11 ? (q - 6) / 2 : 3 + (q - 16) / 2, sn);
                        }
                    }
        __builtin_amdgcn_sched_barrier(0);
        acc[1][3] += tq[1];  // 24 chains: the last one wrote tq[1]
        tq[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
    };

    issue(d0, 0, 0, R_NLD);
#pragma unroll
    for (int n = 0; n < 2; ++n) load_bh(wh[n], n, 0);
#pragma unroll
    for (int it = 0; it < R_NLD; ++it) commit_piece(it, slab);
    __syncthreads();
    int buf = 0;
    for (int p = d0; p < d1; ++p) {
        stage(std::integral_constant<int, 0>{}, p, buf); buf ^= 1;
        if (NCH == 2) { stage(std::integral_constant<int, NCH - 1>{}, p, buf); buf ^= 1; }
    }
    finish(d1 - 1, true);
    if (EPI == 1) flush_stats();
}

// ---- weight packing: [cout/32][tap 9][cin/32][n16 2][part 3][lane 64][8] bf16; element j of lane =
//      part p of src(co = 32 cg + 16 n16 + (lane & 15), ci = 32 cc + 8 (lane >> 4) + j, tap)
__global__ void __launch_bounds__(256)
conv2d_pack_r16_kernel(unsigned short *__restrict__ dst, const float *__restrict__ src, int cin, int cout,
                       long long sn, long long sk, int flip, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63;
    int r = idx >> 9;
    const int p = r % 3; r /= 3;
    const int n = r & 1; r >>= 1;
    const int nch = cin / 32;
    const int cc = r % nch; r /= nch;
    const int tap = r % 9;
    const int cg = r / 9;
    const int co = cg * 32 + n * 16 + (lane & 15);
    const int ci = cc * 32 + 8 * (lane >> 4) + j;
    const float x = src[co * sn + ci * sk + (flip ? 8 - tap : tap)];
    dst[idx] = az_split3_part(x, p);
}

// f16x3 image: [cout/32][tap 9][cin/32][n16 2][part 2][lane 64][8] fp16 of w * 2^k (k from the amax array of w)
__global__ void __launch_bounds__(256)
conv2d_pack_r16_f16_kernel(unsigned short *__restrict__ dst, const float *__restrict__ src, const float *__restrict__ amax,
                           int cin, int cout, long long sn, long long sk, int flip, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const float scale = az_pow2(az_f16_scale_exp(az_amax_read(amax)));  // (before the early exit: a wave-wide read)
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63;
    int r = idx >> 9;
    const int p = r & 1; r >>= 1;
    const int n = r & 1; r >>= 1;
    const int nch = cin / 32;
    const int cc = r % nch; r /= nch;
    const int tap = r % 9;
    const int cg = r / 9;
    const int co = cg * 32 + n * 16 + (lane & 15);
    const int ci = cc * 32 + 8 * (lane >> 4) + j;
    dst[idx] = az_split2_f16_part(src[co * sn + ci * sk + (flip ? 8 - tap : tap)] * scale, p);
}

static bool c2r_shape_ok(int cin, int cout) { return (cin == 32 || cin == 64) && (cout == 32 || cout == 64); }

extern "C" long long az_conv2d_roll_packed_floats(int cin, int cout) {
    if (!c2r_shape_ok(cin, cout)) return AZ_EUNSUPPORTED;
    return 9LL * cin * cout * 3 / 2;
}

extern "C" int az_conv2d_roll_pack(float *packed, const float *w, int cin, int cout, long long stride_out,
                                   long long stride_in, int flip, void *stream) {
    AZ_REQUIRE_PTR(packed); AZ_REQUIRE_PTR(w);
    if (!c2r_shape_ok(cin, cout)) return AZ_EUNSUPPORTED;
    const int total = 9 * cin * cout * 3;
    hipLaunchKernelGGL(conv2d_pack_r16_kernel, dim3((total + 255) / 256), dim3(256), 0, az_stream(stream),
                       reinterpret_cast<unsigned short *>(packed), w, cin, cout, stride_out, stride_in, flip, total);
    return az_launch_status();
}

// image segments per group: one round of workgroups over the chip's 512 slots if the patches allow it
static void c2r_segments(const C2RArgs &a, int &nseg, int &seg_len) {
    // (64 output channels: one workgroup does both channel groups)
    az_c2r_segments((long long)a.G * a.tiles_yb * a.tiles_x, a.N, nseg, seg_len);
}

static int c2r_setup(C2RArgs &a, int groups, int B, int H, int W, int cin, int cout) {
    if (!c2r_shape_ok(cin, cout)) return AZ_EUNSUPPORTED;
    if (groups <= 0 || B <= 0 || B % groups || H <= 0 || W <= 0) return AZ_EINVAL;
    a.G = groups; a.N = B / groups; a.H = H; a.W = W; a.cout = cout;
    a.tiles_yb = (H + R_TY - 1) / R_TY; a.tiles_x = (W + R_TX - 1) / R_TX;
    c2r_segments(a, a.nseg, a.seg_len);
    a.rows = a.nseg * a.tiles_yb * a.tiles_x * 4;
    // one group of one tensor is addressed through a 32-bit buffer offset
    if ((long long)a.N * H * W * 64 * 4 >= 0xffffff00LL) return AZ_EUNSUPPORTED;
    return AZ_OK;
}

extern "C" long long az_conv2d_roll_stats_rows(int groups, int B, int H, int W, int cin, int cout) {
    C2RArgs a{};
    if (int e = c2r_setup(a, groups, B, H, W, cin, cout)) return e;
    return a.rows;
}

// AZ_CONV2D_ROLL_NT4=0: 64 output channels as two channel groups in the grid (two workgroups per patch, each staging
// the slab) instead of four N tiles per wave
static bool c2r_nt4() {
    const int on = az_options().conv2d_roll_nt4;
    return on != 0;
}
template <int EPI, int AR = 0>
static int c2r_launch(const C2RArgs &a, int cin, hipStream_t s) {
    const bool nt4 = a.cout == 64 && c2r_nt4();
    const long long blocks = (long long)a.G * a.nseg * a.tiles_yb * a.tiles_x * (nt4 ? 1 : a.cout / 32);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    const dim3 grid((unsigned)blocks), blk(256);
    if (nt4 && AR == 1 && az_options().conv2d_roll_h) {  // f16x3, 64 output channels: the 3-D kernel's wave layout
        if (cin == 32) hipLaunchKernelGGL((conv2d_roll64_kernel<32, EPI>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((conv2d_roll64_kernel<64, EPI>), grid, blk, 0, s, a);
    } else if (nt4) {
        if (cin == 32) hipLaunchKernelGGL((conv2d_roll_kernel<32, EPI, 4, AR>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((conv2d_roll_kernel<64, EPI, 4, AR>), grid, blk, 0, s, a);
    } else {
        if (cin == 32) hipLaunchKernelGGL((conv2d_roll_kernel<32, EPI, 2, AR>), grid, blk, 0, s, a);
        else hipLaunchKernelGGL((conv2d_roll_kernel<64, EPI, 2, AR>), grid, blk, 0, s, a);
    }
    return az_launch_status();
}

// out = relu?( conv(in) * scale[c] + shift[c] + residual ): dense [B,H,W,cin] -> [B,H,W,cout]; scale / shift / residual optional
extern "C" int az_conv2d_roll_fwd(float *out, const float *in, const float *packed, const float *scale, const float *shift,
                                  const float *residual, int relu, int B, int H, int W, int cin, int cout, void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(in); AZ_REQUIRE_PTR(packed);
    C2RArgs a{};
    if (int e = c2r_setup(a, 1, B, H, W, cin, cout)) return e;
    a.in = in; a.wp = packed; a.out = out; a.scale = scale; a.shift = shift; a.res = residual; a.relu = relu;
    a.out_cs = cout; a.res_cs = cout;
    return residual ? c2r_launch<2>(a, cin, az_stream(stream)) : c2r_launch<0>(a, cin, az_stream(stream));
}

// out = conv(in) (raw) + BatchNorm partials [groups][cout][rows][2], counts [groups][rows], rows = az_conv2d_roll_stats_rows
extern "C" int az_conv2d_roll_fwd_stats(float *out, float *partials, float *counts, const float *in, const float *packed,
                                        int groups, int B, int H, int W, int cin, int cout, void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(in); AZ_REQUIRE_PTR(packed); AZ_REQUIRE_PTR(partials); AZ_REQUIRE_PTR(counts);
    C2RArgs a{};
    if (int e = c2r_setup(a, groups, B, H, W, cin, cout)) return e;
    a.in = in; a.wp = packed; a.out = out; a.part = partials; a.cnt = counts;
    a.out_cs = cout; a.res_cs = cout;
    return c2r_launch<1>(a, cin, az_stream(stream));
}

// ---- f16x3 (include/azhip.h): the same three entry points with the operands' amax arrays ---------------------------
extern "C" int az_conv2d_roll_pack_f16(float *packed, const float *w, const float *w_amax, int cin, int cout,
                                       long long stride_out, long long stride_in, int flip, void *stream) {
    AZ_REQUIRE_PTR(packed); AZ_REQUIRE_PTR(w); AZ_REQUIRE_PTR(w_amax);
    if (!c2r_shape_ok(cin, cout)) return AZ_EUNSUPPORTED;
    const int total = 9 * cin * cout * 2;
    hipLaunchKernelGGL(conv2d_pack_r16_f16_kernel, dim3((total + 255) / 256), dim3(256), 0, az_stream(stream),
                       reinterpret_cast<unsigned short *>(packed), w, w_amax, cin, cout, stride_out, stride_in, flip, total);
    return az_launch_status();
}

extern "C" int az_conv2d_roll_fwd_f16(float *out, const float *in, const float *packed, const float *in_amax,
                                      const float *w_amax, const float *scale, const float *shift, const float *residual,
                                      int relu, int B, int H, int W, int cin, int cout, void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(in); AZ_REQUIRE_PTR(packed); AZ_REQUIRE_PTR(in_amax); AZ_REQUIRE_PTR(w_amax);
    C2RArgs a{};
    if (int e = c2r_setup(a, 1, B, H, W, cin, cout)) return e;
    a.in = in; a.wp = packed; a.out = out; a.scale = scale; a.shift = shift; a.res = residual; a.relu = relu;
    a.out_cs = cout; a.res_cs = cout; a.in_amax = in_amax; a.w_amax = w_amax;
    return residual ? c2r_launch<2, 1>(a, cin, az_stream(stream)) : c2r_launch<0, 1>(a, cin, az_stream(stream));
}

extern "C" int az_conv2d_roll_fwd_stats_f16(float *out, float *partials, float *counts, const float *in,
                                            const float *packed, const float *in_amax, const float *w_amax, int groups,
                                            int B, int H, int W, int cin, int cout, void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(in); AZ_REQUIRE_PTR(packed); AZ_REQUIRE_PTR(partials); AZ_REQUIRE_PTR(counts);
    AZ_REQUIRE_PTR(in_amax); AZ_REQUIRE_PTR(w_amax);
    C2RArgs a{};
    if (int e = c2r_setup(a, groups, B, H, W, cin, cout)) return e;
    a.in = in; a.wp = packed; a.out = out; a.part = partials; a.cnt = counts;
    a.out_cs = cout; a.res_cs = cout; a.in_amax = in_amax; a.w_amax = w_amax;
    return c2r_launch<1, 1>(a, cin, az_stream(stream));
}
