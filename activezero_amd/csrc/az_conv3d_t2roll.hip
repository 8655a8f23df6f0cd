// K5'' (f16x3, 64 -> 32 channels) -- the stride-2 TRANSPOSED 3x3x3 convolution on the depth-rolling machinery of
// az_conv3d_roll.hip: hourglass conv6 (ConvTranspose3d 64 -> 32, V1 -> V0; nets/psmnet/psmnet_3.py:34-58) and the input
// gradient of its stride-2 convolution conv1 (32 -> 64).
//
//   out[2t + p] (p = output parity per dimension, t = coarse index) = sum over the taps
//       p = 0:  k = 1 at coarse t            p = 1:  k = 2 at coarse t,  k = 0 at coarse t + 1
//
// az_conv3d_t2.hip gives a workgroup a 4x16 coarse patch of ONE coarse plane: both planes a phase reads are staged per
// (plane offset, 16-channel chunk) -- eight small stages of 24 MFMAs per wave between barriers, every coarse plane
// staged twice: 0.58 ms (conv6 forward) / 0.70 ms (conv1 input gradient) for 86.6 GFLOP, 0.15-0.18 of the f16x3 roofline.
// Here a workgroup of EIGHT waves owns an 8x16 coarse patch and WALKS the coarse depth:
//   * coarse plane z is fetched, split into its fp16 pair and written to LDS once (two 32-channel chunk slabs of
//     az_conv3d_roll.hip's layout and swizzle, double-buffered: 138 KB, one workgroup per CU) and meets ALL 27 taps in one
//     stage: kd = 1 completes fine plane 2z, kd = 0 completes fine plane 2z - 1 (begun by plane z - 1's kd = 2), kd = 2
//     begins fine plane 2z + 1 -- three sets of four (ph, pw) phase accumulators, the last carried to the next stage;
//   * a wave = 16 output channels x 32 coarse voxels (two 4x4 tiles): 12 phases x 2 tiles x 4 = 96 accumulator registers;
//     operand roles swapped as in r16_chain9 (weights = A operand), so a lane holds four channels of one voxel: 16-byte stores;
//   * a chain = the taps of one (kd group, phase, coarse row offset oh): one or two taps (ow) x three MFMAs per tile, summed
//     from zero, one VALU add per chain and tile into the accumulator, placed under the next chain's first MFMAs; the voxel
//     fragments of a row offset (2 tiles x 2 ow x 2 parts) are read from LDS once per chunk and serve its nine chains;
//     weights come one tap ahead through a buffer resource;
//   * a stage is straight-line code: validity (patch overhang, segment ends, the zero plane behind the last coarse plane)
//     is a buffer bound, never a branch; the next plane is requested at the top of the stage and split / written in its
//     second half; one barrier per stage.
// Measured (B = 4, V1 -> V0, alone): 0.46 ms for either use (az_conv3d_t2.hip: 0.58 / 0.72; 0.49-0.51 before the closing stage
// of a segment was cut down to its nine kd = 0 taps).  Counter pass (profiles/r04_pmc_*_b4.json): 0.27 GB read + 0.80 GB written
// for 0.20 + 0.80 algorithmic, clock 2.19 GHz, MFMA pipe 0.29-0.30 busy.  Timing-only builds (T2R_ABL, 0.46 shipped on that box):
// without the output stores 0.36, with only the first two taps' weight loads per stage 0.31, without slab staging 0.41, without
// fragment reads 0.46; stores + weights + staging all off: 0.163 ms -- the matrix work itself at 0.8 of the pipe.  So the four
// costs ADD: matrix 0.16 + weight fragments 0.15 + output stores 0.10-0.13 + staging 0.05.  The weight term is L1 bandwidth:
// a wave holds two tiles, so every 1 KB fragment feeds six MFMAs (twelve in the stride-1 kernel) and the eight waves of a CU pull
// 864 KB per stage through a 64 B/clk TCP -- 13.5 k cycles beside 10.4 k matrix cycles.  Requesting weights two taps ahead, or
// the first 3 / 6 / 9 taps of a stage in front of the previous stage's stores, changed nothing.  What would: four tiles per wave
// (needs the twelve phase accumulators of 4 tiles: 192 registers) or the weights of a chunk shared through LDS (108 KB beside
// 138 KB of slabs: needs the slabs in a 128-byte voxel layout with a new conflict-free swizzle).
// BatchNorm partials (EPI 1): per-lane shifted running sums of its four channels over everything it stores, merged once
// after the walk: one row per (batch, depth segment, patch, quarter of the patch).
#include <type_traits>

#include "az_conv3d_args.h"
#include "az_roll_common.h"
#include "az_options.h"
#include "az_launch_math.h"

#define T2R_NT 512
#define T2R_NLD 6                        // pieces per thread and plane: 2 chunks x 3 (64 voxels per piece index)
#define T2R_LDS (4 * R_SLAB_BYTES)       // [plane buffer][chunk]
#define T2R_TAPF4 (2 * 2 * 2 * 64)       // float4 per tap of the packed image [tap][cc(2)][n16(2)][part(2)][lane]
#define T2R_CCF4 (2 * 2 * 64)
#ifndef T2R_ABL
#define T2R_ABL 0  // timing-only ablations: 1 no output stores / residual loads, 2 the weights of a stage's first two taps only, 4 no slab staging, 8 no fragment reads
#endif

__host__ __device__ constexpr int t2r_k(int p, int o) { return p == 0 ? 1 : (o == 0 ? 2 : 0); }
// the 27 taps of a stage in chain order: row offset oh, kd group g (0: kd = 0 into the carried set, 1: kd = 1, 2: kd = 2
// into the set the next stage carries), column parity pw, row parity ph >= oh, column offset ow <= pw
struct T2rEnt { int oh, g, ph, pw, ow, tap, first, last, nxc; };  // nxc: the next entry with g == 0 (27: none left in this chunk)
struct T2rTab { T2rEnt e[27]; };
__host__ __device__ constexpr T2rTab t2r_make() {
    T2rTab t{};
    int n = 0;
    for (int oh = 0; oh < 2; ++oh)
        for (int g = 0; g < 3; ++g)
            for (int pw = 0; pw < 2; ++pw)
                for (int ph = oh; ph < 2; ++ph)
                    for (int ow = 0; ow <= pw; ++ow) {
                        const int kd = g == 0 ? 0 : g == 1 ? 1 : 2;
                        t.e[n] = T2rEnt{oh, g, ph, pw, ow, (kd * 3 + t2r_k(ph, oh)) * 3 + t2r_k(pw, ow), ow == 0, ow == pw, 27};
                        ++n;
                    }
    for (int i = 0; i < 27; ++i)
        for (int j = 26; j > i; --j)
            if (t.e[j].g == 0) t.e[i].nxc = j;
    return t;
}

// EPI: 0 = y = relu?(acc * scale + shift) (+ residual when given), 1 = raw output + BatchNorm partials
// PS: the input is a pre-split tensor (az_roll_common.h): the input gradient of hourglass conv1 from its BatchNorm backward
template <int EPI, bool PS = false>
__global__ void __launch_bounds__(T2R_NT, 2)
conv3d_t2roll_kernel(const ConvArgs a) {
    constexpr T2rTab TAB = t2r_make();
    __shared__ __attribute__((aligned(16))) unsigned char slab[T2R_LDS + 64];  // + a sink for the lanes of the partial piece
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv & 1, wm = (wv >> 1) & 1, wx = wv >> 2;  // half of the channels, 4-row half, 8-column half of the patch

    const int lin = a.map_mode >= 1 ? az_xcd_map(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tyb = (a.tiles_y + 1) >> 1;
    int tix, tiy, seg, b;
    az_roll_decode(lin, a.tiles_x, tyb, a.nseg, tix, tiy, seg, b);
    const int c0 = seg * a.seg_len, c1 = min(c0 + a.seg_len, a.Di);  // coarse planes [c0, c1)
    const int ty0 = tiy * R_TY, tx0 = tix * R_TX;

    const unsigned in_bytes = (unsigned)a.Di * a.Hi * a.Wi * 64u * 4u, out_bytes = (unsigned)a.Do * a.Ho * a.Wo * 32u * 4u;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in) + (size_t)b * (in_bytes / 4), 0, in_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * (out_bytes / 4), 0, out_bytes, 0x00020000);
    const bool has_res = EPI == 0 && a.res != nullptr;
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(has_res ? a.res : a.out) + (size_t)b * (out_bytes / 4), 0, has_res ? out_bytes : 0u, 0x00020000);
    const auto rs_part = __builtin_amdgcn_make_buffer_rsrc(EPI == 1 ? a.part : a.out, 0, EPI == 1 ? (unsigned)(a.ntiles * 32 * 2 * 4) : 0u, 0x00020000);
    const auto rs_cnt = __builtin_amdgcn_make_buffer_rsrc(EPI == 1 ? a.cnt : a.out, 0, EPI == 1 ? (unsigned)(a.ntiles * 4) : 0u, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp), 0, 27u * T2R_TAPF4 * 16u, 0x00020000);

    const int ki = az_f16_scale_exp(az_amax_read(a.in_amax));
    const int kw_ = az_f16_scale_exp(az_amax_read(a.w_amax));
    const float in_scale = az_pow2(ki);
    const int out_exp = -(ki + kw_);

    // accumulators: [set][phase ph * 2 + pw][tile]; set 0: carried (fine plane 2z - 1), 1: fine plane 2z, 2: next (2z + 1)
    f32x4 acc[3][4][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[s][p][0] = acc[s][p][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging: coarse plane z = 10 x 18 voxels (rows ty0.., columns tx0..: the halo is on the high side only) x 64
    //      channels.  piece it: chunk it / 3, voxel (tid >> 3) + 64 (it % 3), channels 32 cc + 4 (tid & 7) .. ---------------
    u32x4 pre[T2R_NLD];
    auto issue = [&](int z) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < T2R_NLD; ++it) {
            const int cc = it / 3, vox = (tid >> 3) + 64 * (it % 3);
            const int sy = vox / R_SX, sx = vox - sy * R_SX;
            const int ih = ty0 + sy, iw = tx0 + sx;
            const bool ok = vox < R_SY * R_SX && ih < a.Hi && iw < a.Wi && (unsigned)z < (unsigned)a.Di;
            const unsigned off = (unsigned)((z * a.Hi + ih) * a.Wi + iw) * 256u + (unsigned)cc * 128u + (unsigned)(tid & 7) * 16u;
            pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? off : R_OOB, 0, 0);
        }
    };
    auto commit_piece = [&](int it, unsigned char *plane) __attribute__((always_inline)) {  // plane: the two chunk slabs of one buffer
        const int cc = it / 3, vox = (tid >> 3) + 64 * (it % 3);
        const int sy = vox / R_SX, sx = vox - sy * R_SX;
        const int j = tid & 7;
        unsigned char *dst = plane + cc * R_SLAB_BYTES + (sy * R_SX + sx) * R_VB + ((((j >> 1) ^ ((sy & 1) << 1))) << 4) + (j & 1) * 8;
        unsigned char *dst2 = dst + 64;
        if (it % 3 == 2) {  // (voxels 128..191: only 180 exist -- the others write to the sink; no branch)
            const bool live = vox < R_SY * R_SX;
            dst = live ? dst : slab + T2R_LDS + (tid & 7) * 8;
            dst2 = live ? dst2 : dst;
        }
        uint2 hi, lo;
        az_stage_f16x4<PS>(pre[it], in_scale, hi, lo);
        *reinterpret_cast<uint2 *>(dst) = hi;
        *reinterpret_cast<uint2 *>(dst2) = lo;
    };

    // voxel fragment (B operand): lane -> voxel (row (lane >> 2) & 3, x lane & 3) of a 4x4 tile, channel octet lane >> 4;
    // slab row = 4 wm + tile row + oh: the octet swizzle follows the row's parity
    const int trow = (lane >> 2) & 3, tcol = lane & 3, oct = lane >> 4;
    unsigned xbase[2];
    xbase[0] = ((4 * wm + trow) * R_SX + tcol) * R_VB + ((oct ^ ((trow & 1) << 1)) << 4);
    xbase[1] = ((4 * wm + trow + 1) * R_SX + tcol) * R_VB + ((oct ^ (((trow + 1) & 1) << 1)) << 4);
    const unsigned wlane = (unsigned)(wn * 2 * 64 + lane) * 16u;

    // epilogue constants: this lane's four channels 16 wn + 4 (lane >> 4) + r
    const int cqh = wn * 16 + 4 * (lane >> 4);
    float4 sch = make_float4(1.f, 1.f, 1.f, 1.f), sfh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == 0) {
        if (a.scale) sch = *reinterpret_cast<const float4 *>(a.scale + cqh);
        if (a.shift) sfh = *reinterpret_cast<const float4 *>(a.shift + cqh);
    }
    sch.x = ldexpf(sch.x, out_exp); sch.y = ldexpf(sch.y, out_exp); sch.z = ldexpf(sch.z, out_exp); sch.w = ldexpf(sch.w, out_exp);
    const float osc = ldexpf(1.f, out_exp);  // (EPI 1: the only scale -- a scalar register instead of eight vector ones)
    const float floor_ = (EPI == 0 && a.relu) ? 0.f : -__builtin_inff();
    float hk[4] = {0.f, 0.f, 0.f, 0.f}, hs1[4] = {0.f, 0.f, 0.f, 0.f}, hs2[4] = {0.f, 0.f, 0.f, 0.f};
    int h_n = 0;
    bool h_first = true;

    // store the four (ph, pw) phases of accumulator set `s` as fine plane f (ok: the plane belongs to this segment)
    const int cy = ty0 + 4 * wm + trow;  // this lane's coarse row; its coarse columns: tx0 + 8 wx + 4 m + tcol
    auto finish = [&](int s, int f, bool ok) __attribute__((always_inline)) {
        const bool plane_ok = ok && (unsigned)f < (unsigned)a.Do;
        if (EPI == 1) {  // the shift of the running sums: the first value this lane sees (any value near the data serves)
            const f32x4 c = acc[s][0][0];
            const bool take = h_first && plane_ok;
            hk[0] = take ? c[0] * osc : hk[0]; hk[1] = take ? c[1] * osc : hk[1];
            hk[2] = take ? c[2] * osc : hk[2]; hk[3] = take ? c[3] * osc : hk[3];
            h_first = h_first && !plane_ok;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ph = p >> 1, pw = p & 1;
            const int fy = 2 * cy + ph;
            const bool row_ok = plane_ok && fy < a.Ho;
            const unsigned row_off = (unsigned)((f * a.Ho + fy) * a.Wo) * 128u + (unsigned)cqh * 4u;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int fx = 2 * (tx0 + 8 * wx + 4 * m + tcol) + pw;
                const bool vok = row_ok && fx < a.Wo;
                const unsigned off = vok ? row_off + (unsigned)fx * 128u : R_OOB;
                const f32x4 c = acc[s][p][m];
                float4 y = EPI == 1 ? make_float4(c[0] * osc, c[1] * osc, c[2] * osc, c[3] * osc)
                                    : make_float4(fmaf(c[0], sch.x, sfh.x), fmaf(c[1], sch.y, sfh.y), fmaf(c[2], sch.z, sfh.z), fmaf(c[3], sch.w, sfh.w));
                if (EPI == 0) {
                    const float4 rr = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (T2R_ABL & 1) ? R_OOB : off, 0, 0));
                    y.x += rr.x; y.y += rr.y; y.z += rr.z; y.w += rr.w;
                    y.x = fmaxf(y.x, floor_); y.y = fmaxf(y.y, floor_); y.z = fmaxf(y.z, floor_); y.w = fmaxf(y.w, floor_);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), rs_out, (T2R_ABL & 1) ? R_OOB : off, 0, 0);
                if (EPI == 1) {
                    h_n += vok ? 1 : 0;
                    const float d0 = vok ? y.x - hk[0] : 0.f, d1 = vok ? y.y - hk[1] : 0.f, d2 = vok ? y.z - hk[2] : 0.f, d3 = vok ? y.w - hk[3] : 0.f;
                    hs1[0] += d0; hs1[1] += d1; hs1[2] += d2; hs1[3] += d3;
                    hs2[0] = fmaf(d0, d0, hs2[0]); hs2[1] = fmaf(d1, d1, hs2[1]); hs2[2] = fmaf(d2, d2, hs2[2]); hs2[3] = fmaf(d3, d3, hs2[3]);
                }
            }
        }
    };
    auto rotate = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                acc[0][p][m] = acc[2][p][m];
                acc[1][p][m] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc[2][p][m] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    };

    float4 wk[2][2];  // [tap parity][part]: the weights of the tap being multiplied and of the next one (two taps ahead: no gain)
    auto load_w = [&](float4 (&w)[2], int f4) __attribute__((always_inline)) {  // f4: float4 index of the tap's (chunk) block in the packed image (static)
#pragma unroll
        for (int q = 0; q < 2; ++q)
            w[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, f4 * 16 + q * 1024, 0));
    };
    f32x4 tq[2][2] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};

    // ---- one stage: coarse plane z in plane buffer `buf` ------------------------------------------------------------------
    // CLOSE: the last stage of a segment (plane c1) only completes fine plane 2 c1 - 1: the nine kd = 0 taps, no next plane
    auto stage = [&](auto close_tag, int z, int buf) __attribute__((always_inline)) {
        constexpr bool CLOSE = decltype(close_tag)::value != 0;
        const unsigned char *pl = slab + buf * (2 * R_SLAB_BYTES);
        unsigned char *pn = slab + (buf ^ 1) * (2 * R_SLAB_BYTES);
        // the planes completed by the stage before: fine 2(z-1) - 1 (set 0) and 2(z-1) (set 1)
        finish(0, 2 * z - 3, z - 2 >= c0);
        finish(1, 2 * z - 2, z - 1 >= c0);
        rotate();
        __builtin_amdgcn_sched_barrier(0);
        if (!(T2R_ABL & 4) && !CLOSE) issue(z + 1);
        float4 xf[2][2][2];  // [tile][ow][part] of the current (chunk, oh)
        int chain = 0;       // (static after unrolling)
        int k = 0;           // taps multiplied so far: the parity of the weight buffers
        int oh_have = -1;    // row offset of the fragments in xf
        int pend_g = -1, pend_p = 0;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
            for (int i = 0; i < 27; ++i) {
                constexpr T2rTab T = TAB;
                const T2rEnt e = T.e[i];
                if (CLOSE && e.g != 0) continue;
                __builtin_amdgcn_sched_barrier(0);
                if (oh_have != cc * 2 + e.oh) {  // the fragments of this chunk's row offset oh
                    oh_have = cc * 2 + e.oh;
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int ow = 0; ow < 2; ++ow) {
                            const unsigned char *xp = pl + cc * R_SLAB_BYTES + xbase[e.oh] + (4 * (2 * wx + m) + ow) * R_VB;
                            xf[m][ow][0] = *reinterpret_cast<const float4 *>(xp);
                            xf[m][ow][1] = *reinterpret_cast<const float4 *>(xp + 64);
                        }
                }
                // the next tap's weights (the next chunk's / next stage's first tap at the end)
                {
                    const int nx = CLOSE ? e.nxc : i + 1;                      // next tap of this chunk (27: the chunk is done)
                    const int ni = nx < 27 ? nx : 0, ncc = nx < 27 ? cc : cc + 1;  // (entry 0 has g = 0: first in both forms)
                    if (!(CLOSE && ncc == 2) && !((T2R_ABL & 2) && k > 1)) load_w(wk[(k + 1) & 1], T.e[ni].tap * T2R_TAPF4 + (ncc & 1) * T2R_CCF4);
                }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 &t0 = tq[chain & 1][0], &t1 = tq[chain & 1][1];
                const float4(&w)[2] = wk[k & 1];
                ++k;
                if (e.first) { t0 = f32x4{0.f, 0.f, 0.f, 0.f}; t1 = f32x4{0.f, 0.f, 0.f, 0.f}; }
                t0 = R_MH(t0, w[0], xf[0][e.ow][0]);
                t1 = R_MH(t1, w[0], xf[1][e.ow][0]);
                if (e.first && pend_g >= 0) {  // the chain before: its sums into their accumulators, under this chain's MFMAs
                    acc[pend_g][pend_p][0] += tq[(chain + 1) & 1][0];
                    acc[pend_g][pend_p][1] += tq[(chain + 1) & 1][1];
                }
                t0 = R_MH(t0, w[0], xf[0][e.ow][1]);
                t1 = R_MH(t1, w[0], xf[1][e.ow][1]);
                t0 = R_MH(t0, w[1], xf[0][e.ow][0]);
                t1 = R_MH(t1, w[1], xf[1][e.ow][0]);
                if (e.last) { pend_g = e.g; pend_p = e.ph * 2 + e.pw; ++chain; }
                // the next plane: one piece behind every fourth tap of the second chunk
                if (!CLOSE && cc == 1 && i >= 3 && i < 27 && (i - 3) % 4 == 0 && (i - 3) / 4 < T2R_NLD) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(T2R_ABL & 4)) commit_piece((i - 3) / 4, pn);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[pend_g][pend_p][0] += tq[(chain + 1) & 1][0];
        acc[pend_g][pend_p][1] += tq[(chain + 1) & 1][1];
        __syncthreads();
    };

    // ---- the walk: stages z = c0 .. c1 (plane c1 only closes fine plane 2 c1 - 1; behind the last coarse plane it is zeros)
    issue(c0);
    load_w(wk[0], TAB.e[0].tap * T2R_TAPF4);
#pragma unroll
    for (int it = 0; it < T2R_NLD; ++it) commit_piece(it, slab);
    __syncthreads();
    int buf = 0;
    for (int z = c0; z < c1; ++z) {
        stage(std::integral_constant<int, 0>{}, z, buf);
        buf ^= 1;
    }
    stage(std::integral_constant<int, 1>{}, c1, buf);
    // after stage c1: set 0 holds fine plane 2 c1 - 1 (complete); sets 1 / 2 belong to the next segment's planes
    finish(0, 2 * c1 - 1, c1 - 1 >= c0);

    if (EPI == 1) {
        const unsigned tile_id = (unsigned)((((b * a.nseg + seg) * tyb + tiy) * a.tiles_x + tix) * 4 + wm * 2 + wx);
        float ntot = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float n = (float)h_n;
            float mean = n > 0.f ? hk[r] + hs1[r] / n : 0.f;
            float m2 = n > 0.f ? hs2[r] - hs1[r] * hs1[r] / n : 0.f;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) {  // the 16 voxel lanes of this channel quad
                const float n_o = __shfl_xor(n, off), mean_o = __shfl_xor(mean, off), m2_o = __shfl_xor(m2, off);
                const float nn = n + n_o;
                const float dlt = mean_o - mean;
                const float w_o = nn > 0.f ? n_o / nn : 0.f;
                m2 = m2 + m2_o + dlt * dlt * n * w_o;
                mean = mean + dlt * w_o;
                n = nn;
            }
            ntot = n;
            const unsigned ch = (unsigned)(cqh + r);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, make_float2(n * mean, fmaxf(m2, 0.f))), rs_part,
                                                  (lane & 15) == 0 ? (unsigned)(((size_t)ch * a.ntiles + tile_id) * 8) : R_OOB, 0, 0);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ntot), rs_cnt, (lane == 0 && wn == 0) ? tile_id * 4u : R_OOB, 0, 0);
    }
}

// depth segments: one workgroup per CU (256 slots)
static void t2roll_segments(const ConvArgs &a, int &nseg, int &seg_len) {
    az_t2roll_segments((long long)a.B * ((a.tiles_y + 1) / 2) * a.tiles_x, a.Di, nseg, seg_len);  // (az_launch_math.h: swept on the CPU)
}

long long az_conv3d_t2roll_stats_tiles(const ConvArgs &a) {
    int nseg, seg_len;
    t2roll_segments(a, nseg, seg_len);
    return (long long)a.B * nseg * ((a.tiles_y + 1) / 2) * a.tiles_x * 4;
}

// f16x3, cin = 64, cout = 32, MODE 2 (tiles_y / tiles_x: 4-row / 16-column tiles of the COARSE plane); weights packed by
// az_conv3d_pack_r16_f16(cin = 64, cout = 32)
int az_conv3d_t2roll_launch(ConvArgs a, int epi, hipStream_t s) {
    if (!a.in_amax || !a.w_amax) return AZ_ENULL;
    t2roll_segments(a, a.nseg, a.seg_len);
    const long long blocks = (long long)a.B * a.nseg * ((a.tiles_y + 1) / 2) * a.tiles_x;
    if (epi) a.ntiles = blocks * 4;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    if (!az_fits_buffer_offset((long long)a.Do * a.Ho * a.Wo * 32 * 4) || !az_fits_buffer_offset((long long)a.Di * a.Hi * a.Wi * 64 * 4) ||
        a.ntiles * 256 >= 0xffffff00LL)
        return AZ_EUNSUPPORTED;
    if (a.in_split && epi) return AZ_EUNSUPPORTED;  // (pre-split inputs are gradients: no BatchNorm-partials epilogue)
    if (epi) hipLaunchKernelGGL((conv3d_t2roll_kernel<1>), dim3((unsigned)blocks), dim3(T2R_NT), 0, s, a);
    else if (a.in_split) hipLaunchKernelGGL((conv3d_t2roll_kernel<0, true>), dim3((unsigned)blocks), dim3(T2R_NT), 0, s, a);
    else hipLaunchKernelGGL((conv3d_t2roll_kernel<0>), dim3((unsigned)blocks), dim3(T2R_NT), 0, s, a);
    return az_launch_status();
}
