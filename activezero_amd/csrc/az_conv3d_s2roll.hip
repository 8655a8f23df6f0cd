// K5''' (f16x3, 32 -> 64 channels) -- the stride-2 3x3x3 convolution on depth-rolling workgroups: hourglass conv1
// (nets/psmnet/psmnet_3.py:15-22, V0 -> V1) and the input gradient of its transposed twin conv6 (ConvTranspose3d 64 -> 32,
// nets/psmnet/psmnet_3.py:34-58).  Until round 5 these ran on az_conv3d.hip's one-wave gather kernel: every fine plane
// staged 1.5 times (kd = 0 of one output plane, kd = 2 of another) with a 9 x 17 halo slab per 4 x 8 outputs, one wave
// per SIMD with nothing to hide its staging behind -- 0.46 ms (conv1 forward) / 0.55 ms (conv6 input gradient) for
// 86.6 GFLOP, 0.19 of the f16x3 roofline and 2.3 TB/s of HBM traffic that is all algorithmic (0.80 GB in, 0.20 GB out).
//
//   out[t] = sum_k w[k] in[2t - 1 + k]      (pad 1, per dimension)
//
// A workgroup of FOUR waves owns an 8 x 16 patch of the COARSE output and walks the FINE depth:
//   * fine plane z (17 x 33 voxels x 32 channels with its halo) is fetched, split into its fp16 pair and written to LDS
//     ONCE -- as four PARITY SUB-SLABS (row parity x column parity), each a dense 9 x 17 grid, so that the stride-2 reads of
//     a tap become unit-stride reads of one sub-slab: tap k reads parity (k != 1) at offset (k == 2).  78 KB per workgroup,
//     two workgroups per CU: while one converts and writes its next plane the other multiplies (no double buffer, no
//     prefetch registers -- the second workgroup IS the overlap);
//   * an even plane 2t meets the nine kd = 1 taps of output plane t; an odd plane 2t + 1 the nine kd = 2 taps of plane t
//     and the nine kd = 0 taps of plane t + 1 -- two accumulator sets, the voxel fragments of an odd plane's tap position
//     read from LDS once for both;
//   * a wave = 16 output channels x the whole patch (eight 4 x 4 tiles): a 2 KB weight fragment pair feeds 24 MFMAs (six in
//     az_conv3d_t2roll.hip, whose weight reads through L1 cost as much as its matrix work), the 16 KB of voxel fragments
//     of a tap are read by each of the four waves (256 B/clk: 0.65 of the matrix cycles, 0.43 on odd planes);
//   * operand roles as r16_chain9 (weights = A operand): a lane holds four consecutive channels of one voxel, 16-byte stores.
// LDS layout of a sub-slab voxel (128 B): [part p ^ g][octet o ^ s] 16-byte pieces, g = bit 1 of the column, s = 2 (row & 1):
// the sixteen lanes of every ds_read_b128 group then hit sixteen different 16-byte slots (voxel pitch 8 slots, row pitch
// 17 voxels = 8 mod 16: without g the four x-adjacent voxels of a tile row share two slot quads).
// BatchNorm partials (EPI 1): per-lane shifted running sums of its four channels over everything it stores, merged once
// after the walk: one row per workgroup.
// Measured (B = 4, V0 -> V1, alone): 0.34 ms forward with partials, 0.37 ms input gradient from a pre-split operand (gather
// kernel: 0.59 / 0.55); counter passes 1.01 GB read + 0.20 GB written, clock 2.03 GHz, MFMA pipe 0.42.  Timing-only builds
// (S2R_ABL, profiles/r05v_s2roll_ablations_corrected.txt): without the staging loads 0.226, without fragment reads 0.316,
// without weight loads 0.320, all three off 0.222 -- against 0.122 ms of matrix time.  The plane loads are the exposed term:
// a workgroup cannot request plane z + 1 while it multiplies plane z (vmcnt is one in-order counter: the weight fragments
// requested behind the plane would wait for it; 72 prefetch registers do not exist beside 64 accumulators and two fragment
// sets) and the second workgroup does not fill the gap: delayed starts of the workgroups in a CU's second slot, and three
// workgroups per CU on 4 x 16 patches (-DS2R_TY=4: 168 registers, 43.5 KB), both leave the time where it is.
#include <type_traits>

#include "az_conv3d_args.h"
#include "az_roll_common.h"
#include "az_options.h"
#include "az_launch_math.h"

#define S2R_NT 256
#ifndef S2R_TY
#define S2R_TY 8                           // coarse rows of a patch: 8 (two workgroups per CU) or 4 (three; experiment)
#endif
#define S2R_NH (S2R_TY / 4)                // 4-row halves of the patch
#define S2R_NTILE (4 * S2R_NH)             // 4x4 tiles of the patch = tiles per wave
#define S2R_WGS (S2R_TY == 8 ? 2 : 3)      // workgroups per CU
#define S2R_FY (2 * S2R_TY + 1)            // fine rows / columns of a patch's plane (with the halo)
#define S2R_FX 33                          // (561 voxels per staged plane at 8 patch rows)
#define S2R_NLD (S2R_FY + 1)               // 16-byte pieces per thread and plane: one per fine row + the 33rd column
#define S2R_PITCH 17                       // sub-slab row pitch in voxels
#define S2R_SUB_BYTES ((S2R_TY + 1) * S2R_PITCH * 128)  // 19 584
#define S2R_LDS (4 * S2R_SUB_BYTES)        // 78 336
#define S2R_TAPB (4 * 2 * 64 * 16)         // bytes per tap of the packed image [tap][cout/16 (4)][part (2)][lane][16 B]
#define S2R_SYNC() do { if (!(S2R_ABL & 32)) __syncthreads(); } while (0)
#ifndef S2R_KIND_FIRST  // (experiments: 1 = the first / last plane of a segment multiply all eighteen taps, nine of them dropped)
#define S2R_KIND_FIRST 2
#define S2R_KIND_LAST 3
#endif
#ifndef S2R_ABL
#define S2R_ABL 0  // timing-only ablations: 1 no output stores, 2 no weight loads after the first tap, 4 no slab staging, 8 no fragment reads, 16 no split / LDS writes, 32 no barriers
#endif

// Diagnostic build only (-DS2R_STAMP): per-phase cycle sums of every wave, added to a global array nothing else reads
// (tools/s2roll_stamp_probe.py): 0 staging loads issued -> arrived, 1 split + LDS writes, 2 barrier behind the staging,
// 3 the taps, 4 barrier behind the taps, 5 epilogue, 6 planes, 7 whole wave.
#ifdef S2R_STAMP
__device__ unsigned long long s2r_stamp_sum[8];
#define S2R_T(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); st_acc[i] += t1_ - st_t0; st_t0 = t1_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define S2R_T(i)
#endif

// EPI: 0 = y = relu?(acc * scale + shift) (+ residual when given), 1 = raw output + BatchNorm partials
// PS: the input is a pre-split tensor (az_roll_common.h): the gradient a BatchNorm backward wrote
template <int EPI, bool PS>
__global__ void __launch_bounds__(S2R_NT, S2R_WGS)
conv3d_s2roll_kernel(const ConvArgs a) {
    __shared__ __attribute__((aligned(128))) unsigned char slab[S2R_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's sixteen output channels
#ifdef S2R_STAMP
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long st_k0 = st_t0;
#endif

    const int lin = a.map_mode >= 1 ? az_xcd_map(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    int tix, tiy, seg, b;
    az_roll_decode(lin, a.tiles_x, a.tiles_y, a.nseg, tix, tiy, seg, b);
    const int c0 = seg * a.seg_len, c1 = min(c0 + a.seg_len, a.Do);  // coarse output planes [c0, c1)
    const int ty0 = tiy * S2R_TY, tx0 = tix * 16;                           // coarse patch origin

    const unsigned in_bytes = (unsigned)a.Di * a.Hi * a.Wi * 32u * 4u, out_bytes = (unsigned)a.Do * a.Ho * a.Wo * 64u * 4u;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in) + (size_t)b * (in_bytes / 4), 0, in_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * (out_bytes / 4), 0, out_bytes, 0x00020000);
    const bool has_res = EPI == 0 && a.res != nullptr;
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(has_res ? a.res : a.out) + (size_t)b * (out_bytes / 4), 0, has_res ? out_bytes : 0u, 0x00020000);
    const auto rs_part = __builtin_amdgcn_make_buffer_rsrc(EPI == 1 ? a.part : a.out, 0, EPI == 1 ? (unsigned)(a.ntiles * 64 * 2 * 4) : 0u, 0x00020000);
    const auto rs_cnt = __builtin_amdgcn_make_buffer_rsrc(EPI == 1 ? a.cnt : a.out, 0, EPI == 1 ? (unsigned)(a.ntiles * 4) : 0u, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp), 0, 27u * S2R_TAPB, 0x00020000);

    const int ki = az_f16_scale_exp(az_amax_read(a.in_amax));
    const int kw_ = az_f16_scale_exp(az_amax_read(a.w_amax));
    const float in_scale = az_pow2(ki);
    const int out_exp = -(ki + kw_);

    // accumulators: [set][tile]; set 0: output plane t (the one being completed), set 1: plane t + 1 (begun by kd = 0)
    f32x4 acc[2][S2R_NTILE];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int m = 0; m < S2R_NTILE; ++m) acc[s][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging: fine plane z, rows 2 ty0 - 1 .., columns 2 tx0 - 1 .. (17 x 33 voxels x 8 pieces of four channels).
    //      piece `it` < 17 of thread tid: fine row it, column tid >> 3, channels 4 (tid & 7) .. (a row's 32 voxels = 4 KB in one
    //      instruction of the workgroup; the row is a compile-time constant of the piece, so its address arithmetic is one
    //      add); piece 17: the 33rd column of row tid >> 3 (136 threads) ---------------------------------------------------
    const int pj = tid & 7, pc = tid >> 3;
    const int iw_a = 2 * tx0 - 1 + pc;
    const bool col_ok = (unsigned)iw_a < (unsigned)a.Wi;
    const unsigned in_col = (unsigned)iw_a * 128u + (unsigned)pj * 16u;
    // LDS: [sub-slab (row parity, column parity)][row r][column c]; octet swizzle by the row's parity, part swap by bit 1 of c
    // (the sub-slabs of the ODD columns keep hi and lo the other way round: the sixteen lanes of a ds_write_b64 group are two
    //  x-adjacent voxels, one per column parity, and would otherwise meet in the same 64-byte half of the bank row)
    const unsigned dst_col = (unsigned)((pc & 1) * S2R_SUB_BYTES + (pc >> 1) * 128 + (pj & 1) * 8 + ((((pc >> 2) ^ pc) & 1) * 64));
    const unsigned oct_e = (unsigned)((pj >> 1) << 4), oct_o = (unsigned)(((pj >> 1) ^ 2) << 4);  // rows r even / odd
    const unsigned dst_e = dst_col + oct_e, dst_o = dst_col + oct_o, dst_e_lo = dst_e ^ 64u, dst_o_lo = dst_o ^ 64u;
    auto stage = [&](int z) __attribute__((always_inline)) {
        u32x4 pre[S2R_NLD];
        const bool zok = (unsigned)z < (unsigned)a.Di && !(S2R_ABL & 4);
        const int ih0 = 2 * ty0 - 1;
        const unsigned plane_off = (unsigned)(z * a.Hi) * (unsigned)a.Wi * 128u;
#pragma unroll
        for (int it = 0; it < S2R_FY; ++it) {
            const bool ok = zok && (unsigned)(ih0 + it) < (unsigned)a.Hi && col_ok;
            const unsigned off = plane_off + (unsigned)((ih0 + it) * a.Wi) * 128u + in_col;
            pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? off : R_OOB, 0, 0);
        }
        {   // the 33rd column: row pc (< 17), column 32
            const int iw = 2 * tx0 + 31;
            const bool ok = zok && pc < S2R_FY && (unsigned)(ih0 + pc) < (unsigned)a.Hi && iw < a.Wi;
            const unsigned off = plane_off + (unsigned)((ih0 + pc) * a.Wi + iw) * 128u + (unsigned)pj * 16u;
            pre[S2R_FY] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? off : R_OOB, 0, 0);
        }
#ifdef S2R_STAMP
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        S2R_T(0);
#endif
#pragma unroll
        for (int it = 0; it < S2R_FY; ++it) {
            // fine row 2r + 1 is an even input row (the kh = 1 sub-slabs, row r); fine row 2r an odd one (kh = 0 / 2, row r)
            const int r = it >> 1;
            // (hi at the address with the column's part-swap bit, lo at the one with that bit flipped)
            const unsigned d_hi = ((r & 1) ? dst_o : dst_e), d_lo = ((r & 1) ? dst_o_lo : dst_e_lo);
            const int base = (it & 1) * 2 * S2R_SUB_BYTES + r * S2R_PITCH * 128;
            uint2 hi, lo;
            az_stage_f16x4<PS>(pre[it], in_scale, hi, lo);
            if (!(S2R_ABL & 16)) {
                *reinterpret_cast<uint2 *>(slab + base + d_hi) = hi;
                *reinterpret_cast<uint2 *>(slab + base + d_lo) = lo;
            }
        }
        if (pc < S2R_FY) {
            const int r = pc >> 1;  // column 32 = c 16 of the even-column sub-slabs: part swap bit (16 >> 1) & 1 = 0
            unsigned char *dst = slab + (pc & 1) * 2 * S2R_SUB_BYTES + (r * S2R_PITCH + 16) * 128 + (pj & 1) * 8 + ((r & 1) ? oct_o : oct_e);  // (128-byte aligned base: + 64 below = ^ 64)
            uint2 hi, lo;
            az_stage_f16x4<PS>(pre[S2R_FY], in_scale, hi, lo);
            if (!(S2R_ABL & 16)) {
                *reinterpret_cast<uint2 *>(dst) = hi;
                *reinterpret_cast<uint2 *>(dst + 64) = lo;
            }
        }
    };

    // voxel fragment (B operand): lane -> voxel (row (lane >> 2) & 3, column lane & 3) of a 4 x 4 tile, channel octet lane >> 4;
    // [dh][dw]: the tap's row / column offset inside its sub-slab; part 1 sits at the address with bit 6 flipped
    const int trow = (lane >> 2) & 3, tcol = lane & 3, oct = lane >> 4;
    unsigned xb[2][2][2];  // [part][dh][dw]
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
            const int row = trow + dh, col = tcol + dw;
            xb[0][dh][dw] = (unsigned)((row * S2R_PITCH + col) * 128 + ((oct ^ ((row & 1) << 1)) << 4) + ((col >> 1) & 1) * 64);
            xb[1][dh][dw] = xb[0][dh][dw] ^ 64u;
        }
    const unsigned wlane = (unsigned)(cg * 2 * 64 + lane) * 16u;

    // epilogue constants: this lane's four channels 16 cg + 4 (lane >> 4) + r
    const int cqh = cg * 16 + 4 * (lane >> 4);
    float4 sch = make_float4(1.f, 1.f, 1.f, 1.f), sfh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI == 0) {
        if (a.scale) sch = *reinterpret_cast<const float4 *>(a.scale + cqh);
        if (a.shift) sfh = *reinterpret_cast<const float4 *>(a.shift + cqh);
    }
    sch.x = ldexpf(sch.x, out_exp); sch.y = ldexpf(sch.y, out_exp); sch.z = ldexpf(sch.z, out_exp); sch.w = ldexpf(sch.w, out_exp);
    const float osc = ldexpf(1.f, out_exp);
    const float floor_ = (EPI == 0 && a.relu) ? 0.f : -__builtin_inff();
    float hk[4] = {0.f, 0.f, 0.f, 0.f}, hs1[4] = {0.f, 0.f, 0.f, 0.f}, hs2[4] = {0.f, 0.f, 0.f, 0.f};
    int h_n = 0;
    bool h_first = true;

    // store accumulator set 0 as output plane t
    auto finish = [&](int t) __attribute__((always_inline)) {
        if (EPI == 1) {  // the shift of the running sums: the first value this lane sees (any value near the data serves)
            const f32x4 c = acc[0][0];
            hk[0] = h_first ? c[0] * osc : hk[0]; hk[1] = h_first ? c[1] * osc : hk[1];
            hk[2] = h_first ? c[2] * osc : hk[2]; hk[3] = h_first ? c[3] * osc : hk[3];
            h_first = false;
        }
#pragma unroll
        for (int m = 0; m < S2R_NTILE; ++m) {
            const int oy = ty0 + 4 * (m >> 2) + trow, ox = tx0 + 4 * (m & 3) + tcol;
            const bool vok = oy < a.Ho && ox < a.Wo;
            const unsigned off = vok ? (unsigned)((t * a.Ho + oy) * a.Wo + ox) * 256u + (unsigned)cqh * 4u : R_OOB;
            const f32x4 c = acc[0][m];
            float4 y = EPI == 1 ? make_float4(c[0] * osc, c[1] * osc, c[2] * osc, c[3] * osc)
                                : make_float4(fmaf(c[0], sch.x, sfh.x), fmaf(c[1], sch.y, sfh.y), fmaf(c[2], sch.z, sfh.z), fmaf(c[3], sch.w, sfh.w));
            if (EPI == 0) {
                const float4 rr = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, (S2R_ABL & 1) ? R_OOB : off, 0, 0));
                y.x += rr.x; y.y += rr.y; y.z += rr.z; y.w += rr.w;
                y.x = fmaxf(y.x, floor_); y.y = fmaxf(y.y, floor_); y.z = fmaxf(y.z, floor_); y.w = fmaxf(y.w, floor_);
            }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), rs_out, (S2R_ABL & 1) ? R_OOB : off, 0, 0);
            if (EPI == 1) {
                h_n += vok ? 1 : 0;
                const float d0 = vok ? y.x - hk[0] : 0.f, d1 = vok ? y.y - hk[1] : 0.f, d2 = vok ? y.z - hk[2] : 0.f, d3 = vok ? y.w - hk[3] : 0.f;
                hs1[0] += d0; hs1[1] += d1; hs1[2] += d2; hs1[3] += d3;
                hs2[0] = fmaf(d0, d0, hs2[0]); hs2[1] = fmaf(d1, d1, hs2[1]); hs2[2] = fmaf(d2, d2, hs2[2]); hs2[3] = fmaf(d3, d3, hs2[3]);
            }
        }
#pragma unroll
        for (int m = 0; m < S2R_NTILE; ++m) {
            acc[0][m] = acc[1][m];
            acc[1][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    // weights of in-plane tap position i (kh * 3 + kw) for depth tap kd: [part]
    auto load_w = [&](float4 (&w)[2], int kd, int i) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            w[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, (kd * 9 + i) * S2R_TAPB + q * 1024, 0));
    };
    // the voxel fragments of tap position i for the four tiles of patch half `half`: [tile][part]
    auto load_x = [&](float4 (&x)[4][2], int i, int half) __attribute__((always_inline)) {
        const int kh = i / 3, kw = i - 3 * kh;
        const int sub = (kh == 1 ? 2 : 0) + (kw == 1 ? 1 : 0);
        const int cst = sub * S2R_SUB_BYTES + 4 * half * S2R_PITCH * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (S2R_ABL & 8) {
                // (operands that differ per tile and lane: with one constant the accumulators of the tiles are provably equal
                //  and the compiler keeps ONE chain of MFMAs -- the first ablation run, profiles/r05v_s2roll_ablations.txt)
                x[q][0] = x[q][1] = __builtin_bit_cast(float4, u32x4{0x3c003c00u + (unsigned)(4 * half + q), 0x3c003c00u + (unsigned)lane, 0x3c003c00u, 0x3c003c00u});
            } else {
                x[q][0] = *reinterpret_cast<const float4 *>(slab + xb[kw == 1 ? 1 : 0][kh == 2][kw == 2] + cst + 4 * q * 128);
                x[q][1] = *reinterpret_cast<const float4 *>(slab + xb[kw == 1 ? 0 : 1][kh == 2][kw == 2] + cst + 4 * q * 128);
            }
        }
    };
    // twelve MFMAs: one weight fragment pair x four tiles (the three products of a tile are four MFMAs apart)
    auto mul4 = [&](f32x4 (&c)[S2R_NTILE], int half, const float4 (&w)[2], const float4 (&x)[4][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) c[4 * half + q] = R_MH(c[4 * half + q], w[0], x[q][0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) c[4 * half + q] = R_MH(c[4 * half + q], w[0], x[q][1]);
#pragma unroll
        for (int q = 0; q < 4; ++q) c[4 * half + q] = R_MH(c[4 * half + q], w[1], x[q][0]);
    };

    // ---- the taps of the plane in the slab.  KIND 0: an even plane, kd = 1 into set 0;  1: an odd plane, kd = 2 into set 0 and
    //      kd = 0 into set 1;  2: the first plane of a segment, kd = 0 into set 1 only (its kd = 2 products belong to the segment
    //      before);  3: the last plane, kd = 2 into set 0 only --------------------------------------------------------------------
    auto plane = [&](auto kind_tag) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_tag)::value;
        constexpr bool HAS_A = KIND != 2, HAS_B = KIND == 1 || KIND == 2;
        constexpr int KD_A = KIND == 0 ? 1 : 2;
        float4 wa[2][2], wb[2][2];  // [tap parity][part]: wa: kd = 1 or 2 into set 0; wb: kd = 0 into set 1
        float4 xf[2][4][2];         // [step parity][tile][part]: the fragments of a (tap position, patch half), one step ahead
        if (HAS_A) load_w(wa[0], KD_A, 0);
        if (HAS_B) load_w(wb[0], 0, 0);
        load_x(xf[0], 0, 0);
#pragma unroll
        for (int st = 0; st < 9 * S2R_NH; ++st) {
            const int i = st / S2R_NH, g = st % S2R_NH;
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < 9 * S2R_NH) load_x(xf[(st + 1) & 1], (st + 1) / S2R_NH, (st + 1) % S2R_NH);
            if (g == 0 && i + 1 < 9 && !(S2R_ABL & 2)) {
                if (HAS_A) load_w(wa[(i + 1) & 1], KD_A, i + 1);
                if (HAS_B) load_w(wb[(i + 1) & 1], 0, i + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int wp = (S2R_ABL & 2) ? 0 : (i & 1);
            if (HAS_A) mul4(acc[0], g, wa[wp], xf[st & 1]);
            if (HAS_B) mul4(acc[1], g, wb[wp], xf[st & 1]);
        }
    };

    // ---- the walk: fine planes 2 c0 - 1 .. 2 c1 - 1 ------------------------------------------------------------------------
    stage(2 * c0 - 1);
    __syncthreads();
    plane(std::integral_constant<int, S2R_KIND_FIRST>{});
#pragma unroll
    for (int m = 0; m < S2R_NTILE; ++m) {
        acc[0][m] = acc[1][m];
        acc[1][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int t = c0; t < c1; ++t) {
        S2R_T(3);
        S2R_SYNC();
        S2R_T(4);
        stage(2 * t);
        S2R_T(1);
        S2R_SYNC();
        S2R_T(2);
        plane(std::integral_constant<int, 0>{});
        S2R_T(3);
        S2R_SYNC();
        S2R_T(4);
        stage(2 * t + 1);
        S2R_T(1);
        S2R_SYNC();
        S2R_T(2);
        if (t + 1 < c1) plane(std::integral_constant<int, 1>{});
        else plane(std::integral_constant<int, S2R_KIND_LAST>{});
        S2R_T(3);
        finish(t);
        S2R_T(5);
#ifdef S2R_STAMP
        st_acc[6] += 2;
#endif
    }

#ifdef S2R_STAMP
    st_acc[7] = __builtin_amdgcn_s_memtime() - st_k0;
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&s2r_stamp_sum[i], st_acc[i]);
#endif
    if (EPI == 1) {
        const unsigned tile_id = (unsigned)(((b * a.nseg + seg) * a.tiles_y + tiy) * a.tiles_x + tix);
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
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ntot), rs_cnt, (lane == 0 && cg == 0) ? tile_id * 4u : R_OOB, 0, 0);
    }
}

#ifdef S2R_STAMP
extern "C" int az_debug_s2roll_stamps(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(s2r_stamp_sum), 8 * sizeof(unsigned long long)) != hipSuccess) return AZ_ELAUNCH;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(s2r_stamp_sum), z, sizeof(z)) != hipSuccess) return AZ_ELAUNCH; }
    return AZ_OK;
}
#endif

static void s2roll_geometry(ConvArgs &a) {
    a.tiles_y = (a.Ho + S2R_TY - 1) / S2R_TY;
    a.tiles_x = (a.Wo + 15) / 16;
    az_s2roll_segments((long long)a.B * a.tiles_y * a.tiles_x, a.Do, az_options().s2roll_seglen, a.nseg, a.seg_len, 256 * S2R_WGS);  // (az_launch_math.h: swept on the CPU)
}

long long az_conv3d_s2roll_stats_tiles(ConvArgs a) {
    s2roll_geometry(a);
    return (long long)a.B * a.nseg * a.tiles_y * a.tiles_x;
}

// the shapes the kernel addresses: one batch element of the input and of the output through 32-bit buffer offsets
bool az_conv3d_s2roll_fits(const ConvArgs &a) {
    return az_fits_buffer_offset((long long)a.Do * a.Ho * a.Wo * 64 * 4) && az_fits_buffer_offset((long long)a.Di * a.Hi * a.Wi * 32 * 4);
}

// f16x3, cin = 32, cout = 64, MODE 1; weights packed by az_conv3d_pack_r16_f16(cin = 32, cout = 64)
int az_conv3d_s2roll_launch(ConvArgs a, int epi, hipStream_t s) {
    if (!a.in_amax || !a.w_amax) return AZ_ENULL;
    s2roll_geometry(a);
    const long long blocks = (long long)a.B * a.nseg * a.tiles_y * a.tiles_x;
    if (epi) a.ntiles = blocks;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    if (!az_conv3d_s2roll_fits(a) || a.ntiles * 512 >= 0xffffff00LL) return AZ_EUNSUPPORTED;
    if (a.in_split && epi) return AZ_EUNSUPPORTED;  // (pre-split inputs are gradients: no BatchNorm-partials epilogue)
    if (epi) hipLaunchKernelGGL((conv3d_s2roll_kernel<1, false>), dim3((unsigned)blocks), dim3(S2R_NT), 0, s, a);
    else if (a.in_split) hipLaunchKernelGGL((conv3d_s2roll_kernel<0, true>), dim3((unsigned)blocks), dim3(S2R_NT), 0, s, a);
    else hipLaunchKernelGGL((conv3d_s2roll_kernel<0, false>), dim3((unsigned)blocks), dim3(S2R_NT), 0, s, a);
    return az_launch_status();
}
