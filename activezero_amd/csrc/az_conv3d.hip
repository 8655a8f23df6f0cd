// K4/K5 -- 3x3x3 convolution family of the PSMNet cost aggregation on fp32 MFMA
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 157 TFLOP/s dense peak).
// Reference call sites: nets/psmnet/psmnet_submodule_3.py:44-56 (convbn_3d),
// nets/psmnet/psmnet_3.py:15-58 (hourglass convs / ConvTranspose3d), :87-117.
//
// One kernel template, three index maps (all pad 1, kernel 3):
//   MODE 0  stride-1 conv            in = o - 1 + k
//   MODE 1  stride-2 conv            in = 2o - 1 + k
//   MODE 2  stride-2 TRANSPOSED conv o = 2i - 1 + k, run as 8 output-parity phases:
//           parity 0 -> tap k=1 (i = t), parity 1 -> taps k=2 (i = t), k=0 (i = t+1)
// Forward, input-gradient and transposed variants of every layer reduce to these
// maps with re-packed weights (az_conv3d_pack_weights).
//
// Layout: activations are channels-last NDHWC, so one voxel is 32/64 contiguous
// floats.  Implicit GEMM: M = 32 output voxels (a 4x8 patch), N = 32 output
// channels, K = 27 taps x Cin.  ONE 64-lane wavefront (= one workgroup, no
// barriers) owns a 4x16 (MODE 1: 4x8) output patch for all Cout:
//   * per (input plane, 32-channel chunk) it stages the zero-padded input slab in
//     its private LDS region, voxel stride 36 dwords (bank-conflict padding);
//   * per tap, A fragments come from LDS (4 x ds_read_b128 = 16 k-values of the
//     lane's voxel) and B fragments straight from the L2-resident packed weights
//     (4 x global_load_dwordx4 = the same 16 k-values of the lane's out-channel);
//   * 16 MFMAs per (M-tile, N-tile, tap, chunk); accumulators never leave registers.
// fp32 MFMA is 64 cycles/instruction, so one slab (<= 22 KB) feeds >= 18k cycles of
// matrix work: the kernel is MFMA-bound by construction; LDS/L2 traffic is noise.
//
// Epilogues: (0) y = acc*scale[c] + shift[c] (+residual) (ReLU) -- eval-mode BN folded
// or plain conv; (1) raw store + per-tile per-channel (sum, centred M2) partials for
// train-mode BatchNorm (merged with Chan's formula in fp64 by az_bn3d_finalize).
// SRC 1 synthesises the PSMNet concat cost volume on the fly from the two NHWC
// feature maps (reference psmnet_3.py:149-163) instead of reading a 64-channel tensor.
#include <stdlib.h>

#include "az_roll_common.h"
#include "az_options.h"
#include "az_launch_math.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CV_VS 36  // slab voxel stride in dwords
#define CV_BAND 4 // tile rows per band of the block -> tile order
#define X6_VS 52  // PREC 1 slab voxel stride in dwords
// bf16x6 register policy (measured per shape, tools/conv_ab.py): with two N-tiles or the
// stride-2 map the kernel takes the 1-wave/SIMD budget (512 registers) and prefetches the
// weights one tap ahead; otherwise 2 waves/SIMD without weight prefetch.
#define X6_WIDE(COUT, MODE, PREC) ((PREC) == 1 && (((COUT) == 64 && (MODE) != 2) || (MODE) == 1))
// (Measured and dropped: one wave owning the four (ph, pw) output-parity phases of a transposed-conv patch so that
//  their common coarse slab is staged once -- it needs the 1-wave/SIMD register budget and that costs what the
//  sharing saves: 64->32 V1->V0 0.89 vs 1.00 ms standalone, 156.5 vs 156.8 ms per step.)
// (Tried and dropped: a 3-deep register ring streaming the weights two taps ahead under the
//  1-wave/SIMD budget for every shape -- 2.76 ms vs 2.2 ms on 32->32: one wave per SIMD cannot
//  hide its own commit / epilogue phases, and the fully unrolled ring spills into AGPRs.)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Diagnostic build only (-DCV_STAMP): per-phase cycle sums of every wave, added to a global
// array that nothing else reads (cdna_hip_programming.md, In-kernel stamps).
#ifdef CV_STAMP
__device__ unsigned long long cv_stamp_sum[8];
#define CV_T0() cv_t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#define CV_ACC(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long cv_t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); cv_acc[i] += cv_t1 - cv_t0; __builtin_amdgcn_sched_barrier(0); } while (0)
#define CV_FLUSH() do { if (lane == 0) { for (int i_ = 0; i_ < 5; ++i_) atomicAdd(&cv_stamp_sum[i_], cv_acc[i_]); atomicAdd(&cv_stamp_sum[5], 1ull); \
    atomicAdd(&cv_stamp_sum[6], (unsigned long long)__builtin_amdgcn_s_memtime() - cv_k0); \
    atomicAdd(&cv_stamp_sum[7], (unsigned long long)__builtin_amdgcn_s_memrealtime() - cv_r0); } } while (0)
#else
#define CV_T0()
#define CV_ACC(i)
#define CV_FLUSH()
#endif

#include "az_conv3d_args.h"

template <int CIN, int COUT, int MODE, int EPI, int SRC, int PREC>
__global__ void __launch_bounds__(64, X6_WIDE(COUT, MODE, PREC) ? 1 : 2)
conv3d_gather_kernel(const ConvArgs a) {
    constexpr int NCH = CIN / 32, NR = COUT / 32;
    constexpr int MR = (MODE == 1) ? 1 : 2;
    constexpr int TY = 4, TX = 8 * MR;
    constexpr int SY = (MODE == 0) ? TY + 2 : (MODE == 1) ? 2 * TY + 1 : TY + 1;
    constexpr int SX = (MODE == 0) ? TX + 2 : (MODE == 1) ? 2 * TX + 1 : TX + 1;
    // PREC 0: fp32 slab, 36 dwords per voxel.  PREC 1: three bf16 planes (hi, mid, lo) of 32
    // channels each, 52 dwords per voxel (3 x 64 B + 16 B bank padding).
    // PREC 1, unit-stride reads (MODE 0/2): no padding (48 dwords per voxel); instead the four
    // 16-byte pieces of every 64-byte part are XOR-swizzled with (slab row & 3).  A ds_read_b128
    // is served in four 16-lane groups {0-3,12-15,20-27}, ... = four runs of 4 consecutive x on
    // 4 consecutive slab rows: with a 12-slot voxel stride a run covers the four 64-byte quads
    // of the 256-byte bank row once, and the row swizzle sends the four runs to different
    // 16-byte slots of each quad -> conflict-free (the padded layout was 3-way conflicted and
    // made the LDS, not the MFMA pipe, the bound of the 32-wide kernels).
    // PREC 3 = f16x3 (az_roll_common.h): two scaled fp16 parts in the first two 64-byte parts of the same voxel image
    constexpr bool X16 = (PREC == 3);
    constexpr bool SWZ = (PREC != 0) && (MODE != 1);
    constexpr int VS = (PREC == 0) ? CV_VS : SWZ ? 48 : X6_VS;
    constexpr int SXP = SX;  // LDS row pitch of the slab in voxels
    __shared__ __attribute__((aligned(16))) float slab[SY * SXP * VS];

#ifdef CV_STAMP
    unsigned long long cv_acc[5] = {0, 0, 0, 0, 0}, cv_t0 = 0;
    // whole-wave shader cycles and 100 MHz real-time ticks: in-kernel clock = ratio x 100 MHz
    const unsigned long long cv_k0 = __builtin_amdgcn_s_memtime(), cv_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int lane = threadIdx.x;
    // ---- block -> tile map -------------------------------------------------------------
    // (1) XCD-aware: blocks b and b+8 share an XCD (and its 4 MB L2); hand every XCD one
    //     contiguous chunk of the linear tile order instead of every 8th tile.
    // (2) banded order inside the chunk: x fastest, then CV_BAND tile rows, then the depth
    //     index, then the band -- so the three output planes that read one input plane
    //     (and the row halos inside a band) are processed back to back and hit in L2.
    // Pure performance choice; any map that is a bijection onto the tile set is correct.
    int lin = blockIdx.x;
    if (a.map_mode >= 1) {
        const int nblk = gridDim.x, xcd = blockIdx.x & 7, q8 = nblk >> 3, r8 = nblk & 7;
        lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    }
    int pd = 0, ph = 0, pw = 0;
    if (MODE == 2) {
        const int phase = lin & 7; lin >>= 3;
        pd = phase >> 2; ph = (phase >> 1) & 1; pw = phase & 1;
    }
    const int tix = lin % a.tiles_x; lin /= a.tiles_x;
    int tiy, td, b;
    if (a.map_mode >= 2) {
        // lin indexes (b, band, d, row-in-band); the last band may have fewer rows
        const int per_b = a.Dt * a.tiles_y;
        b = lin / per_b;
        int l = lin - b * per_b;
        const int full = (a.tiles_y / CV_BAND) * CV_BAND * a.Dt;  // blocks in the complete bands
        int band, rows;
        if (l < full) { band = l / (CV_BAND * a.Dt); l -= band * CV_BAND * a.Dt; rows = CV_BAND; }
        else { band = a.tiles_y / CV_BAND; l -= full; rows = a.tiles_y - band * CV_BAND; }
        td = l / rows;
        tiy = band * CV_BAND + (l - td * rows);
    } else {
        tiy = lin % a.tiles_y; lin /= a.tiles_y;
        td = lin % a.Dt;
        b = lin / a.Dt;
    }
    // canonical tile id (rows of the BatchNorm partial buffers)
    const int tile_id = ((((b * a.Dt + td) * a.tiles_y + tiy) * a.tiles_x + tix) << (MODE == 2 ? 3 : 0)) +
                        (MODE == 2 ? (pd * 4 + ph * 2 + pw) : 0);
    const int ty0 = tiy * TY, tx0 = tix * TX;  // tile origin in index space
    // input coordinates of slab voxel (0,0)
    const int ih0 = (MODE == 0) ? ty0 - 1 : (MODE == 1) ? 2 * ty0 - 1 : ty0;
    const int iw0 = (MODE == 0) ? tx0 - 1 : (MODE == 1) ? 2 * tx0 - 1 : tx0;

    f32x16 acc[MR][NR];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][n][e] = 0.f;

    float in_scale = 1.f, osc = 1.f;
    if (X16) {  // wave-uniform power-of-two operand scales from the tensors' largest magnitudes
        const int ki = az_f16_scale_exp(az_amax_read(a.in_amax));
        const int kw = az_f16_scale_exp(az_amax_read(a.w_amax));
        in_scale = az_pow2(ki);
        osc = ldexpf(1.f, -(ki + kw));
    }
    const int nd = (MODE == 2) ? 1 + pd : 3;
    const int nh = (MODE == 2) ? 1 + ph : 3;
    const int nw = (MODE == 2) ? 1 + pw : 3;
    const int row = lane & 31, half = lane >> 5;
    const int rty = row >> 3, rtx = row & 7;
    const float4 *wp4 = reinterpret_cast<const float4 *>(a.wp);

    // ---- software pipeline -------------------------------------------------------------
    // stage = (input plane sd, 32-channel chunk cc).  The slab of stage s+1 is fetched
    // into registers (branch-free, all loads in flight together) while the MFMAs of
    // stage s run; weights are fetched one tap ahead.
    constexpr int NQ = SY * SX * 8;          // float4 pieces of one slab
    constexpr int NLD = (NQ + 63) / 64;      // pieces per lane
    constexpr bool WIDE = X6_WIDE(COUT, MODE, PREC);
    constexpr bool BPIPE = (PREC == 0) ? (NR == 1) : WIDE;
    // stages cover only the input planes that exist (a contiguous sd range): a zero-padding plane
    // contributes nothing, so neither its slab nor its split / LDS traffic is spent -- matters at the
    // depth borders and, above all, for depth-1 volumes (the extractor's 2-D layers: 1 plane of 3)
    int sd_lo = 0, sd_hi = nd - 1;
    {
        const int p0 = (MODE == 0) ? td - 1 : (MODE == 1) ? 2 * td - 1 : td;  // plane of sd = 0
        if (p0 < 0) sd_lo = -p0;                                             // (only MODE 0/1 start at -1)
        if (p0 + sd_hi >= a.Di) sd_hi = a.Di - 1 - p0;
    }
    const int S0 = sd_lo * NCH, NS = (sd_hi + 1) * NCH;  // stage index range [S0, NS)
    float4 pre[NLD];
    unsigned okbits = 0;  // bit it: pre[it] holds real data (else zero padding)

    auto plane_of = [&](int sd) -> int {
        return (MODE == 0) ? td - 1 + sd : (MODE == 1) ? 2 * td - 1 + sd : td + (pd ? sd : 0);
    };
    auto issue = [&](int s) {
        const int sd = s / NCH, cc = s - sd * NCH;
        const int id = plane_of(sd);
        const bool pok = id >= 0 && id < a.Di;
        const int idc = min(max(id, 0), a.Di - 1);
        okbits = 0;
        // every load is issued unconditionally from a clamped (always valid)
        // address, so all NLD requests are in flight together.  Address = wave-uniform
        // 64-bit plane base + 32-bit per-lane offset; a wave-instruction covers 8 slab
        // voxels, so (sy, sx) advance by 8 voxels per iteration without divisions.
        const float *plane0 = (SRC == 0)
            ? a.in + (((size_t)b * a.Di + idc) * a.Hi) * a.Wi * CIN + cc * 32
            : ((cc == 0) ? a.in : a.in2) + ((size_t)b * a.Hi) * a.Wi * 32;
        const int part4 = (lane & 7) * 4;
        int sy = (lane >> 3) / SX, sx = (lane >> 3) - sy * SX;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int ih = ih0 + sy, iw = iw0 + sx;
            const int ihc = min(max(ih, 0), a.Hi - 1), iwc = min(max(iw, 0), a.Wi - 1);
            bool ok = pok && (lane + 64 * it < NQ) && ih == ihc && iw == iwc;
            unsigned off;
            if (SRC == 0) {
                off = (unsigned)(ihc * a.Wi + iwc) * CIN + part4;
            } else {  // concat cost volume: plane id = disparity index
                ok = ok && (iw >= id);
                const int iwr = (cc == 0) ? iwc : max(iwc - idc, 0);
                off = (unsigned)(ihc * a.Wi + iwr) * 32 + part4;
            }
            pre[it] = *reinterpret_cast<const float4 *>(plane0 + off);
            okbits |= ok ? (1u << it) : 0u;
            sx += 8;
            if (sx >= SX) { sx -= SX; ++sy; }
        }
        // (padding is applied at commit time, after the MFMAs this prefetch hides under: touching
        //  the loaded registers here would put the whole memory latency back on the critical path)
    };
    auto commit = [&]() {
        int sy = (lane >> 3) / SX, sx = (lane >> 3) - sy * SX;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int q = lane + 64 * it;
            const int vox = sy * SXP + sx;
            if (!((okbits >> it) & 1u)) pre[it] = make_float4(0.f, 0.f, 0.f, 0.f);  // zero padding
            if (PREC == 0) {
                if (q < NQ) *reinterpret_cast<float4 *>(&slab[vox * CV_VS + (q & 7) * 4]) = pre[it];
            } else if (q < NQ) {
                uint2 hi, mid, lo;
                if (X16) {
                    float4 v = pre[it];
                    v.x *= in_scale; v.y *= in_scale; v.z *= in_scale; v.w *= in_scale;
                    az_split2_f16x4(v, hi, mid);
                    lo = mid;
                } else {
                    az_split3_bf16x4(pre[it], hi, mid, lo);
                }
                unsigned *dst = reinterpret_cast<unsigned *>(slab) + vox * VS +
                                (SWZ ? ((((q & 7) >> 1) ^ (sy & 3)) * 4 + (q & 1) * 2) : (q & 7) * 2);
                *reinterpret_cast<uint2 *>(dst) = hi;
                *reinterpret_cast<uint2 *>(dst + 16) = mid;
                if (!X16) *reinterpret_cast<uint2 *>(dst + 32) = lo;
            }
            sx += 8;
            if (sx >= SX) { sx -= SX; ++sy; }
        }
    };
    // 16-byte operand pieces per (tile, tap, chunk): fp32 -> 4 (k = 16*half + 4j..4j+3);
    // bf16x6 -> 6 = 3 split parts x 2 K16 blocks (k = 16*kb + 8*half + 0..7), index p*2 + kb
    constexpr int NF = (PREC == 0) ? 4 : X16 ? 4 : 6;
    auto load_b = [&](auto &bq, int tap, int cc) {
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int j = 0; j < NF; ++j)
                bq[n][j] = wp4[((((size_t)tap * NCH + cc) * NR + n) * NF + j) * 64 + lane];
    };
    const int ntaps = nh * nw;
    auto tap_of = [&](int kd, int t, int &eh, int &ew) -> int {
        const int sh = t / nw, sw = t - sh * nw;
        const int kh = (MODE == 2) ? (ph ? 2 - 2 * sh : 1) : sh;
        const int kw = (MODE == 2) ? (pw ? 2 - 2 * sw : 1) : sw;
        eh = (MODE == 2) ? (ph ? sh : 0) : sh;
        ew = (MODE == 2) ? (pw ? sw : 0) : sw;
        return (kd * 3 + kh) * 3 + kw;
    };

    // shared by both loop variants
    auto load_a = [&](float4 (&aq)[NF], int m, int eh_, int ew_) {
        const int sy = ((MODE == 1) ? 2 * rty : rty) + eh_;
        const int sx = ((MODE == 1) ? 2 * (rtx + 8 * m) : (rtx + 8 * m)) + ew_;
        if (PREC == 0) {
            const float *ap = &slab[(sy * SXP + sx) * CV_VS + 16 * half];
#pragma unroll
            for (int j = 0; j < 4; ++j) aq[j] = *reinterpret_cast<const float4 *>(ap + 4 * j);
        } else {
            const float *vp = &slab[(sy * SXP + sx) * VS];
            // part f>>1 at +16 dwords; inside a part, piece (K16 block f&1)*2 + half
            const float *ap0 = vp + (SWZ ? ((half ^ (sy & 3)) * 4) : 4 * half);
            const float *ap1 = vp + (SWZ ? (((2 + half) ^ (sy & 3)) * 4) : 4 * half + 8);
#pragma unroll
            for (int f = 0; f < NF; ++f)
                aq[f] = *reinterpret_cast<const float4 *>(((f & 1) ? ap1 : ap0) + (f >> 1) * 16);
        }
    };
    // bf16x6: an MFMA rounds (and floors what it shifts out of) its accumulator at the accumulator's magnitude,
    // so the running sums are kept short (az_common.h).  Two forms:
    //   * two waves per SIMD (NR = 1): every 16-deep K block is summed in a zero-initialised temporary and added
    //     to the accumulator with VALU adds (az_mfma6_now);
    //   * WIDE kernels (one wave per SIMD, > 256 registers: hipcc keeps temporaries in AGPRs, which VALU adds can
    //     only reach through v_accvgpr_read -- measured +29 % on the 64->64 layers): the MFMAs of one STAGE
    //     (slab: 9 taps x 2 blocks) chain into a per-stage temporary `ts`, which is added to the accumulator once
    //     per stage.  The chain inside a stage is 18 blocks long instead of the whole K (108 blocks for 64->64).
    f32x16 ts[MR][NR];  // (dead unless WIDE)
    auto mfma16 = [&](f32x16 (&c)[NR], const float4 (&aq)[NF], const float4 (&bw)[NR][NF]) {
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            if constexpr (PREC == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[j].x, bw[n][j].x, c[n], 0, 0, 0);
                    c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[j].y, bw[n][j].y, c[n], 0, 0, 0);
                    c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[j].z, bw[n][j].z, c[n], 0, 0, 0);
                    c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[j].w, bw[n][j].w, c[n], 0, 0, 0);
                }
            } else if constexpr (X16) {
                // one tap x 32 channels = two K16 blocks: hi*hi twice, then the four cross terms, from zero; one VALU
                // add of the block sum per element (the partner wave of the SIMD fills the gap behind the chain)
                typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define X16_MF(T, A, B) T = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, aq[A]), __builtin_bit_cast(h8, bw[n][B]), T, 0, 0, 0)
                f32x16 t;
#pragma unroll
                for (int e = 0; e < 16; ++e) t[e] = 0.f;
                X16_MF(t, 0, 0); X16_MF(t, 1, 1); X16_MF(t, 0, 2); X16_MF(t, 2, 0); X16_MF(t, 1, 3); X16_MF(t, 3, 1);
#undef X16_MF
                c[n] += t;
                asm volatile("" : "+v"(c[n]));
            } else {
                const float4 a0q[3] = {aq[0], aq[2], aq[4]}, a1q[3] = {aq[1], aq[3], aq[5]};
                const float4 b0q[3] = {bw[n][0], bw[n][2], bw[n][4]}, b1q[3] = {bw[n][1], bw[n][3], bw[n][5]};
                if (WIDE) {  // c = the stage temporary
                    az_mfma6(c[n], a0q, b0q);
                    az_mfma6(c[n], a1q, b1q);
                } else {  // two waves per SIMD at the register limit: one temporary, the partner wave fills the gap
                    az_mfma6_now(c[n], a0q, b0q);
                    az_mfma6_now(c[n], a1q, b1q);
                }
            }
        }
    };

    CV_T0();
    if (S0 < NS) issue(S0);
    CV_ACC(0);
    for (int s = S0; s < NS; ++s) {
        const int sd = s / NCH, cc = s - sd * NCH;
        const int kd = (MODE == 2) ? (pd ? 2 - 2 * sd : 1) : sd;
        const int id = plane_of(sd);
        __syncthreads();  // previous slab fully consumed (single-wave group: fence only)
        CV_T0();
        commit();
        CV_ACC(1);
        __syncthreads();
        float4 bq[NR][NF];
        int eh, ew;
        load_b(bq, tap_of(kd, 0, eh, ew), cc);
        CV_T0();
        if (s + 1 < NS) issue(s + 1);
        CV_ACC(2);
        CV_T0();
        if (id < 0 || id >= a.Di) continue;  // wave-uniform: a zero-padding plane
        // A fragments are read one MFMA block (16*NR instructions) ahead of their use:
        // tile 1's while tile 0 multiplies, the next tap's tile 0 while tile 1 multiplies.
        float4 a0[NF], a1[NF];
        load_a(a0, 0, eh, ew);
        // one tap: `cur` holds this tap's weights, `nxt` receives the next tap's (ping-pong
        // buffers, so the prefetch costs no register copies)
        auto tap_body = [&](int t, float4 (&cur)[NR][NF], float4 (&nxt)[NR][NF]) {
            int eh_, ew_, eh2 = 0, ew2 = 0;
            const int tap = tap_of(kd, t, eh_, ew_);
            if (MR == 2) load_a(a1, 1, eh_, ew_);
            if (t + 1 < ntaps) {
                const int tap2 = tap_of(kd, t + 1, eh2, ew2);
                if (BPIPE) load_b(nxt, tap2, cc);
            }
            if (!BPIPE && t > 0) load_b(cur, tap, cc);
            // hipcc otherwise sinks the LDS reads next to their consumers (register pressure)
            // and every MFMA quad then waits on a just-issued ds_read: pin the written order
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PREC == 1 && WIDE) mfma16(ts[0], a0, cur); else mfma16(acc[0], a0, cur);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < ntaps) load_a(a0, 0, eh2, ew2);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MR == 2) { if constexpr (PREC == 1 && WIDE) mfma16(ts[MR - 1], a1, cur); else mfma16(acc[MR - 1], a1, cur); }
            __builtin_amdgcn_sched_barrier(0);
        };
        if (PREC == 1 && WIDE) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int n = 0; n < NR; ++n)
#pragma unroll
                    for (int e = 0; e < 16; ++e) ts[m][n][e] = 0.f;
        }
        static_assert(!(X16 && WIDE), "f16x3 runs two waves per SIMD");
        if (BPIPE) {
            float4 bq2[NR][NF];
            for (int t = 0; t < ntaps; t += 2) {
                tap_body(t, bq, bq2);
                if (t + 1 < ntaps) tap_body(t + 1, bq2, bq);
            }
        } else {
            for (int t = 0; t < ntaps; ++t) tap_body(t, bq, bq);
        }
        if (PREC == 1 && WIDE) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int n = 0; n < NR; ++n) acc[m][n] += ts[m][n];
        }
        CV_ACC(3);
    }

    CV_T0();
    if (X16) {  // undo the operand scales once, on the finished sums
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int n = 0; n < NR; ++n) acc[m][n] *= osc;
    }
    // ---- epilogue ---------------------------------------------------------------------
    // C/D map of 32x32 MFMA: column (out channel) = lane & 31, row (voxel) =
    // (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    const int od = (MODE == 2) ? 2 * td + pd : td;
    // Addresses: wave-uniform 64-bit plane base + 32-bit offsets.  A tile that lies completely
    // inside the output (every interior tile) takes the branch-free path.
    const size_t plane_el = (((size_t)b * a.Do + od) * a.Ho) * a.Wo * COUT;
    float *outp = a.out + plane_el;
    const float *resp = a.res ? a.res + plane_el : nullptr;
    {
    const int ohs = (MODE == 2) ? 2 : 1;  // output step per index-space step
    const int phq = ph, pwq = pw;
    const int tile_q = tile_id, qb = 0;
    const int oh_base = (MODE == 2) ? 2 * ty0 + phq : ty0, ow_base = (MODE == 2) ? 2 * tx0 + pwq : tx0;
    const bool full = (oh_base + ohs * (TY - 1) < a.Ho) && (ow_base + ohs * (TX - 1) < a.Wo);
    // offset (in floats) of accumulator register r of M-tile m, and its validity
    auto voxel = [&](int m, int r, unsigned &off) -> bool {
        const int vrow = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int oh = oh_base + ohs * (vrow >> 3), ow = ow_base + ohs * ((vrow & 7) + 8 * m);
        off = (unsigned)(oh * a.Wo + ow) * COUT + row;
        return full || ((oh < a.Ho) && (ow < a.Wo));
    };
    if (EPI == 0) {
        float sc[NR], sf[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            sc[n] = a.scale ? a.scale[n * 32 + row] : 1.f;
            sf[n] = a.shift ? a.shift[n * 32 + row] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsigned off;
                if (!voxel(m, r, off)) continue;
#pragma unroll
                for (int n = 0; n < NR; ++n) {
                    float y = acc[qb + m][n][r] * sc[n] + sf[n];
                    if (resp) y += resp[off + n * 32];
                    if (a.relu) y = fmaxf(y, 0.f);
                    outp[off + n * 32] = y;
                }
            }
    } else {
        int nvalid = 0;
        unsigned okmask = 0;  // bit (m*16 + r)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsigned off;
                const bool ok = voxel(m, r, off);
                if (ok) {
                    okmask |= 1u << (m * 16 + r);
                    nvalid++;
#pragma unroll
                    for (int n = 0; n < NR; ++n) outp[off + n * 32] = acc[qb + m][n][r];
                }
            }
        const int ntot = nvalid + __shfl_xor(nvalid, 32);
#pragma unroll
        for (int n = 0; n < NR; ++n) {
            float s = 0.f, m2 = 0.f;
            if (full) {  // wave-uniform: no masking
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += acc[qb + m][n][r];
                s += __shfl_xor(s, 32);
                const float mean = s / (float)(MR * 32);
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float dlt = acc[qb + m][n][r] - mean;
                        m2 += dlt * dlt;
                    }
            } else {
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s += ((okmask >> (m * 16 + r)) & 1u) ? acc[qb + m][n][r] : 0.f;
                s += __shfl_xor(s, 32);
                const float mean = s / (float)max(ntot, 1);
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float dlt = acc[qb + m][n][r] - mean;
                        m2 += ((okmask >> (m * 16 + r)) & 1u) ? dlt * dlt : 0.f;
                    }
            }
            m2 += __shfl_xor(m2, 32);
            if (half == 0) {
                const int co = n * 32 + row;
                *reinterpret_cast<float2 *>(&a.part[((size_t)co * a.ntiles + tile_q) * 2]) = make_float2(s, m2);
            }
        }
        if (lane == 0) a.cnt[tile_q] = (float)ntot;
    }
    }  // phases
    CV_ACC(4);
    CV_FLUSH();
}

// ---- weight packing -----------------------------------------------------------------
// packed[((tap*NCH + cc)*NR + n)*1024 + j*256 + lane*4 + e] =
//     src[(n*32 + (lane&31))*sn + (cc*32 + 16*(lane>>5) + 4*j + e)*sk + (flip ? 26-tap : tap)]
__global__ void __launch_bounds__(256)
conv3d_pack_kernel(float *__restrict__ dst, const float *__restrict__ src, int cin, int cout,
                   long long sn, long long sk, int flip, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int e = idx & 3, lane = (idx >> 2) & 63, j = (idx >> 8) & 3;
    int r = idx >> 10;
    const int nr = cout / 32, nch = cin / 32;
    const int n = r % nr; r /= nr;
    const int cc = r % nch;
    const int tap = r / nch;
    const int co = n * 32 + (lane & 31);
    const int ci = cc * 32 + 16 * (lane >> 5) + 4 * j + e;
    dst[idx] = src[co * sn + ci * sk + (flip ? 26 - tap : tap)];
}

// bf16x6 packing: [tap][cc][n][part(3)][kb(2)][lane(64)][8] bf16, element j of lane =
// part `p` of src(co = n*32 + (lane&31), ci = cc*32 + 16*kb + 8*(lane>>5) + j, tap)
__global__ void __launch_bounds__(256)
conv3d_pack_x6_kernel(unsigned short *__restrict__ dst, const float *__restrict__ src, int cin,
                      int cout, long long sn, long long sk, int flip, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63;
    int r = idx >> 9;
    const int kb = r & 1; r >>= 1;
    const int p = r % 3; r /= 3;
    const int nr = cout / 32, nch = cin / 32;
    const int n = r % nr; r /= nr;
    const int cc = r % nch;
    const int tap = r / nch;
    const int co = n * 32 + (lane & 31);
    const int ci = cc * 32 + 16 * kb + 8 * (lane >> 5) + j;
    const float x = src[co * sn + ci * sk + (flip ? 26 - tap : tap)];
    dst[idx] = az_split3_part(x, p);  // round-to-nearest split, as the activations' (az_common.h)
}

// f16x3 packing for this kernel: [tap][cc][n][part(2)][kb(2)][lane(64)][8] fp16 of w * 2^k (k from max |w|)
__global__ void __launch_bounds__(256)
conv3d_pack_f16_kernel(unsigned short *__restrict__ dst, const float *__restrict__ src, const float *__restrict__ amax,
                       int cin, int cout, long long sn, long long sk, int flip, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const float scale = az_pow2(az_f16_scale_exp(az_amax_read(amax)));  // (before the early exit: a wave-wide read)
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63;
    int r = idx >> 9;
    const int kb = r & 1; r >>= 1;
    const int p = r & 1; r >>= 1;
    const int nr = cout / 32, nch = cin / 32;
    const int n = r % nr; r /= nr;
    const int cc = r % nch;
    const int tap = r / nch;
    const int co = n * 32 + (lane & 31);
    const int ci = cc * 32 + 16 * kb + 8 * (lane >> 5) + j;
    dst[idx] = az_split2_f16_part(src[co * sn + ci * sk + (flip ? 26 - tap : tap)] * scale, p);
}

template <int CIN, int COUT, int MODE, int EPI, int SRC, int PREC>
static int launch_conv(const ConvArgs &a, hipStream_t s) {
    long long blocks = (long long)a.B * a.Dt * a.tiles_y * a.tiles_x *
                       (MODE == 2 ? 8 : 1);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    hipLaunchKernelGGL((conv3d_gather_kernel<CIN, COUT, MODE, EPI, SRC, PREC>),
                       dim3((unsigned)blocks), dim3(64), 0, s, a);
    return az_launch_status();
}

// AZ_CONV_M128=0 keeps the 4x16-patch kernel for the bf16x6 stride-1 32-output-channel layers
static bool conv_m128_enabled() { return az_options().conv_m128 != 0; }

template <int MODE, int EPI, int PREC>
static int dispatch_channels(const ConvArgs &a, int cin, int cout, int src, hipStream_t s) {
    if (PREC == 1 && MODE == 0 && cout == 32 && conv_m128_enabled())
        return az_conv3d_m128_launch(a, cin, EPI, src, s);
    if ((PREC == 1 || PREC == 3) && MODE == 2 && cout == 32 && src == 0 && (cin == 32 || cin == 64))
        return az_conv3d_t2_launch(a, cin, EPI, s);  // (f16x3 when a.in_amax / a.w_amax are set)
    if (src == 1) {
        if (MODE != 0 || cin != 64) return AZ_EUNSUPPORTED;
        if (cout == 32) return launch_conv<64, 32, 0, EPI, 1, PREC>(a, s);
        return AZ_EUNSUPPORTED;
    }
    if (cin == 32 && cout == 32) return launch_conv<32, 32, MODE, EPI, 0, PREC>(a, s);
    if (cin == 64 && cout == 32) return launch_conv<64, 32, MODE, EPI, 0, PREC>(a, s);
    if (cin == 32 && cout == 64) return launch_conv<32, 64, MODE, EPI, 0, PREC>(a, s);
    if (cin == 64 && cout == 64) return launch_conv<64, 64, MODE, EPI, 0, PREC>(a, s);
    return AZ_EUNSUPPORTED;
}

template <int EPI>
static int dispatch_mode(const ConvArgs &a, int mode, int precision, int cin, int cout, int src,
                         hipStream_t s) {
    if (precision == 0) {
        if (mode == 0) return dispatch_channels<0, EPI, 0>(a, cin, cout, src, s);
        if (mode == 1) return dispatch_channels<1, EPI, 0>(a, cin, cout, src, s);
        return dispatch_channels<2, EPI, 0>(a, cin, cout, src, s);
    }
    if (precision == 1) {
        if (mode == 0) return dispatch_channels<0, EPI, 1>(a, cin, cout, src, s);
        if (mode == 1) return dispatch_channels<1, EPI, 1>(a, cin, cout, src, s);
        return dispatch_channels<2, EPI, 1>(a, cin, cout, src, s);
    }
    if (precision == 3) {  // f16x3 on this kernel (shapes the depth-rolling kernel does not take)
        if (src != 0) return AZ_EUNSUPPORTED;
        if (mode == 0) return dispatch_channels<0, EPI, 3>(a, cin, cout, src, s);
        if (mode == 1) return dispatch_channels<1, EPI, 3>(a, cin, cout, src, s);
        return dispatch_channels<2, EPI, 3>(a, cin, cout, src, s);
    }
    if (precision == 2) {  // bf16x6 arithmetic, weights packed for the depth-rolling 16x16x32 kernel
        if (mode != 0 || cout != 32 || src != 0) return AZ_EUNSUPPORTED;
        return az_conv3d_roll_launch(a, cin, EPI, s);
    }
    return AZ_EINVAL;
}

static void conv_out_dims(int mode, int Di, int Hi, int Wi, int &Do, int &Ho, int &Wo) {
    if (mode == 0) { Do = Di; Ho = Hi; Wo = Wi; }
    else if (mode == 1) { Do = (Di - 1) / 2 + 1; Ho = (Hi - 1) / 2 + 1; Wo = (Wi - 1) / 2 + 1; }
    else { Do = 2 * Di; Ho = 2 * Hi; Wo = 2 * Wi; }
}

static int conv_tiles(int mode, int Di, int Hi, int Wi, int &Dt, int &ty, int &tx) {
    int Do, Ho, Wo;
    conv_out_dims(mode, Di, Hi, Wi, Do, Ho, Wo);
    const int TX = (mode == 1) ? 8 : 16;
    if (mode == 2) { Dt = Di; ty = (Hi + 3) / 4; tx = (Wi + TX - 1) / TX; }
    else { Dt = Do; ty = (Ho + 3) / 4; tx = (Wo + TX - 1) / TX; }
    return 0;
}

#ifdef CV_STAMP
extern "C" int az_debug_conv_stamps(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(cv_stamp_sum), 8 * sizeof(unsigned long long)) != hipSuccess) return AZ_ELAUNCH;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(cv_stamp_sum), z, sizeof(z)) != hipSuccess) return AZ_ELAUNCH; }
    return AZ_OK;
}
#endif

extern "C" long long az_conv3d_num_tiles(int mode, int B, int Di, int Hi, int Wi) {
    if (mode < 0 || mode > 2 || B <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0) return AZ_EINVAL;
    int Dt, ty, tx;
    conv_tiles(mode, Di, Hi, Wi, Dt, ty, tx);
    return (long long)B * Dt * ty * tx * (mode == 2 ? 8 : 1);
}

extern "C" long long az_conv3d_packed_floats(int cin, int cout, int precision) {
    if (cin % 32 || cout % 32 || cin <= 0 || cout <= 0 || precision < 0 || precision > 2) return AZ_EINVAL;
    if (precision == 2 && cout != 32) return AZ_EINVAL;
    return precision == 0 ? 27LL * cin * cout : 27LL * cin * cout * 3 / 2;
}

extern "C" int az_conv3d_pack_weights(float *packed, const float *w, int cin, int cout,
                                      long long stride_out, long long stride_in, int flip,
                                      int precision, void *stream) {
    AZ_REQUIRE_PTR(packed); AZ_REQUIRE_PTR(w);
    if (cin % 32 || cout % 32 || cin <= 0 || cout <= 0) return AZ_EUNSUPPORTED;
    if (precision == 0) {
        const int total = 27 * cin * cout;
        hipLaunchKernelGGL(conv3d_pack_kernel, dim3((total + 255) / 256), dim3(256), 0,
                           az_stream(stream), packed, w, cin, cout, stride_out, stride_in, flip, total);
    } else if (precision == 1) {
        const int total = 27 * cin * cout * 3;
        hipLaunchKernelGGL(conv3d_pack_x6_kernel, dim3((total + 255) / 256), dim3(256), 0,
                           az_stream(stream), reinterpret_cast<unsigned short *>(packed), w, cin, cout,
                           stride_out, stride_in, flip, total);
    } else if (precision == 2) {
        return az_conv3d_pack_r16(packed, w, cin, cout, stride_out, stride_in, flip, az_stream(stream));
    } else {
        return AZ_EINVAL;
    }
    return az_launch_status();
}

static int conv_map_mode() { return az_options().conv_map; }

static int conv_common(ConvArgs &a, int mode, int B, int cin, int Di, int Hi, int Wi, int src) {
    if (mode < 0 || mode > 2 || B <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0) return AZ_EINVAL;
    a.B = B; a.Di = Di; a.Hi = Hi; a.Wi = Wi;
    a.map_mode = conv_map_mode();
    conv_out_dims(mode, Di, Hi, Wi, a.Do, a.Ho, a.Wo);
    conv_tiles(mode, Di, Hi, Wi, a.Dt, a.tiles_y, a.tiles_x);
    (void)cin; (void)src;
    return AZ_OK;
}

extern "C" int az_conv3d_fwd(float *out, const float *in, const float *in2,
                             const float *packed_w, const float *scale, const float *shift,
                             const float *residual, int relu, int mode, int src, int precision,
                             int B, int cin, int cout, int Di, int Hi, int Wi, void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(in); AZ_REQUIRE_PTR(packed_w);
    if (src == 1) AZ_REQUIRE_PTR(in2);
    ConvArgs a{};
    if (int e = conv_common(a, mode, B, cin, Di, Hi, Wi, src)) return e;
    a.in = in; a.in2 = in2; a.wp = packed_w; a.out = out;
    a.scale = scale; a.shift = shift; a.res = residual; a.relu = relu;
    return dispatch_mode<0>(a, mode, precision, cin, cout, src, az_stream(stream));
}

// rows of the partial buffers az_conv3d_fwd_stats fills when called with the same arguments (the depth-rolling
// kernel of precision 2 writes one entry per depth SEGMENT and tile, the others one per output depth and tile)
extern "C" long long az_conv3d_stats_tiles(int mode, int precision, int B, int cin, int cout, int Di, int Hi, int Wi) {
    if (precision == 2) {
        if (mode != 0 || cout != 32) return AZ_EUNSUPPORTED;
        ConvArgs a{};
        if (int e = conv_common(a, mode, B, cin, Di, Hi, Wi, 0)) return e;
        return az_conv3d_roll_stats_tiles(a);
    }
    return az_conv3d_num_tiles(mode, B, Di, Hi, Wi);
}

extern "C" int az_conv3d_fwd_stats(float *out, float *partials, float *counts, const float *in,
                                   const float *in2, const float *packed_w, int mode, int src,
                                   int precision, int B, int cin, int cout, int Di, int Hi,
                                   int Wi, void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(in); AZ_REQUIRE_PTR(packed_w);
    AZ_REQUIRE_PTR(partials); AZ_REQUIRE_PTR(counts);
    if (src == 1) AZ_REQUIRE_PTR(in2);
    ConvArgs a{};
    if (int e = conv_common(a, mode, B, cin, Di, Hi, Wi, src)) return e;
    a.in = in; a.in2 = in2; a.wp = packed_w; a.out = out; a.part = partials; a.cnt = counts;
    a.ntiles = az_conv3d_stats_tiles(mode, precision, B, cin, cout, Di, Hi, Wi);
    if (a.ntiles < 0) return (int)a.ntiles;
    return dispatch_mode<1>(a, mode, precision, cin, cout, src, az_stream(stream));
}

// ---- f16x3: the input-gradient launches (az_roll_common.h) -----------------------------------------------------------
extern "C" long long az_conv3d_packed_floats_f16(int cin, int cout) {
    if (cin % 32 || cout % 32 || cin <= 0 || cout <= 0) return AZ_EINVAL;
    return 27LL * cin * cout;  // two fp16 parts per weight
}

// which kernel serves (mode, cout) in f16x3 -- the two read different packed-weight layouts
// the stride-1 layers with 32 output channels on the depth-rolling kernel of az_conv3d_roll.hip; since round 5 the 64 -> 64
// ones too, as two workgroups per patch (AZ_CONV_ROLL64=0: this file's gather kernel, 0.61 ms at V1 against 0.4x)
static bool f16_on_roll64(int mode, int cin, int cout) { return mode == 0 && cin == 64 && cout == 64 && az_options().conv_roll64 != 0; }
static bool f16_on_roll(int mode, int cin, int cout) { return (mode == 0 && cout == 32) || f16_on_roll64(mode, cin, cout); }
// the transposed 64 -> 32 layers on the depth-rolling kernel of az_conv3d_t2roll.hip (AZ_CONV_T2ROLL=0: az_conv3d_t2.hip)
static bool f16_on_t2roll(int mode, int cin, int cout) { return mode == 2 && cin == 64 && cout == 32 && az_options().conv_t2roll != 0; }
// the stride-2 32 -> 64 layers on the depth-rolling kernel of az_conv3d_s2roll.hip (AZ_CONV_S2ROLL=0: the gather kernel here)
static bool f16_on_s2roll(int mode, int cin, int cout) { return mode == 1 && cin == 32 && cout == 64 && az_options().conv_s2roll != 0; }
static bool f16_roll_layout(int mode, int cin, int cout) { return f16_on_roll(mode, cin, cout) || f16_on_t2roll(mode, cin, cout) || f16_on_s2roll(mode, cin, cout); }

// the layout az_conv3d_pack_weights_f16 writes for (mode, cin, cout): AZ_PACK_3D_ROLL or AZ_PACK_3D_GATHER (az_pack_f16_multi's `kind`)
extern "C" int az_conv3d_f16_layout(int mode, int cin, int cout) {
    if (cin % 32 || cout % 32 || cin <= 0 || cout <= 0 || mode < 0 || mode > 2) return AZ_EUNSUPPORTED;
    if (f16_on_roll64(mode, cin, cout)) return AZ_PACK_3D_ROLL2;
    return f16_roll_layout(mode, cin, cout) ? AZ_PACK_3D_ROLL : AZ_PACK_3D_GATHER;
}

extern "C" int az_conv3d_pack_weights_f16(float *packed, const float *w, const float *w_amax, int cin, int cout,
                                          long long stride_out, long long stride_in, int flip, int mode, void *stream) {
    AZ_REQUIRE_PTR(packed); AZ_REQUIRE_PTR(w); AZ_REQUIRE_PTR(w_amax);
    if (cin % 32 || cout % 32 || cin <= 0 || cout <= 0 || mode < 0 || mode > 2) return AZ_EUNSUPPORTED;
    if (f16_roll_layout(mode, cin, cout))
        return az_conv3d_pack_r16_f16(packed, w, w_amax, cin, cout, stride_out, stride_in, flip, az_stream(stream),
                                      f16_on_roll64(mode, cin, cout) ? 2 : 1);
    const int total = 27 * cin * cout * 2;
    hipLaunchKernelGGL(conv3d_pack_f16_kernel, dim3((total + 255) / 256), dim3(256), 0, az_stream(stream),
                       reinterpret_cast<unsigned short *>(packed), w, w_amax, cin, cout, stride_out, stride_in, flip, total);
    return az_launch_status();
}

static int conv_f16_dispatch(ConvArgs &a, int mode, int cin, int cout, int epi, hipStream_t s) {
    if (f16_on_roll(mode, cin, cout)) return az_conv3d_roll_launch_f16(a, cin, epi, s, cout);
    if (f16_on_t2roll(mode, cin, cout)) {
        const int rc = az_conv3d_t2roll_launch(a, epi, s);
        if (rc != AZ_EUNSUPPORTED) return rc;
        return AZ_EUNSUPPORTED;  // (the weights are packed for that kernel: the caller takes the bf16x6 route)
    }
    if (f16_on_s2roll(mode, cin, cout)) return az_conv3d_s2roll_launch(a, epi, s);  // (AZ_EUNSUPPORTED: as above)
    return epi ? dispatch_mode<1>(a, mode, 3, cin, cout, 0, s) : dispatch_mode<0>(a, mode, 3, cin, cout, 0, s);
}

// 1: an az_conv3d_fwd_f16 launch of this shape may read a pre-split input (the depth-rolling kernels, which stage by copy,
// under the size conditions of their launchers); 0: it needs the fp32 tensor
extern "C" int az_conv3d_fwd_f16_split_ok(int mode, int B, int cin, int cout, int Di, int Hi, int Wi) {
    if (mode < 0 || mode > 2 || B <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0) return 0;
    if (f16_on_roll(mode, cin, cout))
        return (cin == 32 || cin == 64) && az_fits_buffer_offset((long long)Di * Hi * Wi * (cin > cout ? cin : cout) * 4) ? 1 : 0;
    if (f16_on_t2roll(mode, cin, cout))
        return az_fits_buffer_offset(8LL * Di * Hi * Wi * 32 * 4) && az_fits_buffer_offset((long long)Di * Hi * Wi * 64 * 4) ? 1 : 0;
    if (f16_on_s2roll(mode, cin, cout)) return az_fits_buffer_offset((long long)Di * Hi * Wi * 32 * 4) ? 1 : 0;
    return 0;
}

extern "C" int az_conv3d_fwd_f16(float *out, const float *in, const float *packed_w, const float *in_amax,
                                 const float *w_amax, int in_split, const float *scale, const float *shift,
                                 const float *residual, int relu, int mode, int B, int cin, int cout, int Di, int Hi, int Wi,
                                 void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(in); AZ_REQUIRE_PTR(packed_w); AZ_REQUIRE_PTR(in_amax); AZ_REQUIRE_PTR(w_amax);
    ConvArgs a{};
    if (int e = conv_common(a, mode, B, cin, Di, Hi, Wi, 0)) return e;
    if (in_split && !az_conv3d_fwd_f16_split_ok(mode, B, cin, cout, Di, Hi, Wi)) return AZ_EUNSUPPORTED;
    a.in = in; a.wp = packed_w; a.out = out; a.scale = scale; a.shift = shift; a.res = residual; a.relu = relu;
    a.in_amax = in_amax; a.w_amax = w_amax; a.in_split = in_split ? 1 : 0;
    return conv_f16_dispatch(a, mode, cin, cout, 0, az_stream(stream));
}

extern "C" long long az_conv3d_stats_tiles_f16(int mode, int B, int cin, int cout, int Di, int Hi, int Wi) {
    if (mode < 0 || mode > 2) return AZ_EINVAL;
    if (f16_on_roll64(mode, cin, cout)) {
        ConvArgs a{};
        if (int e = conv_common(a, mode, B, cin, Di, Hi, Wi, 0)) return e;
        return az_conv3d_roll_stats_tiles(a, 64);
    }
    if (f16_on_roll(mode, cin, cout)) return az_conv3d_stats_tiles(mode, 2, B, cin, cout, Di, Hi, Wi);
    if (f16_on_t2roll(mode, cin, cout)) {
        ConvArgs a{};
        if (int e = conv_common(a, mode, B, cin, Di, Hi, Wi, 0)) return e;
        return az_conv3d_t2roll_stats_tiles(a);
    }
    if (f16_on_s2roll(mode, cin, cout)) {
        ConvArgs a{};
        if (int e = conv_common(a, mode, B, cin, Di, Hi, Wi, 0)) return e;
        return az_conv3d_s2roll_stats_tiles(a);
    }
    return az_conv3d_num_tiles(mode, B, Di, Hi, Wi);
}

extern "C" int az_conv3d_fwd_stats_f16(float *out, float *partials, float *counts, const float *in, const float *packed_w,
                                       const float *in_amax, const float *w_amax, int mode, int B, int cin, int cout,
                                       int Di, int Hi, int Wi, void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(in); AZ_REQUIRE_PTR(packed_w); AZ_REQUIRE_PTR(in_amax); AZ_REQUIRE_PTR(w_amax);
    AZ_REQUIRE_PTR(partials); AZ_REQUIRE_PTR(counts);
    ConvArgs a{};
    if (int e = conv_common(a, mode, B, cin, Di, Hi, Wi, 0)) return e;
    a.in = in; a.wp = packed_w; a.out = out; a.part = partials; a.cnt = counts; a.in_amax = in_amax; a.w_amax = w_amax;
    a.ntiles = az_conv3d_stats_tiles_f16(mode, B, cin, cout, Di, Hi, Wi);
    if (a.ntiles < 0) return (int)a.ntiles;
    return conv_f16_dispatch(a, mode, cin, cout, 1, az_stream(stream));
}
