// K4'' (bf16x6, stride 1, 32 output channels) -- the V0 layers of the cost aggregation and their input
// gradients (nets/psmnet/psmnet_3.py:87-117 dres0..dres4 / classif first convs; psmnet_submodule_3.py:44-56),
// re-tiled around what az_conv3d_m128.hip's counters and ablations showed (profiles/r02_clock_pipe_ablation.md):
// that kernel splits every input plane three times (once per output depth that reads it, 19 % of its time),
// pays 16 accumulate-adds per six MFMAs, and holds 1.75-1.86 GHz on the 32x32x16 shape.
//
//   * DEPTH ROLLING.  A workgroup owns an 8x16 (y, x) patch and WALKS a depth segment: every input plane is
//     fetched, split into its bf16 triplet and written to LDS ONCE, and feeds the three output depths that read
//     it (kd = 0, 1, 2) from three live accumulator sets.  Planes staged per output: (L + 2) / L instead of 3.
//   * v_mfma_f32_16x16x32_bf16.  K = 32 per instruction = one 32-channel chunk per tap: one accumulate-add per
//     element per 32-deep block (half the adds of the 32x32x16 form, and one accumulator rounding per K32 block
//     instead of per K16 block); the chip holds a higher clock on this shape (MI355X_MICROARCH.md, DVFS item 7).
//   * FOUR WAVES = 2 halves of the output channels x 2 halves of the patch.  A wave accumulates 16 channels x 64
//     voxels x 3 depth slots = 48 registers, so the kernel keeps 2 waves per SIMD with room for the software
//     pipeline (the first version of this file gave a wave 128 voxels: 96 accumulator registers, the prefetched
//     slab spilled to scratch, and every reload waited -- vmcnt is one in-order counter -- for the output
//     stores' acknowledgements: 40 % of the kernel was stage-boundary time, profiles/r03_roll_kernel_notes.md).
//   * LDS slab: 10 x 18 voxels x [3 parts][32 ch] bf16 = 192 B per voxel, 34.5 KB, DOUBLE-BUFFERED (69 KB per
//     workgroup, two workgroups per CU): the next plane is split and written while this one is multiplied, in
//     the shadow of the MFMAs; one barrier per stage.  An M tile is a 4 x 4 voxel square: lane l reads voxel
//     (l & 15) = (row l>>2 & 3, x l & 3), channel octet l >> 4.  Four x-adjacent voxels step through the four
//     64-B bank quarters (192 B = 3 quarters) and the octet index is XORed with 2 on odd slab rows, so that the
//     16-lane groups of a ds_read_b128 touch every bank exactly once for ANY tap shift (SQ_LDS_BANK_CONFLICT 0).
//   * A STAGE (one plane x one 32-channel chunk: 27 taps x 4 tiles x 6 MFMAs per wave) IS STRAIGHT-LINE CODE:
//     all three kd are always computed (at the two ends of a segment a slot then holds a partial sum that is
//     never stored: 4 % more MFMAs at 48 planes), the epilogue of the finished depth sits inside the stage with
//     lane validity expressed through buffer-instruction bounds (no branch), so hipcc counts vmcnt exactly:
//     nothing waits for the slab prefetch or for store acknowledgements before it has to.  One A fragment read
//     from LDS feeds the three kd (18 MFMAs).
//
// Arithmetic: az_common.h's bf16x6 product (exact 3-way RNE split, six MFMAs per block summed from zero,
// largest terms first, one VALU add per block into the fp32 accumulator).  Accumulation order per output:
// input plane (kd) ascending, 32-channel chunk, kh, kw -- the order of az_conv3d_m128.hip with K32 blocks.
#include <type_traits>

#include "az_conv3d_args.h"

#include "az_roll_common.h"
#include "az_options.h"
#include "az_launch_math.h"

// Diagnostic build only (-DR16_STAMP): shader cycles per wave, summed into a buffer nothing else reads
// (tools/roll_stamp_probe.py): [0] prologue, [1] stage bodies, [2] stage-end barriers, [3] tail, [8] kernel, [9] waves.
#ifndef R16H_ABL
#define R16H_ABL 0  // timing-only builds of the f16x3 stage: 1 no slab staging, 2 no weight refills, 4 no output stores, 8 no fragment refills
#endif
#ifdef R16_STAMP
__device__ unsigned long long r16_stamp_sum[10];
extern "C" int az_debug_roll_stamps(unsigned long long *out10, int reset) {
    if (hipMemcpyFromSymbol(out10, HIP_SYMBOL(r16_stamp_sum), sizeof(r16_stamp_sum)) != hipSuccess) return AZ_ELAUNCH;
    if (reset) { unsigned long long z[10] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(r16_stamp_sum), z, sizeof(z)) != hipSuccess) return AZ_ELAUNCH; }
    return AZ_OK;
}
#define R16_T(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_[i] += now_ - tl_; tl_ = now_; } while (0)
#else
#define R16_T(i) do { } while (0)
#endif

// EPI: 0 = y = relu?(acc * scale + shift), 2 = the same + residual, 1 = raw output + BatchNorm partials
// AR: 0 = bf16x6 (three bf16 parts, six MFMAs per block), 1 = f16x3 (two scaled fp16 parts, three MFMAs: input
// gradients only, az_roll_common.h); the slab keeps its 192-byte voxels either way (the third 64-byte part is then
// unused), so addresses and the conflict-free swizzle are shared.
// PS (AR = 1 only): the input is a pre-split tensor (az_roll_common.h): staging copies its two fp16 parts.
// COUT = 64 (f16x3, round 5: the 64 -> 64 layers at V1 / V2 and their input gradients): TWO workgroups per patch and
// segment, one per half of the output channels -- adjacent in the launch order, so the second finds the planes the first
// fetched in L2; the packed weights hold the two halves one after the other (AZ_PACK_3D_ROLL2), each in the 32-channel
// layout, the output / residual voxels are 256 bytes apart and every channel index carries the half's 32.
template <int CIN, int EPI, int AR = 0, bool PS = false, int COUT = 32>
__global__ void __launch_bounds__(256, 2)
conv3d_roll_kernel(const ConvArgs a) {
    static_assert(COUT == 32 || (COUT == 64 && AR == 1), "64 output channels: the f16x3 form only");
    constexpr unsigned OVB = COUT * 4u;      // bytes per output voxel
    constexpr int NCH = CIN / 32;            // 32-channel chunks per plane
    constexpr int NP = AR ? 2 : 3;           // parts per operand
    constexpr int TAPF4 = NCH * 2 * NP * 64; // float4 per tap in the packed image: [tap][cc][n16][part][lane]

    __shared__ __attribute__((aligned(16))) unsigned char slab[2 * R_SLAB_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wv & 1, wm = wv >> 1;  // half of the output channels, half (4 rows) of the patch
#ifdef R16_STAMP
    unsigned long long st_[4] = {0, 0, 0, 0};
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
    unsigned long long tl_ = t0_;
#endif

    // ---- block -> (batch, depth segment, patch): contiguous chunk of the linear order per XCD, x fastest,
    //      so that the workgroups resident on an XCD are a compact (y, x) region walking the same depths ----
    int lin = a.map_mode >= 1 ? az_xcd_map(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int co0 = COUT == 64 ? (lin & 1) * 32 : 0;  // first output channel of this workgroup
    if (COUT == 64) lin >>= 1;
    const int tyb = (a.tiles_y + 1) >> 1;  // 8-row patch rows (a.tiles_y counts 4-row tiles)
    int tix, tiy, seg, b;
    az_roll_decode(lin, a.tiles_x, tyb, a.nseg, tix, tiy, seg, b);
    const int d0 = seg * a.seg_len, d1 = min(d0 + a.seg_len, a.Do);  // outputs [d0, d1)
    const int ty0 = tiy * R_TY, tx0 = tix * R_TX;
    const int ih0 = ty0 - 1, iw0 = tx0 - 1;

    // buffer resources: lane validity (zero padding of the input, patches that overhang the volume, depths
    // outside the segment) is an out-of-range offset, never a branch
    const unsigned in_bytes = (unsigned)a.Di * a.Hi * a.Wi * CIN * 4u, out_bytes = (unsigned)a.Do * a.Ho * a.Wo * OVB;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in) + (size_t)b * (in_bytes / 4), 0, in_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)b * (out_bytes / 4), 0, out_bytes, 0x00020000);
    const auto rs_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(EPI == 2 ? a.res : a.out) + (size_t)b * (out_bytes / 4), 0, EPI == 2 ? out_bytes : 0u, 0x00020000);
    const auto rs_part = __builtin_amdgcn_make_buffer_rsrc(EPI == 1 ? a.part : a.out, 0,
                                                           EPI == 1 ? (unsigned)(a.ntiles * COUT * 2 * 4) : 0u, 0x00020000);
    const auto rs_cnt = __builtin_amdgcn_make_buffer_rsrc(EPI == 1 ? a.cnt : a.out, 0,
                                                          EPI == 1 ? (unsigned)(a.ntiles * 4) : 0u, 0x00020000);

    // f16x3: power-of-two operand scales from the tensors' largest magnitudes (device scalars)
    float in_scale = 1.f;
    int out_exp = 0;
    if (AR) {
        // (wave-uniform: kept in scalar registers)
        const int ki = az_f16_scale_exp(az_amax_read(a.in_amax));
        const int kw = az_f16_scale_exp(az_amax_read(a.w_amax));
        in_scale = az_pow2(ki);
        out_exp = -(ki + kw);
    }

    f32x4 acc[3][4];  // [slot: kd = 0 -> output p+1, 1 -> p, 2 -> p-1][4x4-voxel tile of this wave's 4x16 half patch]
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[s][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging: one plane chunk = 10 x 18 voxels x 32 channels fp32 -> bf16 triplets in LDS -------------
    // piece idx = tid + 256 it: voxel idx >> 3 (sy = voxel / 18, sx = voxel % 18), channels 4 (idx & 7) ..
    u32x4 pre[R_NLD];
    auto issue = [&](int p, int cc, int it0 = 0, int it1 = R_NLD) {  // pieces [it0, it1) (static)
        int tid = threadIdx.x;
        // (f16x3: at its register limit -- recompute the piece offsets from the thread index at every call instead of
        //  keeping six hoisted offsets and their validity masks alive across the stage)
        if (AR) asm volatile("" : "+v"(tid));
        int sy = 0, sx = tid >> 3;  // (tid >> 3 is 0..31)
        if (sx >= R_SX) { sx -= R_SX; ++sy; }
#pragma unroll
        for (int it = 0; it < R_NLD; ++it) {
            const int ih = ih0 + sy, iw = iw0 + sx;
            const bool ok = (tid + 256 * it < R_NQ) && (unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi &&
                            (unsigned)p < (unsigned)a.Di;
            const unsigned off = (unsigned)((p * a.Hi + ih) * a.Wi + iw) * (CIN * 4) + cc * 128 + (tid & 7) * 16;
            if (it >= it0 && it < it1) pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? off : R_OOB, 0, 0);
            sx += 14; ++sy;  // 32 voxels on = one slab row + 14
            if (sx >= R_SX) { sx -= R_SX; ++sy; }
        }
    };
    auto commit_piece = [&](int it, unsigned char *dstbuf) {
        // (sy, sx) of piece `it`: static `it`, so this is a handful of integer instructions on tid.  The last
        // piece exists for 160 of the 256 threads only; the others re-write their previous piece (same bytes, same
        // place) instead of branching, so that a stage stays one basic block.
        int tid = threadIdx.x;
        if (AR) asm volatile("" : "+v"(tid));  // (see issue)
        const bool live = (tid + 256 * it < R_NQ);
        const int ite = (it > 0 && !live) ? it - 1 : it;
        const int vox = (tid >> 3) + 32 * ite;
        const int sy = vox / R_SX, sx = vox - sy * R_SX;
        const int j = tid & 7;  // channels 4j..4j+3: octet j >> 1, 8-byte half j & 1
        u32x4 raw = pre[it];
        if (it > 0 && 256 * it + 255 >= R_NQ) {  // (static: only the last piece)
            raw[0] = live ? raw[0] : pre[it - 1][0]; raw[1] = live ? raw[1] : pre[it - 1][1];
            raw[2] = live ? raw[2] : pre[it - 1][2]; raw[3] = live ? raw[3] : pre[it - 1][3];
        }
        unsigned char *dst = dstbuf + (sy * R_SX + sx) * R_VB + ((((j >> 1) ^ ((sy & 1) << 1))) << 4) + (j & 1) * 8;
        if (AR) {
            uint2 hi, lo;
            az_stage_f16x4<PS>(raw, in_scale, hi, lo);
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

    // ---- operands ------------------------------------------------------------------------------------------
    // A: lane -> voxel (row (lane >> 2) & 3, x lane & 3) of a 4x4 tile, channel octet lane >> 4; the octet
    // swizzle depends on the slab row's parity = (kh + tile row) & 1 (this wave's tile rows start at 4 wm: even)
    const int trow = (lane >> 2) & 3, tcol = lane & 3, oct = lane >> 4;
    unsigned abase[2];
    abase[0] = ((4 * wm + trow) * R_SX + tcol) * R_VB + ((oct ^ ((trow & 1) << 1)) << 4);
    abase[1] = ((4 * wm + trow) * R_SX + tcol) * R_VB + ((oct ^ (((trow + 1) & 1) << 1)) << 4);
    // B: packed [tap][cc][n16][part][lane] float4 (conv3d_pack_r16_kernel), this wave's 16 output channels.  Read
    // through a buffer resource: one lane-offset register, the tap's byte offset travels in an SGPR / the immediate
    // (with flat addresses hipcc hoists the 27 x 64-bit tap addresses out of the walk and spills them)
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.wp) + (size_t)(co0 / 32) * (27u * CIN * 32u * 2u * NP / 4u), 0,
                                                        27u * CIN * 32u * 2u * NP, 0x00020000);
    const unsigned wlane = (unsigned)(wn * NP * 64 + lane) * 16u;
    auto load_b = [&](float4 (&bq)[NP], int tap_f4) {  // tap_f4: float4 index of the tap's first fragment (static)
#pragma unroll
        for (int p = 0; p < NP; ++p)
            bq[p] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, tap_f4 * 16 + p * 1024, 0));
    };
    // per-channel epilogue constants, loaded once (four consecutive channels per lane after the quad transpose)
    const int cq = co0 + wn * 16 + (lane & 12);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
    if (EPI != 1) {
        if (a.scale) sc = *reinterpret_cast<const float4 *>(a.scale + cq);
        if (a.shift) sf = *reinterpret_cast<const float4 *>(a.shift + cq);
        if (AR) { sc.x = ldexpf(sc.x, out_exp); sc.y = ldexpf(sc.y, out_exp); sc.z = ldexpf(sc.z, out_exp); sc.w = ldexpf(sc.w, out_exp); }
    }
    const float floor_ = a.relu ? 0.f : -__builtin_inff();

    // validity of the 16 accumulator elements of every lane for the BatchNorm partials -- the same for every depth: the
    // lane's row is inside the volume and element e = 4 m + r (its x offset in the patch) is left of the right edge,
    // i.e. e < st_nv with ONE per-lane count (sixteen 64-bit lane masks in SGPRs made hipcc spill scalars into the stage)
    int st_nv = 0;
    if (EPI == 1) st_nv = (ty0 + 4 * wm + (lane >> 4) < a.Ho) ? min(max(a.Wo - tx0, 0), 16) : 0;
    // BatchNorm partials (EPI 1): ONE entry per channel for the wave's 4x16 half patch over the whole depth segment.
    // Every lane keeps running sums of its 16 accumulator elements about a shift of its own (the first valid value
    // it sees: shifted-data algorithm, no cancellation in sum-of-squares minus squared-sum) -- 64 VALU instructions per
    // finished plane and no cross-lane traffic; lanes and row groups are merged once, after the walk (Chan's formula).
    // The first version reduced every plane to a (sum, centred M2) entry of its own (~100 VALU and four cross-lane sums
    // per plane, 9 spilled registers, 97 920 partial rows per V0 layer for the finalize kernels to merge; this form: no
    // spills, 4 080 rows, no pre-merge launch).  Both cost nothing measurable in the kernel itself: 1.42 ms with or
    // without statistics (profiles/r03_roll_kernel_notes.md section 6 -- the 0.2 ms "epilogue gap" quoted earlier was
    // the timing script measuring this kernel first, before the clocks had settled).
    const int st_nl = st_nv;  // this lane's valid elements per plane
    float st_k = 0.f, st_s1 = 0.f, st_s2 = 0.f;
    int st_planes = 0;
    bool st_first = true;

    // ---- epilogue of a finished output depth (slot 2); `ok`: the depth belongs to this segment ---------------
    auto finish = [&](int o, bool ok) {
        const int oh = ty0 + 4 * wm + (lane >> 4);  // row of the 4x4 tiles this lane's accumulator registers belong to
        const bool row_ok = ok && oh < a.Ho;
        const unsigned row_off = (unsigned)((o * a.Ho + oh) * a.Wo) * OVB + (unsigned)cq * 4u;
        if (EPI != 1) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int ow = tx0 + 4 * m + (lane & 3);
                const f32x4 v = r16_quad_transpose(acc[2][m], lane);
                const unsigned off = (row_ok && ow < a.Wo) ? row_off + (unsigned)ow * OVB : R_OOB;
                float4 y = make_float4(v[0] * sc.x + sf.x, v[1] * sc.y + sf.y, v[2] * sc.z + sf.z, v[3] * sc.w + sf.w);
                if (EPI == 2) {
                    const float4 rr = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, off, 0, 0));
                    y.x += rr.x; y.y += rr.y; y.z += rr.z; y.w += rr.w;
                }
                y.x = fmaxf(y.x, floor_); y.y = fmaxf(y.y, floor_); y.z = fmaxf(y.z, floor_); y.w = fmaxf(y.w, floor_);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), rs_out, off, 0, 0);
            }
        } else {
            // raw output; BatchNorm running sums (see st_k above)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x4 vt = r16_quad_transpose(acc[2][m], lane);
                const int owt = tx0 + 4 * m + (lane & 3);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vt), rs_out,
                                                       (row_ok && owt < a.Wo) ? row_off + (unsigned)owt * OVB : R_OOB, 0, 0);
            }
            st_k = (ok && st_first) ? acc[2][0][0] : st_k;  // (element (0,0) is valid whenever any of the lane's is)
            st_first = st_first && !ok;
            st_planes += ok ? 1 : 0;
            const int nv_p = ok ? st_nv : 0;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dlt = (m * 4 + r < nv_p) ? acc[2][m][r] - st_k : 0.f;
                    st_s1 += dlt;
                    st_s2 = fmaf(dlt, dlt, st_s2);
                }
        }
    };

    // after the walk: merge the lanes of a channel (the four row groups of the 16x16 C layout) and write the entry
    auto flush_stats = [&]() {
        float n = (float)(st_nl * st_planes);
        float mean = n > 0.f ? st_k + st_s1 / n : 0.f;
        float m2 = n > 0.f ? st_s2 - st_s1 * st_s1 / n : 0.f;
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            const float n_o = __shfl_xor(n, off), mean_o = __shfl_xor(mean, off), m2_o = __shfl_xor(m2, off);
            const float nn = n + n_o;
            const float dlt = mean_o - mean;
            const float w_o = nn > 0.f ? n_o / nn : 0.f;
            m2 = m2 + m2_o + dlt * dlt * n * w_o;
            mean = mean + dlt * w_o;
            n = nn;
        }
        const int tiy4 = 2 * tiy + wm;
        const bool tile_ok = tiy4 < a.tiles_y;
        const unsigned tile_id = (unsigned)(((b * a.nseg + seg) * a.tiles_y + tiy4) * a.tiles_x + tix);
        const unsigned ch = co0 + wn * 16 + (lane & 15);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, make_float2(n * mean, fmaxf(m2, 0.f))), rs_part,
                                              (tile_ok && lane < 16 && !R16_NOPART) ? (unsigned)(((size_t)ch * a.ntiles + tile_id) * 8) : R_OOB, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, n), rs_cnt,
                                              (tile_ok && lane == 0 && wn == 0 && co0 == 0) ? tile_id * 4u : R_OOB, 0, 0);
    };

    auto rotate = [&]() {  // what was output p (slot 1) becomes output (p+1) - 1 of the next plane, ...
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[2][m] = acc[1][m];
            acc[1][m] = acc[0][m];
            acc[0][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    // ---- one stage: plane p, chunk CC, slab in buffer `buf`; straight-line -------------------------------------
    // order: (kh, kw) outer, tile, kd inner: one A fragment load feeds the 18 MFMAs of the three kd (a third of the
    // LDS reads of a kd-outer order: 1.47 -> 1.435 ms); the three kd weights of a (kh, kw) are double-buffered in
    // registers, requested one (kh, kw) ahead (72 MFMAs).  Slot 2 is complete only at the end of the stage: its
    // epilogue (and the rotation) open the NEXT plane's first stage, and its stores have that whole stage to land.
    float4 wk[2][3][NP];
    f32x4 tq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    auto stage_s = [&](auto cc_tag, int p, int buf) {
        constexpr int CC = decltype(cc_tag)::value;
        constexpr bool LAST = (CC == NCH - 1);
        constexpr int CCN = (CC + 1) % NCH;
        const unsigned char *sl = slab + buf * R_SLAB_BYTES;
        unsigned char *sn = slab + (buf ^ 1) * R_SLAB_BYTES;
        constexpr int wcur = CC * (2 * NP * 64), wnxt = CCN * (2 * NP * 64);
        const int pn = LAST ? p + 1 : p;
        if (CC == 0) {
            finish(p - 2, p - 2 >= d0);
            rotate();
        }
        float4 av[2][NP];
        auto load_a = [&](float4 (&aq)[NP], int m, int kh, int kw) {
            const unsigned char *ap = sl + abase[kh & 1] + (kh * R_SX + 4 * m + kw) * R_VB;
#pragma unroll
            for (int q = 0; q < NP; ++q) aq[q] = *reinterpret_cast<const float4 *>(ap + 64 * q);
        };
        load_a(av[0], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int kh = j / 3, kw = j % 3;
            __builtin_amdgcn_sched_barrier(0);
            // the three kd weights of the next (kh, kw) -- of the next stage's first one at the end
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
                load_b(wk[(j + 1) & 1][kd], (j + 1 < 9 ? wcur + (kd * 9 + j + 1) * TAPF4 : wnxt + (kd * 9) * TAPF4));
            if (j == 0) {
                __builtin_amdgcn_sched_barrier(0);
                issue(pn, CCN);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int t = j * 4 + m;  // tile step of the stage: A buffers alternate
                __builtin_amdgcn_sched_barrier(0);
                if (m + 1 < 4) load_a(av[(t + 1) & 1], m + 1, kh, kw);
                else if (j + 1 < 9) load_a(av[(t + 1) & 1], 0, (j + 1) / 3, (j + 1) % 3);
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const int st = t * 3 + kd;  // step of the stage: temporaries alternate
                    __builtin_amdgcn_sched_barrier(0);
                    // the temporary carried in belongs to the step before: (m, kd-1), (m-1, 2) or the previous
                    // (kh, kw)'s (3, 2); the stage starts with a zero temporary
                    f32x4 &prev = kd > 0 ? acc[kd - 1][m] : (m > 0 ? acc[2][m - 1] : acc[2][3]);
                    if constexpr (AR == 0) r16_step(tq[st & 1], av[t & 1], wk[j & 1][kd], prev, tq[(st + 1) & 1]);
                }
                if (j >= 5 && j < 8 && (m & 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    commit_piece((j - 5) * 2 + (m >> 1), sn);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[2][3] += tq[1];  // the stage's last temporary (108 steps: the last one wrote tq[1])
        tq[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        // nine (kh, kw) per stage: the buffer the next stage's first weights landed in becomes buffer 0
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int q = 0; q < NP; ++q) wk[0][kd][q] = wk[1][kd][q];
        R16_T(1);
        __syncthreads();
        R16_T(2);
    };

    // ---- f16x3 (AR = 1): epilogue and stage ------------------------------------------------------------------
    // accumulator layout (r16_chain9): lane = voxel (row (lane >> 2) & 3, x lane & 3) of the 4x4 tile, registers = the
    // four channels 16 wn + 4 (lane >> 4) + r: no transpose.  y = acc * 2^out_exp (+ residual); no affine map, no ReLU
    // (an input gradient has neither).
    const int cqh = co0 + wn * 16 + 4 * (lane >> 4);
    float4 sch = make_float4(1.f, 1.f, 1.f, 1.f), sfh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (AR) {
        if (EPI != 1) {
            if (a.scale) sch = *reinterpret_cast<const float4 *>(a.scale + cqh);
            if (a.shift) sfh = *reinterpret_cast<const float4 *>(a.shift + cqh);
        }
        // the power-of-two unscaling rides on the affine map (out_exp beyond the float range: the true result is, too)
        sch.x = ldexpf(sch.x, out_exp); sch.y = ldexpf(sch.y, out_exp); sch.z = ldexpf(sch.z, out_exp); sch.w = ldexpf(sch.w, out_exp);
    }
    // BatchNorm partials (EPI 1), this layout: per lane running sums of its FOUR channels over its voxel of each tile
    // (shifted-data sums as in the bf16x6 form above); lanes of a channel quad are merged once, after the walk
    float hk[4] = {0.f, 0.f, 0.f, 0.f}, hs1[4] = {0.f, 0.f, 0.f, 0.f}, hs2[4] = {0.f, 0.f, 0.f, 0.f};
    int h_n = 0;
    bool h_first = true;
    auto finish_h = [&](int o, bool ok) {
        const int oh = ty0 + 4 * wm + ((lane >> 2) & 3);
        const bool row_ok = ok && oh < a.Ho;
        const unsigned row_off = (unsigned)((o * a.Ho + oh) * a.Wo) * OVB + (unsigned)cqh * 4u;
        if (EPI == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) hk[r] = (row_ok && h_first) ? acc[2][0][r] * sch.x : hk[r];  // (sch.x = 2^out_exp here)
            h_first = h_first && !row_ok;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int ow = tx0 + 4 * m + (lane & 3);
            const bool vok = row_ok && ow < a.Wo;
            const unsigned off = vok ? row_off + (unsigned)ow * OVB : R_OOB;
            float4 y = make_float4(fmaf(acc[2][m][0], sch.x, sfh.x), fmaf(acc[2][m][1], sch.y, sfh.y),
                                   fmaf(acc[2][m][2], sch.z, sfh.z), fmaf(acc[2][m][3], sch.w, sfh.w));
            if (EPI == 2) {
                const float4 rr = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, off, 0, 0));
                y.x += rr.x; y.y += rr.y; y.z += rr.z; y.w += rr.w;
            }
            if (EPI != 1) { y.x = fmaxf(y.x, floor_); y.y = fmaxf(y.y, floor_); y.z = fmaxf(y.z, floor_); y.w = fmaxf(y.w, floor_); }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), rs_out, (R16H_ABL & 4) ? R_OOB : off, 0, 0);
            if (EPI == 1) {
                h_n += vok ? 1 : 0;
                const float v[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dlt = vok ? v[r] - hk[r] : 0.f;
                    hs1[r] += dlt;
                    hs2[r] = fmaf(dlt, dlt, hs2[r]);
                }
            }
        }
    };
    auto flush_stats_h = [&]() {
        const int tiy4 = 2 * tiy + wm;
        const bool tile_ok = tiy4 < a.tiles_y;
        const unsigned tile_id = (unsigned)(((b * a.nseg + seg) * a.tiles_y + tiy4) * a.tiles_x + tix);
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
                                                  (tile_ok && (lane & 15) == 0 && !R16_NOPART) ? (unsigned)(((size_t)ch * a.ntiles + tile_id) * 8) : R_OOB, 0, 0);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ntot), rs_cnt,
                                              (tile_ok && lane == 0 && wn == 0 && co0 == 0) ? tile_id * 4u : R_OOB, 0, 0);
    };
    // A stage = 36 chains of nine MFMAs (r16_chain9: K = the three kw taps of a (kd, kh) row), ordered kh, tile pair,
    // kd, tile: 144 accumulate-adds per stage (the K32-block form needs 432 and was VALU-issue bound at 0.94 ms: an
    // MFMA leaves 8 of its 16 cycles to the vector issue of both waves of the SIMD).  Registers: the voxel fragments of
    // a tile pair (2 x 3 kw x 2 parts = 48) are read from LDS once and serve the three kd; the weights of a kh row
    // (3 kd x 3 kw x 2 parts = 72) stay for both pairs, and each kd's set is replaced by the next row's right after
    // its last use (four chains = 36 MFMAs before its next one).
    float4 wh[3][3][2];
    auto load_bh = [&](float4 (&bq)[3][2], int tap0_f4) {
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                bq[kw][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, (tap0_f4 + kw * TAPF4) * 16 + q * 1024, 0));
    };
    auto stage_h = [&](auto cc_tag, int p, int buf) {
        constexpr int CC = decltype(cc_tag)::value;
        constexpr bool LAST = (CC == NCH - 1);
        constexpr int CCN = (CC + 1) % NCH;
        const unsigned char *sl = slab + buf * R_SLAB_BYTES;
        unsigned char *sn = slab + (buf ^ 1) * R_SLAB_BYTES;
        constexpr int wcur = CC * (2 * NP * 64), wnxt = CCN * (2 * NP * 64);
        const int pn = LAST ? p + 1 : p;
        if (CC == 0) {
            finish_h(p - 2, p - 2 >= d0);
            rotate();
        }
        float4 ah[2][3][2];
        auto load_ah = [&](int m, int kh) {
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
                for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm) {
                        const int m = 2 * pr + mm;
                        const int q = ((kh * 2 + pr) * 3 + kd) * 2 + mm;
                        __builtin_amdgcn_sched_barrier(0);
                        // the temporary carried in belongs to the chain before (a zero at the stage start)
                        f32x4 &prev = mm > 0 ? acc[kd][m - 1] : kd > 0 ? acc[kd - 1][m + 1] : pr > 0 ? acc[2][1] : acc[2][3];
                        r16_chain9(tq[q & 1], ah[mm], wh[kd], prev, tq[(q + 1) & 1]);
                        if (kd == 2 && !(kh == 2 && pr == 1) && !(R16H_ABL & 8)) {  // this tile's fragments: the next pair's / next row's
                            __builtin_amdgcn_sched_barrier(0);
                            load_ah(pr == 0 ? m + 2 : mm, pr == 0 ? kh : kh + 1);
                        }
                        if (pr == 1 && mm == 1 && !(R16H_ABL & 2)) {  // last use of this row's (kd) weights: the next row's, or the next stage's first
                            __builtin_amdgcn_sched_barrier(0);
                            load_bh(wh[kd], kh < 2 ? wcur + (kd * 9 + (kh + 1) * 3) * TAPF4 : wnxt + (kd * 9) * TAPF4);
                        }
                        // the next slab in two halves of three pieces (12 registers in flight instead of 24): requested at
                        // chains 1 / 16, split and written at chains 10, 12, 14 / 26, 28, 30
                        if ((q == 1 || q == 16) && !(R16H_ABL & 1)) {
                            __builtin_amdgcn_sched_barrier(0);
                            issue(pn, CCN, q == 1 ? 0 : 3, q == 1 ? 3 : R_NLD);
                        }
                        if (((q >= 10 && q <= 14 && !(q & 1)) || (q >= 26 && q <= 30 && !(q & 1))) && !(R16H_ABL & 1)) {
                            __builtin_amdgcn_sched_barrier(0);
                            commit_piece(q < 16 ? (q - 10) / 2 : 3 + (q - 26) / 2, sn);
                        }
                    }
        __builtin_amdgcn_sched_barrier(0);
        acc[2][3] += tq[1];  // the stage's last temporary (36 chains: the last one wrote tq[1])
        tq[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        R16_T(1);
        __syncthreads();
        R16_T(2);
    };

    // ---- the walk ------------------------------------------------------------------------------------------
    // planes p_first .. p_last (those of d0-1 .. d1 inside the volume); plane p adds kd to output p + 1 - kd.
    const int p_first = max(d0 - 1, 0), p_last = min(d1, a.Di - 1);
    issue(p_first, 0);
    if constexpr (AR) {
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) load_bh(wh[kd], (kd * 9) * TAPF4);
    } else {
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) load_b(wk[0][kd], (kd * 9) * TAPF4);
    }
#pragma unroll
    for (int it = 0; it < R_NLD; ++it) commit_piece(it, slab);
    __syncthreads();
    R16_T(0);
    int buf = 0;
    for (int p = p_first; p <= p_last; ++p) {
        if constexpr (AR) {
            stage_h(std::integral_constant<int, 0>{}, p, buf); buf ^= 1;
            if (NCH == 2) { stage_h(std::integral_constant<int, NCH - 1>{}, p, buf); buf ^= 1; }
        } else {
            stage_s(std::integral_constant<int, 0>{}, p, buf); buf ^= 1;
            if (NCH == 2) { stage_s(std::integral_constant<int, NCH - 1>{}, p, buf); buf ^= 1; }
        }
    }
    // (output p-1 is finished at the top of stage p+1: the last processed plane's is still in slot 2)
    if constexpr (AR) finish_h(p_last - 1, p_last - 1 >= d0); else finish(p_last - 1, p_last - 1 >= d0);
    rotate();
    // the last output of a segment that ends at the volume's last plane has no plane behind it
    if (p_last < d1) { if constexpr (AR) finish_h(p_last, p_last >= d0); else finish(p_last, p_last >= d0); }
    if (EPI == 1) { if constexpr (AR) flush_stats_h(); else flush_stats(); }
#ifdef R16_STAMP
    R16_T(3);
    if (lane == 0) {
        for (int i = 0; i < 4; ++i) atomicAdd(&r16_stamp_sum[i], st_[i]);
        atomicAdd(&r16_stamp_sum[8], (unsigned long long)__builtin_amdgcn_s_memtime() - t0_);
        atomicAdd(&r16_stamp_sum[9], 1ull);
    }
#endif
}

// ---- weight packing: [tap][cc32][n16][part(3)][lane(64)][8] bf16, element j of lane =
//      part `p` of src(co = n16*16 + (lane & 15), ci = cc*32 + 8*(lane >> 4) + j, tap)  (the B operand of
//      v_mfma_f32_16x16x32_bf16: lane holds column lane & 15, k = 8 (lane >> 4) + j)
__global__ void __launch_bounds__(256)
conv3d_pack_r16_kernel(unsigned short *__restrict__ dst, const float *__restrict__ src, int cin, int cout,
                       long long sn, long long sk, int flip, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63;
    int r = idx >> 9;
    const int p = r % 3; r /= 3;
    const int nn = cout / 16, nch = cin / 32;
    const int n = r % nn; r /= nn;
    const int cc = r % nch;
    const int tap = r / nch;
    const int co = n * 16 + (lane & 15);
    const int ci = cc * 32 + 8 * (lane >> 4) + j;
    const float x = src[co * sn + ci * sk + (flip ? 26 - tap : tap)];
    dst[idx] = az_split3_part(x, p);
}

int az_conv3d_pack_r16(float *packed, const float *w, int cin, int cout, long long stride_out, long long stride_in,
                       int flip, hipStream_t s) {
    if (cin % 32 || cout != 32 || cin <= 0) return AZ_EUNSUPPORTED;
    const int total = 27 * cin * cout * 3;
    hipLaunchKernelGGL(conv3d_pack_r16_kernel, dim3((total + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<unsigned short *>(packed), w, cin, cout, stride_out, stride_in, flip, total);
    return az_launch_status();
}

// f16x3 packing: [tap][cc32][n16][part(2)][lane(64)][8] fp16 of w * 2^k, k from the tensor's largest magnitude
__global__ void __launch_bounds__(256)
conv3d_pack_r16_f16_kernel(unsigned short *__restrict__ dst, const float *__restrict__ src, const float *__restrict__ amax,
                           int cin, int cout, long long sn, long long sk, int flip, int total, int halves) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const float scale = az_pow2(az_f16_scale_exp(az_amax_read(amax)));  // (before the early exit: a wave-wide read)
    if (idx >= total) return;
    const int j = idx & 7, lane = (idx >> 3) & 63;
    // halves = 2 (AZ_PACK_3D_ROLL2): the image of output channels 0..31, then that of 32..63, each in the 32-channel layout
    const int half = halves == 2 ? idx / (total / 2) : 0;
    int r = (halves == 2 ? idx % (total / 2) : idx) >> 9;
    const int p = r & 1; r >>= 1;
    const int nn = (halves == 2 ? 32 : cout) / 16, nch = cin / 32;
    const int n = r % nn; r /= nn;
    const int cc = r % nch;
    const int tap = r / nch;
    const int co = half * 32 + n * 16 + (lane & 15);
    const int ci = cc * 32 + 8 * (lane >> 4) + j;
    dst[idx] = az_split2_f16_part(src[co * sn + ci * sk + (flip ? 26 - tap : tap)] * scale, p);
}

int az_conv3d_pack_r16_f16(float *packed, const float *w, const float *w_amax, int cin, int cout, long long stride_out,
                           long long stride_in, int flip, hipStream_t s, int halves) {
    if (cin % 32 || cout % 16 || cin <= 0 || cout <= 0 || (halves == 2 && cout != 64)) return AZ_EUNSUPPORTED;
    const int total = 27 * cin * cout * 2;
    hipLaunchKernelGGL(conv3d_pack_r16_f16_kernel, dim3((total + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<unsigned short *>(packed), w, w_amax, cin, cout, stride_out, stride_in, flip, total, halves);
    return az_launch_status();
}

// depth segments: one round of workgroups over the chip's 512 slots (256 CUs x 2) if the patches allow it,
// otherwise the split that minimises rounds x (planes walked per workgroup)
static void roll_segments(const ConvArgs &a, int &nseg, int &seg_len, int halves = 1) {
    az_roll_segments((long long)a.B * ((a.tiles_y + 1) / 2) * a.tiles_x * halves, a.Do, az_options().roll_seglen, nseg, seg_len);
}

// rows of the BatchNorm partial buffers of a roll launch: one per (batch, depth segment, 4x16 tile)
long long az_conv3d_roll_stats_tiles(const ConvArgs &a, int cout) {
    int nseg, seg_len;
    roll_segments(a, nseg, seg_len, cout / 32);
    return (long long)a.B * nseg * a.tiles_y * a.tiles_x;
}

template <int CIN, int EPI, int AR = 0, bool PS = false, int COUT = 32>
static int launch_roll(ConvArgs a, hipStream_t s) {
    roll_segments(a, a.nseg, a.seg_len, COUT / 32);
    if (EPI == 1) a.ntiles = (long long)a.B * a.nseg * a.tiles_y * a.tiles_x;
    const long long blocks = (long long)a.B * a.nseg * ((a.tiles_y + 1) / 2) * a.tiles_x * (COUT / 32);
    if (blocks <= 0 || blocks > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    // the kernel addresses one batch element of a tensor through a 32-bit buffer offset
    if (!az_fits_buffer_offset((long long)a.Di * a.Hi * a.Wi * (CIN > COUT ? CIN : COUT) * 4) || a.ntiles * 2 * COUT * 4 >= 0xffffff00LL)
        return AZ_EUNSUPPORTED;
    hipLaunchKernelGGL((conv3d_roll_kernel<CIN, EPI, AR, PS, COUT>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    return az_launch_status();
}

int az_conv3d_roll_launch_f16(const ConvArgs &a, int cin, int epi, hipStream_t s, int cout) {
    if (!a.in_amax || !a.w_amax) return AZ_ENULL;
    const int e = epi ? 1 : (a.res ? 2 : 0);
    if (cout == 64) {  // 64 -> 64: two workgroups per patch (AZ_PACK_3D_ROLL2 weights)
        if (cin != 64) return AZ_EUNSUPPORTED;
        if (a.in_split) {
            if (e == 1) return AZ_EUNSUPPORTED;
            return e == 2 ? launch_roll<64, 2, 1, true, 64>(a, s) : launch_roll<64, 0, 1, true, 64>(a, s);
        }
        return e == 1 ? launch_roll<64, 1, 1, false, 64>(a, s) : e == 2 ? launch_roll<64, 2, 1, false, 64>(a, s) : launch_roll<64, 0, 1, false, 64>(a, s);
    }
    if (cout != 32) return AZ_EUNSUPPORTED;
    if (a.in_split) {  // pre-split input: the input gradients of the 32 -> 32 / 32 -> 64 layers (no BatchNorm-partials epilogue)
        if (e == 1) return AZ_EUNSUPPORTED;
        if (cin == 32) return e == 2 ? launch_roll<32, 2, 1, true>(a, s) : launch_roll<32, 0, 1, true>(a, s);
        if (cin == 64) return e == 2 ? launch_roll<64, 2, 1, true>(a, s) : launch_roll<64, 0, 1, true>(a, s);
        return AZ_EUNSUPPORTED;
    }
    if (cin == 32) return e == 1 ? launch_roll<32, 1, 1>(a, s) : e == 2 ? launch_roll<32, 2, 1>(a, s) : launch_roll<32, 0, 1>(a, s);
    if (cin == 64) return e == 1 ? launch_roll<64, 1, 1>(a, s) : e == 2 ? launch_roll<64, 2, 1>(a, s) : launch_roll<64, 0, 1>(a, s);
    return AZ_EUNSUPPORTED;
}

int az_conv3d_roll_launch(const ConvArgs &a, int cin, int epi, hipStream_t s) {
    const int e = epi ? 1 : (a.res ? 2 : 0);
    if (cin == 32) return e == 1 ? launch_roll<32, 1>(a, s) : e == 2 ? launch_roll<32, 2>(a, s) : launch_roll<32, 0>(a, s);
    if (cin == 64) return e == 1 ? launch_roll<64, 1>(a, s) : e == 2 ? launch_roll<64, 2>(a, s) : launch_roll<64, 0>(a, s);
    return AZ_EUNSUPPORTED;
}
