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
//   * OUTPUT CHANNELS SPLIT OVER TWO WAVES.  16 output channels x 128 voxels per wave = 32 accumulator
//     registers per depth slot, 96 for the three slots: the kernel keeps 2 waves per SIMD (the partner covers
//     LDS / weight latency and the staging burst) although it carries three outputs.  The two waves of a
//     workgroup stage the shared slab cooperatively (half the split work each) and read the same A fragments.
//   * LDS slab: 10 x 18 voxels x [3 parts][32 ch] bf16 = 192 B per voxel, 34.5 KB, four workgroups per CU.  An
//     M tile is a 4 x 4 voxel square: lane l reads voxel (l & 15) = (row l>>2 & 3, x l & 3), channel octet
//     l >> 4.  Four x-adjacent voxels step through the four 64-B bank quarters (192 B = 3 quarters), and the
//     octet index is XORed with 2 on odd slab rows, so that the 16-lane groups of a ds_read_b128
//     ({0-3, 12-15, 20-27}, ...) touch every bank exactly once for ANY tap shift: conflict-free, no padding.
//
// Arithmetic: az_common.h's bf16x6 product (exact 3-way RNE split, six MFMAs per block summed from zero,
// largest terms first, one VALU add per block into the fp32 accumulator).  Accumulation order per output:
// input plane (kd) ascending, 32-channel chunk, kh, kw -- the order of az_conv3d_m128.hip with K32 blocks.
#include "az_conv3d_args.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// timing-only ablation builds (tools/build_variant.sh): bit 0 no staging after the first stage, 1 no barriers,
// 2 weights loaded once, 3 A fragments loaded once per row, 4 no epilogue.  0 = the shipped kernel.
#ifndef R16_ABL
#define R16_ABL 0
#endif

#define R_TY 8
#define R_TX 16
#define R_SY 10
#define R_SX 18
#define R_VB 192                          // bytes per slab voxel
#define R_SLAB_BYTES (R_SY * R_SX * R_VB)  // 34 560
#define R_NQ (R_SY * R_SX * 8)             // 16-byte fp32 pieces of one plane chunk (8 per voxel)
#define R_NLD ((R_NQ + 127) / 128)         // 12 per thread

#define R_MF(ACC, A, B) __builtin_amdgcn_mfma_f32_16x16x32_bf16( \
        __builtin_bit_cast(az_bf16x8, aq[A]), __builtin_bit_cast(az_bf16x8, bq[B]), ACC, 0, 0, 0)

// tnew = block product (six MFMAs from zero, largest terms first);  cprev += tprev with the four adds placed
// between the MFMAs (the temporary of the tile before): the matrix pipe never waits for them.
__device__ __forceinline__ void r16_step(f32x4 &tnew, const float4 (&aq)[3], const float4 (&bq)[3], f32x4 &cprev,
                                         const f32x4 &tprev) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    float c0 = cprev[0], c1 = cprev[1], c2 = cprev[2], c3 = cprev[3];
    t = R_MF(t, 0, 0);
    c0 += tprev[0];
    asm volatile("" : "+v"(c0));  // one scalar add per gap: packed adds beside MFMAs cost more than they save
    t = R_MF(t, 0, 1);
    c1 += tprev[1];
    asm volatile("" : "+v"(c1));
    t = R_MF(t, 1, 0);
    c2 += tprev[2];
    asm volatile("" : "+v"(c2));
    t = R_MF(t, 1, 1);
    c3 += tprev[3];
    asm volatile("" : "+v"(c3));
    t = R_MF(t, 0, 2);
    t = R_MF(t, 2, 0);
    tnew = t;
    cprev[0] = c0; cprev[1] = c1; cprev[2] = c2; cprev[3] = c3;
    // pin the interleave: MFMA, add, MFMA, add, MFMA, add, MFMA, add, MFMA, MFMA
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
}
__device__ __forceinline__ void r16_first(f32x4 &tnew, const float4 (&aq)[3], const float4 (&bq)[3]) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    t = R_MF(t, 0, 0); t = R_MF(t, 0, 1); t = R_MF(t, 1, 0); t = R_MF(t, 1, 1); t = R_MF(t, 0, 2); t = R_MF(t, 2, 0);
    tnew = t;
}

// 4x4 transpose across the four lanes of a quad: in: lane q holds M[q][0..3]; out: lane q holds M[0..3][q].
// (C layout of the 16x16 MFMA: lane = output channel, registers = four x-adjacent voxels; after the transpose a
//  lane holds four consecutive channels of ONE voxel: a 16-byte store, 64 contiguous bytes per quad.)
template <int CTRL>
__device__ __forceinline__ float r16_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ f32x4 r16_quad_transpose(const f32x4 &v, int lane) {
    const bool o1 = lane & 1, o2 = lane & 2;
    // 1x1 blocks between lanes q and q^1
    const float s01 = r16_dpp<0xB1>(o1 ? v[0] : v[1]);  // quad_perm [1,0,3,2]
    const float s23 = r16_dpp<0xB1>(o1 ? v[2] : v[3]);
    const float a0 = o1 ? s01 : v[0], a1 = o1 ? v[1] : s01, a2 = o1 ? s23 : v[2], a3 = o1 ? v[3] : s23;
    // 2x2 blocks between lanes q and q^2
    const float t0 = r16_dpp<0x4E>(o2 ? a0 : a2);       // quad_perm [2,3,0,1]
    const float t1 = r16_dpp<0x4E>(o2 ? a1 : a3);
    return f32x4{o2 ? t0 : a0, o2 ? t1 : a1, o2 ? a2 : t0, o2 ? a3 : t1};
}

// Diagnostic build only (-DR16_STAMP): shader cycles per phase of the walk, summed over all waves into a buffer
// nothing else reads (tools/roll_stamp_probe.py): [0] first barrier, [1] commit, [2] second barrier, [3] issue,
// [4] rows, [5] after the rows up to the epilogue, [6] epilogue, [7] rotation + loop, [8] whole kernel, [9] waves.
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

template <int S> struct r16_slot { static constexpr int value = S; };

template <int CIN, int EPI>
__global__ void __launch_bounds__(128, 2)
conv3d_roll_kernel(const ConvArgs a) {
    constexpr int NCH = CIN / 32;            // 32-channel chunks per plane
    constexpr int TAPF4 = NCH * 2 * 3 * 64;  // float4 per tap in the packed image: [tap][cc][n16][part][lane]
    __shared__ __attribute__((aligned(16))) unsigned char slab[R_SLAB_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
#ifdef R16_STAMP
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
    unsigned long long tl_ = t0_;
#endif

    // ---- block -> (batch, depth segment, patch): contiguous chunk of the linear order per XCD, x fastest,
    //      so that the workgroups resident on an XCD are a compact (y, x) region walking the same depths ----
    int lin = blockIdx.x;
    if (a.map_mode >= 1) {
        const int nblk = gridDim.x, xcd = blockIdx.x & 7, q8 = nblk >> 3, r8 = nblk & 7;
        lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    }
    const int tyb = (a.tiles_y + 1) >> 1;  // 8-row patch rows (a.tiles_y counts 4-row tiles)
    const int tix = lin % a.tiles_x; lin /= a.tiles_x;
    const int tiy = lin % tyb; lin /= tyb;
    const int seg = lin % a.nseg;
    const int b = lin / a.nseg;
    const int d0 = seg * a.seg_len, d1 = min(d0 + a.seg_len, a.Do);  // outputs [d0, d1)
    const int ty0 = tiy * R_TY, tx0 = tix * R_TX;
    const int ih0 = ty0 - 1, iw0 = tx0 - 1;

    // Workgroups that share a CU would otherwise run in lockstep for the whole walk (same program, same work, started
    // together): every wave of the CU splits its slab at the same time while the matrix pipe idles, then all of
    // them contend for it (measured: staging + epilogue fully exposed, 23 % of the kernel).  Blocks b and b + 256
    // land on the same CU when the chip is filled round-robin (8 XCDs x 32 CUs): give the four residents of a CU
    // four different phases, a quarter of a stage apart.  Speed only: nothing depends on the placement.
    if (a.stagger) {
        const int phase = (blockIdx.x >> 8) & 3;
        for (int i = 0; i < phase * a.stagger; ++i) __builtin_amdgcn_s_sleep(64);  // 64 x 64 cycles each
    }

    f32x4 acc[3][8];  // [slot: kd = 0 -> output p+1, 1 -> p, 2 -> p-1][4x4-voxel tile of the 8x16 patch]
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[s][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging: one plane chunk = 10 x 18 voxels x 32 channels fp32 -> bf16 triplets in LDS -------------
    float4 pre[R_NLD];
    unsigned okbits = 0;
    auto issue = [&](int p, int cc) {
        const float *plane0 = a.in + (((size_t)b * a.Di + p) * a.Hi) * a.Wi * CIN + cc * 32 + (tid & 7) * 4;
        int sy = 0, sx = tid >> 3;
        // (opaque start: otherwise the 12 offsets and masks below are hoisted out of the walk as loop invariants
        //  and live -- spilled -- through it; recomputing them costs ~100 VALU per stage)
        asm volatile("" : "+v"(sx));
        okbits = 0;
#pragma unroll
        for (int it = 0; it < R_NLD; ++it) {
            const int ih = ih0 + sy, iw = iw0 + sx;
            const int ihc = min(max(ih, 0), a.Hi - 1), iwc = min(max(iw, 0), a.Wi - 1);
            const bool ok = (tid + 128 * it < R_NQ) && ih == ihc && iw == iwc;
            pre[it] = *reinterpret_cast<const float4 *>(plane0 + (unsigned)(ihc * a.Wi + iwc) * CIN);
            okbits |= ok ? (1u << it) : 0u;
            sx += 16;
            if (sx >= R_SX) { sx -= R_SX; ++sy; }
        }
    };
    auto commit = [&]() {
        int sy = 0, sx = tid >> 3;
        asm volatile("" : "+v"(sx));  // as in issue()
        const int j = tid & 7;  // channels 4j..4j+3: octet j >> 1, 8-byte half j & 1
#pragma unroll
        for (int it = 0; it < R_NLD; ++it) {
            if (!((okbits >> it) & 1u)) pre[it] = make_float4(0.f, 0.f, 0.f, 0.f);  // zero padding
            if (tid + 128 * it < R_NQ) {
                uint2 hi, mid, lo;
                az_split3_bf16x4(pre[it], hi, mid, lo);
                unsigned char *dst = slab + (sy * R_SX + sx) * R_VB + ((((j >> 1) ^ ((sy & 1) << 1))) << 4) + (j & 1) * 8;
                *reinterpret_cast<uint2 *>(dst) = hi;
                *reinterpret_cast<uint2 *>(dst + 64) = mid;
                *reinterpret_cast<uint2 *>(dst + 128) = lo;
            }
            sx += 16;
            if (sx >= R_SX) { sx -= R_SX; ++sy; }
        }
    };

    // ---- operands ------------------------------------------------------------------------------------------
    // A: lane -> voxel (row (lane >> 2) & 3, x lane & 3) of a 4x4 tile, channel octet lane >> 4; the octet
    // swizzle depends on the slab row's parity = (kh + tile row) & 1 (tile origins are even rows)
    const int trow = (lane >> 2) & 3, tcol = lane & 3, oct = lane >> 4;
    unsigned abase[2];
    abase[0] = (trow * R_SX + tcol) * R_VB + ((oct ^ ((trow & 1) << 1)) << 4);
    abase[1] = (trow * R_SX + tcol) * R_VB + ((oct ^ (((trow + 1) & 1) << 1)) << 4);
    // B: packed [tap][cc][n16][part][lane] float4 (conv3d_pack_r16_kernel), this wave's 16 output channels
    const float4 *wp4 = reinterpret_cast<const float4 *>(a.wp) + wn * 3 * 64 + lane;
    auto load_b = [&](float4 (&bq)[3], const float4 *tap) {
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[p] = tap[p * 64];
    };
    auto wrow = [&](int kd, int kh, int cc) -> const float4 * {  // first tap (kw = 0) of a (kd, kh) row
        return wp4 + (size_t)((kd * 9 + kh * 3) * NCH + cc) * (2 * 3 * 64);
    };

    float4 ring[3][3];  // weights of kw = 0, 1, 2 of the row in progress; refilled two taps ahead
    f32x4 tq[2];

    // one (kd, kh) row: 3 taps x 8 tiles x 6 MFMAs into accumulator slot S.  `wr` = this row's weights (ring[0],
    // ring[1] already requested), `wn_` = the row that follows in execution order (its kw = 0, 1 are requested here).
    auto row = [&](auto slot, int kh, const float4 *wr, const float4 *wn_) {
        constexpr int S = decltype(slot)::value;
        const unsigned char *ab = slab + abase[kh & 1] + kh * (R_SX * R_VB);
        auto load_a = [&](float4 (&aq)[3], int m, int kw) {
            const unsigned char *ap = ab + ((4 * (m >> 2)) * R_SX + 4 * (m & 3) + kw) * R_VB;
#pragma unroll
            for (int p = 0; p < 3; ++p) aq[p] = *reinterpret_cast<const float4 *>(ap + 64 * p);
        };
        float4 a0[3], a1[3];
        load_a(a0, 0, 0);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            __builtin_amdgcn_sched_barrier(0);
            if (!(R16_ABL & 4)) {
                if (kw == 0) load_b(ring[2], wr + 2 * TAPF4);
                else load_b(ring[kw - 1], wn_ + (kw - 1) * TAPF4);
            }
#pragma unroll
            for (int m = 0; m < 8; m += 2) {
                __builtin_amdgcn_sched_barrier(0);  // program order as written: A fragments one tile ahead
                if (!(R16_ABL & 8)) load_a(a1, m + 1, kw);
                else { if (kw == 0 && m == 0) load_a(a1, 1, 0); asm volatile("" : "+v"(a0[0].x), "+v"(a1[0].x)); }
                __builtin_amdgcn_sched_barrier(0);
                if (kw == 0 && m == 0) r16_first(tq[0], a0, ring[kw]);
                else r16_step(tq[0], a0, ring[kw], acc[S][(m + 7) & 7], tq[1]);
                __builtin_amdgcn_sched_barrier(0);
                if (!(R16_ABL & 8)) {
                    if (m + 2 < 8) load_a(a0, m + 2, kw);
                    else if (kw < 2) load_a(a0, 0, kw + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
                r16_step(tq[1], a1, ring[kw], acc[S][m], tq[0]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[S][7] += tq[1];  // the row's last temporary
        asm volatile("" : "+v"(acc[S][7]));
    };

    // ---- epilogue of a finished output depth (slot 2) ------------------------------------------------------
    auto finish = [&](int o) {
        int ln = lane;
        asm volatile("" : "+v"(ln));  // per-lane constants of the epilogue are rebuilt here, not kept through the walk
        const int ch = wn * 16 + (ln & 15);  // this lane's output channel
        const int vrow = ln >> 4;            // row of the 4x4 tile this lane's accumulator registers belong to
        const size_t plane_el = (((size_t)b * a.Do + o) * a.Ho) * a.Wo * 32;
        float *outp = a.out + plane_el;
        // after the quad transpose this lane holds channels cq..cq+3 of the voxel (row vrow, x = lane & 3) of a tile
        const int cq = wn * 16 + (ln & 12);
        if (EPI == 0) {
            const float *resp = a.res ? a.res + plane_el : nullptr;
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sf = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.scale) sc = *reinterpret_cast<const float4 *>(a.scale + cq);
            if (a.shift) sf = *reinterpret_cast<const float4 *>(a.shift + cq);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int oh = ty0 + 4 * (m >> 2) + vrow, ow = tx0 + 4 * (m & 3) + (ln & 3);
                const f32x4 v = r16_quad_transpose(acc[2][m], ln);
                if (oh >= a.Ho || ow >= a.Wo) continue;
                const unsigned off = (unsigned)(oh * a.Wo + ow) * 32 + cq;
                float4 y = make_float4(v[0] * sc.x + sf.x, v[1] * sc.y + sf.y, v[2] * sc.z + sf.z, v[3] * sc.w + sf.w);
                if (resp) {
                    const float4 rr = *reinterpret_cast<const float4 *>(resp + off);
                    y.x += rr.x; y.y += rr.y; y.z += rr.z; y.w += rr.w;
                }
                if (a.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
                *reinterpret_cast<float4 *>(outp + off) = y;
            }
        } else {
            // raw output + BatchNorm partials at az_conv3d.hip's granularity: one (sum, centred M2, count) entry
            // per 4x16 half of the patch and channel, under that kernel's canonical tile id
#pragma unroll
            for (int my = 0; my < 2; ++my) {
                const int tiy4 = 2 * tiy + my;
                if (tiy4 >= a.tiles_y) continue;  // workgroup-uniform: the lower half lies outside the volume
                const int oh = ty0 + 4 * my + vrow;
                unsigned okmask = 0;  // bit (mx * 4 + r)
                int nvalid = 0;
                float sm = 0.f;
#pragma unroll
                for (int mx = 0; mx < 4; ++mx) {
                    const f32x4 vt = r16_quad_transpose(acc[2][my * 4 + mx], ln);
                    const int owt = tx0 + 4 * mx + (ln & 3);
                    if (oh < a.Ho && owt < a.Wo)
                        *reinterpret_cast<float4 *>(outp + (unsigned)(oh * a.Wo + owt) * 32 + cq) =
                            make_float4(vt[0], vt[1], vt[2], vt[3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ow = tx0 + 4 * mx + r;
                        if (oh < a.Ho && ow < a.Wo) {
                            okmask |= 1u << (mx * 4 + r);
                            ++nvalid;
                            sm += acc[2][my * 4 + mx][r];
                        }
                    }
                }
                nvalid += __shfl_xor(nvalid, 16); nvalid += __shfl_xor(nvalid, 32);
                sm += __shfl_xor(sm, 16); sm += __shfl_xor(sm, 32);
                const float mean = sm / (float)max(nvalid, 1);
                float m2 = 0.f;
#pragma unroll
                for (int mx = 0; mx < 4; ++mx)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float dlt = acc[2][my * 4 + mx][r] - mean;
                        m2 += ((okmask >> (mx * 4 + r)) & 1u) ? dlt * dlt : 0.f;
                    }
                m2 += __shfl_xor(m2, 16); m2 += __shfl_xor(m2, 32);
                const int tile_id = ((b * a.Dt + o) * a.tiles_y + tiy4) * a.tiles_x + tix;
                if (vrow == 0)
                    *reinterpret_cast<float2 *>(&a.part[((size_t)ch * a.ntiles + tile_id) * 2]) = make_float2(sm, m2);
                if (tid == 0) a.cnt[tile_id] = (float)nvalid;
            }
        }
    };

    // ---- the walk ------------------------------------------------------------------------------------------
    // planes p = d0-1 .. d1; plane p adds (kd) to output p + 1 - kd for the outputs of this segment; planes
    // outside the volume are zero padding and skipped outright.  After plane p output p-1 is complete.
    const int p_first = max(d0 - 1, 0), p_last = min(d1, a.Di - 1);
    auto kd_lo = [&](int p) { return max(0, p + 2 - d1); };
    auto kd_hi = [&](int p) { return min(2, p + 1 - d0); };
    issue(p_first, 0);
    {
        const float4 *w0 = wrow(kd_lo(p_first), 0, 0);
        load_b(ring[0], w0);
        load_b(ring[1], w0 + TAPF4);
    }
    for (int p = d0 - 1; p <= d1; ++p) {
        if (p >= p_first && p <= p_last) {
            const int lo = kd_lo(p), hi = kd_hi(p);
            for (int cc = 0; cc < NCH; ++cc) {
                const bool stage_it = !(R16_ABL & 1) || (p == p_first && cc == 0);
                R16_T(7);
                if (!(R16_ABL & 2)) __syncthreads();  // the previous slab has been consumed by both waves
                R16_T(0);
                if (stage_it) commit();
                R16_T(1);
                if (!(R16_ABL & 2)) __syncthreads();
                R16_T(2);
                // next stage in execution order (the last one re-requests itself: valid, cache-hot addresses)
                int pn = p, ccn = cc + 1;
                if (ccn == NCH) { ccn = 0; pn = p + 1; }
                if (pn > p_last) { pn = p; ccn = cc; }
                if (!(R16_ABL & 1)) issue(pn, ccn);
                R16_T(3);
                const bool more = !(pn == p && ccn == cc);
                const int lon = kd_lo(pn);
#pragma unroll 1
                for (int kh = 0; kh < 3; ++kh) {
                    // rows of this kh in execution order: kd = lo .. hi; the row after the last one is the first
                    // row of the next kh, or of the next stage
                    const float4 *after = (kh < 2) ? wrow(lo, kh + 1, cc) : (more ? wrow(lon, 0, ccn) : wrow(lo, 0, cc));
                    if (lo == 0) row(r16_slot<0>{}, kh, wrow(0, kh, cc), hi >= 1 ? wrow(1, kh, cc) : after);
                    if (lo <= 1 && hi >= 1) row(r16_slot<1>{}, kh, wrow(1, kh, cc), hi >= 2 ? wrow(2, kh, cc) : after);
                    if (hi == 2) row(r16_slot<2>{}, kh, wrow(2, kh, cc), after);
                }
                R16_T(4);
            }
        }
        R16_T(5);
        if (!(R16_ABL & 16) || p - 1 == d1 - 1) { if (p - 1 >= d0) finish(p - 1); }
        R16_T(6);
        // rotate the depth slots: what was output p (slot 1) becomes output (p+1) - 1 of the next plane, ...
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[2][m] = acc[1][m];
            acc[1][m] = acc[0][m];
            acc[0][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        R16_T(7);
    }
#ifdef R16_STAMP
    R16_T(7);
    if (lane == 0) {
        for (int i = 0; i < 8; ++i) atomicAdd(&r16_stamp_sum[i], st_[i]);
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

// depth segments: one round of workgroups over the chip's 1024 slots (256 CUs x 4) if the patches allow it,
// otherwise the split that minimises rounds x (planes staged per workgroup)
static void roll_segments(const ConvArgs &a, int &nseg, int &seg_len) {
    const long long patches = (long long)a.B * ((a.tiles_y + 1) / 2) * a.tiles_x;
    const char *e = getenv("AZ_ROLL_SEGLEN");
    if (e && atoi(e) > 0) {
        seg_len = min(atoi(e), a.Do);
        nseg = (a.Do + seg_len - 1) / seg_len;
        return;
    }
    long long best = -1;
    nseg = 1; seg_len = a.Do;
    for (int n = 1; n <= a.Do; ++n) {
        const int len = (a.Do + n - 1) / n;
        const int nn = (a.Do + len - 1) / len;
        if (nn != n) continue;
        const long long rounds = (patches * n + 1023) / 1024;
        const long long cost = rounds * (len + 2) * 3 + 4;  // +: fixed cost per workgroup
        if (best < 0 || cost < best) { best = cost; nseg = n; seg_len = len; }
    }
}

template <int CIN, int EPI>
static int launch_roll(ConvArgs a, hipStream_t s) {
    roll_segments(a, a.nseg, a.seg_len);
    {   // quarter of a stage in units of 4096 cycles: a stage is 27 x 48 x (CIN / 32) MFMAs of 16 cycles per wave,
        // two waves per SIMD
        const char *e = getenv("AZ_ROLL_STAGGER");
        a.stagger = e ? atoi(e) : (27 * 48 * (CIN / 32) * 16 * 2 / 4) / 4096;
    }
    const long long blocks = (long long)a.B * a.nseg * ((a.tiles_y + 1) / 2) * a.tiles_x;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    hipLaunchKernelGGL((conv3d_roll_kernel<CIN, EPI>), dim3((unsigned)blocks), dim3(128), 0, s, a);
    return az_launch_status();
}

int az_conv3d_roll_launch(const ConvArgs &a, int cin, int epi, hipStream_t s) {
    if (cin == 32) return epi ? launch_roll<32, 1>(a, s) : launch_roll<32, 0>(a, s);
    if (cin == 64) return epi ? launch_roll<64, 1>(a, s) : launch_roll<64, 0>(a, s);
    return AZ_EUNSUPPORTED;
}
