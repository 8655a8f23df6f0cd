// K4 (bf16x6, stride 1, 32 output channels) -- the V0 layers of the cost aggregation, which
// carry most of its FLOPs (nets/psmnet/psmnet_3.py:87-117 dres0..dres4 / classif first convs
// and their input gradients).  Same implicit GEMM and same packed weights as az_conv3d.hip,
// re-tiled after the counters showed where that kernel's time goes (SQ_WAIT_INST_ANY 51 % of
// wave cycles, SQ_WAIT_INST_LDS 1 %: waves sit in s_waitcnt vmcnt on the per-tap weight
// fragments, 16 GB of L1/L2 weight reads per launch):
//   * one wave owns an 8x16 output patch (M = 128 voxels, four 32x32 MFMA tiles) instead of
//     4x16, so every weight fragment fetched feeds twice the MFMAs (weight bytes per launch
//     halved) and the slab halo shrinks from 1.69 to 1.41 voxels staged per output voxel;
//   * K is walked in 16-channel chunks: a tap needs 3 weight registers-quads (hi/mid/lo) instead
//     of 6, which pays for a 3-slot ring that keeps the weights TWO taps ahead of their use,
//     across stage boundaries, inside the 2-waves/SIMD register budget;
//   * slab: 10x18 voxels x 3 parts x 16 ch bf16 = 96 B per voxel, row pitch 20 voxels, the two
//     16-byte halves of a part swapped on odd rows: the four runs of a ds_read_b128 lane group
//     (4 consecutive x on 4 consecutive rows) cover the eight even/odd slot pairs exactly once,
//     so A-fragment reads are bank-conflict free without padding (19.2 KB -> 8 waves per CU).
// Arithmetic is identical to az_conv3d.hip PREC 1 (six bf16 MFMAs per K16 block, smallest
// terms first, fp32 accumulate) except for the order in which K16 blocks are accumulated.
#include "az_conv3d_args.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define M1_TY 8
#define M1_TX 16
#define M1_SY 10
#define M1_SX 18
#define M1_SXP 20  // LDS row pitch in voxels (== 4 mod 8, see above)
#define M1_VS 24   // dwords per slab voxel
#define M1_BAND 2  // 8-row tile rows per band of the block -> tile order (16 rows, as CV_BAND)

// Diagnostic build only (-DCV_STAMP): whole-wave shader cycles and 100 MHz real-time ticks, summed
// into a buffer nothing else reads; their ratio x 100 MHz is the clock the chip holds in this kernel.
#ifdef CV_STAMP
__device__ unsigned long long m1_stamp_sum[4];
extern "C" int az_debug_m128_stamps(unsigned long long *out4, int reset) {
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(m1_stamp_sum), sizeof(m1_stamp_sum)) != hipSuccess) return AZ_ELAUNCH;
    if (reset) { unsigned long long z[4] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(m1_stamp_sum), z, sizeof(z)) != hipSuccess) return AZ_ELAUNCH; }
    return AZ_OK;
}
#endif

template <int CIN, int EPI, int SRC>
__global__ void __launch_bounds__(64, 2)
conv3d_m128_kernel(const ConvArgs a) {
#ifdef CV_STAMP
    const unsigned long long m1_k0 = __builtin_amdgcn_s_memtime(), m1_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int NCH = CIN / 16, NCH32 = CIN / 32;
    __shared__ __attribute__((aligned(16))) float slab[M1_SY * M1_SXP * M1_VS];
    const int lane = threadIdx.x;

    // ---- block -> tile (XCD chunking + banded order, as az_conv3d.hip) ----------------------
    int lin = blockIdx.x;
    if (a.map_mode >= 1) {
        const int nblk = gridDim.x, xcd = blockIdx.x & 7, q8 = nblk >> 3, r8 = nblk & 7;
        lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    }
    const int tyb = (a.tiles_y + 1) >> 1;  // 8-row tile rows (a.tiles_y counts 4-row tiles)
    const int tix = lin % a.tiles_x; lin /= a.tiles_x;
    int tiy, td, b;
    if (a.map_mode >= 2) {
        const int per_b = a.Dt * tyb;
        b = lin / per_b;
        int l = lin - b * per_b;
        const int full = (tyb / M1_BAND) * M1_BAND * a.Dt;
        int band, rows;
        if (l < full) { band = l / (M1_BAND * a.Dt); l -= band * M1_BAND * a.Dt; rows = M1_BAND; }
        else { band = tyb / M1_BAND; l -= full; rows = tyb - band * M1_BAND; }
        td = l / rows;
        tiy = band * M1_BAND + (l - td * rows);
    } else {
        tiy = lin % tyb; lin /= tyb;
        td = lin % a.Dt;
        b = lin / a.Dt;
    }
    const int ty0 = tiy * M1_TY, tx0 = tix * M1_TX;
    const int ih0 = ty0 - 1, iw0 = tx0 - 1;

    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;

    const int row = lane & 31, half = lane >> 5;
    const int rty = row >> 3, rtx = row & 7;
    const float4 *wp4 = reinterpret_cast<const float4 *>(a.wp);

    // stages = (valid input plane, 16-channel chunk); zero-padding planes are skipped outright
    const int sd_lo = max(0, 1 - td), sd_hi = min(2, a.Di - td);
    const int NS = (sd_hi - sd_lo + 1) * NCH;

    constexpr int NQ = M1_SY * M1_SX * 4;    // 16-byte pieces of one slab (4 per voxel)
    constexpr int NLD = (NQ + 63) / 64;      // 12
    float4 pre[NLD];
    unsigned okbits = 0;

    auto issue = [&](int s) {
        const int sdi = s / NCH, cc = s - sdi * NCH;
        const int id = td - 1 + sd_lo + sdi;  // always a valid plane
        okbits = 0;
        const float *plane0 = (SRC == 0)
            ? a.in + (((size_t)b * a.Di + id) * a.Hi) * a.Wi * CIN + cc * 16
            : ((cc < 2) ? a.in : a.in2) + ((size_t)b * a.Hi) * a.Wi * 32 + (cc & 1) * 16;
        const int part4 = (lane & 3) * 4;
        int sy = 0, sx = lane >> 2;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int ih = ih0 + sy, iw = iw0 + sx;
            const int ihc = min(max(ih, 0), a.Hi - 1), iwc = min(max(iw, 0), a.Wi - 1);
            bool ok = (lane + 64 * it < NQ) && ih == ihc && iw == iwc;
            unsigned off;
            if (SRC == 0) {
                off = (unsigned)(ihc * a.Wi + iwc) * CIN + part4;
            } else {  // concat cost volume (psmnet_3.py:149-163): plane = disparity index
                ok = ok && (iw >= id);
                const int iwr = (cc < 2) ? iwc : max(iwc - id, 0);
                off = (unsigned)(ihc * a.Wi + iwr) * 32 + part4;
            }
            pre[it] = *reinterpret_cast<const float4 *>(plane0 + off);
            okbits |= ok ? (1u << it) : 0u;
            sx += 16;
            if (sx >= M1_SX) { sx -= M1_SX; ++sy; }
        }
    };
    auto commit = [&]() {
        int sy = 0, sx = lane >> 2;
        const int j = lane & 3;  // channels 4j..4j+3 of the chunk: half j>>1, 8-byte slot j&1
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            if (!((okbits >> it) & 1u)) pre[it] = make_float4(0.f, 0.f, 0.f, 0.f);  // zero padding
            if (lane + 64 * it < NQ) {
                uint2 hi, mid, lo;
                az_split3_bf16x4(pre[it], hi, mid, lo);
                unsigned *dst = reinterpret_cast<unsigned *>(slab) + (sy * M1_SXP + sx) * M1_VS +
                                (((j >> 1) ^ (sy & 1)) * 4) + (j & 1) * 2;
                *reinterpret_cast<uint2 *>(dst) = hi;
                *reinterpret_cast<uint2 *>(dst + 8) = mid;
                *reinterpret_cast<uint2 *>(dst + 16) = lo;
            }
            sx += 16;
            if (sx >= M1_SX) { sx -= M1_SX; ++sy; }
        }
    };
    // weights of (stage s, tap t): packed [tap][cc32][n=0][part*2 + kb][lane] float4 (az_conv3d.hip)
    auto wbase = [&](int s) -> const float4 * {
        const int sdi = s / NCH, cc = s - sdi * NCH;
        const int kd = sd_lo + sdi;
        return wp4 + ((size_t)(kd * 9 * NCH32 + (cc >> 1)) * 6 + (cc & 1)) * 64;  // wave-uniform
    };
    auto load_b = [&](float4 (&bq)[3], const float4 *base, int t) {
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[p] = (base + (t * NCH32 * 6 + p * 2) * 64)[lane];
    };
    // A fragments: two per-lane base addresses (row parity decides the half swap), everything
    // else is a compile-time offset of the ds_read
    const float *abase[2];
    abase[0] = &slab[(rty * M1_SXP + rtx) * M1_VS + ((half ^ (rty & 1)) * 4)];
    abase[1] = &slab[(rty * M1_SXP + rtx) * M1_VS + ((half ^ ((rty + 1) & 1)) * 4)];
    auto load_a = [&](float4 (&aq)[3], int m, int eh, int ew) {
        const float *ap = abase[eh & 1] + ((4 * (m >> 1) + eh) * M1_SXP + 8 * (m & 1) + ew) * M1_VS;
#pragma unroll
        for (int p = 0; p < 3; ++p) aq[p] = *reinterpret_cast<const float4 *>(ap + 8 * p);
    };
    // fp32 product on the bf16 pipe, one accumulator rounding per K16 block (az_common.h az_mfma6_step)
    // (M1_PIPE 1 = two alternating temporaries, adds one block late: needs 32 more registers than the
    //  2-waves/SIMD budget has left here; 0 = one temporary, added at once: the adds wait for the block's
    //  last MFMA and the SIMD's other wave fills the gap)
#ifndef M1_PIPE
#define M1_PIPE 0
#endif
    f32x16 t0, t1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { t0[e] = 0.f; t1[e] = 0.f; }
    auto step = [&](int cur, f32x16 &tn, const f32x16 &tp, const float4 (&aq)[3], const float4 (&bq)[3]) {
        if (M1_PIPE) az_mfma6_step(tn, aq, bq, acc[(cur + 3) & 3], tp);
        else az_mfma6_now(acc[cur], aq, bq);
    };

    // ---- software pipeline ------------------------------------------------------------------
    // slab of stage s+1: global -> registers during the taps of stage s, split + LDS at its end;
    // weights: ring[t % 3] holds tap t, fetched two taps ahead (9 taps per stage keep the slot
    // index static across stages).
    float4 ring[3][3];
    issue(0);
    const float4 *bcur = wbase(0);
    load_b(ring[0], bcur, 0);
    load_b(ring[1], bcur, 1);
    for (int s = 0; s < NS; ++s) {
        const float4 *bnext = (s + 1 < NS) ? wbase(s + 1) : bcur;
        __syncthreads();  // previous slab fully consumed (single-wave group: fence only)
        commit();
        __syncthreads();
        float4 a0[3], a1[3];
        load_a(a0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int eh = t / 3, ew = t - 3 * eh;
            __builtin_amdgcn_sched_barrier(0);
            load_a(a1, 1, eh, ew);
            // (both prefetches are unconditional -- the last stage re-reads valid, cache-hot
            //  addresses -- so the loop body is branch-free and hipcc counts vmcnt exactly:
            //  a wave-uniform branch here made every tap-0 wait drain the slab prefetch)
            if (t + 2 < 9) load_b(ring[(t + 2) % 3], bcur, t + 2);
            else load_b(ring[(t + 2) % 3], bnext, t + 2 - 9);
            // slab of the next stage, AFTER this tap's weight request: vmcnt retires in order,
            // so only weights requested from tap 1 on (used from tap 3 on) queue behind it
            if (t == 0) {
                __builtin_amdgcn_sched_barrier(0);  // keep the request order as written
                issue(min(s + 1, NS - 1));
            }
            __builtin_amdgcn_sched_barrier(0);
            step(0, t0, t1, a0, ring[t % 3]);
            __builtin_amdgcn_sched_barrier(0);
            load_a(a0, 2, eh, ew);
            __builtin_amdgcn_sched_barrier(0);
            step(1, t1, t0, a1, ring[t % 3]);
            __builtin_amdgcn_sched_barrier(0);
            load_a(a1, 3, eh, ew);
            __builtin_amdgcn_sched_barrier(0);
            step(2, t0, t1, a0, ring[t % 3]);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) load_a(a0, 0, (t + 1) / 3, (t + 1) % 3);
            __builtin_amdgcn_sched_barrier(0);
            step(3, t1, t0, a1, ring[t % 3]);
        }
        bcur = bnext;
    }

    if (M1_PIPE) acc[3] += t1;  // the last block's temporary
    // ---- epilogue -----------------------------------------------------------------------------
    // C/D map of the 32x32 MFMA: column (out channel) = lane & 31, row (voxel of the 4x8 tile) =
    // (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); M-tile m covers rows 4(m>>1).., cols 8(m&1)..
    const size_t plane_el = (((size_t)b * a.Do + td) * a.Ho) * a.Wo * 32;
    float *outp = a.out + plane_el;
    const float *resp = a.res ? a.res + plane_el : nullptr;
    const bool full = (ty0 + M1_TY - 1 < a.Ho) && (tx0 + M1_TX - 1 < a.Wo);
    auto voxel = [&](int m, int r, unsigned &off) -> bool {
        const int vrow = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int oh = ty0 + 4 * (m >> 1) + (vrow >> 3), ow = tx0 + 8 * (m & 1) + (vrow & 7);
        off = (unsigned)(oh * a.Wo + ow) * 32 + row;
        return full || ((oh < a.Ho) && (ow < a.Wo));
    };
    if (EPI == 0) {
        const float sc = a.scale ? a.scale[row] : 1.f, sf = a.shift ? a.shift[row] : 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsigned off;
                if (!voxel(m, r, off)) continue;
                float y = acc[m][r] * sc + sf;
                if (resp) y += resp[off];
                if (a.relu) y = fmaxf(y, 0.f);
                outp[off] = y;
            }
    } else {
        // BatchNorm partials keep az_conv3d.hip's granularity: one (sum, centred M2, count) entry
        // per 4x16 half of the patch, under that kernel's canonical tile id
#pragma unroll
        for (int my = 0; my < 2; ++my) {
            const int tiy4 = 2 * tiy + my;
            if (tiy4 >= a.tiles_y) continue;  // wave-uniform: the lower half lies outside the volume
            int nvalid = 0;
            unsigned okmask = 0;  // bit (mx*16 + r)
#pragma unroll
            for (int mx = 0; mx < 2; ++mx)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    unsigned off;
                    if (voxel(my * 2 + mx, r, off)) {
                        okmask |= 1u << (mx * 16 + r);
                        nvalid++;
                        outp[off] = acc[my * 2 + mx][r];
                    }
                }
            const int ntot = nvalid + __shfl_xor(nvalid, 32);
            float sm = 0.f, m2 = 0.f;
#pragma unroll
            for (int mx = 0; mx < 2; ++mx)
#pragma unroll
                for (int r = 0; r < 16; ++r) sm += ((okmask >> (mx * 16 + r)) & 1u) ? acc[my * 2 + mx][r] : 0.f;
            sm += __shfl_xor(sm, 32);
            const float mean = sm / (float)max(ntot, 1);
#pragma unroll
            for (int mx = 0; mx < 2; ++mx)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dlt = acc[my * 2 + mx][r] - mean;
                    m2 += ((okmask >> (mx * 16 + r)) & 1u) ? dlt * dlt : 0.f;
                }
            m2 += __shfl_xor(m2, 32);
            const int tile_id = ((b * a.Dt + td) * a.tiles_y + tiy4) * a.tiles_x + tix;
            if (half == 0)
                *reinterpret_cast<float2 *>(&a.part[((size_t)row * a.ntiles + tile_id) * 2]) = make_float2(sm, m2);
            if (lane == 0) a.cnt[tile_id] = (float)ntot;
        }
    }
#ifdef CV_STAMP
    if (threadIdx.x == 0) {
        atomicAdd(&m1_stamp_sum[0], (unsigned long long)__builtin_amdgcn_s_memtime() - m1_k0);
        atomicAdd(&m1_stamp_sum[1], (unsigned long long)__builtin_amdgcn_s_memrealtime() - m1_r0);
        atomicAdd(&m1_stamp_sum[2], 1ull);
    }
#endif
}

template <int CIN, int EPI, int SRC>
static int launch_m128(const ConvArgs &a, hipStream_t s) {
    const long long blocks = (long long)a.B * a.Dt * ((a.tiles_y + 1) / 2) * a.tiles_x;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    hipLaunchKernelGGL((conv3d_m128_kernel<CIN, EPI, SRC>), dim3((unsigned)blocks), dim3(64), 0, s, a);
    return az_launch_status();
}

int az_conv3d_m128_launch(const ConvArgs &a, int cin, int epi, int src, hipStream_t s) {
    if (src == 1) {
        if (cin != 64) return AZ_EUNSUPPORTED;
        return epi ? launch_m128<64, 1, 1>(a, s) : launch_m128<64, 0, 1>(a, s);
    }
    if (cin == 32) return epi ? launch_m128<32, 1, 0>(a, s) : launch_m128<32, 0, 0>(a, s);
    if (cin == 64) return epi ? launch_m128<64, 1, 0>(a, s) : launch_m128<64, 0, 0>(a, s);
    return AZ_EUNSUPPORTED;
}
