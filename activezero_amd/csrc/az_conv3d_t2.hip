// K5 (bf16x6, 32 output channels) -- stride-2 TRANSPOSED 3x3x3 convolution: the hourglass's conv6
// (ConvTranspose3d 64 -> 32, V1 -> V0; reference nets/psmnet/psmnet_3.py:34-58) and the input gradient of its
// stride-2 convolution conv1 (32 -> 64), the two V0-sized launches that az_conv3d.hip's MODE 2 served at 0.19
// of the MFMA roofline: there every one of the 8 output-parity phases of a coarse patch was a workgroup of its
// own and staged the SAME coarse slab again, the lightest phase running 24 MFMAs per staged slab.
//
//   out[2t + p] (p = output parity per dimension, t = coarse index) = sum over the taps
//       p = 0:  k = 1 at coarse t            p = 1:  k = 2 at coarse t,  k = 0 at coarse t + 1
//
// so phase (pd, ph, pw) reads coarse offsets (od, oh, ow) <= (pd, ph, pw) -- 27 (phase, offset) pairs = the
// 27 taps.  Here ONE workgroup of four waves owns a 4x16 coarse patch of one coarse plane for ALL eight
// phases: the (plane offset od, 16-channel chunk) slab -- 5x17 voxels -- is staged once (double-buffered,
// cooperative split as in az_conv2d.hip) and every wave multiplies it for its own phases:
//       wave 0: (1,1,1)                  8 taps        wave 2: (1,0,1) (0,0,1)            6 taps
//       wave 1: (0,1,1) (1,0,0)          6 taps        wave 3: (1,1,0) (0,1,0) (0,0,0)    7 taps
// An A fragment (coarse voxel, offset) is read once per wave and shared by its phases; weights (the packed
// image of az_conv3d_pack_weights) are fetched one block ahead.  Arithmetic: az_mfma6_step (az_common.h).
// BatchNorm partials keep az_conv3d.hip's tile ids (8 phase tiles per coarse patch), so az_bn3d_finalize is
// unchanged.
#include <type_traits>

#include "az_conv3d_args.h"
#include "az_roll_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));  // (= az_f32x16h)

#define T2_PITCH 20  // LDS row pitch in voxels (== 4 mod 8)
#define T2_VS 24     // dwords per slab voxel
#define T2_SY 5
#define T2_SX 17
#define T2_BAND 4

// phases of wave w (code = pd*4 + ph*2 + pw; -1 = none)
__host__ __device__ constexpr int t2_phase(int w, int q) {
    constexpr int tab[4][3] = {{7, -1, -1}, {3, 4, -1}, {5, 1, -1}, {6, 2, 0}};
    return tab[w][q];
}
__host__ __device__ constexpr int t2_kidx(int p, int o) { return p == 0 ? 1 : (o == 0 ? 2 : 0); }
// the i-th (slot q, oh, ow) block of wave w in a stage with plane offset od, ordered by (oh, ow, q);
// returns q | oh << 4 | ow << 8 | tap << 12, or -1 past the end
__host__ __device__ constexpr int t2_block(int w, int od, int i) {
    int n = 0;
    for (int oh = 0; oh < 2; ++oh)
        for (int ow = 0; ow < 2; ++ow)
            for (int q = 0; q < 3; ++q) {
                const int p = t2_phase(w, q);
                if (p < 0) continue;
                const int pd = p >> 2, ph = (p >> 1) & 1, pw = p & 1;
                if (od > pd || oh > ph || ow > pw) continue;
                if (n == i) return q | (oh << 4) | (ow << 8) | (((t2_kidx(pd, od) * 3 + t2_kidx(ph, oh)) * 3 + t2_kidx(pw, ow)) << 12);
                ++n;
            }
    return -1;
}
__host__ __device__ constexpr int t2_count(int w, int od) {
    int n = 0;
    while (t2_block(w, od, n) >= 0) ++n;
    return n;
}

// AR: 0 = bf16x6, 1 = f16x3 (az_roll_common.h: two scaled fp16 parts in the first two parts of the same slab image,
// three MFMAs per block; weights in the gather kernel's f16x3 packing)
template <int CIN, int EPI, int WV, int AR>
__device__ __forceinline__ void t2_wave(const ConvArgs &a, float *slab, int b, int td, int tiy, int tix) {
    constexpr int NCH = CIN / 16, NCH32 = CIN / 32;
    constexpr int NP = AR ? 2 : 3, NF = AR ? 4 : 6;
    float in_scale = 1.f, osc = 1.f;
    if (AR) {
        const int ki = az_f16_scale_exp(az_amax_read(a.in_amax)), kw_ = az_f16_scale_exp(az_amax_read(a.w_amax));
        in_scale = az_pow2(ki);
        osc = ldexpf(1.f, -(ki + kw_));
    }
    constexpr int SLAB = T2_SY * T2_PITCH * T2_VS;
    constexpr int NQ = T2_SY * T2_SX * 4, NLD = (NQ + 255) / 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int ty0 = tiy * 4, tx0 = tix * 16;
    const int row = lane & 31, half = lane >> 5;
    const int rty = row >> 3, rtx = row & 7;
    const float4 *wp4 = reinterpret_cast<const float4 *>(a.wp);

    f32x16 acc[3][2];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q][m][e] = 0.f;

    float4 pre[NLD];
    unsigned okbits = 0;
    const int nplanes = (td + 1 < a.Di) ? 2 : 1;   // plane td + 1 beyond the volume contributes nothing
    const int NS = nplanes * NCH;
    auto issue = [&](int s) {
        const int od = s / NCH, cc = s - od * NCH;
        const float *plane0 = a.in + (((size_t)b * a.Di + td + od) * a.Hi) * a.Wi * CIN + cc * 16;
        okbits = 0;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int q = tid + 256 * it, vox = q >> 2, j = q & 3;
            const int sy = vox / T2_SX, sx = vox - sy * T2_SX;
            const int ih = ty0 + sy, iw = tx0 + sx;
            const int ihc = min(ih, a.Hi - 1), iwc = min(iw, a.Wi - 1);
            const bool ok = (q < NQ) && ih == ihc && iw == iwc;
            pre[it] = *reinterpret_cast<const float4 *>(plane0 + (unsigned)(ihc * a.Wi + iwc) * CIN + j * 4);
            okbits |= ok ? (1u << it) : 0u;
        }
    };
    auto commit = [&](int buf) {
        unsigned *sb = reinterpret_cast<unsigned *>(slab) + buf * SLAB;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int q = tid + 256 * it, vox = q >> 2, j = q & 3;
            const int sy = vox / T2_SX, sx = vox - sy * T2_SX;
            if (!((okbits >> it) & 1u)) pre[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < NQ) {
                uint2 hi, mid, lo;
                if (AR) {
                    az_split2_f16x4(make_float4(pre[it].x * in_scale, pre[it].y * in_scale, pre[it].z * in_scale, pre[it].w * in_scale), hi, mid);
                    lo = mid;
                } else {
                    az_split3_bf16x4(pre[it], hi, mid, lo);
                }
                unsigned *dst = sb + (sy * T2_PITCH + sx) * T2_VS + (((j >> 1) ^ (sy & 1)) * 4) + (j & 1) * 2;
                *reinterpret_cast<uint2 *>(dst) = hi;
                *reinterpret_cast<uint2 *>(dst + 8) = mid;
                if (!AR) *reinterpret_cast<uint2 *>(dst + 16) = lo;
            }
        }
    };
    // packed weights [tap][cin/32][n = 0][part*2 + kb][lane] float4 (az_conv3d.hip), kb = odd 16-channel chunk
    auto load_b = [&](float4 (&bq)[3], int tap, int cc) {
        const float4 *p = wp4 + ((size_t)(tap * NCH32 + (cc >> 1)) * NF + (cc & 1)) * 64 + lane;
#pragma unroll
        for (int k = 0; k < NP; ++k) bq[k] = p[k * 2 * 64];
    };
    const float *abase[2];
    abase[0] = &slab[(rty * T2_PITCH + rtx) * T2_VS + ((half ^ (rty & 1)) * 4)];
    abase[1] = &slab[(rty * T2_PITCH + rtx) * T2_VS + ((half ^ ((rty + 1) & 1)) * 4)];
    auto load_a = [&](float4 (&aq)[3], int buf, int m, int oh, int ow) {
        const float *ap = abase[oh & 1] + buf * SLAB + (oh * T2_PITCH + 8 * m + ow) * T2_VS;
#pragma unroll
        for (int p = 0; p < NP; ++p) aq[p] = *reinterpret_cast<const float4 *>(ap + 8 * p);
    };

    issue(0);
    commit(0);
    __syncthreads();
    // Block products are summed in two alternating temporaries; an accumulator tile receives its finished
    // temporary under the NEXT block's MFMAs (az_mfma6_step).  The pending tile at a stage's start is the last
    // tile of this wave's previous stage, whose plane offset PREV is a template tag too, so every target is a
    // compile-time register range.
    f32x16 t0, t1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { t0[e] = 0.f; t1[e] = 0.f; }
    auto stage = [&](auto od_tag, auto prev_tag, int cc, int s) {
        constexpr int OD = decltype(od_tag)::value, PREV = decltype(prev_tag)::value;
        constexpr int N = t2_count(WV, OD);
        constexpr int QLAST_PREV = t2_block(WV, PREV, t2_count(WV, PREV) - 1) & 15;
        const int buf = s & 1;
        float4 bq[2][3], a0[3], a1[3];
        load_b(bq[0], t2_block(WV, OD, 0) >> 12, cc);
        __builtin_amdgcn_sched_barrier(0);
        issue(min(s + 1, NS - 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int blk = t2_block(WV, OD, i), prv = i ? t2_block(WV, OD, i - 1) : -1;
            const int q = blk & 15, oh = (blk >> 4) & 15, ow = (blk >> 8) & 15;
            const int qprev = i ? (prv & 15) : QLAST_PREV;
            if (i + 1 < N) load_b(bq[(i + 1) & 1], t2_block(WV, OD, i + 1) >> 12, cc);
            if (i == 0 || ((blk >> 4) & 255) != ((prv >> 4) & 255)) {  // a new (oh, ow): fetch its fragments
                load_a(a0, buf, 0, oh, ow);
                load_a(a1, buf, 1, oh, ow);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (AR) az_mfma3_step(t0, a0, bq[i & 1], acc[qprev][1], t1); else az_mfma6_step(t0, a0, bq[i & 1], acc[qprev][1], t1);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (AR) az_mfma3_step(t1, a1, bq[i & 1], acc[q][0], t0); else az_mfma6_step(t1, a1, bq[i & 1], acc[q][0], t0);
            __builtin_amdgcn_sched_barrier(0);
        }
        commit(buf ^ 1);
        __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    for (int cc = 0; cc < NCH; ++cc) stage(I0{}, I0{}, cc, cc);
    if (nplanes == 2) {
        stage(I1{}, I0{}, 0, NCH);
        for (int cc = 1; cc < NCH; ++cc) stage(I1{}, I1{}, cc, NCH + cc);
        acc[t2_block(WV, 1, t2_count(WV, 1) - 1) & 15][1] += t1;
    } else {
        acc[t2_block(WV, 0, t2_count(WV, 0) - 1) & 15][1] += t1;
    }

    if (AR) {  // undo the operand scales on the finished sums
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[q][m] *= osc;
    }
    // ---- epilogue: one 4x16 coarse patch per phase -> fine voxels (2t + p) ---------------------------------
    const float sc = (EPI == 0 && a.scale) ? a.scale[row] : 1.f, sf = (EPI == 0 && a.shift) ? a.shift[row] : 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int p = t2_phase(WV, q);
        if (p < 0) continue;
        const int pd = p >> 2, ph = (p >> 1) & 1, pw = p & 1;
        const int od_out = 2 * td + pd;
        const size_t plane_el = (((size_t)b * a.Do + od_out) * a.Ho) * a.Wo * 32;
        float *outp = a.out + plane_el;
        const float *resp = (EPI == 0 && a.res) ? a.res + plane_el : nullptr;
        int nvalid = 0;
        unsigned okmask = 0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int vrow = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int th = ty0 + (vrow >> 3), tw = tx0 + 8 * m + (vrow & 7);
                if (th >= a.Hi || tw >= a.Wi) continue;
                const unsigned off = (unsigned)((2 * th + ph) * a.Wo + 2 * tw + pw) * 32 + row;
                if (EPI == 0) {
                    float y = acc[q][m][r] * sc + sf;
                    if (resp) y += resp[off];
                    if (a.relu) y = fmaxf(y, 0.f);
                    outp[off] = y;
                } else {
                    outp[off] = acc[q][m][r];
                    okmask |= 1u << (m * 16 + r);
                    ++nvalid;
                }
            }
        if (EPI == 1) {
            const int ntot = nvalid + __shfl_xor(nvalid, 32);
            float sm = 0.f, m2 = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) sm += ((okmask >> (m * 16 + r)) & 1u) ? acc[q][m][r] : 0.f;
            sm += __shfl_xor(sm, 32);
            const float mean = sm / (float)max(ntot, 1);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dlt = acc[q][m][r] - mean;
                    m2 += ((okmask >> (m * 16 + r)) & 1u) ? dlt * dlt : 0.f;
                }
            m2 += __shfl_xor(m2, 32);
            const int tile_id = ((((b * a.Dt + td) * a.tiles_y + tiy) * a.tiles_x + tix) << 3) + p;
            if (half == 0)
                *reinterpret_cast<float2 *>(&a.part[((size_t)row * a.ntiles + tile_id) * 2]) = make_float2(sm, m2);
            if (lane == 0) a.cnt[tile_id] = (float)ntot;
        }
    }
}

template <int CIN, int EPI, int AR>
__global__ void __launch_bounds__(256, 2)
conv3d_t2_kernel(const ConvArgs a) {
    __shared__ __attribute__((aligned(16))) float slab[2 * T2_SY * T2_PITCH * T2_VS];
    int lin = blockIdx.x;
    if (a.map_mode >= 1) {
        const int nblk = gridDim.x, xcd = blockIdx.x & 7, q8 = nblk >> 3, r8 = nblk & 7;
        lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    }
    const int tix = lin % a.tiles_x; lin /= a.tiles_x;
    int tiy, td, b;
    if (a.map_mode >= 2) {  // banded: x fastest, then T2_BAND tile rows, then the depth index
        const int per_b = a.Dt * a.tiles_y;
        b = lin / per_b;
        int l = lin - b * per_b;
        const int full = (a.tiles_y / T2_BAND) * T2_BAND * a.Dt;
        int band, rows;
        if (l < full) { band = l / (T2_BAND * a.Dt); l -= band * T2_BAND * a.Dt; rows = T2_BAND; }
        else { band = a.tiles_y / T2_BAND; l -= full; rows = a.tiles_y - band * T2_BAND; }
        td = l / rows;
        tiy = band * T2_BAND + (l - td * rows);
    } else {
        tiy = lin % a.tiles_y; lin /= a.tiles_y;
        td = lin % a.Dt;
        b = lin / a.Dt;
    }
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wv == 0) t2_wave<CIN, EPI, 0, AR>(a, slab, b, td, tiy, tix);
    else if (wv == 1) t2_wave<CIN, EPI, 1, AR>(a, slab, b, td, tiy, tix);
    else if (wv == 2) t2_wave<CIN, EPI, 2, AR>(a, slab, b, td, tiy, tix);
    else t2_wave<CIN, EPI, 3, AR>(a, slab, b, td, tiy, tix);
}

int az_conv3d_t2_launch(const ConvArgs &a, int cin, int epi, hipStream_t s) {
    const long long blocks = (long long)a.B * a.Dt * a.tiles_y * a.tiles_x;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    if ((long long)a.Ho * a.Wo * 32 > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    const bool f16 = a.in_amax != nullptr && a.w_amax != nullptr;
#define T2_LAUNCH(CIN, EPI) do { if (f16) hipLaunchKernelGGL((conv3d_t2_kernel<CIN, EPI, 1>), dim3((unsigned)blocks), dim3(256), 0, s, a); \
                                 else hipLaunchKernelGGL((conv3d_t2_kernel<CIN, EPI, 0>), dim3((unsigned)blocks), dim3(256), 0, s, a); } while (0)
    if (cin == 64) { if (epi) T2_LAUNCH(64, 1); else T2_LAUNCH(64, 0); }
    else if (cin == 32) { if (epi) T2_LAUNCH(32, 1); else T2_LAUNCH(32, 0); }
    else return AZ_EUNSUPPORTED;
#undef T2_LAUNCH
    return az_launch_status();
}
