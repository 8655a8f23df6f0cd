// K13 -- stride-1 "same" 2-D convolution family on the bf16x6 MFMA path: the 2-D ResNet + SPP feature
// extractor in front of the cost volume (reference nets/psmnet/psmnet_submodule_3.py:13-41, 92-220:
// convbn / BasicBlock / FeatureExtraction) and the three 2-D convolutions of the factored cost-volume
// convolution (az_costconv.hip; reference nets/psmnet/psmnet_3.py:149-166).  Forward AND input gradient:
// the input gradient of a stride-1 layer is the same convolution with the taps flipped and the channel
// roles swapped, which is only a different weight packing (az_conv2d_pack_weights).
//
// Layout: activations channels-last [B,H,W,C] fp32 (torch.channels_last memory); weights are read from
// a packed bf16 triplet image [tap][16-ch chunk][32-cout tile][part][lane] made per call.
//
// Implicit GEMM, M = an 8x16 pixel patch (128 rows, four 32x32 MFMA tiles), K = taps x Cin walked in
// 16-channel chunks.  Unlike the 3-D kernels (one wave = one workgroup, N = 32), a WORKGROUP of NW waves
// owns the patch for 32*NW output channels: the waves stage ONE zero-padded input slab cooperatively
// (so the exact 3-way bf16 split -- the VALU cost of this arithmetic -- is paid once per 32*NW output
// channels, not once per 32) and every wave multiplies it with its own 32-channel weight tile:
//   * slab: (8+2HY) x (16+2HX) pixels x 3 parts x 16 ch bf16 = 96 B per pixel, row pitch 20 pixels, the two
//     16-byte halves of a part swapped on odd rows (conflict-free ds_read_b128, as az_conv3d_m128.hip);
//     DOUBLE-buffered: chunk s+1 is fetched into registers under the MFMAs of chunk s, split and written
//     to the other buffer behind them -> one workgroup barrier per chunk;
//   * weights: three 16-byte fragments per tap and wave straight from L2, kept two taps ahead in a
//     3-slot register ring (taps % 3 == 0 keeps the slot index static across chunks);
//   * 24 MFMAs per (tap, chunk, wave): six v_mfma_f32_32x32x16_bf16 per 32x32x16 block (az_conv3d.hip).
// Taps are compile-time (KH x KW, dilation): 3x3 d1, 3x3 d2 (layer4), 1x1, 3x5 (the right-image kernels
// of the factored cost-volume convolution).
// Epilogue: y = acc (* scale[c] + shift[c]) (+ residual) (ReLU) -- training stores the raw sums; an
// eval-mode caller folds BatchNorm here.  Output pixel stride is a parameter, so a producer can write
// straight into a channel slice of a wider tensor.
#include "az_pack_f16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef C2_ABL
#define C2_ABL 0  // timing-only builds: 1 weight fragments fetched once per workgroup (no refills), 2 no slab staging after the first chunk
#endif
#define C2_TY 8
#define C2_TX 16
#define C2_PITCH 20  // LDS row pitch in pixels (== 4 mod 8: az_conv3d_m128.hip)
#define C2_VS 24     // dwords per slab pixel: 3 parts x 16 ch bf16

struct C2Args {
    const float *in;   // [B,H,W,*] pixel stride in_cs floats, channels [0, cin) used
    const float *wp;   // packed weights
    float *out;        // [B,H,W,*] pixel stride out_cs floats, channels [0, cout) written
    const float *scale, *shift, *res;  // optional epilogue operands (res: pixel stride res_cs)
    const float *gz, *gh;              // act 4 (GRU combine): update gate z and previous state h, strides gz_cs / gh_cs
    int gz_cs, gh_cs;
    int B, H, W;
    int cin, cout;     // cin % 16 == 0, cout % 32 == 0
    int in_cs, out_cs, res_cs;
    int tiles_y, tiles_x, ngroups;  // patch tiles and cout groups of 32*NW
    int relu;  // epilogue activation: 0 none, 1 ReLU, 2 sigmoid, 3 tanh, 4 GRU combine (1-z)*h + z*tanh(.)
    // STATS instantiations: BatchNorm partials of the raw output, one (sum, M2 about the patch mean) per channel
    // and patch in az_bn2d_fwd's layout [group][cout][stiles][2], counts [group][stiles]
    float *spart, *scnt;
    int sgroups;       // statistic groups: consecutive B / sgroups images each
    long long stiles;  // patches per group
    const float *in_amax, *w_amax;  // PARTS 2 (f16x3): device scalars max |in|, max |w| (az_roll_common.h)
};

// PARTS = 3: the bf16x6 arithmetic above (fp32-class).  PARTS = 1: plain bf16 operands (round-to-nearest), one
// MFMA per 16-deep block, fp32 accumulation -- the arithmetic of the reference's autocast region around the
// RAFT-Stereo GRU update (nets/raft/raft_stereo.py:98,142-172; nets/raft/update.py:19-41), same slab / weight
// pipeline with the mid / lo parts left out.  PARTS = 2: f16x3 (az_roll_common.h) -- two scaled fp16 parts in the
// first two parts of the same slab image, three MFMAs per block.
// H1 (with PARTS = 1): the ONE part is fp16 instead of bf16 -- "f16x1", the reference's real autocast arithmetic for the GRU
// block (torch.cuda.amp.autocast is float16 on CUDA, raft_stereo.py:14): operands rounded to 11 bits once, optionally scaled
// by a power of two from an amax array (in_amax / w_amax, either may be null = scale 1: the stand-in for GradScaler on the
// gradient operands, train.py:303-309), one v_mfma_f32_32x32x16_f16 per block, fp32 accumulation.
template <int NW, int KH, int KW, int DIL, int PARTS = 3, bool STATS = false, bool H1 = false>
__global__ void __launch_bounds__(64 * NW, 2)
conv2d_same_kernel(const C2Args a) {
    static_assert(!H1 || PARTS == 1, "H1 is the fp16 form of the one-part arithmetic");
    constexpr int T = KH * KW;
    constexpr int HY = DIL * (KH - 1) / 2, HX = DIL * (KW - 1) / 2;
    constexpr int SY = C2_TY + 2 * HY, SX = C2_TX + 2 * HX;
    static_assert(SX <= C2_PITCH, "slab row does not fit the LDS pitch");
    constexpr int SLAB = SY * C2_PITCH * C2_VS;  // dwords per buffer
    constexpr int NTHR = 64 * NW;
    constexpr int NQ = SY * SX * 4;              // 16-byte pieces of one slab (4 per pixel)
    constexpr int NLD = (NQ + NTHR - 1) / NTHR;
    // one wave per workgroup (32 output channels): a single slab buffer (19 KB -> 8 workgroups per CU; two
    // buffers would let LDS, not registers, cap the CU at 4 waves)
    constexpr int NBUF = (NW == 1) ? 1 : 2;
    __shared__ __attribute__((aligned(16))) float slab[NBUF * SLAB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- block -> (cout group, patch): XCD-chunked linear order, cout groups of a patch adjacent ----
    int lin = blockIdx.x;
    {
        const int nblk = gridDim.x, xcd = blockIdx.x & 7, q8 = nblk >> 3, r8 = nblk & 7;
        lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    }
    const int grp = lin % a.ngroups; lin /= a.ngroups;
    const int tix = lin % a.tiles_x; lin /= a.tiles_x;
    const int tiy = lin % a.tiles_y;
    const int b = lin / a.tiles_y;
    const int ty0 = tiy * C2_TY, tx0 = tix * C2_TX;
    const int ih0 = ty0 - HY, iw0 = tx0 - HX;
    const int ntile = grp * NW + wv;               // this wave's 32-channel output tile
    const int NT = a.cout >> 5, NCH = a.cin >> 4;  // cout tiles, 16-channel chunks

    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
    float in_scale = 1.f, osc = 1.f;
    if (PARTS == 2) {  // wave-uniform power-of-two operand scales
        const int ki = az_f16_scale_exp(az_amax_read(a.in_amax));
        const int kw_ = az_f16_scale_exp(az_amax_read(a.w_amax));
        in_scale = az_pow2(ki);
        osc = ldexpf(1.f, -(ki + kw_));
    }
    if (H1) {  // (either amax may be absent: that operand is rounded as it is, the way autocast does)
        const int ki = a.in_amax ? az_f16_scale_exp(az_amax_read(a.in_amax)) : 0;
        const int kw_ = a.w_amax ? az_f16_scale_exp(az_amax_read(a.w_amax)) : 0;
        in_scale = az_pow2(ki);
        osc = ldexpf(1.f, -(ki + kw_));
    }

    const int row = lane & 31, half = lane >> 5;
    const int rty = row >> 3, rtx = row & 7;
    const float4 *wp4 = reinterpret_cast<const float4 *>(a.wp);
    const float *img = a.in + (size_t)b * a.H * a.W * a.in_cs;

    float4 pre[NLD];
    unsigned okbits = 0;
    auto issue = [&](int cc) {
        okbits = 0;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int q = tid + NTHR * it, pix = q >> 2, j = q & 3;
            const int sy = pix / SX, sx = pix - sy * SX;
            const int ih = ih0 + sy, iw = iw0 + sx;
            const int ihc = min(max(ih, 0), a.H - 1), iwc = min(max(iw, 0), a.W - 1);
            const bool ok = (q < NQ) && ih == ihc && iw == iwc;
            // loads are unconditional from clamped (valid) addresses: all in flight together; the zero
            // padding is applied at commit time
            pre[it] = *reinterpret_cast<const float4 *>(img + (unsigned)(ihc * a.W + iwc) * (unsigned)a.in_cs + cc * 16 + j * 4);
            okbits |= ok ? (1u << it) : 0u;
        }
    };
    auto commit = [&](int buf) {
        unsigned *sb = reinterpret_cast<unsigned *>(slab) + buf * SLAB;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int q = tid + NTHR * it, pix = q >> 2, j = q & 3;
            const int sy = pix / SX, sx = pix - sy * SX;
            if (!((okbits >> it) & 1u)) pre[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < NQ) {
                unsigned *dst = sb + (sy * C2_PITCH + sx) * C2_VS + (((j >> 1) ^ (sy & 1)) * 4) + (j & 1) * 2;
                if (PARTS == 3) {
                    uint2 hi, mid, lo;
                    az_split3_bf16x4(pre[it], hi, mid, lo);
                    *reinterpret_cast<uint2 *>(dst) = hi;
                    *reinterpret_cast<uint2 *>(dst + 8) = mid;
                    *reinterpret_cast<uint2 *>(dst + 16) = lo;
                } else if (PARTS == 2) {
                    uint2 hi, lo;
                    az_split2_f16x4(make_float4(pre[it].x * in_scale, pre[it].y * in_scale, pre[it].z * in_scale, pre[it].w * in_scale), hi, lo);
                    *reinterpret_cast<uint2 *>(dst) = hi;
                    *reinterpret_cast<uint2 *>(dst + 8) = lo;
                } else if (H1) {
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(az_pk_f16(pre[it].x * in_scale, pre[it].y * in_scale),
                                                                 az_pk_f16(pre[it].z * in_scale, pre[it].w * in_scale));
                } else {
                    *reinterpret_cast<uint2 *>(dst) = make_uint2(az_pk_bf16(pre[it].x, pre[it].y), az_pk_bf16(pre[it].z, pre[it].w));
                }
            }
        }
    };
    // packed weights: [tap][chunk][ntile][part][lane] float4
    bool abl_first = true;
    auto load_b = [&](float4 (&bq)[3], int cc, int t) {
        if ((C2_ABL & 1) && !abl_first) return;
        if ((C2_ABL & 1) && cc == 0 && t >= 2) abl_first = false;
        const float4 *p = wp4 + (((size_t)t * NCH + cc) * NT + ntile) * PARTS * 64 + lane;
#pragma unroll
        for (int k = 0; k < PARTS; ++k) bq[k] = p[k * 64];
    };
    // A fragments: two per-lane bases (row parity decides the half swap), compile-time offsets otherwise
    const float *abase[2];
    abase[0] = &slab[(rty * C2_PITCH + rtx) * C2_VS + ((half ^ (rty & 1)) * 4)];
    abase[1] = &slab[(rty * C2_PITCH + rtx) * C2_VS + ((half ^ ((rty + 1) & 1)) * 4)];
    auto load_a = [&](float4 (&aq)[3], int buf, int m, int oy, int ox) {
        const float *ap = abase[oy & 1] + buf * SLAB + ((4 * (m >> 1) + oy) * C2_PITCH + 8 * (m & 1) + ox) * C2_VS;
#pragma unroll
        for (int p = 0; p < PARTS; ++p) aq[p] = *reinterpret_cast<const float4 *>(ap + 8 * p);
    };
    // block products are summed in two alternating temporaries; the accumulators take each finished
    // temporary one block later (az_common.h az_mfma6_step): acc[3] of a tap is completed under the next
    // tap's first block, the very last one after the loop
    // (one wave per workgroup sits at the 2-waves/SIMD register limit: single temporary, added at once)
    constexpr bool PIPE = (NW > 1);
    f32x16 t0, t1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { t0[e] = 0.f; t1[e] = 0.f; }
    // block `cur` of the rotation 0,1,2,3: multiply into tn, complete accumulator (cur+3)%4 with tp
    auto step = [&](int cur, f32x16 &tn, const f32x16 &tp, const float4 (&aq)[3], const float4 (&bq)[3]) {
        // (price of the temporaries, measured against the plain six-MFMA chain into acc: +2.4 % on the
        //  extractor's forward, 10.09 vs 9.85 ms; profiles/r02_conv2d_layers_hip.txt)
        if (PARTS == 1 && H1)
            acc[cur] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(az_f16x8, aq[0]),
                                                              __builtin_bit_cast(az_f16x8, bq[0]), acc[cur], 0, 0, 0);
        else if (PARTS == 1)
            acc[cur] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(az_bf16x8, aq[0]),
                                                               __builtin_bit_cast(az_bf16x8, bq[0]), acc[cur], 0, 0, 0);
        else if (PARTS == 2 && PIPE) az_mfma3_step(tn, aq, bq, acc[(cur + 3) & 3], tp);
        else if (PARTS == 2) az_mfma3_now(acc[cur], aq, bq);
        else if (PIPE) az_mfma6_step(tn, aq, bq, acc[(cur + 3) & 3], tp);
        else az_mfma6_now(acc[cur], aq, bq);
    };

    // ---- pipeline -----------------------------------------------------------------------------------
    constexpr bool LATE = (NBUF == 1);  // single buffer: the next slab is written after this chunk's last read
    issue(0);
    commit(0);
    __syncthreads();
    if constexpr (T % 3 == 0) {
        float4 ring[3][3];
        load_b(ring[0], 0, 0);
        load_b(ring[1], 0, 1);
        for (int s = 0; s < NCH; ++s) {
            const int buf = (NBUF == 2) ? (s & 1) : 0;
            const int sn = min(s + 1, NCH - 1);  // (the last chunk re-fetches valid, cache-hot data: branch-free loop body)
            float4 a0[3], a1[3];
            load_a(a0, buf, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int oy = (t / KW) * DIL, ox = (t % KW) * DIL;
                __builtin_amdgcn_sched_barrier(0);
                load_a(a1, buf, 1, oy, ox);
                if (t + 2 < T) load_b(ring[(t + 2) % 3], s, t + 2);
                else load_b(ring[(t + 2) % 3], sn, t + 2 - T);
                if (t == 0) {
                    __builtin_amdgcn_sched_barrier(0);  // slab request after this tap's weight request (vmcnt order)
                    issue(sn);
                }
                __builtin_amdgcn_sched_barrier(0);
                step(0, t0, t1, a0, ring[t % 3]);
                __builtin_amdgcn_sched_barrier(0);
                load_a(a0, buf, 2, oy, ox);
                __builtin_amdgcn_sched_barrier(0);
                step(1, t1, t0, a1, ring[t % 3]);
                __builtin_amdgcn_sched_barrier(0);
                load_a(a1, buf, 3, oy, ox);
                __builtin_amdgcn_sched_barrier(0);
                step(2, t0, t1, a0, ring[t % 3]);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < T) load_a(a0, buf, 0, ((t + 1) / KW) * DIL, ((t + 1) % KW) * DIL);
                if (t == T - 1 && !LATE) {
                    // the next chunk's slab: split + LDS write into the other buffer (nobody reads it
                    // during this chunk) under the last MFMA block of this one
                    __builtin_amdgcn_sched_barrier(0);
                    commit((buf ^ 1) & (NBUF - 1));
                }
                __builtin_amdgcn_sched_barrier(0);
                step(3, t1, t0, a1, ring[t % 3]);
            }
            if (LATE) { __syncthreads(); commit(0); }
            __syncthreads();
        }
    } else {
        // single-tap layers (1x1): one weight fragment set per chunk, fetched one chunk ahead
        static_assert(T % 3 == 0 || T == 1, "tap count must be 1 or a multiple of 3");
        float4 bc[3], bn[3];
        load_b(bc, 0, 0);
        for (int s = 0; s < NCH; ++s) {
            const int buf = (NBUF == 2) ? (s & 1) : 0;
            const int sn = min(s + 1, NCH - 1);
            float4 a0[3], a1[3];
            load_a(a0, buf, 0, 0, 0);
            load_a(a1, buf, 1, 0, 0);
            load_b(bn, sn, 0);
            __builtin_amdgcn_sched_barrier(0);
            issue(sn);
            __builtin_amdgcn_sched_barrier(0);
            step(0, t0, t1, a0, bc);
            __builtin_amdgcn_sched_barrier(0);
            load_a(a0, buf, 2, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            step(1, t1, t0, a1, bc);
            __builtin_amdgcn_sched_barrier(0);
            load_a(a1, buf, 3, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            step(2, t0, t1, a0, bc);
            __builtin_amdgcn_sched_barrier(0);
            step(3, t1, t0, a1, bc);
            __builtin_amdgcn_sched_barrier(0);
            if (LATE) __syncthreads();
            commit((buf ^ 1) & (NBUF - 1));
#pragma unroll
            for (int k = 0; k < 3; ++k) bc[k] = bn[k];
            __syncthreads();
        }
    }

    if (PIPE && PARTS != 1) acc[3] += t1;  // the last block's temporary
    if (PARTS == 2 || H1) {  // undo the operand scales once, on the finished sums
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] *= osc;
    }
    // ---- epilogue: C/D map of the 32x32 MFMA -- column (out channel) = lane & 31, row (pixel of the 4x8
    // tile) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); M-tile m covers rows 4(m>>1).., cols 8(m&1)..
    const int co = ntile * 32 + row;
    // per-lane base at the patch's pixel (ty0, tx0 + 4*half); every (m, r) then adds a wave-uniform offset
    // (rows and columns of the C/D map are compile-time): scalar address arithmetic instead of a 64-bit
    // multiply chain per output element
    const unsigned pix0 = (unsigned)(ty0 * a.W + tx0 + 4 * half);
    float *outp = a.out + (size_t)b * a.H * a.W * a.out_cs + co + (size_t)pix0 * a.out_cs;
    const float *resp = a.res ? a.res + (size_t)b * a.H * a.W * a.res_cs + co + (size_t)pix0 * a.res_cs : nullptr;
    const float sc = a.scale ? a.scale[co] : 1.f, sf = a.shift ? a.shift[co] : 0.f;
    const bool full = (ty0 + C2_TY <= a.H) && (tx0 + C2_TX <= a.W);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cy = 4 * (m >> 1) + (r >> 2), cx = 8 * (m & 1) + (r & 3);  // compile-time
            if (!(full || (ty0 + cy < a.H && tx0 + cx + 4 * half < a.W))) continue;
            const unsigned dpix = (unsigned)(cy * a.W + cx);  // wave-uniform
            const unsigned pix = pix0 + dpix;
            float y = acc[m][r] * sc + sf;
            if (resp) y += resp[dpix * (unsigned)a.res_cs];
            if (a.relu == 1) y = fmaxf(y, 0.f);
            if constexpr (PARTS == 1) {  // the gate activations exist only in the plain-bf16 (GRU) instantiations
            if (a.relu == 2) y = 1.f / (1.f + __expf(-y));
            else if (a.relu >= 3) {
                const float e2 = __expf(-2.f * fabsf(y));  // tanh without overflow
                const float th = (1.f - e2) / (1.f + e2);
                y = y < 0.f ? -th : th;
                if (a.relu == 4) {  // GRU state update (update.py:40): h' = (1 - z) h + z q
                    const size_t ib = (size_t)b * a.H * a.W;
                    const float z = a.gz[(ib + pix) * a.gz_cs + co], hp = a.gh[(ib + pix) * a.gh_cs + co];
                    y = (1.f - z) * hp + z * y;
                }
            }
            }
            outp[dpix * (unsigned)a.out_cs] = y;
        }
    if constexpr (STATS) {
        // BatchNorm partials of this wave's 32 channels over the patch (the layer's BatchNorm then skips its own
        // statistics pass over the tensor): lane = channel co, its 64 values + the other half-wave's 64
        float sm = 0.f;
        int n = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cy = 4 * (m >> 1) + (r >> 2), cx = 8 * (m & 1) + (r & 3);
                const bool ok = full || (ty0 + cy < a.H && tx0 + cx + 4 * half < a.W);
                sm += ok ? acc[m][r] : 0.f;
                n += ok ? 1 : 0;
            }
        sm += __shfl_xor(sm, 32);
        n += __shfl_xor(n, 32);
        const float mean = sm / (float)max(n, 1);
        float m2 = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cy = 4 * (m >> 1) + (r >> 2), cx = 8 * (m & 1) + (r & 3);
                const bool ok = full || (ty0 + cy < a.H && tx0 + cx + 4 * half < a.W);
                const float d = acc[m][r] - mean;
                m2 = ok ? fmaf(d, d, m2) : m2;
            }
        m2 += __shfl_xor(m2, 32);
        const int per = a.B / a.sgroups, g = b / per;
        const long long tile = ((long long)(b - g * per) * a.tiles_y + tiy) * a.tiles_x + tix;
        if (half == 0)
            *reinterpret_cast<float2 *>(&a.spart[(((size_t)g * a.cout + co) * a.stiles + tile) * 2]) = make_float2(sm, m2);
        if (ntile == 0 && lane == 0) a.scnt[(size_t)g * a.stiles + tile] = (float)n;
    }
}

// ---- weight packing ---------------------------------------------------------------------------------
// packed[(((t*NCH + cc)*NT + n)*3 + p)*64 + lane][j] (bf16) = part p of
//     w[(n*32 + (lane&31)) * s_co + (cc*16 + 8*(lane>>5) + j) * s_ci + (flip ? T-1-t : t)]
// zero where the operation's channel index exceeds the tensor's (co >= co_real / ci >= ci_real): layers
// whose channel counts are not multiples of 32 / 16 are padded here, not in the activations' producers.
__global__ void __launch_bounds__(256)
conv2d_pack_kernel(unsigned short *__restrict__ dst, const float *__restrict__ src, int cin, int cout,
                   int ci_real, int co_real, long long s_co, long long s_ci, int taps, int flip, int parts,
                   long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    long long r = idx >> 9;
    const int p = (int)(r % parts); r /= parts;
    const int nt = cout / 32, nch = cin / 16;
    const int n = (int)(r % nt); r /= nt;
    const int cc = (int)(r % nch);
    const int t = (int)(r / nch);
    const int co = n * 32 + (lane & 31), ci = cc * 16 + 8 * (lane >> 5) + j;
    float x = 0.f;
    if (co < co_real && ci < ci_real) x = src[co * s_co + ci * s_ci + (flip ? taps - 1 - t : t)];
    dst[idx] = az_split3_part(x, p);  // round-to-nearest split, as the activations' (az_common.h)
}

// f16x3 image: [tap][cin/16][cout/32][part 2][64 lanes][8] fp16 of w * 2^k (k from max |w|): az_pack_f16.h, AZ_PACK_2D_SAME
__global__ void __launch_bounds__(256)
conv2d_pack_f16_kernel(const AzPackDesc d, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const float scale = az_pow2(az_f16_scale_exp(az_amax_read(d.amax)));  // (before the early exit: a wave-wide read)
    if (idx >= total) return;
    reinterpret_cast<unsigned short *>(d.dst)[idx] = az_pack_f16_elem(d, idx, scale);
}

extern "C" int az_conv2d_pack_weights_f16(float *packed, const float *w, const float *w_amax, int cin, int cout,
                                          int ci_real, int co_real, long long stride_out, long long stride_in, int kh,
                                          int kw, int flip, void *stream) {
    AZ_REQUIRE_PTR(packed); AZ_REQUIRE_PTR(w); AZ_REQUIRE_PTR(w_amax);
    if (cin <= 0 || cout <= 0 || cin % 16 || cout % 32 || kh <= 0 || kw <= 0) return AZ_EUNSUPPORTED;
    AZ_REQUIRE(ci_real > 0 && ci_real <= cin && co_real > 0 && co_real <= cout);
    const long long total = (long long)kh * kw * cin * cout * 2;
    AzPackDesc d{};
    d.dst = packed; d.src = w; d.amax = w_amax; d.s_co = stride_out; d.s_ci = stride_in; d.kind = AZ_PACK_2D_SAME;
    d.cin = cin; d.cout = cout; d.ci_real = ci_real; d.co_real = co_real; d.taps = kh * kw; d.flip = flip;
    hipLaunchKernelGGL(conv2d_pack_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, az_stream(stream), d, total);
    return az_launch_status();
}

extern "C" long long az_conv2d_packed_floats(int cin, int cout, int kh, int kw) {
    if (cin <= 0 || cout <= 0 || cin % 16 || cout % 32 || kh <= 0 || kw <= 0) return AZ_EINVAL;
    return (long long)kh * kw * cin * cout * 3 / 2;  // three bf16 per weight
}

extern "C" int az_conv2d_pack_weights(float *packed, const float *w, int cin, int cout, int ci_real, int co_real,
                                      long long stride_out, long long stride_in, int kh, int kw, int flip,
                                      void *stream) {
    AZ_REQUIRE_PTR(packed); AZ_REQUIRE_PTR(w);
    if (az_conv2d_packed_floats(cin, cout, kh, kw) < 0) return AZ_EUNSUPPORTED;
    AZ_REQUIRE(ci_real > 0 && ci_real <= cin && co_real > 0 && co_real <= cout);
    const long long total = (long long)kh * kw * cin * cout * 3;
    hipLaunchKernelGGL(conv2d_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, az_stream(stream),
                       reinterpret_cast<unsigned short *>(packed), w, cin, cout, ci_real, co_real, stride_out,
                       stride_in, kh * kw, flip, 3, total);
    return az_launch_status();
}

// plain bf16 image (one part): [tap][cin/16][cout/32][64 lanes][8] bf16 = kh*kw*cin*cout/2 floats
static int pack_bf16(float *packed, const float *w, int cin, int cout, long long stride_out, long long stride_in, int kh,
                     int kw, int flip, void *stream) {
    AZ_REQUIRE_PTR(packed); AZ_REQUIRE_PTR(w);
    if (az_conv2d_packed_floats(cin, cout, kh, kw) < 0) return AZ_EUNSUPPORTED;
    const long long total = (long long)kh * kw * cin * cout;
    hipLaunchKernelGGL(conv2d_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, az_stream(stream),
                       reinterpret_cast<unsigned short *>(packed), w, cin, cout, cin, cout, stride_out, stride_in,
                       kh * kw, flip, 1, total);
    return az_launch_status();
}
// f16x1 image: [tap][cin/16][cout/32][lane 64][8] fp16 of w * 2^k (k from w's amax array, 0 without one)
__global__ void __launch_bounds__(256)
conv2d_pack_h1_kernel(unsigned short *__restrict__ dst, const float *__restrict__ src, const float *__restrict__ amax, int cin,
                      int cout, long long s_co, long long s_ci, int taps, int flip, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const float scale = amax ? az_pow2(az_f16_scale_exp(az_amax_read(amax))) : 1.f;  // (before the early exit: a wave-wide read)
    if (idx >= total) return;
    const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    long long r = idx >> 9;
    const int nt = cout / 32, nch = cin / 16;
    const int n = (int)(r % nt); r /= nt;
    const int cc = (int)(r % nch);
    const int t = (int)(r / nch);
    const int co = n * 32 + (lane & 31), ci = cc * 16 + 8 * (lane >> 5) + j;
    dst[idx] = __builtin_bit_cast(unsigned short, (_Float16)(src[co * s_co + ci * s_ci + (flip ? taps - 1 - t : t)] * scale));
}
/* one-part fp16 image of a 3x3 weight (az_conv2d_h1_fwd); flip = 1 with (cin, cout, stride_out, stride_in) swapped: the
 * image of the layer's input gradient.  w_amax may be NULL (no scaling). */
extern "C" int az_conv2d_pack_weights_h1(float *packed, const float *w, const float *w_amax, int cin, int cout,
                                         long long stride_out, long long stride_in, int flip, void *stream) {
    AZ_REQUIRE_PTR(packed); AZ_REQUIRE_PTR(w);
    if (az_conv2d_packed_floats(cin, cout, 3, 3) < 0) return AZ_EUNSUPPORTED;
    const long long total = 9LL * cin * cout;
    hipLaunchKernelGGL(conv2d_pack_h1_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, az_stream(stream),
                       reinterpret_cast<unsigned short *>(packed), w, w_amax, cin, cout, stride_out, stride_in, 9, flip, total);
    return az_launch_status();
}
extern "C" int az_conv2d_pack_weights_bf16(float *packed, const float *w, int cin, int cout, long long stride_out,
                                           long long stride_in, int kh, int kw, void *stream) {
    return pack_bf16(packed, w, cin, cout, stride_out, stride_in, kh, kw, 0, stream);
}
/* the same with the taps flipped: with (cin, cout, stride_out, stride_in) swapped this is the packing of the layer's
 * input gradient (az_conv2d_bf16_fwd on the output gradient) */
extern "C" int az_conv2d_pack_weights_bf16_flipped(float *packed, const float *w, int cin, int cout, long long stride_out,
                                                   long long stride_in, int kh, int kw, void *stream) {
    return pack_bf16(packed, w, cin, cout, stride_out, stride_in, kh, kw, 1, stream);
}

template <int NW, int KH, int KW, int DIL, int PARTS = 3, bool STATS = false, bool H1 = false>
static int launch_c2(C2Args a, hipStream_t s) {
    a.ngroups = (a.cout / 32) / NW;
    const long long blocks = (long long)a.B * a.tiles_y * a.tiles_x * a.ngroups;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    hipLaunchKernelGGL((conv2d_same_kernel<NW, KH, KW, DIL, PARTS, STATS, H1>), dim3((unsigned)blocks), dim3(64 * NW), 0, s, a);
    return az_launch_status();
}

template <int KH, int KW, int DIL, int PARTS = 3, bool STATS = false, bool H1 = false>
static int dispatch_nw(const C2Args &a, hipStream_t s) {
    const int nt = a.cout / 32;
    if (nt % 4 == 0) return launch_c2<4, KH, KW, DIL, PARTS, STATS, H1>(a, s);
    if (nt % 3 == 0) return launch_c2<3, KH, KW, DIL, PARTS, STATS, H1>(a, s);
    if (nt % 2 == 0) return launch_c2<2, KH, KW, DIL, PARTS, STATS, H1>(a, s);
    return launch_c2<1, KH, KW, DIL, PARTS, STATS, H1>(a, s);
}

extern "C" int az_conv2d_fwd(float *out, const float *in, const float *packed_w, const float *scale,
                             const float *shift, const float *residual, int relu, int B, int H, int W,
                             int cin, int cout, int in_cstride, int out_cstride, int res_cstride, int kh, int kw,
                             int dilation, void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(in); AZ_REQUIRE_PTR(packed_w);
    AZ_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0 && cout > 0);
    if (cin % 16 || cout % 32) return AZ_EUNSUPPORTED;
    AZ_REQUIRE(in_cstride >= cin && out_cstride >= cout && in_cstride % 4 == 0);
    AZ_REQUIRE(residual == nullptr || res_cstride >= cout);
    {   // per-image bases are 64-bit; offsets inside one image are 32-bit
        const long long cs = in_cstride > out_cstride ? in_cstride : out_cstride;
        if ((long long)H * W * (cs > res_cstride ? cs : res_cstride) > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    }
    C2Args a{};
    a.in = in; a.wp = packed_w; a.out = out; a.scale = scale; a.shift = shift; a.res = residual; a.relu = relu ? 1 : 0;
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.cout = cout;
    a.in_cs = in_cstride; a.out_cs = out_cstride; a.res_cs = res_cstride;
    a.tiles_y = (H + C2_TY - 1) / C2_TY; a.tiles_x = (W + C2_TX - 1) / C2_TX;
    hipStream_t s = az_stream(stream);
    if (kh == 3 && kw == 3 && dilation == 1) return dispatch_nw<3, 3, 1>(a, s);
    if (kh == 3 && kw == 3 && dilation == 2) return dispatch_nw<3, 3, 2>(a, s);
    if (kh == 1 && kw == 1) return dispatch_nw<1, 1, 1>(a, s);
    if (kh == 3 && kw == 5 && dilation == 1) return dispatch_nw<3, 5, 1>(a, s);
    return AZ_EUNSUPPORTED;
}

/* az_conv2d_fwd without epilogue operands, plus the BatchNorm partials of the (raw) output: partials
 * [groups][cout][tiles][2] = (sum, M2 about the patch mean) per channel and 8x16 patch, counts [groups][tiles],
 * tiles = az_conv2d_stats_tiles(B, H, W, groups) -- what az_bn2d_fwd takes in place of its own statistics pass.
 * 3x3 (dilation 1, 2) and 1x1 layers. */
extern "C" long long az_conv2d_stats_tiles(int B, int H, int W, int groups) {
    if (B <= 0 || H <= 0 || W <= 0 || groups <= 0 || B % groups) return AZ_EINVAL;
    return (long long)(B / groups) * ((H + C2_TY - 1) / C2_TY) * ((W + C2_TX - 1) / C2_TX);
}

extern "C" int az_conv2d_fwd_stats(float *out, float *partials, float *counts, const float *in, const float *packed_w,
                                   int groups, int B, int H, int W, int cin, int cout, int in_cstride,
                                   int out_cstride, int kh, int kw, int dilation, void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(partials); AZ_REQUIRE_PTR(counts); AZ_REQUIRE_PTR(in); AZ_REQUIRE_PTR(packed_w);
    AZ_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0 && cout > 0 && groups > 0 && B % groups == 0);
    if (cin % 16 || cout % 32) return AZ_EUNSUPPORTED;
    AZ_REQUIRE(in_cstride >= cin && out_cstride >= cout && in_cstride % 4 == 0);
    if ((long long)H * W * (in_cstride > out_cstride ? in_cstride : out_cstride) > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    C2Args a{};
    a.in = in; a.wp = packed_w; a.out = out;
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.cout = cout;
    a.in_cs = in_cstride; a.out_cs = out_cstride;
    a.tiles_y = (H + C2_TY - 1) / C2_TY; a.tiles_x = (W + C2_TX - 1) / C2_TX;
    a.spart = partials; a.scnt = counts; a.sgroups = groups; a.stiles = az_conv2d_stats_tiles(B, H, W, groups);
    hipStream_t s = az_stream(stream);
    if (kh == 3 && kw == 3 && dilation == 1) return dispatch_nw<3, 3, 1, 3, true>(a, s);
    if (kh == 3 && kw == 3 && dilation == 2) return dispatch_nw<3, 3, 2, 3, true>(a, s);
    if (kh == 1 && kw == 1) return dispatch_nw<1, 1, 1, 3, true>(a, s);
    return AZ_EUNSUPPORTED;
}

/* 3x3 stride-1 "same" convolution with plain bf16 operands (one MFMA per block, fp32 accumulation, fp32 in / out):
 * out = act(conv(in) + bias[co] + residual); act 0 none, 1 ReLU, 2 sigmoid, 3 tanh, 4 = the GRU state update
 * (1 - z) * h + z * tanh(.) with gate z and previous state h read at gate_z / gate_h. */
static int conv2d_onepart_fwd(float *out, const float *in, const float *packed_w, const float *bias,
                              const float *residual, const float *gate_z, const float *gate_h, int act, int B,
                              int H, int W, int cin, int cout, int in_cstride, int out_cstride, int res_cstride,
                              int z_cstride, int h_cstride, void *stream, bool h1, const float *in_amax, const float *w_amax) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(in); AZ_REQUIRE_PTR(packed_w);
    AZ_REQUIRE(B > 0 && H > 0 && W > 0 && cin > 0 && cout > 0 && act >= 0 && act <= 4);
    if (cin % 16 || cout % 32) return AZ_EUNSUPPORTED;
    AZ_REQUIRE(in_cstride >= cin && out_cstride >= cout && in_cstride % 4 == 0);
    AZ_REQUIRE(residual == nullptr || res_cstride >= cout);
    if (act == 4) { AZ_REQUIRE_PTR(gate_z); AZ_REQUIRE_PTR(gate_h); AZ_REQUIRE(z_cstride >= cout && h_cstride >= cout); }
    {
        long long cs = in_cstride > out_cstride ? in_cstride : out_cstride;
        if (res_cstride > cs) cs = res_cstride;
        if ((long long)H * W * cs > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    }
    C2Args a{};
    a.in = in; a.wp = packed_w; a.out = out; a.scale = nullptr; a.shift = bias; a.res = residual; a.relu = act;
    a.gz = gate_z; a.gh = gate_h; a.gz_cs = z_cstride; a.gh_cs = h_cstride;
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.cout = cout;
    a.in_cs = in_cstride; a.out_cs = out_cstride; a.res_cs = res_cstride;
    a.tiles_y = (H + C2_TY - 1) / C2_TY; a.tiles_x = (W + C2_TX - 1) / C2_TX;
    a.in_amax = in_amax; a.w_amax = w_amax;
    if (h1) return dispatch_nw<3, 3, 1, 1, false, true>(a, az_stream(stream));
    return dispatch_nw<3, 3, 1, 1>(a, az_stream(stream));
}
extern "C" int az_conv2d_bf16_fwd(float *out, const float *in, const float *packed_w, const float *bias,
                                  const float *residual, const float *gate_z, const float *gate_h, int act, int B,
                                  int H, int W, int cin, int cout, int in_cstride, int out_cstride, int res_cstride,
                                  int z_cstride, int h_cstride, void *stream) {
    return conv2d_onepart_fwd(out, in, packed_w, bias, residual, gate_z, gate_h, act, B, H, W, cin, cout, in_cstride, out_cstride,
                              res_cstride, z_cstride, h_cstride, stream, false, nullptr, nullptr);
}
/* the same with ONE FP16 part per operand ("f16x1": what torch.cuda.amp.autocast computes on CUDA, raft_stereo.py:14): weights
 * packed by az_conv2d_pack_weights_h1; in_amax / w_amax (either may be NULL = scale 1): amax arrays for a power-of-two operand
 * scale -- the gradient operands of the backward launches get one, in place of the reference's GradScaler */
extern "C" int az_conv2d_h1_fwd(float *out, const float *in, const float *packed_w, const float *in_amax, const float *w_amax,
                                const float *bias, const float *residual, const float *gate_z, const float *gate_h, int act, int B,
                                int H, int W, int cin, int cout, int in_cstride, int out_cstride, int res_cstride,
                                int z_cstride, int h_cstride, void *stream) {
    return conv2d_onepart_fwd(out, in, packed_w, bias, residual, gate_z, gate_h, act, B, H, W, cin, cout, in_cstride, out_cstride,
                              res_cstride, z_cstride, h_cstride, stream, true, in_amax, w_amax);
}

/* az_conv2d_fwd / az_conv2d_fwd_stats on the f16x3 arithmetic (include/azhip.h): in_amax / w_amax = device scalars
 * max |in|, max |w|; weights packed by az_conv2d_pack_weights_f16. */
static int c2_f16_args(C2Args &a, float *out, const float *in, const float *packed_w, const float *in_amax,
                       const float *w_amax, int B, int H, int W, int cin, int cout, int in_cstride, int out_cstride,
                       int res_cstride) {
    if (!out || !in || !packed_w || !in_amax || !w_amax) return AZ_ENULL;
    if (!(B > 0 && H > 0 && W > 0 && cin > 0 && cout > 0)) return AZ_EINVAL;
    if (cin % 16 || cout % 32) return AZ_EUNSUPPORTED;
    if (!(in_cstride >= cin && out_cstride >= cout && in_cstride % 4 == 0)) return AZ_EINVAL;
    long long cs = in_cstride > out_cstride ? in_cstride : out_cstride;
    if (res_cstride > cs) cs = res_cstride;
    if ((long long)H * W * cs > 0x7fffffffLL) return AZ_EUNSUPPORTED;
    a.in = in; a.wp = packed_w; a.out = out; a.in_amax = in_amax; a.w_amax = w_amax;
    a.B = B; a.H = H; a.W = W; a.cin = cin; a.cout = cout;
    a.in_cs = in_cstride; a.out_cs = out_cstride; a.res_cs = res_cstride;
    a.tiles_y = (H + C2_TY - 1) / C2_TY; a.tiles_x = (W + C2_TX - 1) / C2_TX;
    return AZ_OK;
}

extern "C" int az_conv2d_fwd_f16(float *out, const float *in, const float *packed_w, const float *in_amax,
                                 const float *w_amax, const float *scale, const float *shift, const float *residual,
                                 int relu, int B, int H, int W, int cin, int cout, int in_cstride, int out_cstride,
                                 int res_cstride, int kh, int kw, int dilation, void *stream) {
    C2Args a{};
    if (int e = c2_f16_args(a, out, in, packed_w, in_amax, w_amax, B, H, W, cin, cout, in_cstride, out_cstride,
                            residual ? res_cstride : 0)) return e;
    AZ_REQUIRE(residual == nullptr || res_cstride >= cout);
    a.scale = scale; a.shift = shift; a.res = residual; a.relu = relu ? 1 : 0;
    hipStream_t s = az_stream(stream);
    if (kh == 3 && kw == 3 && dilation == 1) return dispatch_nw<3, 3, 1, 2>(a, s);
    if (kh == 3 && kw == 3 && dilation == 2) return dispatch_nw<3, 3, 2, 2>(a, s);
    if (kh == 1 && kw == 1) return dispatch_nw<1, 1, 1, 2>(a, s);
    if (kh == 3 && kw == 5 && dilation == 1) return dispatch_nw<3, 5, 1, 2>(a, s);
    return AZ_EUNSUPPORTED;
}

extern "C" int az_conv2d_fwd_stats_f16(float *out, float *partials, float *counts, const float *in, const float *packed_w,
                                       const float *in_amax, const float *w_amax, int groups, int B, int H, int W,
                                       int cin, int cout, int in_cstride, int out_cstride, int kh, int kw, int dilation,
                                       void *stream) {
    AZ_REQUIRE_PTR(partials); AZ_REQUIRE_PTR(counts);
    AZ_REQUIRE(groups > 0 && B > 0 && B % groups == 0);
    C2Args a{};
    if (int e = c2_f16_args(a, out, in, packed_w, in_amax, w_amax, B, H, W, cin, cout, in_cstride, out_cstride, 0)) return e;
    a.spart = partials; a.scnt = counts; a.sgroups = groups; a.stiles = az_conv2d_stats_tiles(B, H, W, groups);
    hipStream_t s = az_stream(stream);
    if (kh == 3 && kw == 3 && dilation == 1) return dispatch_nw<3, 3, 1, 2, true>(a, s);
    if (kh == 3 && kw == 3 && dilation == 2) return dispatch_nw<3, 3, 2, 2, true>(a, s);
    if (kh == 1 && kw == 1) return dispatch_nw<1, 1, 1, 2, true>(a, s);
    return AZ_EUNSUPPORTED;
}
