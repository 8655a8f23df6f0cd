// Weight gradient of the stride-1 3x3x3 layers (bf16x6): the V0 32 -> 32 layers and, as 2 x 2 tiles of 32 x 32 channels in
// the grid, the 64 -> 64 layers of the hourglasses (0.94 -> 0.84 ms at 1/8 resolution, 0.44 -> 0.50 of the roofline),
// rebuilt the way az_conv3d_roll.hip rebuilt the
// forward / input-gradient kernel (psmnet_3.py:87-117 dres0..dres4 / classif convs; G = dW[co][ci][27]):
//
//   G[m][n][kd,kh,kw] = sum over positions (b, d, h, w) of  coarse[b,d,h,w][m] * fine[b, d-1+kd, h-1+kh, w-1+kw][n]
//
// az_conv3d_wgrad.hip gives a wave ONE kd (nine 32x32 accumulators = 144 registers): the coarse row is fetched, split
// into its bf16 triplet and written to LDS by three waves, every fine row by three waves too, and that staging sits
// between two barriers of a one-wave workgroup (20-30 % of the kernel, profiles/r02_clock_pipe_ablation.md section 7).
// Here, on v_mfma_f32_16x16x32_bf16:
//   * a wave owns a 16 x 16 block (half of the coarse channels x half of the fine channels) of ALL 27 taps: 27 x 4 = 108
//     accumulator registers; four waves = one workgroup cover 32 x 32 and share every staged byte;
//   * K = 32 positions per MFMA = two adjacent coarse rows x 16 positions (240 = 15 x 16: no padded positions); a step
//     is 27 taps x 6 MFMAs per wave and needs, staged, 2 coarse rows and the 4-row window of three fine planes, of which
//     only 2 coarse + 3 x 2 fine rows are new: 140 positions per 162 MFMAs x 4 waves (was 34 per 54 MFMAs x 1 wave);
//   * the coarse pair is double-buffered, each fine plane keeps a ring of six rows: the next step's rows are split and
//     written while this step is multiplied (one barrier per step); loads, validity (zero padding in h / w / d) and
//     therefore vmcnt bookkeeping go through buffer instructions with out-of-range offsets: a step is one basic block;
//   * workgroups are persistent: each walks a list of (batch, coarse depth, 16-position chunk) columns and flushes its
//     27 x 32 x 32 block once, with float atomics, into the tap-major workspace of az_conv3d_wgrad.hip.
// Arithmetic per K block: the six-MFMA chain of az_conv3d_wgrad.hip (smallest terms first, fp32 accumulate).
//
// Measured (B=4, V0; profiles/r03_roll_kernel_notes.md section 5): 1.65-1.74 ms against 1.76-1.80 ms for the one-kd-per-
// wave kernel on the same boxes -- staging is now a third per MFMA and hidden, LDS bank conflicts are gone (46 % of the
// LDS cycles before the row-half swap), but a 16x16 block per tap gives every fine fragment exactly ONE use: 168
// transposing reads per 162 MFMAs, twice the LDS instructions per matrix cycle of the 32x32x16 form, and the kernel is
// bound by instruction issue (4.1 instructions per 16-cycle MFMA; +4 VALU per tap cost +15 %, fragments two taps ahead
// with recomputed addresses +13 %).
#include <stdlib.h>

#include "az_roll_common.h"
#include "az_options.h"
#include "az_launch_math.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

#define W16_POS 16                       // positions of a coarse row per step
#define W16_FW (W16_POS + 2)             // fine positions per staged row
#define W16_ROWB 64                      // bytes of one (position, 32 channels) bf16 row
#define W16_CBUF (3 * 32 * W16_ROWB)     // one coarse buffer: [part][k = 2 rows x 16][32 ch]      6 144 B
#define W16_FROW (3 * W16_FW * W16_ROWB) // one fine row: [part][18 positions][32 ch]              3 456 B
#define W16_RING 6
#define W16_LDS (2 * W16_CBUF + 3 * W16_RING * W16_FROW)  // 74 496 B
#define W16_NPOS (2 * W16_POS + 3 * 2 * W16_FW)           // positions staged per step: 140
#define W16_NQ (W16_NPOS * 8)                              // float4 pieces: 1 120
#define W16_NLD ((W16_NQ + 255) / 256)                     // 5 per thread
#define W16_OOB 0xffffff00u
#ifndef W16_TWO_CHAINS
#define W16_TWO_CHAINS 0  // experiment: two independent MFMA chains per tap
#endif

struct Wg16Args {
    const float *coarse, *fine;
    float *ws;  // [27][CM][CN]
    int B, D, H, W;
    int CM, CN;  // channels of coarse / fine (32 or 64 each): one workgroup owns ONE 32 x 32 (m, n) tile
    int nwchunk;
    long long ncols;  // B * D * nwchunk columns of work
    int wgs;          // persistent workgroups per tile
    int xcd;          // XCD-chunked column order
    const float *coarse_amax, *fine_amax;  // f16x3 (AR = 1): device scalars max |coarse|, max |fine|
};

// AR: 0 = bf16x6 (three bf16 parts, six MFMAs per tap), 1 = f16x3 (two scaled fp16 parts, three MFMAs; az_roll_common.h).
// The LDS images keep their three-part strides either way (the third part is unused with AR = 1).
// PSM (AR = 1): bit 0 = the coarse operand, bit 1 = the fine operand is a pre-split tensor (az_roll_common.h).
template <int AR, int PSM = 0>
__global__ void __launch_bounds__(256, 2)
conv3d_wgrad_r16_kernel(const Wg16Args a) {
    constexpr int NP = AR ? 2 : 3;
    __shared__ __attribute__((aligned(16))) unsigned char lds[W16_LDS + 64];  // + a sink for the lanes of a partial piece
    unsigned char *const cbuf = lds;                  // [2][W16_CBUF]
    unsigned char *const fring = lds + 2 * W16_CBUF;  // [plane kd][slot][W16_FROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = wv >> 1, ni = wv & 1;
    // 64-channel operands: 2 x 2 (or 2 x 1 / 1 x 2) tiles in the grid; the workgroups of tile t are blockIdx.x in [t wgs, (t + 1) wgs)
    const int ntn = a.CN >> 5;
    // workgroups b and b + 8 share an XCD (and its L2): each XCD gets a contiguous run of the column list -- neighbouring depths
    // of one chunk, which read the same fine planes -- instead of every eighth column (AZ_WGRAD_R16_XCD=0: the linear order)
    const int tile = blockIdx.x / a.wgs, wgl = blockIdx.x - tile * a.wgs;
    const int wg0 = (a.xcd && !(a.wgs & 7)) ? az_xcd_map(wgl, a.wgs) : wgl;
    const int m0 = (tile / ntn) * 32, n0 = (tile % ntn) * 32;

    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float c_scale = 1.f, f_scale = 1.f, o_scale = 1.f;
    if (AR) {  // wave-uniform power-of-two operand scales
        const int kc = az_f16_scale_exp(az_amax_read(a.coarse_amax));
        const int kf = az_f16_scale_exp(az_amax_read(a.fine_amax));
        c_scale = az_pow2(kc); f_scale = az_pow2(kf);
        o_scale = ldexpf(1.f, -(kc + kf));
    }

    // transposing-read geometry (ds_read_b64_tr_b16 on a [k][32 ch] bf16 image): a 16-lane group (= one K octet) reads
    // 4 k-rows x 16 channels; lane 4q + p supplies the address of row q, channels 4p..4p+3 and receives channel (lane & 15)
    // Banks: a 64-B row is 16 banks, so rows r and r + 8 of an image fall on the same banks, and the two octets of a
    // 32-lane half read exactly such rows (k-rows 8 apart): a 2-way conflict on every read (measured: 46 % of the LDS
    // cycles).  The two 32-byte channel halves of a row are therefore swapped on rows with bit 3 set (writes and reads
    // XOR the in-row byte offset with ((row >> 3) & 1) << 5).
    const int oct = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const unsigned a_lane = (unsigned)(8 * oct + tq) * W16_ROWB + (((unsigned)(16 * mi + 4 * tp) * 2) ^ ((unsigned)(oct & 1) << 5));
    // fine: octet -> (row of the pair, position 8 (oct & 1) ..): the row-in-window part is added per step (ring slots);
    // the LDS row of a read is position + kw (+ 4 for the second half of a fragment): offsets per kw, built once
    const int rp = oct >> 1;
    unsigned b_off[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const unsigned rowi = (unsigned)(8 * (oct & 1) + tq + kw + 4 * h2);
            b_off[kw][h2] = rowi * W16_ROWB + (((unsigned)(16 * ni + 4 * tp) * 2) ^ (((rowi >> 3) & 1u) << 5));
        }

    auto frag2 = [&](const unsigned char *lo, const unsigned char *hi) -> az_bf16x8 {  // k rows 0..3 at lo, 4..7 at hi
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lo));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(hi));
        s16x8 v;
        v[0] = lo4[0]; v[1] = lo4[1]; v[2] = lo4[2]; v[3] = lo4[3];
        v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
        return __builtin_bit_cast(az_bf16x8, v);
    };

    const unsigned vb_c = (unsigned)a.CM * 4u, vb_f = (unsigned)a.CN * 4u;  // bytes per voxel
    const unsigned plane_c = (unsigned)a.H * a.W * vb_c, plane_f = (unsigned)a.H * a.W * vb_f;
    for (long long col = wg0; col < a.ncols; col += a.wgs) {
        // column -> (w chunk, coarse depth, batch); consecutive columns = consecutive depths of one chunk: the
        // workgroups resident together read neighbouring planes of the same rows (one L2 serves the three kd)
        long long r_ = col;
        const int cd = (int)(r_ % a.D); r_ /= a.D;
        const int wc = (int)(r_ % a.nwchunk);
        const int b = (int)(r_ / a.nwchunk);
        const int cw0 = wc * W16_POS;
        const unsigned vol_c = (unsigned)a.D * plane_c, vol_f = (unsigned)a.D * plane_f;
        const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.coarse) + (size_t)b * (vol_c / 4) + m0, 0, vol_c, 0x00020000);
        const auto rs_f = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.fine) + (size_t)b * (vol_f / 4) + n0, 0, vol_f, 0x00020000);

        // ---- staging set of step s: coarse rows 2s, 2s+1 (buffer s & 1); fine rows 2s-1+{2,3}... see below ----------
        // piece q = tid + 256 it of a set:  q < 256: coarse (row pair, 16 positions, 8 float4 per position);
        // q >= 256: fine, plane kd = (q - 256) / 288, new row j = .. / 144, position .. / 8  (18 x 8 = 144 per row)
        u32x4 pre[W16_NLD];
        auto issue = [&](int crow0, bool with_coarse, int frow0) {  // coarse rows crow0, crow0+1; fine rows frow0, frow0+1
#pragma unroll
            for (int it = 0; it < W16_NLD; ++it) {
                const int q = tid + 256 * it;
                unsigned off = W16_OOB;
                if (it == 0) {  // q < 256: coarse
                    const int k = q >> 3, row = crow0 + (k >> 4), cw = cw0 + (k & 15);
                    if (with_coarse && row < a.H && cw < a.W)
                        off = (unsigned)cd * plane_c + (unsigned)(row * a.W + cw) * vb_c + (unsigned)(q & 7) * 16u;
                    pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, off, 0, 0);
                } else {
                    const int f = q - 256;
                    const int kd = f / 288, g = f - kd * 288;
                    const int j = g / 144, pp = (g - j * 144) >> 3;
                    const int fd = cd - 1 + kd, fr = frow0 + j, fw = cw0 - 1 + pp;
                    if (q < W16_NQ && (unsigned)fd < (unsigned)a.D && (unsigned)fr < (unsigned)a.H && (unsigned)fw < (unsigned)a.W)
                        off = (unsigned)fd * plane_f + (unsigned)(fr * a.W + fw) * vb_f + (unsigned)(q & 7) * 16u;
                    pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_f, off, 0, 0);
                }
            }
        };
        auto commit_piece = [&](int it, int cbuf_idx, int frow0) {
            const int q = tid + 256 * it;
            uint2 hi, mid, lo;
            if (AR) {
                if (it == 0) az_stage_f16x4<(PSM & 1) != 0>(pre[it], c_scale, hi, mid);
                else az_stage_f16x4<(PSM & 2) != 0>(pre[it], f_scale, hi, mid);
                lo = mid;
            } else {
                az_split3_bf16x4(__builtin_bit_cast(float4, pre[it]), hi, mid, lo);
            }
            unsigned char *dst;
            unsigned part_stride;
            if (it == 0) {
                dst = cbuf + cbuf_idx * W16_CBUF + (q >> 3) * W16_ROWB + (((q & 7) * 8) ^ ((((q >> 3) >> 3) & 1) << 5));
                part_stride = 32 * W16_ROWB;
            } else {
                const int f = q - 256;
                const int kd = f / 288, g = f - kd * 288;
                const int j = g / 144, pp = (g - j * 144) >> 3;
                const int slot = (frow0 + j + W16_RING) % W16_RING;
                dst = fring + (kd * W16_RING + slot) * W16_FROW + pp * W16_ROWB + (((q & 7) * 8) ^ (((pp >> 3) & 1) << 5));
                part_stride = W16_FW * W16_ROWB;
            }
            if (q >= W16_NQ) { dst = lds + W16_LDS + (tid & 7) * 8; part_stride = 0; }  // (no branch: a step stays one block)
            *reinterpret_cast<uint2 *>(dst) = hi;
            *reinterpret_cast<uint2 *>(dst + part_stride) = mid;
            if (!AR) *reinterpret_cast<uint2 *>(dst + 2 * part_stride) = lo;
        };

        // ---- prologue: the window of step 0 (fine rows -1 .. 2, coarse rows 0, 1), then the request for step 1 -------
        __syncthreads();  // the previous column's last step no longer reads
        issue(0, false, -1);
#pragma unroll
        for (int it = 0; it < W16_NLD; ++it) commit_piece(it, 0, -1);
        issue(0, true, 1);
#pragma unroll
        for (int it = 0; it < W16_NLD; ++it) commit_piece(it, 0, 1);
        issue(2, true, 3);
        __syncthreads();

        const int nsteps = (a.H + 1) / 2;
        for (int s = 0; s < nsteps; ++s) {
            const int ch = 2 * s;
            const unsigned char *ca = cbuf + (s & 1) * W16_CBUF + a_lane;
            // ring slots of fine rows ch-1 .. ch+2 (wave-uniform), then this lane's (rows rp + kh)
            int slot[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) slot[r] = (ch - 1 + r + W16_RING) % W16_RING;
            unsigned fb[3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) fb[kh] = (unsigned)(rp ? slot[kh + 1] : slot[kh]) * W16_FROW;

            az_bf16x8 af[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) af[p] = frag2(ca + p * 32 * W16_ROWB, ca + p * 32 * W16_ROWB + 4 * W16_ROWB);
            az_bf16x8 bf[2][NP];
            auto load_b = [&](az_bf16x8 (&bq)[NP], int t) {
                const int kd = t / 9, kh = (t % 9) / 3, kw = t % 3;
                const unsigned char *fp = fring + kd * (W16_RING * W16_FROW) + fb[kh];
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    bq[p] = frag2(fp + b_off[kw][0] + p * W16_FW * W16_ROWB, fp + b_off[kw][1] + p * W16_FW * W16_ROWB);
            };
            load_b(bf[0], 0);
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < 27) load_b(bf[(t + 1) & 1], t + 1);
                __builtin_amdgcn_sched_barrier(0);
                f32x4 c = acc[t];
                const az_bf16x8(&bq)[NP] = bf[t & 1];
                if constexpr (AR) {  // lo*hi, hi*lo, hi*hi chained into the running accumulator (smallest first, as below)
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(az_f16x8, af[1]), __builtin_bit_cast(az_f16x8, bq[0]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(az_f16x8, af[0]), __builtin_bit_cast(az_f16x8, bq[1]), c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(az_f16x8, af[0]), __builtin_bit_cast(az_f16x8, bq[0]), c, 0, 0, 0);
                } else {
#if W16_TWO_CHAINS
                f32x4 u = {0.f, 0.f, 0.f, 0.f};
                u = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bq[0], u, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bq[2], c, 0, 0, 0);
                u = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bq[1], u, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bq[0], c, 0, 0, 0);
                u = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bq[1], u, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bq[0], c, 0, 0, 0);
                c += u;
#else
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2], bq[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bq[2], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bq[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1], bq[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bq[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bq[0], c, 0, 0, 0);
#endif
                }
                acc[t] = c;
                // the set of step s+1 (requested a step ago): one piece after each of the taps 1, 3, 5, 7, 9; then the
                // request for step s+2 (17 taps + the next step's first ones to land)
                if (t >= 1 && t <= 9 && (t & 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    commit_piece((t - 1) / 2, (s + 1) & 1, ch + 3);
                }
                if (t == 10) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue(ch + 4, true, ch + 5);
                }
            }
            __syncthreads();  // next step's rows written by all four waves; this step's no longer read
        }
    }
    // D[i][j]: i = coarse channel 16 mi + 4 (lane >> 4) + r, j = fine channel 16 ni + (lane & 15)
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * mi + 4 * (lane >> 4) + r;
            atomicAdd(&a.ws[((size_t)t * a.CM + m) * a.CN + n0 + 16 * ni + (lane & 15)], AR ? acc[t][r] * o_scale : acc[t][r]);
        }
}

// ---- round 5: the same kernel with TAPS, not channel blocks, split over the four waves ("wide tile") -------------------
// Counted from the ISA of the kernel above (f16x3, one step of one wave): 81 MFMAs (16 cycles each) against 114 transposing
// LDS reads -- every 16 x 16 block of every tap fetches its own fine fragment, and each fragment is fetched by the two waves
// that share its channel half.  One ds_read_b64_tr_b16 moves 512 bytes, the LDS delivers 128 bytes per clock and CU: eight
// resident waves need 8 x 114 x 4 = 3 648 LDS cycles per step for 2 x 81 x 16 = 2 592 matrix cycles per SIMD.  The kernel
// is LDS-bandwidth bound (counter pass: matrix pipe 0.445 busy), and no scheduling inside it can change that.
// Here a wave owns the WHOLE 32 x 32 channel tile of SEVEN taps (wave w: taps 7w .. 7w + 6; the last wave's seventh is a
// duplicate that is not flushed) on v_mfma_f32_32x32x16_f16: a fragment read feeds 32 x 32 x 16 products instead of
// 16 x 16 x 32 (same bytes, twice the flops), every fine fragment is read by exactly one wave, the coarse fragment of a
// 16-position sub-block by all four.  Per wave and step: 8 + 7 x 2 x 4 = 64 reads for 42 MFMAs of 32 cycles -- 2 048 LDS
// cycles per CU and step against 2 688 matrix cycles per SIMD.  Accumulators: 7 x 16 = 112 registers (108 before).
// Staging, rings, buffer-instruction bounds, persistent columns and the LDS images (and their row-half swap: four
// consecutive 64-byte rows cover the 64 banks once, whichever rows are swapped) are the kernel's above, unchanged.
// MEASURED (profiles/r05l_wgrad_wide_vs_narrow.md), and why it is NOT the default (AZ_WGRAD_R16_WIDE=1 selects it): the
// arithmetic above holds -- SQ_WAIT_INST_LDS 1.6e8 -> 3.9e7, SQ_LDS_IDX_ACTIVE 2.3e8 -> 1.3e8, bank conflicts 1.4e7 -> 0,
// wave cycles 9.2e8 -> 8.0e8 (-13 %) -- and the kernel takes the SAME time, 1.12 against 1.11 ms alone, 85.6 against 85.4 ms
// per step: the chip gives the saved cycles back as clock (the 32x32x16 shape holds a lower clock than 16x16x32 at equal
// flops, MI355X_MICROARCH.md DVFS notes).  At this matrix rate the V0 weight gradient is bound by what the part sustains
// under fp16 MFMA load, not by its LDS traffic.
template <int PSM>
__global__ void __launch_bounds__(256, 2)
conv3d_wgrad_r16w_kernel(const Wg16Args a) {
    constexpr int NP = 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[W16_LDS + 64];  // + a sink for the lanes of a partial piece
    unsigned char *const cbuf = lds;                  // [2][W16_CBUF]
    unsigned char *const fring = lds + 2 * W16_CBUF;  // [plane kd][slot][W16_FROW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntn = a.CN >> 5;
    const int tile = blockIdx.x / a.wgs, wgl = blockIdx.x - tile * a.wgs;
    const int wg0 = (a.xcd && !(a.wgs & 7)) ? az_xcd_map(wgl, a.wgs) : wgl;
    const int m0 = (tile / ntn) * 32, n0 = (tile % ntn) * 32;

    az_f32x16h acc[7];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const int kc = az_f16_scale_exp(az_amax_read(a.coarse_amax));
    const int kf = az_f16_scale_exp(az_amax_read(a.fine_amax));
    const float c_scale = az_pow2(kc), f_scale = az_pow2(kf), o_scale = ldexpf(1.f, -(kc + kf));

    // transposing-read geometry for the 32x32x16 operands on a [k][32 ch] fp16 image (az_conv2d_wgrad.hip): lane l holds
    // channel 16 ((l >> 4) & 1) + (l & 15), k = 8 (l >> 5) + j; a 16-lane group reads 4 k-rows x 16 channels, lane 4q + p of
    // it supplies the address of row q, channels 4p .. 4p + 3.  Rows with bit 3 set have their 32-byte halves swapped.
    const int g8 = lane >> 5, half = (lane >> 4) & 1, tq = (lane & 15) >> 2, tp = lane & 3;
    const unsigned chan_b = (unsigned)(16 * half + 4 * tp) * 2u;
    // coarse: k-row 16 s + 8 g8 + tq (+ 4): bit 3 of the row = g8
    const unsigned a_lane = (unsigned)(8 * g8 + tq) * W16_ROWB + (chan_b ^ ((unsigned)g8 << 5));
    // fine: position row kw + 8 g8 + tq (+ 4)
    unsigned b_off[3][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const unsigned rowi = (unsigned)(8 * g8 + tq + kw + 4 * h2);
            b_off[kw][h2] = rowi * W16_ROWB + (chan_b ^ (((rowi >> 3) & 1u) << 5));
        }
    // this wave's taps 7 wv + i (clamped to 26): a step must stay straight-line code although wv is a run-time value, so a
    // tap is reduced, once, to what its reads need -- a per-lane byte offset (plane kd + the two row groups of shift kw) in two
    // vector registers, and kh in two bits of one scalar
    unsigned boff[7][2], khpack = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int t = min(7 * wv + i, 26);
        const int kd = t / 9, kh = (t % 9) / 3, kw = t % 3;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
            boff[i][h2] = (unsigned)kd * (W16_RING * W16_FROW) + (kw == 0 ? b_off[0][h2] : kw == 1 ? b_off[1][h2] : b_off[2][h2]);
        khpack |= (unsigned)kh << (2 * i);
    }
    auto frag2 = [&](const unsigned char *lo, const unsigned char *hi) -> az_f16x8 {  // k rows 0..3 at lo, 4..7 at hi
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lo));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(hi));
        s16x8 v;
        v[0] = lo4[0]; v[1] = lo4[1]; v[2] = lo4[2]; v[3] = lo4[3];
        v[4] = hi4[0]; v[5] = hi4[1]; v[6] = hi4[2]; v[7] = hi4[3];
        return __builtin_bit_cast(az_f16x8, v);
    };

    const unsigned vb_c = (unsigned)a.CM * 4u, vb_f = (unsigned)a.CN * 4u;  // bytes per voxel
    const unsigned plane_c = (unsigned)a.H * a.W * vb_c, plane_f = (unsigned)a.H * a.W * vb_f;
    for (long long col = wg0; col < a.ncols; col += a.wgs) {
        long long r_ = col;
        const int cd = (int)(r_ % a.D); r_ /= a.D;
        const int wc = (int)(r_ % a.nwchunk);
        const int b = (int)(r_ / a.nwchunk);
        const int cw0 = wc * W16_POS;
        const unsigned vol_c = (unsigned)a.D * plane_c, vol_f = (unsigned)a.D * plane_f;
        const auto rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.coarse) + (size_t)b * (vol_c / 4) + m0, 0, vol_c, 0x00020000);
        const auto rs_f = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.fine) + (size_t)b * (vol_f / 4) + n0, 0, vol_f, 0x00020000);

        // ---- staging: exactly the kernel above (pieces, validity, ring slots) -------------------------------------------------
        u32x4 pre[W16_NLD];
        auto issue = [&](int crow0, bool with_coarse, int frow0) {
#pragma unroll
            for (int it = 0; it < W16_NLD; ++it) {
                const int q = tid + 256 * it;
                unsigned off = W16_OOB;
                if (it == 0) {
                    const int k = q >> 3, row = crow0 + (k >> 4), cw = cw0 + (k & 15);
                    if (with_coarse && row < a.H && cw < a.W)
                        off = (unsigned)cd * plane_c + (unsigned)(row * a.W + cw) * vb_c + (unsigned)(q & 7) * 16u;
                    pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_c, off, 0, 0);
                } else {
                    const int f = q - 256;
                    const int kd = f / 288, g = f - kd * 288;
                    const int j = g / 144, pp = (g - j * 144) >> 3;
                    const int fd = cd - 1 + kd, fr = frow0 + j, fw = cw0 - 1 + pp;
                    if (q < W16_NQ && (unsigned)fd < (unsigned)a.D && (unsigned)fr < (unsigned)a.H && (unsigned)fw < (unsigned)a.W)
                        off = (unsigned)fd * plane_f + (unsigned)(fr * a.W + fw) * vb_f + (unsigned)(q & 7) * 16u;
                    pre[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_f, off, 0, 0);
                }
            }
        };
        auto commit_piece = [&](int it, int cbuf_idx, int frow0) {
            const int q = tid + 256 * it;
            uint2 hi, lo;
            if (it == 0) az_stage_f16x4<(PSM & 1) != 0>(pre[it], c_scale, hi, lo);
            else az_stage_f16x4<(PSM & 2) != 0>(pre[it], f_scale, hi, lo);
            unsigned char *dst;
            unsigned part_stride;
            if (it == 0) {
                dst = cbuf + cbuf_idx * W16_CBUF + (q >> 3) * W16_ROWB + (((q & 7) * 8) ^ ((((q >> 3) >> 3) & 1) << 5));
                part_stride = 32 * W16_ROWB;
            } else {
                const int f = q - 256;
                const int kd = f / 288, g = f - kd * 288;
                const int j = g / 144, pp = (g - j * 144) >> 3;
                const int slot = (frow0 + j + W16_RING) % W16_RING;
                dst = fring + (kd * W16_RING + slot) * W16_FROW + pp * W16_ROWB + (((q & 7) * 8) ^ (((pp >> 3) & 1) << 5));
                part_stride = W16_FW * W16_ROWB;
            }
            if (q >= W16_NQ) { dst = lds + W16_LDS + (tid & 7) * 8; part_stride = 0; }  // (no branch: a step stays one block)
            *reinterpret_cast<uint2 *>(dst) = hi;
            *reinterpret_cast<uint2 *>(dst + part_stride) = lo;
        };

        __syncthreads();  // the previous column's last step no longer reads
        issue(0, false, -1);
#pragma unroll
        for (int it = 0; it < W16_NLD; ++it) commit_piece(it, 0, -1);
        issue(0, true, 1);
#pragma unroll
        for (int it = 0; it < W16_NLD; ++it) commit_piece(it, 0, 1);
        issue(2, true, 3);
        __syncthreads();

        const int nsteps = (a.H + 1) / 2;
        for (int s = 0; s < nsteps; ++s) {
            const int ch = 2 * s;
            const unsigned char *ca = cbuf + (s & 1) * W16_CBUF + a_lane;
            const int slot0 = (ch - 1 + W16_RING) % W16_RING;  // ring slot of fine row ch - 1 (wave-uniform)
            // coarse fragments of the two sub-blocks (k = the 16 positions of coarse row ch + sb), both parts
            az_f16x8 af[2][NP];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const unsigned char *cp = ca + sb * 16 * W16_ROWB + p * 32 * W16_ROWB;
                    af[sb][p] = frag2(cp, cp + 4 * W16_ROWB);
                }
            az_f16x8 bf[2][NP];
            auto load_b = [&](az_f16x8 (&bq)[NP], int u) {  // unit u = 2 i + sb: tap i of this wave, sub-block sb
                const int i = u >> 1, sb = u & 1;
                int r = slot0 + (int)((khpack >> (2 * i)) & 3u) + sb;  // fine row ch - 1 + kh + sb
                r = r >= W16_RING ? r - W16_RING : r;
                const unsigned char *fp = fring + (unsigned)r * W16_FROW;
#pragma unroll
                for (int p = 0; p < NP; ++p) bq[p] = frag2(fp + boff[i][0] + p * W16_FW * W16_ROWB, fp + boff[i][1] + p * W16_FW * W16_ROWB);
            };
            load_b(bf[0], 0);
#pragma unroll
            for (int u = 0; u < 14; ++u) {
                __builtin_amdgcn_sched_barrier(0);
                if (u + 1 < 14) load_b(bf[(u + 1) & 1], u + 1);
                __builtin_amdgcn_sched_barrier(0);
                az_f32x16h c = acc[u >> 1];
                const az_f16x8(&bq)[NP] = bf[u & 1];
                const az_f16x8(&aq)[NP] = af[u & 1];
                // lo*hi, hi*lo, hi*hi chained into the running accumulator (smallest first, as the kernel above)
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[1], bq[0], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[0], bq[1], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[0], bq[0], c, 0, 0, 0);
                acc[u >> 1] = c;
                // the set of step s+1 (requested a step ago): one piece after each of the units 1, 3, 5, 7, 9; then the
                // request for step s+2
                if (u >= 1 && u <= 9 && (u & 1)) {
                    __builtin_amdgcn_sched_barrier(0);
                    commit_piece((u - 1) / 2, (s + 1) & 1, ch + 3);
                }
                if (u == 10) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue(ch + 4, true, ch + 5);
                }
            }
            __syncthreads();  // next step's rows written by all four waves; this step's no longer read
        }
    }
    // D[i][j]: i = coarse channel (r & 3) + 8 (r >> 2) + 4 (lane >> 5), j = fine channel lane & 31; the last wave's seventh
    // tap is the duplicate of tap 26
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int t = 7 * wv + i;
        if (t > 26) break;  // (wave-uniform)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            atomicAdd(&a.ws[((size_t)t * a.CM + m) * a.CN + n0 + (lane & 31)], acc[i][r] * o_scale);
        }
    }
}

// persistent workgroups: at most 512 resident (2 per CU); the count that balances the columns best
int az_conv3d_wgrad_r16_launch(float *ws, const float *coarse, const float *fine, int B, int cm, int cn, int D, int H, int W, hipStream_t s,
                               const float *coarse_amax, const float *fine_amax, int split_mask) {
    if (!((cm == 32 || cm == 64) && (cn == 32 || cn == 64))) return AZ_EUNSUPPORTED;
    Wg16Args a{};
    a.coarse = coarse; a.fine = fine; a.ws = ws; a.coarse_amax = coarse_amax; a.fine_amax = fine_amax;
    a.B = B; a.D = D; a.H = H; a.W = W; a.CM = cm; a.CN = cn;
    const int ntiles = (cm / 32) * (cn / 32);
    const int slots = 512 / ntiles;  // resident workgroups per tile (two per CU)
    a.nwchunk = (W + W16_POS - 1) / W16_POS;
    a.ncols = (long long)B * D * a.nwchunk;
    if (!az_fits_buffer_offset((long long)D * H * W * (cm > cn ? cm : cn) * 4)) return AZ_EUNSUPPORTED;  // one batch element through a 32-bit offset
    const int best = az_wgrad16_workgroups(a.ncols, slots, ntiles, az_options().wgrad_r16_wgs);
    a.wgs = best;
    a.xcd = az_options().wgrad_r16_xcd;
    const dim3 grid((unsigned)(a.wgs * ntiles));
#ifdef WG16_SKIP  // timing-only build: the launch is left out (the workspace stays zero: those weights do not move) -- what the step
                  // would gain if these weight gradients were free (tools/abl_step_sensitivity.sh); -DWG16_SKIP=1: the V0 shapes only
    if (WG16_SKIP == 2 || (long long)D * H * W >= 1000000) return AZ_OK;
#endif
    if (!(coarse_amax && fine_amax)) {
        if (split_mask) return AZ_EINVAL;
        hipLaunchKernelGGL(conv3d_wgrad_r16_kernel<0>, grid, dim3(256), 0, s, a);
    } else if (az_options().wgrad_r16_wide) {  // taps split over the waves, 32x32x16 tiles (AZ_WGRAD_R16_WIDE=0: the kernel above)
        if (split_mask == 0) hipLaunchKernelGGL(conv3d_wgrad_r16w_kernel<0>, grid, dim3(256), 0, s, a);
        else if (split_mask == 1) hipLaunchKernelGGL(conv3d_wgrad_r16w_kernel<1>, grid, dim3(256), 0, s, a);
        else if (split_mask == 2) hipLaunchKernelGGL(conv3d_wgrad_r16w_kernel<2>, grid, dim3(256), 0, s, a);
        else if (split_mask == 3) hipLaunchKernelGGL(conv3d_wgrad_r16w_kernel<3>, grid, dim3(256), 0, s, a);
        else return AZ_EINVAL;
    } else if (split_mask == 0) hipLaunchKernelGGL((conv3d_wgrad_r16_kernel<1, 0>), grid, dim3(256), 0, s, a);
    else if (split_mask == 1) hipLaunchKernelGGL((conv3d_wgrad_r16_kernel<1, 1>), grid, dim3(256), 0, s, a);
    else if (split_mask == 2) hipLaunchKernelGGL((conv3d_wgrad_r16_kernel<1, 2>), grid, dim3(256), 0, s, a);
    else if (split_mask == 3) hipLaunchKernelGGL((conv3d_wgrad_r16_kernel<1, 3>), grid, dim3(256), 0, s, a);
    else return AZ_EINVAL;
    return az_launch_status();
}
