// The 32 -> 1 classifier convolutions of PSMNet (reference
// nets/psmnet/psmnet_3.py:103-117, :177-179: classifN[2] = Conv3d(32,1,3,pad 1) and the
// running sums cost2 = classif2(out2) + cost1, cost3 = classif3(out3) + cost2).
//
// N = 1 is not MFMA-shaped (and fp32 MFMA runs at the VALU rate anyway): these are
// VALU kernels, 2*27*32 flops per voxel against 128 B of input -> HBM/LDS bound.
//   fwd   : block = 8x16 output voxels marching over a depth segment (chosen per launch); each 10x18x32ch input
//           slab is staged once in LDS (voxel stride 36 dwords) and feeds three output planes
//           through rolling accumulators; a lane owns a channel pair of 8 adjacent voxels (weights in
//           registers), the 16 lanes of a voxel group reduce with DPP.  Optional fused "+ previous cost".
//   dgrad : gin[v][c] = sum_k gout[v+1-k] * w[c][k]; a lane owns one channel pair of EIGHT voxels adjacent
//           in x: each row of the 3x10x(32+2) gout halo tile in LDS is read once as 10 values and serves
//           8 voxels x 3 kw taps -- 27 wide LDS reads per 216 packed FMAs (the one-voxel-per-lane version
//           issued 27 ds_read_b32 per 54 and was LDS-issue bound); the lane's 54 weights live in
//           registers; every voxel's 128 B is written by 16 adjacent lanes.
//   wgrad : gw[c][k] = sum_v in[v][c] * gout[v+1-k]; a lane owns one channel quad of FOUR adjacent voxels
//           (6 tile values per row: one ds_read_b128 + one ds_read_b64), 108 register
//           accumulators per lane, LDS reduction per block, 864 float atomics per block.
#include "az_common.h"

#define C1_TH 8
#define C1_TW 32
#define C1_VS 36
#define C1_GP 36  // row pitch of the gout tile (dgrad / wgrad): rows start 16-byte aligned
// the multiply-adds of dgrad / wgrad are written on channel PAIRS: v_pk_fma_f32 (two fp32 lanes per issue slot,
// the scalar factor broadcast by op_sel); spelled out because hipcc's own pairing of the scalar form needs
// ~0.7 v_mov per FMA to line the operands up
typedef float c1_f2 __attribute__((ext_vector_type(2)));
#define C1_FMA2(a, b, c) __builtin_elementwise_fma(a, b, c)


// Marches along depth: input plane `id` is staged once and feeds the three output planes
// id+1 (kd=0), id (kd=1), id-1 (kd=2) through three rolling accumulators, so the slab and every
// LDS read are shared by three taps (the first version re-staged three planes per output plane:
// 3x the global and LDS traffic, 0.75 ms at B=4 full size).
// Lane mapping (second version; the first gave every lane ONE voxel and all 32 channels with scalar weights:
// 72 ds_read_b128 per 432 packed FMAs, LDS-issue bound at 0.47 ms): a lane owns one channel PAIR of EIGHT
// voxels adjacent in x.  A tile row is read once as 10 float2 (30 ds_read_b64 per plane) and serves
// 8 voxels x 3 kw x 3 kd = 216 packed FMAs against the lane's 27 weight pairs in registers; the 16 lanes of
// a voxel group then sum their channel pairs with four DPP row steps per output value.
#define C1F_TH 8
#define C1F_TW 16
template <int CTRL>
__device__ __forceinline__ float c1_dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__global__ void __launch_bounds__(256)
c1_fwd_kernel(float *__restrict__ out, const float *__restrict__ in, const float *__restrict__ w,
              const float *__restrict__ addend, const float *__restrict__ in_scale,
              const float *__restrict__ in_shift, int D, int H, int W, int tiles_x, int dseg) {
    constexpr int SX = C1F_TW + 2, SY = C1F_TH + 2;
    __shared__ __attribute__((aligned(16))) float slab[SY * SX * C1_VS];
    const int tx0 = (blockIdx.x % tiles_x) * C1F_TW, ty0 = (blockIdx.x / tiles_x) * C1F_TH;
    const int d0 = blockIdx.y * dseg, d1 = min(d0 + dseg, D), b = blockIdx.z;
    const int cp = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int ly = grp >> 1, lx0 = (grp & 1) * 8;
    const int oh = ty0 + ly;
    c1_f2 wp[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wp[t] = c1_f2{w[(cp * 2 + 0) * 27 + t], w[(cp * 2 + 1) * 27 + t]};
    c1_f2 a_prev[8], a_cur[8], a_next[8];  // output planes id-1, id, id+1
#pragma unroll
    for (int j = 0; j < 8; ++j) { a_prev[j] = c1_f2{0.f, 0.f}; a_cur[j] = c1_f2{0.f, 0.f}; a_next[j] = c1_f2{0.f, 0.f}; }
    // plane id+1 travels global -> registers while plane id is multiplied; it is written to LDS after the
    // block has finished reading plane id
    constexpr int NQ = SY * SX * 8, NLD = (NQ + 255) / 256;
    float4 pre[NLD];
    // in_scale / in_shift given: the operand is relu(in * scale + shift), the train-mode BatchNorm + ReLU of the
    // layer in front (psmnet_3.py:103-117 classifN[0..1]) applied HERE instead of by a pass of its own over
    // the 802 MB tensor; the zero padding stays zero.  A thread's channel quad is fixed (256 % 8 == 0).
    const bool affine = in_scale != nullptr;
    float4 isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = make_float4(0.f, 0.f, 0.f, 0.f);
    if (affine) {
        isc = reinterpret_cast<const float4 *>(in_scale)[threadIdx.x & 7];
        ish = reinterpret_cast<const float4 *>(in_shift)[threadIdx.x & 7];
    }
    unsigned okbits = 0;
    auto issue = [&](int id) {
        okbits = 0;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int q = threadIdx.x + 256 * it, v = q >> 3, part = q & 7;
            const int sy = v / SX, sx = v - sy * SX;
            const int ih = ty0 - 1 + sy, iw = tx0 - 1 + sx;
            pre[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < NQ && id >= 0 && id < D && ih >= 0 && ih < H && iw >= 0 && iw < W) {
                pre[it] = *reinterpret_cast<const float4 *>(
                    in + ((((size_t)b * D + id) * H + ih) * W + iw) * 32 + part * 4);
                okbits |= 1u << it;
            }
        }
    };
    issue(d0 - 1);
    for (int id = d0 - 1; id <= d1; ++id) {
        if (id >= 0 && id < D) {  // block-uniform; planes outside the volume are zero padding
            __syncthreads();
#pragma unroll
            for (int it = 0; it < NLD; ++it) {
                const int q = threadIdx.x + 256 * it;
                float4 v = pre[it];
                if (affine && ((okbits >> it) & 1u)) {  // (fmaf + max exactly as az_bn3d_apply rounds)
                    v.x = fmaxf(fmaf(v.x, isc.x, ish.x), 0.f); v.y = fmaxf(fmaf(v.y, isc.y, ish.y), 0.f);
                    v.z = fmaxf(fmaf(v.z, isc.z, ish.z), 0.f); v.w = fmaxf(fmaf(v.w, isc.w, ish.w), 0.f);
                }
                if (q < NQ) *reinterpret_cast<float4 *>(&slab[(q >> 3) * C1_VS + (q & 7) * 4]) = v;
            }
            __syncthreads();
        }
        if (id < d1) issue(id + 1);
        if (id >= 0 && id < D) {
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                c1_f2 r[10];
                const float *row = &slab[((ly + kh) * SX + lx0) * C1_VS + cp * 2];
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const float2 x = *reinterpret_cast<const float2 *>(row + i * C1_VS);
                    r[i] = c1_f2{x.x, x.y};
                }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        a_next[j] = C1_FMA2(r[j + kw], wp[kh * 3 + kw], a_next[j]);       // kd = 0 -> plane id+1
                        a_cur[j] = C1_FMA2(r[j + kw], wp[9 + kh * 3 + kw], a_cur[j]);     // kd = 1 -> id
                        a_prev[j] = C1_FMA2(r[j + kw], wp[18 + kh * 3 + kw], a_prev[j]);  // kd = 2 -> id-1
                    }
            }
        }
        const int od = id - 1;  // complete now: it has seen planes od-1, od, od+1
        if (od >= d0 && od < d1) {  // block-uniform
            // sum the 16 channel pairs of each voxel: xor 1, xor 2 inside quads, then the two row mirrors
            float mine = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = a_prev[j][0] + a_prev[j][1];
                v = c1_dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
                v = c1_dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
                v = c1_dpp_add<0x141>(v);  // row_half_mirror
                v = c1_dpp_add<0x140>(v);  // row_mirror
                mine = (cp == j) ? v : mine;
            }
            const int ow = tx0 + lx0 + cp;
            if (cp < 8 && oh < H && ow < W) {
                const size_t o = (((size_t)b * D + od) * H + oh) * W + ow;
                out[o] = addend ? mine + addend[o] : mine;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { a_prev[j] = a_cur[j]; a_cur[j] = a_next[j]; a_next[j] = c1_f2{0.f, 0.f}; }
    }
}

// shared by dgrad / wgrad: stage gout[b][od-1..od+1][ty0-1..][tx0-1..] (zero padded)
__device__ __forceinline__ void c1_load_gtile(float *gt, const float *__restrict__ gout, int b,
                                              int od, int ty0, int tx0, int D, int H, int W) {
    constexpr int SX = C1_TW + 2, SY = C1_TH + 2;
    for (int q = threadIdx.x; q < 3 * SY * SX; q += 256) {
        const int sx = q % SX;
        const int r = q / SX;
        const int sy = r % SY, sd = r / SY;
        const int gd = od - 1 + sd, gh = ty0 - 1 + sy, gw = tx0 - 1 + sx;
        gt[r * C1_GP + sx] = (gd >= 0 && gd < D && gh >= 0 && gh < H && gw >= 0 && gw < W)
                                 ? gout[(((size_t)b * D + gd) * H + gh) * W + gw] : 0.f;
    }
}
// the 6 tile values under 4 adjacent voxels (lx0 .. lx0+3, lx0 % 4 == 0) of row (pd, sy): r[j + 2 - kw] is
// gout at voxel j for tap kw
__device__ __forceinline__ void c1_row6(float (&r)[6], const float *gt, int pd, int sy, int lx0) {
    const float *row = static_cast<const float *>(__builtin_assume_aligned(gt + (pd * (C1_TH + 2) + sy) * C1_GP + lx0, 16));
    const float4 a = *reinterpret_cast<const float4 *>(row);
    const float2 c = *reinterpret_cast<const float2 *>(row + 4);
    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = c.x; r[5] = c.y;
}

__global__ void __launch_bounds__(256)
c1_dgrad_kernel(float *__restrict__ gin, const float *__restrict__ gout,
                const float *__restrict__ w, int D, int H, int W, int tiles_x) {
    constexpr int SY = C1_TH + 2;
    __shared__ __attribute__((aligned(16))) float gt[3 * SY * C1_GP];
    const int tx0 = (blockIdx.x % tiles_x) * C1_TW, ty0 = (blockIdx.x / tiles_x) * C1_TH;
    const int od = blockIdx.y, b = blockIdx.z;
    c1_load_gtile(gt, gout, b, od, ty0, tx0, D, H, W);
    // a lane owns one channel PAIR (54 weight registers) of EIGHT voxels adjacent in x; 16 lanes write a
    // voxel's 128 B together
    const int cp = threadIdx.x & 15, grp = threadIdx.x >> 4;  // 16 groups of 8 voxels per pass
    c1_f2 wp[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wp[t] = c1_f2{w[(cp * 2 + 0) * 27 + t], w[(cp * 2 + 1) * 27 + t]};
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < C1_TH * C1_TW / 128; ++p) {  // 4 tile rows per pass
        const int gi = p * 16 + grp, ly = gi >> 2, lx0 = (gi & 3) * 8;
        const int oh = ty0 + ly;
        c1_f2 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = c1_f2{0.f, 0.f};
#pragma unroll
        for (int kd = 0; kd < 3; ++kd)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                // gin[v] += gout[v + 1 - k] * w[k]; tile-local index of v+1-k is (+1) shifted
                float r[12];
                {
                    const float *row = static_cast<const float *>(__builtin_assume_aligned(
                        gt + ((2 - kd) * SY + ly + 2 - kh) * C1_GP + lx0, 16));
                    const float4 a = *reinterpret_cast<const float4 *>(row);
                    const float4 c = *reinterpret_cast<const float4 *>(row + 4);
                    const float2 e = *reinterpret_cast<const float2 *>(row + 8);
                    r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = c.x; r[5] = c.y; r[6] = c.z; r[7] = c.w;
                    r[8] = e.x; r[9] = e.y;
                }
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int t = (kd * 3 + kh) * 3 + kw;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const c1_f2 g2 = {r[j + 2 - kw], r[j + 2 - kw]};
                        acc[j] = C1_FMA2(g2, wp[t], acc[j]);
                    }
                }
            }
        // (without this the compiler sinks each voxel's 27 FMAs into its own bounds-checked store below and keeps
        //  all nine tile rows live: 184 registers, 114 v_mov per pass)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(acc[j]));
        if (oh < H) {
            float *dst = gin + ((((size_t)b * D + od) * H + oh) * W + tx0 + lx0) * 32 + cp * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (tx0 + lx0 + j < W) *reinterpret_cast<float2 *>(dst + j * 32) = make_float2(acc[j][0], acc[j][1]);
        }
    }
}

__global__ void __launch_bounds__(256)
c1_wgrad_kernel(float *__restrict__ gw, const float *__restrict__ in,
                const float *__restrict__ gout, const float *__restrict__ in_scale,
                const float *__restrict__ in_shift, int D, int H, int W, int tiles_x, int tiles_y,
                long long ntiles) {
    constexpr int SY = C1_TH + 2;
    __shared__ __attribute__((aligned(16))) float gt[3 * SY * C1_GP];
    __shared__ float red[864];
    for (int q = threadIdx.x; q < 864; q += 256) red[q] = 0.f;
    const int quad = threadIdx.x & 7, grp = threadIdx.x >> 3;
    const bool affine = in_scale != nullptr;
    float4 isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = make_float4(0.f, 0.f, 0.f, 0.f);
    if (affine) {
        isc = reinterpret_cast<const float4 *>(in_scale)[quad];
        ish = reinterpret_cast<const float4 *>(in_shift)[quad];
    }
    c1_f2 a01[27], a23[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) { a01[t] = c1_f2{0.f, 0.f}; a23[t] = c1_f2{0.f, 0.f}; }
    // a block walks a strided list of (b, d, tile) items and keeps its 108 partial sums per
    // lane in registers, so the 864 global atomics are paid once per block, not once per tile
    for (long long item = blockIdx.x; item < ntiles; item += gridDim.x) {
        long long r = item;
        const int tx0 = (int)(r % tiles_x) * C1_TW; r /= tiles_x;
        const int ty0 = (int)(r % tiles_y) * C1_TH; r /= tiles_y;
        const int od = (int)(r % D);
        const int b = (int)(r / D);
        __syncthreads();
        c1_load_gtile(gt, gout, b, od, ty0, tx0, D, H, W);
        __syncthreads();
#pragma unroll 1
        for (int p = 0; p < C1_TH * C1_TW / 128; ++p) {
            const int gi = p * 32 + grp, ly = gi >> 3, lx0 = (gi & 7) * 4;
            const int ih = ty0 + ly;
            c1_f2 x01[4], x23[4];
            const float *src = in + ((((size_t)b * D + od) * H + ih) * W + tx0 + lx0) * 32 + quad * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ih < H && tx0 + lx0 + j < W) {
                    x = *reinterpret_cast<const float4 *>(src + j * 32);
                    if (affine) {  // the operand is relu(in * scale + shift), see c1_fwd_kernel
                        x.x = fmaxf(fmaf(x.x, isc.x, ish.x), 0.f); x.y = fmaxf(fmaf(x.y, isc.y, ish.y), 0.f);
                        x.z = fmaxf(fmaf(x.z, isc.z, ish.z), 0.f); x.w = fmaxf(fmaf(x.w, isc.w, ish.w), 0.f);
                    }
                }
                x01[j] = c1_f2{x.x, x.y}; x23[j] = c1_f2{x.z, x.w};
            }
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    // out[o] = sum_k in[o - 1 + k] w[k]  =>  gw[k] += in[v] * gout[v + 1 - k]
                    float r[6];
                    c1_row6(r, gt, 2 - kd, ly + 2 - kh, lx0);
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const int t = (kd * 3 + kh) * 3 + kw;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const c1_f2 g2 = {r[j + 2 - kw], r[j + 2 - kw]};
                            // (the broadcast factor FIRST: hipcc keeps the operand order, and v_pk_fma_f32 with the
                            //  high register of a pair selected for src1 -- op_sel:[0,1,0], what an odd-numbered g
                            //  register becomes in second place -- returns wrong values in lanes 48..63 while a wave of
                            //  another kernel issues MFMAs on the same SIMD: tools/probes/pkfma_corun.hip,
                            //  profiles/r03_pkfma_corun.md; tests/test_isa_lint_cpu.py keeps the form out of the library)
#ifdef C1_WGRAD_OLD_OPERAND_ORDER  // (the form that fails: only to show that the tests see it -- tools/build_variant.sh)
                            a01[t] = C1_FMA2(x01[j], g2, a01[t]);
                            a23[t] = C1_FMA2(x23[j], g2, a23[t]);
#else
                            a01[t] = C1_FMA2(g2, x01[j], a01[t]);
                            a23[t] = C1_FMA2(g2, x23[j], a23[t]);
#endif
                        }
                    }
                }
        }
    }
    // reduce the 32 voxel-lanes that share a channel quad: lanes with equal (lane & 7);
    // xor-shuffle over lane bits 3..5 inside the wave, then LDS atomics across the 4 waves.
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            float v = (c < 2 ? a01[t] : a23[t])[c & 1];
            v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            if ((threadIdx.x & 63) < 8) atomicAdd(&red[(quad * 4 + c) * 27 + t], v);
        }
    __syncthreads();
    for (int q = threadIdx.x; q < 864; q += 256) atomicAdd(&gw[q], red[q]);
}

static int c1_check(int B, int D, int H, int W) {
    if (!(B > 0 && D > 0 && H > 0 && W > 0)) return AZ_EINVAL;
    if (B > 65535 || D > 65535) return AZ_EUNSUPPORTED;
    return AZ_OK;
}

extern "C" int az_conv3d_c1_fwd(float *logits, const float *in, const float *w,
                                const float *addend, const float *in_scale, const float *in_shift, int B, int D,
                                int H, int W, void *stream) {
    AZ_REQUIRE_PTR(logits); AZ_REQUIRE_PTR(in); AZ_REQUIRE_PTR(w);
    if ((in_scale == nullptr) != (in_shift == nullptr)) return AZ_EINVAL;
    if (int e = c1_check(B, D, H, W)) return e;
    const int tiles_x = (W + C1F_TW - 1) / C1F_TW, tiles_y = (H + C1F_TH - 1) / C1F_TH;
    // depth segment per block: every segment stages 2 halo planes, and the grid should fill a whole
    // number of residency rounds (3 blocks per CU x 256 CUs by registers); minimise rounds x planes
    int best = 1;
    long long best_cost = -1;
    for (int nseg = 1; nseg <= D; ++nseg) {
        const int dseg = (D + nseg - 1) / nseg;
        const long long blocks = (long long)tiles_x * tiles_y * B * ((D + dseg - 1) / dseg);
        const long long cost = ((blocks + 767) / 768) * (dseg + 2);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = dseg; }
    }
    hipLaunchKernelGGL(c1_fwd_kernel, dim3(tiles_x * tiles_y, (D + best - 1) / best, B), dim3(256), 0,
                       az_stream(stream), logits, in, w, addend, in_scale, in_shift, D, H, W, tiles_x, best);
    return az_launch_status();
}

extern "C" int az_conv3d_c1_dgrad(float *grad_in, const float *grad_logits, const float *w,
                                  int B, int D, int H, int W, void *stream) {
    AZ_REQUIRE_PTR(grad_in); AZ_REQUIRE_PTR(grad_logits); AZ_REQUIRE_PTR(w);
    if (int e = c1_check(B, D, H, W)) return e;
    const int tiles_x = (W + C1_TW - 1) / C1_TW, tiles_y = (H + C1_TH - 1) / C1_TH;
    hipLaunchKernelGGL(c1_dgrad_kernel, dim3(tiles_x * tiles_y, D, B), dim3(256), 0,
                       az_stream(stream), grad_in, grad_logits, w, D, H, W, tiles_x);
    return az_launch_status();
}

extern "C" int az_conv3d_c1_wgrad(float *grad_w, const float *in, const float *grad_logits,
                                  const float *in_scale, const float *in_shift, int B, int D, int H, int W,
                                  void *stream) {
    AZ_REQUIRE_PTR(grad_w); AZ_REQUIRE_PTR(in); AZ_REQUIRE_PTR(grad_logits);
    if ((in_scale == nullptr) != (in_shift == nullptr)) return AZ_EINVAL;
    if (int e = c1_check(B, D, H, W)) return e;
    if (hipMemsetAsync(grad_w, 0, 864 * sizeof(float), az_stream(stream)) != hipSuccess)
        return AZ_ELAUNCH;
    const int tiles_x = (W + C1_TW - 1) / C1_TW, tiles_y = (H + C1_TH - 1) / C1_TH;
    const long long ntiles = (long long)B * D * tiles_y * tiles_x;
    const unsigned grid = (unsigned)(ntiles < 2048 ? ntiles : 2048);
    hipLaunchKernelGGL(c1_wgrad_kernel, dim3(grid), dim3(256), 0, az_stream(stream), grad_w, in,
                       grad_logits, in_scale, in_shift, D, H, W, tiles_x, tiles_y, ntiles);
    return az_launch_status();
}
