// Shared by the depth-rolling convolution kernels (az_conv3d_roll.hip, az_conv2d_roll.hip): slab geometry, the
// bf16x6 step on v_mfma_f32_16x16x32_bf16 and the quad transpose of its C layout.
#pragma once
#include "az_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define R_TY 8
#define R_TX 16
#define R_SY 10
#define R_SX 18
#define R_VB 192                          // bytes per slab voxel
#define R_SLAB_BYTES (R_SY * R_SX * R_VB)  // 34 560
#define R_NQ (R_SY * R_SX * 8)             // 16-byte fp32 pieces of one plane chunk (8 per voxel)
#define R_NLD ((R_NQ + 255) / 256)         // 6 per thread
#ifndef R16_NOPART
#define R16_NOPART 0  // timing-only build: BatchNorm partials not written
#endif
#define R_OOB 0xffffff00u                  // a buffer offset beyond every tensor: loads return 0, stores are dropped

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

// ---- "f16x3": the backward kernels' arithmetic (round 4) -----------------------------------------------------------
// Both operands are scaled by a power of two taken from the tensor's largest magnitude (so that the largest element
// lands in [2^14, 2^15): az_f16_scale_exp) and split into TWO fp16 parts, hi = fp16(x), lo = fp16(x - hi), both
// round-to-nearest: x = hi + lo up to 2^-22 |x| (11 + 11 significand bits; elements more than 2^17 below the
// tensor's largest lose low bits of `lo` to the fp16 subnormal spacing: an ABSOLUTE error of 2^-39 of the largest
// element).  Three products -- hi*hi, hi*lo, lo*hi, largest first -- are summed from zero by
// v_mfma_f32_16x16x32_f16 and the block sum is added to the fp32 accumulator by VALU adds, as in r16_step; the
// dropped lo*lo term is <= 2^-22 |x y|.  Half the matrix instructions of bf16x6 for a per-product error of about
// 2^-22 instead of 2^-24.  Since round 4 the default arithmetic of every matrix kernel of the path, forward passes
// included (tools/f16x3_probe.py has the per-layer errors against fp64 next to torch's fp32
// convolution); the inference-time 2-D extractor alone stays on bf16x6.  The guaranteed bound and what a loose or a
// stale amax costs: include/azhip.h ("CONTRACT of a caller-supplied amax"), tests/test_gpu_f16x3_contract.py.
typedef _Float16 az_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 az_f16x2 __attribute__((ext_vector_type(2)));
// k with 2^k * amax in [2^14, 2^15) (clamped so that 2^k is a normal float; amax = 0, inf or nan: any k works or
// nothing does)
__host__ __device__ __forceinline__ int az_f16_scale_exp(float amax) {
    const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127;
    const int k = 14 - e;
    return k < -126 ? -126 : (k > 127 ? 127 : k);
}
__host__ __device__ __forceinline__ float az_pow2(int k) { return __builtin_bit_cast(float, (unsigned)(k + 127) << 23); }

__device__ __forceinline__ void az_split2_f16_pair(float x0, float x1, unsigned &hi, unsigned &lo) {
    az_f16x2 h, l;
    h[0] = (_Float16)x0; h[1] = (_Float16)x1;
    l[0] = (_Float16)(x0 - (float)h[0]); l[1] = (_Float16)(x1 - (float)h[1]);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}
// two values -> one packed fp16 pair, round-to-nearest (the one-part "f16x1" operands of the RAFT-Stereo GRU convolutions)
__device__ __forceinline__ unsigned az_pk_f16(float a, float b) {
    az_f16x2 p;
    p[0] = (_Float16)a; p[1] = (_Float16)b;
    return __builtin_bit_cast(unsigned, p);
}
// four values (already scaled) -> two 8-byte pieces
__device__ __forceinline__ void az_split2_f16x4(const float4 &v, uint2 &hi, uint2 &lo) {
    az_split2_f16_pair(v.x, v.y, hi.x, lo.x);
    az_split2_f16_pair(v.z, v.w, hi.y, lo.y);
}
// ---- "pre-split" operands (round 5; include/azhip.h, "S2 format") -------------------------------------------------------
// A tensor whose only readers are f16x3 matrix kernels can be STORED the way those kernels stage it: every aligned group
// of four channels (16 bytes) holds  hi(c0) hi(c1) | hi(c2) hi(c3) | lo(c0) lo(c1) | lo(c2) lo(c3)  -- the two fp16 parts
// of x * 2^k, k = az_f16_scale_exp(amax) with the amax array the producer wrote next to it -- in place of the four floats.
// Same bytes in HBM; the consumer's staging is then a copy (two 8-byte LDS writes per piece) instead of four multiplies,
// six conversions and four subtractions, and the bits it multiplies are the ones it would have computed itself from the
// fp32 tensor with that amax.  The producer is the BatchNorm-backward apply pass (az_bn3d.hip), whose amax is an upper
// BOUND of max |dx| known before the first element is written.
// az_stage_f16x4<PS>: one staged 16-byte piece -> its (hi, lo) 8-byte halves; PS = the tensor is pre-split.
template <bool PS>
__device__ __forceinline__ void az_stage_f16x4(const u32x4 &raw, float scale, uint2 &hi, uint2 &lo) {
    if (PS) {
        hi = uint2{raw[0], raw[1]};
        lo = uint2{raw[2], raw[3]};
    } else {
        float4 v = __builtin_bit_cast(float4, raw);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        az_split2_f16x4(v, hi, lo);
    }
}
// the producer's side: four values (already scaled) -> the 16 bytes above
__device__ __forceinline__ float4 az_presplit_f16x4(const float4 &v) {
    uint2 hi, lo;
    az_split2_f16x4(v, hi, lo);
    return __builtin_bit_cast(float4, u32x4{hi.x, hi.y, lo.x, lo.y});
}
__device__ __forceinline__ unsigned short az_split2_f16_part(float x, int p) {
    const _Float16 h = (_Float16)x;
    const _Float16 l = (_Float16)(x - (float)h);
    return __builtin_bit_cast(unsigned short, p == 0 ? h : l);
}

// f16x3 on the 32x32x16 shape (az_conv2d.hip, az_conv3d.hip): tnew = one K16 block (hi*hi, hi*lo, lo*hi from zero);
// cprev += tprev with the sixteen adds spread behind the second and third MFMA (the temporary of the block before is
// complete by then: no wait states)
typedef float az_f32x16h __attribute__((ext_vector_type(16)));
#define AZ_H3(ACC, A, B) __builtin_amdgcn_mfma_f32_32x32x16_f16( \
        __builtin_bit_cast(az_f16x8, aq[A]), __builtin_bit_cast(az_f16x8, bq[B]), ACC, 0, 0, 0)
__device__ __forceinline__ void az_mfma3_step(az_f32x16h &tnew, const float4 (&aq)[3], const float4 (&bq)[3],
                                              az_f32x16h &cprev, const az_f32x16h &tprev) {
    az_f32x16h t;
#pragma unroll
    for (int e = 0; e < 16; ++e) t[e] = 0.f;
    t = AZ_H3(t, 0, 0);
    t = AZ_H3(t, 0, 1);
#pragma unroll
    for (int e = 0; e < 8; ++e) cprev[e] += tprev[e];
    t = AZ_H3(t, 1, 0);
#pragma unroll
    for (int e = 8; e < 16; ++e) cprev[e] += tprev[e];
    tnew = t;
    asm volatile("" : "+v"(cprev));
    __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 8, 0);
}
__device__ __forceinline__ void az_mfma3_now(az_f32x16h &c, const float4 (&aq)[3], const float4 (&bq)[3]) {
    az_f32x16h t;
#pragma unroll
    for (int e = 0; e < 16; ++e) t[e] = 0.f;
    t = AZ_H3(t, 0, 0); t = AZ_H3(t, 0, 1); t = AZ_H3(t, 1, 0);
    c += t;
    asm volatile("" : "+v"(c));
}

// f16x3 form of r16_step (one K32 block: hi*hi, hi*lo, lo*hi from zero; unswapped operand roles, C layout as
// r16_step): the four adds of the temporary before sit behind the second and third MFMA
#define R_MH3(ACC, A, B) __builtin_amdgcn_mfma_f32_16x16x32_f16( \
        __builtin_bit_cast(az_f16x8, aq[A]), __builtin_bit_cast(az_f16x8, bq[B]), ACC, 0, 0, 0)
__device__ __forceinline__ void r16_step3(f32x4 &tnew, const float4 (&aq)[3], const float4 (&bq)[3], f32x4 &cprev,
                                          const f32x4 &tprev) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    float c0 = cprev[0], c1 = cprev[1], c2 = cprev[2], c3 = cprev[3];
    t = R_MH3(t, 0, 0);
    t = R_MH3(t, 0, 1);
    c0 += tprev[0];
    asm volatile("" : "+v"(c0));
    c1 += tprev[1];
    asm volatile("" : "+v"(c1));
    t = R_MH3(t, 1, 0);
    c2 += tprev[2];
    asm volatile("" : "+v"(c2));
    c3 += tprev[3];
    asm volatile("" : "+v"(c3));
    tnew = t;
    cprev[0] = c0; cprev[1] = c1; cprev[2] = c2; cprev[3] = c3;
    __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
}

// One f16x3 chain: the three kw taps of a (kd, kh) row for one 4x4-voxel tile, K = 3 x 32, NINE MFMAs summed from zero
// (the three hi*hi products first, then the six cross terms), while the temporary of the chain before is added to its
// accumulator (four VALU adds, placed behind the second and third MFMA: the previous chain's last result is then
// complete and no wait states are inserted).  One accumulator rounding per 96-deep block.
// OPERAND ROLES ARE SWAPPED against r16_step: the WEIGHT fragment is the A operand (rows = 16 output channels), the
// voxel fragment the B operand (columns = 16 voxels), so that the result holds, per lane, FOUR CONSECUTIVE CHANNELS
// (4 (lane >> 4) + r) of ONE voxel (lane & 15): a 16-byte store without the quad transpose (both fragments have
// the same register layout -- index lane & 15, k = 8 (lane >> 4) + j -- so the swap costs nothing).
#define R_MH(ACC, W, X) __builtin_amdgcn_mfma_f32_16x16x32_f16( \
        __builtin_bit_cast(az_f16x8, W), __builtin_bit_cast(az_f16x8, X), ACC, 0, 0, 0)
__device__ __forceinline__ void r16_chain9(f32x4 &tnew, const float4 (&x)[3][2], const float4 (&w)[3][2], f32x4 &cprev,
                                           const f32x4 &tprev) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    float c0 = cprev[0], c1 = cprev[1], c2 = cprev[2], c3 = cprev[3];
    t = R_MH(t, w[0][0], x[0][0]);
    t = R_MH(t, w[1][0], x[1][0]);
    c0 += tprev[0];
    asm volatile("" : "+v"(c0));
    c1 += tprev[1];
    asm volatile("" : "+v"(c1));
    t = R_MH(t, w[2][0], x[2][0]);
    c2 += tprev[2];
    asm volatile("" : "+v"(c2));
    c3 += tprev[3];
    asm volatile("" : "+v"(c3));
    t = R_MH(t, w[0][0], x[0][1]);
    t = R_MH(t, w[0][1], x[0][0]);
    t = R_MH(t, w[1][0], x[1][1]);
    t = R_MH(t, w[1][1], x[1][0]);
    t = R_MH(t, w[2][0], x[2][1]);
    t = R_MH(t, w[2][1], x[2][0]);
    tnew = t;
    cprev[0] = c0; cprev[1] = c1; cprev[2] = c2; cprev[3] = c3;
    __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 6, 0);
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
    const float s01 = r16_dpp<0xB1>(o1 ? v[0] : v[1]);  // quad_perm [1,0,3,2]
    const float s23 = r16_dpp<0xB1>(o1 ? v[2] : v[3]);
    const float a0 = o1 ? s01 : v[0], a1 = o1 ? v[1] : s01, a2 = o1 ? s23 : v[2], a3 = o1 ? v[3] : s23;
    const float t0 = r16_dpp<0x4E>(o2 ? a0 : a2);       // quad_perm [2,3,0,1]
    const float t1 = r16_dpp<0x4E>(o2 ? a1 : a3);
    return f32x4{o2 ? t0 : a0, o2 ? t1 : a1, o2 ? a2 : t0, o2 ? a3 : t1};
}

