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

