// Shared helpers for the gfx950 kernels behind include/azhip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/azhip.h"

#define AZ_WAVE 64

#define AZ_REQUIRE_PTR(p) \
    do {                  \
        if ((p) == nullptr) return AZ_ENULL; \
    } while (0)
#define AZ_REQUIRE(cond)               \
    do {                               \
        if (!(cond)) return AZ_EINVAL; \
    } while (0)

static inline int az_launch_status() {
    return hipGetLastError() == hipSuccess ? AZ_OK : AZ_ELAUNCH;
}

static inline hipStream_t az_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// grid size for an element-wise / grid-stride launch: enough blocks to fill
// 256 CUs x 8 blocks, never more than the work needs.
static inline unsigned az_grid_for(long long work_items, int block) {
    long long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256LL * 16) g = 256LL * 16;
    return (unsigned)g;
}

// Branch-free guarded 16-byte load: the address is forced to `safe` (any valid 16-byte
// location) and the value to zero when !ok.  Written with scalar selects on purpose --
// a float4 ?: makes hipcc spill through scratch memory and serialise the loads.
__device__ __forceinline__ float4 az_ld16_or_zero(const float *base, size_t offset, bool ok) {
    const float4 t = *reinterpret_cast<const float4 *>(base + (ok ? offset : (size_t)0));
    float4 r;
    r.x = ok ? t.x : 0.f;
    r.y = ok ? t.y : 0.f;
    r.z = ok ? t.z : 0.f;
    r.w = ok ? t.w : 0.f;
    return r;
}

// ---- max |x| of a tensor, taken by the kernel that writes it (the operand scale of the f16x3 kernels that read it
// next; az_absmax.hip is the stand-alone pass).  |x| is compared as a bit pattern: non-negative floats order like
// unsigned integers; inf / NaN elements are left out (az_finite_abs_bits).
// The maximum lives in AZ_AMAX_SLOTS words AZ_AMAX_STRIDE floats (256 B) apart -- an "amax array" is
// AZ_AMAX_FLOATS = 1024 floats, all ZERO before the producing launch; a workgroup adds its maximum to the slot of its
// block index with ONE atomic, readers take the largest slot.  Measured on a BatchNorm apply of a 67 MB tensor (20 us
// alone; tools/amax_cost_probe.py): one atomic per WAVE into one word or into 64 adjacent words +73 us (16 384 atomics
// drain through one memory channel at ~4 ns each, whatever the word), with every slot already at the maximum -- no atomic
// issued, see below -- +0.1 us.  An atomic is issued only when the block's maximum beats what its slot holds; that read
// bypasses L1 and may be stale (another XCD's L2), which only costs an atomic that changes nothing.
// |x| as a bit pattern, ZERO for inf / NaN: an amax is the largest FINITE magnitude of its tensor.  (With a non-finite
// amax the power-of-two scale would flush every finite element of the tensor to zero -- one bad voxel would wipe out all
// outputs; this way the bad element alone becomes inf / NaN in fp16 and spoils exactly the outputs that read it, as it
// does in an fp32 convolution.)
__device__ __forceinline__ unsigned az_finite_abs_bits(float x) {
    const unsigned b = __float_as_uint(x) & 0x7fffffffu;
    return b < 0x7f800000u ? b : 0u;
}
__device__ __forceinline__ void az_amax_acc(unsigned &am, const float4 &o) {
    am = max(max(am, az_finite_abs_bits(o.x)), max(az_finite_abs_bits(o.y), max(az_finite_abs_bits(o.z), az_finite_abs_bits(o.w))));
}
// called by EVERY thread of the workgroup (a barrier inside); blockDim.x <= 1024
__device__ __forceinline__ void az_amax_flush(unsigned *dst, unsigned am) {
    __shared__ unsigned az_amax_wave[16];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) am = max(am, (unsigned)__shfl_xor((int)am, off));
    if ((threadIdx.x & 63) == 0) az_amax_wave[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; ++w) am = max(am, az_amax_wave[w]);
        unsigned *slot = dst + ((blockIdx.x + blockIdx.y * 5u) & (AZ_AMAX_SLOTS - 1)) * AZ_AMAX_STRIDE;
        if (am > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, am);
    }
}
// (MODE.FP16_OVFL -- fp32 -> fp16 conversions that saturate at +-65504 instead of overflowing to inf -- is NOT used to
// soften a too-small amax: with the bit set v_mfma_f32_16x16x32_f16 itself clamps an inf operand's products to
// +-FLT_MAX and drops a NaN operand silently (tools/probes/fp16_ovfl_probe.hip, profiles/r05_fp16_ovfl_probe.txt), and
// a wave cannot keep the bit on for its conversions and off for its MFMAs without pinning their order.  A stale amax
// therefore overflows loudly -- inf / NaN in the outputs that read the element; AZ_DEBUG_AMAX=1 names the tensor.)
// the tensor's max |x| from its amax array; wave-uniform (call with all 64 lanes active)
__device__ __forceinline__ float az_amax_read(const float *p) {
    unsigned v = reinterpret_cast<const unsigned *>(p)[(threadIdx.x & (AZ_AMAX_SLOTS - 1)) * AZ_AMAX_STRIDE];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, off));
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((int)v));
}

// Exact 3-way split of fp32 values into bf16 parts, ROUND-TO-NEAREST-EVEN at every level:
//   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid);   x == hi + mid + lo exactly
// (8 + 8 + 8 signed significand bits cover fp32's 24).  The three partial products the bf16x6 arithmetic
// drops (mid*lo', lo*mid', lo*lo') are then <= 2^-26 |x y| and of either sign.  A TRUNCATING split
// (mask the low 16 bits) costs the same instructions but leaves residuals that all carry the sign of x:
// the dropped terms become a one-sided 2^-24 |x y| bias, which random-walk tolerances hide but anything that
// averages many outputs (the 64x64 SPP pooling in front of a high-gain BatchNorm, measured: the extractor's
// output error was 10x the fp32 reference's although every single layer matched it) turns into the
// dominant error.  v_cvt_pk_bf16_f32 converts two values per instruction.
typedef __bf16 az_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned az_pk_bf16(float a, float b) {
    az_bf16x2 p;
    p[0] = (__bf16)a;  // -O3: one v_cvt_pk_bf16_f32 (RNE) for the pair
    p[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ void az_split3_pair(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
    hi = az_pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
    mid = az_pk_bf16(r0, r1);
    const float q0 = r0 - __uint_as_float(mid << 16), q1 = r1 - __uint_as_float(mid & 0xffff0000u);
    lo = az_pk_bf16(q0, q1);
}
// four values -> three 8-byte pieces (4 x bf16 each)
__device__ __forceinline__ void az_split3_bf16x4(const float4 &v, uint2 &hi, uint2 &mid, uint2 &lo) {
    az_split3_pair(v.x, v.y, hi.x, mid.x, lo.x);
    az_split3_pair(v.z, v.w, hi.y, mid.y, lo.y);
}
// one value, host-or-device scalar form for the weight packers: part p of x as a bf16 bit pattern
__device__ __forceinline__ unsigned short az_split3_part(float x, int p) {
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    return __builtin_bit_cast(unsigned short, p == 0 ? h : p == 1 ? m : l);
}

// ---- one 32x32x16 block of the bf16x6 arithmetic ------------------------------------------------------
// fp32-class product on the bf16 pipe: (ah+am+al)(bh+bm+bl) without the three terms below 2^-26, six
// v_mfma_f32_32x32x16_bf16.  aq/bq: the hi, mid, lo parts' 8-element fragments.  Two measured properties of
// the instruction shape how the six are combined (tools/parity_probe*.py, tools/rz_probe.py):
//
// (1) An MFMA rounds its fp32 accumulator after every instruction, at the accumulator's magnitude.  Chaining
//     all six into the RUNNING accumulator costs six such roundings per 16-deep K block, five of which only
//     add terms <= 2^-8 of the block's value: the extractor's layers then sat at 2.7x (K = 1152) to 3.6x
//     (K = 2880) the error of torch's own fp32 convolution.  So the six partial products of a block are summed
//     in a ZERO-initialised temporary (the MFMA takes the inline constant 0 as C) and the accumulator receives
//     the finished temporary with plain VALU adds: one rounding of the accumulator per K block.
// (2) When the addend C is much smaller than the products (or the reverse) the MFMA aligns the small operand
//     to the large one and FLOORS the bits it shifts out -- towards minus infinity, for either sign.  With the
//     customary "smallest terms first" order the last instruction adds the hi*hi products to a C that holds
//     the 2^-8 / 2^-16 terms: every block result came out low by ~2^-28 of sum|a b| (signed mean relative error
//     -1.2e-7 on mixed-sign data, 8 times the -1.5e-8 of the order below; hi*hi alone from zero: -1.4e-10).  A bias
//     of that size is invisible per layer, but it survives averaging: through the 64x64 SPP pooling in front of a
//     high-gain BatchNorm it made the extractor's output error 5x the fp32 reference's and the D = 192 disparity
//     error 4x.  LARGEST TERMS FIRST costs nothing and removes it: hi*hi starts from zero, each later group is
//     added to a C at least as large as its products, whose own low bits stay inside the adder.
//
// (Measured, profiles/r02_clock_pipe_ablation.md: against the plain chain the adds are free in the multi-wave 2-D
// kernels and cost 10 % in the one-wave V0 kernel; issued as v_pk_add_f32 they are no faster -- hipcc itself
// unpacks packed adds that sit in an MFMA's shadow.)
// az_mfma6_step overlaps the adds with the next block's MFMAs (two temporaries in flight = 32 registers);
// az_mfma6_now is the single-temporary form for kernels at their register limit; az_mfma6 is the plain chain
// into a running accumulator (weight-gradient kernels).
typedef float az_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 az_bf16x8 __attribute__((ext_vector_type(8)));
#define AZ_X6(ACC, A, B) __builtin_amdgcn_mfma_f32_32x32x16_bf16( \
        __builtin_bit_cast(az_bf16x8, aq[A]), __builtin_bit_cast(az_bf16x8, bq[B]), ACC, 0, 0, 0)
__device__ __forceinline__ void az_mfma6(az_f32x16 &c, const float4 (&aq)[3], const float4 (&bq)[3]) {
    c = AZ_X6(c, 0, 0); c = AZ_X6(c, 0, 1); c = AZ_X6(c, 1, 0); c = AZ_X6(c, 1, 1); c = AZ_X6(c, 0, 2);
    c = AZ_X6(c, 2, 0);
}
// c += (block product summed from zero).  The adds wait for the block's last MFMA; the SIMD's other wave
// covers the gap.
__device__ __forceinline__ void az_mfma6_now(az_f32x16 &c, const float4 (&aq)[3], const float4 (&bq)[3]) {
    az_f32x16 t;
#pragma unroll
    for (int e = 0; e < 16; ++e) t[e] = 0.f;
    t = AZ_X6(t, 0, 0); t = AZ_X6(t, 0, 1); t = AZ_X6(t, 1, 0); t = AZ_X6(t, 1, 1); t = AZ_X6(t, 0, 2);
    t = AZ_X6(t, 2, 0);
    c += t;
    asm volatile("" : "+v"(c));  // keeps the adds here (see az_mfma6_step)
}
// tnew = block product (from zero);  cprev += tprev  (tprev: the temporary of the block before), the adds
// interleaved between this block's MFMAs: the matrix pipe never waits for them
__device__ __forceinline__ void az_mfma6_step(az_f32x16 &tnew, const float4 (&aq)[3], const float4 (&bq)[3],
                                              az_f32x16 &cprev, const az_f32x16 &tprev) {
    az_f32x16 t;
#pragma unroll
    for (int e = 0; e < 16; ++e) t[e] = 0.f;
    t = AZ_X6(t, 0, 0);
#pragma unroll
    for (int e = 0; e < 3; ++e) cprev[e] += tprev[e];
    t = AZ_X6(t, 0, 1);
#pragma unroll
    for (int e = 3; e < 6; ++e) cprev[e] += tprev[e];
    t = AZ_X6(t, 1, 0);
#pragma unroll
    for (int e = 6; e < 9; ++e) cprev[e] += tprev[e];
    t = AZ_X6(t, 1, 1);
#pragma unroll
    for (int e = 9; e < 12; ++e) cprev[e] += tprev[e];
    t = AZ_X6(t, 0, 2);
#pragma unroll
    for (int e = 12; e < 16; ++e) cprev[e] += tprev[e];
    t = AZ_X6(t, 2, 0);
    tnew = t;
    // the adds must stay HERE: without a use at this point LLVM sinks each add down to the next add of the
    // same accumulator (a whole tap later), every temporary stays live and the kernel spills ~600 registers
    asm volatile("" : "+v"(cprev));
    // pin the interleave: one MFMA, then three (four) of the independent adds, six times
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
}
