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

// Exact 3-way split of four fp32 values into bf16 parts (truncation: 8+8+8 significand bits,
// x == hi + mid + lo exactly), packed 4 x bf16 = 8 bytes per part.  v_perm_b32 picks the high
// halves of two registers, so no masking is needed for packing; only the two remainders are.
__device__ __forceinline__ void az_split3_bf16x4(const float4 &v, uint2 &hi, uint2 &mid, uint2 &lo) {
    const float xs[4] = {v.x, v.y, v.z, v.w};
    float r1[4], r2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r1[e] = xs[e] - __uint_as_float(__float_as_uint(xs[e]) & 0xffff0000u);
        r2[e] = r1[e] - __uint_as_float(__float_as_uint(r1[e]) & 0xffff0000u);
    }
    // perm(S0, S1, sel): bytes 0-3 of the pool are S1, 4-7 are S0 -> {S1.hi16, S0.hi16}
#define AZ_HI2(a, b) __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u)
    hi = make_uint2(AZ_HI2(xs[0], xs[1]), AZ_HI2(xs[2], xs[3]));
    mid = make_uint2(AZ_HI2(r1[0], r1[1]), AZ_HI2(r1[2], r1[3]));
    lo = make_uint2(AZ_HI2(r2[0], r2[1]), AZ_HI2(r2[2], r2[3]));
#undef AZ_HI2
}
