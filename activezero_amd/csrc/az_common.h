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
