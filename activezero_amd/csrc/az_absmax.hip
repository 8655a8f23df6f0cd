// max |x| of a tensor into an amax array (az_common.h, include/azhip.h) -- the operand scale of the f16x3 arithmetic of the backward kernels
// (az_roll_common.h): a pure stream, one atomic per workgroup.  |x| is compared as its bit pattern (non-negative floats
// order like unsigned integers; inf / NaN elements are left out: the largest FINITE magnitude, az_common.h).
#include "az_common.h"

__global__ void __launch_bounds__(256)
absmax_kernel(unsigned *__restrict__ out, const float *__restrict__ x, long long n) {
    const long long n4 = n >> 2;
    unsigned m = 0;
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    const f32x4v *x4 = reinterpret_cast<const f32x4v *>(x);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4v v = __builtin_nontemporal_load(x4 + i);
        m = max(max(m, az_finite_abs_bits(v.x)), max(az_finite_abs_bits(v.y), max(az_finite_abs_bits(v.z), az_finite_abs_bits(v.w))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, az_finite_abs_bits(x[(n4 << 2) + threadIdx.x]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, off));
    __shared__ unsigned wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {  // (slots and the conditional atomic: az_common.h, az_amax_flush)
        const unsigned bm = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        unsigned *slot = out + (blockIdx.x & (AZ_AMAX_SLOTS - 1)) * AZ_AMAX_STRIDE;
        if (bm > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, bm);
    }
}

extern "C" int az_absmax(float *amax, const float *x, long long n, void *stream) {
    AZ_REQUIRE_PTR(amax); AZ_REQUIRE_PTR(x);
    AZ_REQUIRE(n > 0);
    if (reinterpret_cast<uintptr_t>(x) & 15) return AZ_EINVAL;
    hipStream_t s = az_stream(stream);
    if (hipMemsetAsync(amax, 0, AZ_AMAX_FLOATS * sizeof(float), s) != hipSuccess) return AZ_ELAUNCH;
    hipLaunchKernelGGL(absmax_kernel, dim3(az_grid_for((n + 3) / 4, 256)), dim3(256), 0, s,
                       reinterpret_cast<unsigned *>(amax), x, n);
    return az_launch_status();
}
