// What MODE.FP16_OVFL does on gfx950, for the saturating operand conversion of the f16x3 kernels (az_common.h):
//   (a) v_cvt_f16_f32 of {1e6, -1e6, 65504, 7e4, inf, -inf, nan, 1.0} with the bit off / on;
//   (b) v_mfma_f32_16x16x32_f16 with an fp16 inf, an fp16 NaN or 65504 in one A row, bit off / on: what the fp32
//       result of that row (and of its neighbours) is.
// hipcc --offload-arch=gfx950 -O2 tools/probes/fp16_ovfl_probe.hip -o fp16_ovfl_probe && ./fp16_ovfl_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int ON>
__global__ void cvt_kernel(unsigned short *o, const float *x, int n) {
    if (ON) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
    const int i = threadIdx.x;
    if (i < n) { const _Float16 h = (_Float16)x[i]; o[i] = __builtin_bit_cast(unsigned short, h); }
}

// A[16 x 32] row-major fp16 bits in `a` (row r, k), B = all ones; out[16 x 16]
template <int ON>
__global__ void mfma_kernel(float *out, const unsigned short *a) {
    if (ON) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
    const int lane = threadIdx.x;
    f16x8 av, bv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        av[j] = __builtin_bit_cast(_Float16, a[(lane & 15) * 32 + 8 * (lane >> 4) + j]);
        bv[j] = (_Float16)1.0f;
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
    // C layout: lane holds column lane & 15, rows 4 (lane >> 4) + r
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];
}

int main() {
    const float xs[8] = {1e6f, -1e6f, 65504.f, 7e4f, __builtin_inff(), -__builtin_inff(), __builtin_nanf(""), 1.0f};
    float *dx; unsigned short *dout; CHECK(hipMalloc(&dx, sizeof(xs))); CHECK(hipMalloc(&dout, 16));
    CHECK(hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice));
    for (int on = 0; on < 2; ++on) {
        if (on) hipLaunchKernelGGL(cvt_kernel<1>, dim3(1), dim3(64), 0, 0, dout, dx, 8);
        else hipLaunchKernelGGL(cvt_kernel<0>, dim3(1), dim3(64), 0, 0, dout, dx, 8);
        unsigned short h[8]; CHECK(hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost));
        printf("cvt FP16_OVFL=%d:", on);
        for (int i = 0; i < 8; ++i) printf("  %g->0x%04x", xs[i], h[i]);
        printf("\n");
    }
    unsigned short a[16 * 32]; unsigned short *da; float *dc; CHECK(hipMalloc(&da, sizeof(a))); CHECK(hipMalloc(&dc, 1024));
    const unsigned short specials[4] = {0x7c00 /* inf */, 0x7e00 /* nan */, 0x7bff /* 65504 */, 0xfc00 /* -inf */};
    const char *names[4] = {"inf", "nan", "65504", "-inf"};
    for (int s = 0; s < 4; ++s)
        for (int on = 0; on < 2; ++on) {
            for (int i = 0; i < 16 * 32; ++i) a[i] = 0x3c00;  // 1.0
            a[5 * 32 + 3] = specials[s];                       // row 5
            CHECK(hipMemcpy(da, a, sizeof(a), hipMemcpyHostToDevice));
            if (on) hipLaunchKernelGGL(mfma_kernel<1>, dim3(1), dim3(64), 0, 0, dc, da);
            else hipLaunchKernelGGL(mfma_kernel<0>, dim3(1), dim3(64), 0, 0, dc, da);
            float c[256]; CHECK(hipMemcpy(c, dc, 1024, hipMemcpyDeviceToHost));
            printf("mfma A[5][3]=%s FP16_OVFL=%d: row 4 -> %g, row 5 -> %g (bits 0x%08x), row 6 -> %g\n", names[s], on, c[4 * 16], c[5 * 16],
                   *reinterpret_cast<unsigned *>(&c[5 * 16]), c[6 * 16]);
        }
    return 0;
}
