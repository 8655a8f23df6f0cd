// K1/K2 -- integer scatter warp (apply_disparity_cu), reference
// utils/warp_ops.py:22-45.
//
// The reference gives every (n,c,y) row to ONE thread that walks the row
// serially; the last writer of a destination wins (smallest j for disp >= 0,
// largest j for disp <= 0).  Here one 64-lane wavefront owns one (n,y) row for
// all C channels:
//   pass 1  winner[t] = min (pos) / max (neg) over { j : j + disp[j] == t }
//           by LDS integer atomics -- order independent, hence deterministic;
//   pass 2  dst[c][t] = winner[t] valid ? src[c][winner[t]] : 0, coalesced
//           stores, zero fill fused.
// HBM traffic = disp once + src once + dst once: 4*H*W*(2C+1) bytes.
#include "az_common.h"

template <bool POS>
__global__ void __launch_bounds__(256)
warp_scatter_kernel(float *__restrict__ dst, const float *__restrict__ src,
                    const int *__restrict__ disp, int C, int H, int W, int rows) {
    extern __shared__ int winner_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int *winner = winner_all + wave * W;
    const int row = blockIdx.x * 4 + wave;  // (n, y)
    const bool active = row < rows;
    const int none = POS ? 0x7fffffff : -1;
    for (int t = lane; t < W; t += 64) winner[t] = none;
    __syncthreads();
    if (active) {
        const int *drow = disp + (size_t)row * W;
        for (int j = lane; j < W; j += 64) {
            const int t = j + drow[j];
            if (t >= 0 && t < W) {
                if (POS) atomicMin(&winner[t], j);
                else atomicMax(&winner[t], j);
            }
        }
    }
    __syncthreads();
    if (!active) return;
    const int n = row / H, y = row - n * H;
    for (int c = 0; c < C; ++c) {
        const size_t base = (((size_t)n * C + c) * H + y) * W;
        for (int t = lane; t < W; t += 64) {
            const int j = winner[t];
            dst[base + t] = (j != none) ? src[base + j] : 0.0f;
        }
    }
}

extern "C" int az_warp_scatter(float *dst, const float *src, const int32_t *disp, int N, int C,
                               int H, int W, int sign, void *stream) {
    AZ_REQUIRE_PTR(dst); AZ_REQUIRE_PTR(src); AZ_REQUIRE_PTR(disp);
    AZ_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && sign != 0);
    if ((size_t)W * 4 * sizeof(int) > 64 * 1024) return AZ_EUNSUPPORTED;  // W <= 4096
    const int rows = N * H;
    const unsigned grid = (rows + 3) / 4;
    const size_t lds = (size_t)W * 4 * sizeof(int);
    if (sign > 0)
        hipLaunchKernelGGL(warp_scatter_kernel<true>, dim3(grid), dim3(256), lds,
                           az_stream(stream), dst, src, disp, C, H, W, rows);
    else
        hipLaunchKernelGGL(warp_scatter_kernel<false>, dim3(grid), dim3(256), lds,
                           az_stream(stream), dst, src, disp, C, H, W, rows);
    return az_launch_status();
}
