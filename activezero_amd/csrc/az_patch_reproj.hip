// K8 -- fused patch reprojection loss (get_reproj_error_patch), reference
// utils/reprojection.py:99-127 (Unfold(ps) of L and R -> apply_disparity on the
// C*ps*ps tap channels with the CENTRE pixel's disparity -> masked MSE -> Fold).
//
// The reference materialises three [B, C*ps^2, H, W] tensors (253 MB each per
// sample at 544x960, ps=11).  Here one thread owns one pixel:
//   warped(u,v) = sum_{a,b in {0,1}} wy_a wx_b R0[y0+a+u][x0+b+v]
// with R0 the zero-padded right pattern (Unfold's padding) and wy_a / wx_b set
// to 0 when the bilinear corner (y0+a, x0+b) itself is outside the image
// (grid_sample's zero padding).  Rows are walked with a sliding pair of
// horizontally interpolated rows, so a pixel costs (ps+1)^2 R loads and ps^2 L
// loads, all served by L1/L2; HBM traffic is L + R + disp + mask once.
// sum((warped - L)^2) and the element count are reduced per block and
// accumulated in fp64 (acc[0], acc[1]); loss = acc[0] / acc[1].
//
// Backward (w.r.t. the disparity only -- the patterns are data):
//   d loss / d disp = gloss * 2/count * sign * sum_taps diff * d warped / d ix.
#include "az_common.h"
#include "az_options.h"

#define PR_MAX_PS 15

__device__ __forceinline__ float pr_linspace01(int i, int n) {
    const float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? (step * (float)i) : (1.0f - step * (float)(n - 1 - i));
}

struct PrGeom {
    int x0, y0;
    float wx0, wx1, wy0, wy1;  // corner weights, zeroed for out-of-image corners
    float dwx0, dwx1;          // d(wx)/d(ix) with the same validity (-1/+1 or 0)
};

__device__ __forceinline__ PrGeom pr_geom(int i, int j, float disp, int H, int W) {
#pragma clang fp contract(off)
    const float gx = pr_linspace01(j, W) + disp / (float)W;
    const float gy = pr_linspace01(i, H);
    const float nx = 2.0f * gx - 1.0f, ny = 2.0f * gy - 1.0f;
    const float ix = ((nx + 1.0f) * (float)W - 1.0f) / 2.0f;
    const float iy = ((ny + 1.0f) * (float)H - 1.0f) / 2.0f;
    const float fx = floorf(ix), fy = floorf(iy);
    PrGeom g;
    g.x0 = (int)fminf(fmaxf(fx, -2.0f), (float)W + 1.0f);
    g.y0 = (int)fminf(fmaxf(fy, -2.0f), (float)H + 1.0f);
    const float tx = ix - fx, ty = iy - fy;
    const bool vx0 = g.x0 >= 0 && g.x0 < W, vx1 = g.x0 + 1 >= 0 && g.x0 + 1 < W;
    const bool vy0 = g.y0 >= 0 && g.y0 < H, vy1 = g.y0 + 1 >= 0 && g.y0 + 1 < H;
    g.wx0 = vx0 ? 1.f - tx : 0.f;
    g.wx1 = vx1 ? tx : 0.f;
    g.wy0 = vy0 ? 1.f - ty : 0.f;
    g.wy1 = vy1 ? ty : 0.f;
    g.dwx0 = vx0 ? -1.f : 0.f;
    g.dwx1 = vx1 ? 1.f : 0.f;
    return g;
}

__device__ __forceinline__ float pr_ld(const float *__restrict__ img, int y, int x, int H, int W) {
    return (y >= 0 && y < H && x >= 0 && x < W) ? img[(size_t)y * W + x] : 0.f;
}

// MODE 0: forward (sum of squared differences), MODE 1: backward (d/d disp)
template <int MODE>
__global__ void __launch_bounds__(256)
patch_reproj_kernel(double *__restrict__ acc, float *__restrict__ gdisp,
                    const float *__restrict__ gloss, const float *__restrict__ L,
                    const float *__restrict__ R, const float *__restrict__ disp,
                    const uint8_t *__restrict__ mask, int C, int H, int W, int ps,
                    float sign, long long total) {
    const int r = ps / 2;
    float local = 0.f;
    unsigned local_n = 0;
    float bwd_scale = 0.f;
    if (MODE == 1) bwd_scale = (float)((double)gloss[0] * 2.0 / acc[1]) * sign;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int j = idx % W;
        const long long q = idx / W;
        const int i = q % H;
        const long long b = q / H;
        if (mask && !mask[idx]) {
            if (MODE == 1) gdisp[idx] = 0.f;
            continue;
        }
        const PrGeom g = pr_geom(i, j, sign * disp[idx], H, W);
        float pix = 0.f;
        for (int c = 0; c < C; ++c) {
            const float *Rc = R + ((size_t)b * C + c) * H * W;
            const float *Lc = L + ((size_t)b * C + c) * H * W;
            float prev[PR_MAX_PS], dprev[PR_MAX_PS];
            for (int t = -r; t <= r + 1; ++t) {
                const int y = g.y0 + t;
                float cur[PR_MAX_PS], dcur[PR_MAX_PS];
                float a = pr_ld(Rc, y, g.x0 - r, H, W);
#pragma unroll
                for (int v = 0; v < PR_MAX_PS; ++v) {
                    if (v < ps) {
                        const float bnext = pr_ld(Rc, y, g.x0 - r + v + 1, H, W);
                        cur[v] = g.wx0 * a + g.wx1 * bnext;
                        if (MODE == 1) dcur[v] = g.dwx0 * a + g.dwx1 * bnext;
                        a = bnext;
                    }
                }
                if (t > -r) {
                    const int u = t - 1;  // patch row offset
#pragma unroll
                    for (int v = 0; v < PR_MAX_PS; ++v) {
                        if (v < ps) {
                            const float warped = g.wy0 * prev[v] + g.wy1 * cur[v];
                            const float diff = warped - pr_ld(Lc, i + u, j + v - r, H, W);
                            if (MODE == 0) pix += diff * diff;
                            else pix += diff * (g.wy0 * dprev[v] + g.wy1 * dcur[v]);
                        }
                    }
                }
#pragma unroll
                for (int v = 0; v < PR_MAX_PS; ++v) {
                    prev[v] = cur[v];
                    if (MODE == 1) dprev[v] = dcur[v];
                }
            }
        }
        if (MODE == 0) {
            local += pix;
            local_n += 1;
        } else {
            gdisp[idx] = pix * bwd_scale;
        }
    }
    if (MODE == 0) {
        // wave reduce, then block reduce through LDS, one fp64 atomic per block
        __shared__ float s_sum[4];
        __shared__ unsigned s_cnt[4];
        for (int o = 32; o > 0; o >>= 1) {
            local += __shfl_xor(local, o);
            local_n += __shfl_xor(local_n, o);
        }
        if ((threadIdx.x & 63) == 0) {
            s_sum[threadIdx.x >> 6] = local;
            s_cnt[threadIdx.x >> 6] = local_n;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double s = (double)s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
            const double n = (double)s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            atomicAdd(&acc[0], s);
            atomicAdd(&acc[1], n * (double)(C * ps * ps));
        }
    }
}


// ---- block-tiled form -----------------------------------------------------------------------------------------
// The per-pixel kernel above issues (ps+1)^2 + ps^2 = 265 bounds-checked scalar global loads per pixel at ps = 11:
// 72 % of its wave cycles are SQ_WAIT_ANY on L1 gathers (profiles/r02_sq_softargmin_patch_reproj.md), 0.53 ms for 27 MB.
// Here a workgroup owns a band of TR image rows at FULL width and keeps, zero-padded, in LDS:
//   R rows [i0 - 1 - r, i0 + TR + r]  (the bilinear base row y0 = floor(i H / (H-1) - 0.5) is i - 1 or i; the window
//          of a pixel sits at x0 = floor(j W / (W-1) + disp - 0.5): data dependent, anywhere in the row -- hence full rows),
//   L rows [i0 - r, i0 + TR - 1 + r],
// so the inner loops read LDS without a bounds check.  A thread takes four x-adjacent pixels of one row: they share
// y0 and the L window (ps + 3 values per patch row instead of 4 ps); each keeps its own pair of horizontally
// interpolated R rows.  Per-pixel arithmetic (order c, patch row, v) is that of the kernel above.  HBM traffic:
// every image row is read (TR + 2r + 2) / TR times from L2 instead of (ps+1)^2 times from L1.
// Shapes whose bands do not fit in 160 KB of LDS (W > ~1000 at ps = 11) keep the per-pixel kernel.
// PSM: compile-time bound of ps (register arrays), PR_K: pixels per thread (4 forward; 2 backward, which also carries
// the derivative rows)
template <int MODE, int PSM, int PR_K>
__global__ void __launch_bounds__(512)
patch_reproj_tiled_kernel(double *__restrict__ acc, float *__restrict__ gdisp, const float *__restrict__ gloss,
                          const float *__restrict__ L, const float *__restrict__ R, const float *__restrict__ disp,
                          const uint8_t *__restrict__ mask, int C, int H, int W, int ps, float sign, int TR, int nbands) {
    extern __shared__ __attribute__((aligned(16))) float pr_lds[];
    const int r = ps / 2, PX = r + 3;
    const int RS = W + 2 * PX, LS = W + 2 * r + PR_K;  // row strides (floats); L has PR_K spare columns on the right
    const int NR = TR + 2 * r + 2, NL = TR + 2 * r;
    float *Rb = pr_lds, *Lb = pr_lds + (size_t)NR * RS;
    const int b = blockIdx.x / nbands, band = blockIdx.x % nbands;
    const int i0 = band * TR, rows = min(TR, H - i0);
    const int ra = i0 - 1 - r, la = i0 - r;  // first image row of each band
    float local = 0.f;
    unsigned local_n = 0;
    float bwd_scale = 0.f;
    if (MODE == 1) bwd_scale = (float)((double)gloss[0] * 2.0 / acc[1]) * sign;
    const int groups = (W + PR_K - 1) / PR_K;

    for (int c = 0; c < C; ++c) {
        const float *Rc = R + ((size_t)b * C + c) * H * W;
        const float *Lc = L + ((size_t)b * C + c) * H * W;
        __syncthreads();  // the previous channel's bands are no longer read
        for (int q = threadIdx.x; q < NR * RS; q += blockDim.x) {
            const int yy = ra + q / RS, xx = q % RS - PX;
            Rb[q] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? Rc[(size_t)yy * W + xx] : 0.f;
        }
        for (int q = threadIdx.x; q < NL * LS; q += blockDim.x) {
            const int yy = la + q / LS, xx = q % LS - r;
            Lb[q] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? Lc[(size_t)yy * W + xx] : 0.f;
        }
        __syncthreads();
        for (int item = threadIdx.x; item < rows * groups; item += blockDim.x) {
            const int ii = item / groups, j0 = (item - ii * groups) * PR_K;
            const int i = i0 + ii;
            PrGeom g[PR_K];
            bool live[PR_K];
            float pix[PR_K];
#pragma unroll
            for (int k = 0; k < PR_K; ++k) {
                const int j = j0 + k;
                const long long idx = ((long long)b * H + i) * W + min(j, W - 1);
                live[k] = j < W && !(mask && !mask[idx]);
                g[k] = pr_geom(i, min(j, W - 1), sign * disp[idx], H, W);
                pix[k] = 0.f;
            }
            const int y0 = g[0].y0;  // the same for the whole row
            float prev[PR_K][PSM], dprev[PR_K][PSM];
            for (int t = -r; t <= r + 1; ++t) {
                const float *rrow = Rb + (size_t)(y0 + t - ra) * RS + PX - r;
                float cur[PR_K][PSM], dcur[PR_K][PSM];
#pragma unroll
                for (int k = 0; k < PR_K; ++k) {
                    const float *rp = rrow + g[k].x0;
                    float a = rp[0];
#pragma unroll
                    for (int v = 0; v < PSM; ++v) {
                        if (v < ps) {
                            const float bnext = rp[v + 1];
                            cur[k][v] = g[k].wx0 * a + g[k].wx1 * bnext;
                            if (MODE == 1) dcur[k][v] = g[k].dwx0 * a + g[k].dwx1 * bnext;
                            a = bnext;
                        }
                    }
                }
                if (t > -r) {
                    const int u = t - 1;  // patch row offset
                    const float *lp = Lb + (size_t)(i + u - la) * LS + j0;  // column j0 - r of the image
                    float lw[PSM + PR_K - 1];
#pragma unroll
                    for (int v = 0; v < PSM + PR_K - 1; ++v)
                        if (v < ps + PR_K - 1) lw[v] = lp[v];
#pragma unroll
                    for (int k = 0; k < PR_K; ++k) {
#pragma unroll
                        for (int v = 0; v < PSM; ++v) {
                            if (v < ps) {
                                const float warped = g[k].wy0 * prev[k][v] + g[k].wy1 * cur[k][v];
                                const float diff = warped - lw[v + k];
                                if (MODE == 0) pix[k] += diff * diff;
                                else pix[k] += diff * (g[k].wy0 * dprev[k][v] + g[k].wy1 * dcur[k][v]);
                            }
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < PR_K; ++k)
#pragma unroll
                    for (int v = 0; v < PSM; ++v) {
                        prev[k][v] = cur[k][v];
                        if (MODE == 1) dprev[k][v] = dcur[k][v];
                    }
            }
#pragma unroll
            for (int k = 0; k < PR_K; ++k) {
                const int j = j0 + k;
                if (j >= W) continue;
                const long long idx = ((long long)b * H + i) * W + j;
                if (MODE == 0) {
                    if (live[k]) { local += pix[k]; local_n += (c == 0) ? 1u : 0u; }
                } else {
                    // channels accumulate into the gradient (C is 1 for the IR patterns)
                    const float gv = live[k] ? pix[k] * bwd_scale : 0.f;
                    gdisp[idx] = (c == 0) ? gv : gdisp[idx] + gv;
                }
            }
        }
    }
    if (MODE == 0) {
        __shared__ float s_sum[8];
        __shared__ unsigned s_cnt[8];
        for (int o = 32; o > 0; o >>= 1) {
            local += __shfl_xor(local, o);
            local_n += __shfl_xor(local_n, o);
        }
        if ((threadIdx.x & 63) == 0) {
            s_sum[threadIdx.x >> 6] = local;
            s_cnt[threadIdx.x >> 6] = local_n;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double s_ = 0.0, n_ = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { s_ += (double)s_sum[w]; n_ += (double)s_cnt[w]; }
            atomicAdd(&acc[0], s_);
            atomicAdd(&acc[1], n_ * (double)(C * ps * ps));
        }
    }
}

// rows per band: the LDS the two bands need must fit (160 KB per CU, one workgroup per CU); among the fitting
// heights the one that leaves the fewest idle CUs in the last round of workgroups
static int pr_band_rows(int B, int H, int W, int ps, int pr_k, size_t *lds_bytes) {
    const int r = ps / 2;
    const size_t RS = W + 2 * (r + 3), LS = W + 2 * r + pr_k;
    int best = 0;
    double best_eff = 0.0;
    for (int tr = 1; tr <= 32 && tr <= H; ++tr) {
        const size_t need = ((size_t)(tr + 2 * r + 2) * RS + (size_t)(tr + 2 * r) * LS) * sizeof(float);
        if (need > 158 * 1024) break;
        const long long blocks = (long long)B * ((H + tr - 1) / tr);
        const long long rounds = (blocks + 255) / 256;
        // cost ~ rounds x (rows computed + rows staged); efficiency = useful rows / that
        const double eff = (double)B * H / (double)(rounds * 256) / (double)(tr + 0.15 * (2 * tr + 4 * r + 2));
        if (eff > best_eff) { best_eff = eff; best = tr; *lds_bytes = need; }
    }
    return best;
}
static bool pr_tiled_enabled() { return az_options().patch_tiled != 0; }
template <int MODE, int PSM, int K>
static bool pr_launch_tiled_k(double *acc, float *gdisp, const float *gloss, const float *L, const float *R,
                              const float *disp, const uint8_t *mask, int B, int C, int H, int W, int ps, float sign,
                              hipStream_t s);
template <int MODE, int PSM>
static bool pr_launch_tiled_ps(double *acc, float *gdisp, const float *gloss, const float *L, const float *R,
                               const float *disp, const uint8_t *mask, int B, int C, int H, int W, int ps, float sign,
                               hipStream_t s) {
    const int kk = az_options().patch_k;  // 1: one pixel per thread (A/B: same forward time, slower backward)
    if (kk == 1) return pr_launch_tiled_k<MODE, PSM, 1>(acc, gdisp, gloss, L, R, disp, mask, B, C, H, W, ps, sign, s);
    return pr_launch_tiled_k<MODE, PSM, (MODE == 0 ? 4 : 2)>(acc, gdisp, gloss, L, R, disp, mask, B, C, H, W, ps, sign, s);
}
template <int MODE, int PSM, int K>
static bool pr_launch_tiled_k(double *acc, float *gdisp, const float *gloss, const float *L, const float *R,
                              const float *disp, const uint8_t *mask, int B, int C, int H, int W, int ps, float sign,
                              hipStream_t s) {
    size_t lds = 0;
    const int tr = pr_band_rows(B, H, W, ps, K, &lds);
    if (tr <= 0) return false;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&patch_reproj_tiled_kernel<MODE, PSM, K>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024) != hipSuccess) {
            (void)hipGetLastError();  // (static + dynamic LDS must stay within the CU's 160 KB)
            return false;
        }
        attr_set = true;
    }
    const int nbands = (H + tr - 1) / tr;
    hipLaunchKernelGGL((patch_reproj_tiled_kernel<MODE, PSM, K>), dim3((unsigned)(B * nbands)), dim3(512), lds, s, acc, gdisp,
                       gloss, L, R, disp, mask, C, H, W, ps, sign, tr, nbands);
    return true;
}
template <int MODE>
static bool pr_launch_tiled(double *acc, float *gdisp, const float *gloss, const float *L, const float *R, const float *disp,
                            const uint8_t *mask, int B, int C, int H, int W, int ps, float sign, hipStream_t s) {
    if (!pr_tiled_enabled()) return false;
    if (ps <= 5) return pr_launch_tiled_ps<MODE, 5>(acc, gdisp, gloss, L, R, disp, mask, B, C, H, W, ps, sign, s);
    if (ps <= 11) return pr_launch_tiled_ps<MODE, 11>(acc, gdisp, gloss, L, R, disp, mask, B, C, H, W, ps, sign, s);
    return pr_launch_tiled_ps<MODE, PR_MAX_PS>(acc, gdisp, gloss, L, R, disp, mask, B, C, H, W, ps, sign, s);
}

// Fold visualisation: vis[b,c,y,x] = sum_{u,v} warped_{(c,u,v)}(y-u, x-v)
__global__ void __launch_bounds__(256)
patch_reproj_vis_kernel(float *__restrict__ vis, const float *__restrict__ R,
                        const float *__restrict__ disp, int C, int H, int W, int ps, float sign,
                        long long total) {
    const int r = ps / 2;
    for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int x = idx % W;
        const long long q = idx / W;
        const int y = q % H;
        const long long b = q / H;
        for (int c = 0; c < C; ++c) {
            const float *Rc = R + ((size_t)b * C + c) * H * W;
            float acc = 0.f;
            for (int u = -r; u <= r; ++u) {
                const int i = y - u;
                if (i < 0 || i >= H) continue;
                for (int v = -r; v <= r; ++v) {
                    const int j = x - v;
                    if (j < 0 || j >= W) continue;
                    const PrGeom g = pr_geom(i, j, sign * disp[((size_t)b * H + i) * W + j], H, W);
                    const float top = g.wx0 * pr_ld(Rc, g.y0 + u, g.x0 + v, H, W) +
                                      g.wx1 * pr_ld(Rc, g.y0 + u, g.x0 + 1 + v, H, W);
                    const float bot = g.wx0 * pr_ld(Rc, g.y0 + 1 + u, g.x0 + v, H, W) +
                                      g.wx1 * pr_ld(Rc, g.y0 + 1 + u, g.x0 + 1 + v, H, W);
                    acc += g.wy0 * top + g.wy1 * bot;
                }
            }
            vis[((size_t)b * C + c) * H * W + (size_t)y * W + x] = acc;
        }
    }
}

static int pr_check(int B, int C, int H, int W, int ps) {
    if (!(B > 0 && C > 0 && H > 1 && W > 1)) return AZ_EINVAL;
    if (ps < 1 || (ps & 1) == 0) return AZ_EINVAL;
    if (ps > PR_MAX_PS) return AZ_EUNSUPPORTED;
    return AZ_OK;
}

extern "C" int az_patch_reproj_fwd(double *acc, const float *L, const float *R,
                                   const float *disp, const uint8_t *mask, int B, int C, int H,
                                   int W, int ps, float sign, void *stream) {
    AZ_REQUIRE_PTR(acc); AZ_REQUIRE_PTR(L); AZ_REQUIRE_PTR(R); AZ_REQUIRE_PTR(disp);
    if (int e = pr_check(B, C, H, W, ps)) return e;
    if (hipMemsetAsync(acc, 0, 2 * sizeof(double), az_stream(stream)) != hipSuccess)
        return AZ_ELAUNCH;
    const long long total = (long long)B * H * W;
    if (pr_launch_tiled<0>(acc, nullptr, nullptr, L, R, disp, mask, B, C, H, W, ps, sign, az_stream(stream)))
        return az_launch_status();
    hipLaunchKernelGGL(patch_reproj_kernel<0>, dim3(az_grid_for(total, 256)), dim3(256), 0,
                       az_stream(stream), acc, (float *)nullptr, (const float *)nullptr, L, R,
                       disp, mask, C, H, W, ps, sign, total);
    return az_launch_status();
}

extern "C" int az_patch_reproj_bwd(float *grad_disp, const float *grad_loss, const double *acc,
                                   const float *L, const float *R, const float *disp,
                                   const uint8_t *mask, int B, int C, int H, int W, int ps,
                                   float sign, void *stream) {
    AZ_REQUIRE_PTR(grad_disp); AZ_REQUIRE_PTR(grad_loss); AZ_REQUIRE_PTR(acc);
    AZ_REQUIRE_PTR(L); AZ_REQUIRE_PTR(R); AZ_REQUIRE_PTR(disp);
    if (int e = pr_check(B, C, H, W, ps)) return e;
    const long long total = (long long)B * H * W;
    if (pr_launch_tiled<1>(const_cast<double *>(acc), grad_disp, grad_loss, L, R, disp, mask, B, C, H, W, ps, sign,
                           az_stream(stream)))
        return az_launch_status();
    hipLaunchKernelGGL(patch_reproj_kernel<1>, dim3(az_grid_for(total, 256)), dim3(256), 0,
                       az_stream(stream), const_cast<double *>(acc), grad_disp, grad_loss, L, R,
                       disp, mask, C, H, W, ps, sign, total);
    return az_launch_status();
}

extern "C" int az_patch_reproj_vis(float *vis, const float *R, const float *disp, int B, int C,
                                   int H, int W, int ps, float sign, void *stream) {
    AZ_REQUIRE_PTR(vis); AZ_REQUIRE_PTR(R); AZ_REQUIRE_PTR(disp);
    if (int e = pr_check(B, C, H, W, ps)) return e;
    const long long total = (long long)B * H * W;
    hipLaunchKernelGGL(patch_reproj_vis_kernel, dim3(az_grid_for(total, 256)), dim3(256), 0,
                       az_stream(stream), vis, R, disp, C, H, W, ps, sign, total);
    return az_launch_status();
}
