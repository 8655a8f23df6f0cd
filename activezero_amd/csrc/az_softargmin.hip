// K6 -- fused soft-argmin head: trilinear x4 upsample + softmax over D +
// disparity expectation, reference nets/psmnet/psmnet_3.py:184-215 and
// nets/psmnet/psmnet_submodule_3.py:80-89.
//
// The reference materialises [B,D,H,W] three times per head (401 MB each at
// 544x960, D=192).  Here a block owns a 4-row x 64-column tile of the output;
// the (3 x 18 x d) low-resolution logits it touches are staged once in LDS and
// every thread walks the D axis of its own pixel in registers:
//   pass A  v_k = bilinear(y,x) of plane k, M = max_D u_D      (softmax shift)
//   pass B  u_D = lerp_D(v_k, v_k+1);  s += exp(u_D - M);  t += D * exp(u_D - M)
//   out = t / s
// align_corners=False with an exact x4 scale gives the fixed phase weights
// (dst = 4k+r): r0 -> (.375,.625) of (k-1,k), r1 -> (.125,.875), r2 -> (.875,.125)
// of (k,k+1), r3 -> (.625,.375); border indices clamp (PyTorch
// area_pixel_compute_source_index + min(i0+1, size-1)).
// Algorithmic HBM bytes per head: 4*(d*h*w + H*W); the kernel is VALU-issue bound (see sa_stats_planes).
//
// Backward reads the per-pixel softmax shift / normaliser the forward saved (8 B per pixel; it
// recomputes them when given none), forms
//   d out / d u_D = p_D * (D - out)
// folds the D-lerp back onto the 48 planes, reduces the wave's 64 pixels onto their 18 x-cells
// with shuffles (quad xor, then +-4 lanes), accumulates the tile's (3 x 18 x d) gradient in LDS
// (one conflict-free ds_add_f32 lane per cell and wave) and flushes it with one global float
// atomic per LDS cell.
#include "az_common.h"

#define SA_TY 4
#define SA_TX 64
#define SA_LY 3   // low-res rows touched by 4 output rows
#define SA_LX 18  // low-res cols touched by 64 output cols

__device__ __forceinline__ void sa_load_tile(float *tile, const float *__restrict__ logits,
                                             int b, int d, int h, int w, int ybase, int xbase) {
    const int n = d * SA_LY * SA_LX;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const int lx = e % SA_LX;
        const int r = e / SA_LX;
        const int ly = r % SA_LY, k = r / SA_LY;
        const int gy = min(max(ybase + ly, 0), h - 1);
        const int gx = min(max(xbase + lx, 0), w - 1);
        tile[e] = logits[(((size_t)b * d + k) * h + gy) * w + gx];
    }
}

// exp(x) for x <= 0 on the hardware exp2 (backward pass only)
__device__ __forceinline__ float sa_exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

struct SaPix {
    int ly0, lx0;    // tile-local index of the lower cell; the upper one is +1
    float wy1, wx1;  // weight of the upper cell (lower gets 1 - w)
};

// Source cell / weight of PyTorch's linear x4 upsampling (align_corners=False),
// in UNCLAMPED form: dst = 4q+r reads cells (q-1,q) for r<2 and (q,q+1) for r>=2.
// Border clamping is applied where the tile is loaded from / flushed to global
// memory (cell -1 aliases cell 0, cell `size` aliases size-1), which is the same
// linear map as PyTorch's clamped indices.
__device__ __forceinline__ void sa_axis(int dst, int base, int &l0, float &lam1) {
    const int q = dst >> 2, r = dst & 3;
    l0 = ((r < 2) ? q - 1 : q) - base;
    lam1 = (r < 2) ? (r == 0 ? 0.625f : 0.875f) : (r == 2 ? 0.125f : 0.375f);
}

__device__ __forceinline__ float sa_plane(const float *tile, int k, const SaPix &p) {
    const float *t = tile + k * (SA_LY * SA_LX);
    const float *c = t + p.ly0 * SA_LX + p.lx0;
    const float a00 = c[0], a01 = c[1], a10 = c[SA_LX], a11 = c[SA_LX + 1];
    const float wx0 = 1.f - p.wx1, wy0 = 1.f - p.wy1;
    return wy0 * (wx0 * a00 + p.wx1 * a01) + p.wy1 * (wx0 * a10 + p.wx1 * a11);
}

// softmax statistics of one pixel: M (shift), s = sum exp, t = sum D*exp.
// The counters of the first version (profiles/r02_sq_softargmin_patch_reproj.md: 5 245 VALU instructions per
// pixel = 27 per disparity, 61 % of wave cycles in dependent-issue stalls) shaped this one:
//   * every plane value v_k (a bilinear read of the LDS tile, ~11 instructions) is evaluated ONCE and kept in
//     registers (48 planes at D = 192) instead of once per pass;
//   * the shift stays the exact maximum of the upsampled logits, taken from the registers;
//   * planes are pre-scaled once, v'_k = (v_k - M) * log2(e); the lerp is linear, so exp(u - M) = exp2(lerp(v'))
//     is ONE v_exp_f32 (1 ulp) per disparity instead of expf's range reduction: 5 instructions per disparity;
//   * four independent (s, t) accumulator pairs break the dependent add chain.
template <int DP>
__device__ __forceinline__ void sa_stats_planes(const float *tile, const SaPix &p, float &M, float &s, float &t) {
    float v[DP];
#pragma unroll
    for (int k = 0; k < DP; ++k) v[k] = sa_plane(tile, k, p);
    // exact maximum of the UPSAMPLED logits (what F.softmax subtracts; a plane maximum would let every
    // upsampled level underflow when one plane towers over its neighbours): both end planes and, per plane
    // pair, the two outer lerp phases (the lerp is linear in the phase weight)
    M = fmaxf(v[0], v[DP - 1]);
#pragma unroll
    for (int k = 0; k + 1 < DP; ++k)
        M = fmaxf(M, fmaxf(0.875f * v[k] + 0.125f * v[k + 1], 0.125f * v[k] + 0.875f * v[k + 1]));
#pragma unroll
    for (int k = 0; k < DP; ++k) v[k] = (v[k] - M) * 1.4426950408889634f;
    float sa[4] = {0.f, 0.f, 0.f, 0.f}, ta[4] = {0.f, 0.f, 0.f, 0.f};
    {   // D = 0, 1 -> plane 0 (source index clamped to 0)
        const float e = __builtin_amdgcn_exp2f(v[0]);
        sa[0] += e + e;
        ta[0] += e;
    }
#pragma unroll
    for (int k = 0; k + 1 < DP; ++k) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float w1 = 0.125f + 0.25f * m;
            const float e = __builtin_amdgcn_exp2f((1.f - w1) * v[k] + w1 * v[k + 1]);
            sa[m] += e;
            ta[m] = fmaf((float)(4 * k + 2 + m), e, ta[m]);
        }
    }
    {   // D = 4d-2, 4d-1 -> plane d-1 (upper index clamped)
        const float e = __builtin_amdgcn_exp2f(v[DP - 1]);
        sa[1] += e + e;
        ta[1] += (float)(4 * DP - 2) * e + (float)(4 * DP - 1) * e;
    }
    s = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    t = (ta[0] + ta[1]) + (ta[2] + ta[3]);
}

// generic depth (any d): two passes over the LDS tile, same shift rule and exp2 arithmetic
__device__ __forceinline__ void sa_stats(const float *tile, int d, const SaPix &p, float &M,
                                         float &s, float &t) {
    if (d == 48) { sa_stats_planes<48>(tile, p, M, s, t); return; }  // D = 192
    if (d == 16) { sa_stats_planes<16>(tile, p, M, s, t); return; }  // D = 64 (BASELINE configs[0])
    {
        float a0 = sa_plane(tile, 0, p);
        M = a0;
        for (int k = 0; k + 1 < d; ++k) {
            const float a1 = sa_plane(tile, k + 1, p);
            M = fmaxf(M, fmaxf(0.875f * a0 + 0.125f * a1, 0.125f * a0 + 0.875f * a1));
            a0 = a1;
        }
        M = fmaxf(M, a0);
    }
    const float L2E = 1.4426950408889634f;
    s = 0.f;
    t = 0.f;
    float v0 = (sa_plane(tile, 0, p) - M) * L2E;
    {
        const float e = __builtin_amdgcn_exp2f(v0);
        s += e + e;
        t += e;
    }
    for (int k = 0; k + 1 < d; ++k) {
        const float v1 = (sa_plane(tile, k + 1, p) - M) * L2E;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const float w1 = 0.125f + 0.25f * m;
            const float e = __builtin_amdgcn_exp2f((1.f - w1) * v0 + w1 * v1);
            s += e;
            t = fmaf((float)(4 * k + 2 + m), e, t);
        }
        v0 = v1;
    }
    {
        const float e = __builtin_amdgcn_exp2f(v0);
        s += e + e;
        t += (float)(4 * d - 2) * e + (float)(4 * d - 1) * e;
    }
}

__global__ void __launch_bounds__(256)
softargmin_fwd_kernel(float *__restrict__ out, float2 *__restrict__ stats,
                      const float *__restrict__ logits, int d, int h, int w, int tiles_x) {
    extern __shared__ float tile[];
    const int H = 4 * h, W = 4 * w;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int b = blockIdx.y;
    const int ybase = ty * (SA_TY / 4) - 1, xbase = tx * (SA_TX / 4) - 1;
    sa_load_tile(tile, logits, b, d, h, w, ybase, xbase);
    __syncthreads();
    const int Y = ty * SA_TY + (threadIdx.x >> 6), X = tx * SA_TX + (threadIdx.x & 63);
    if (X >= W || Y >= H) return;
    SaPix p;
    sa_axis(Y, ybase, p.ly0, p.wy1);
    sa_axis(X, xbase, p.lx0, p.wx1);
    float M, s, t;
    sa_stats(tile, d, p, M, s, t);
    out[((size_t)b * H + Y) * W + X] = t / s;
    if (stats) stats[((size_t)b * H + Y) * W + X] = make_float2(M, s);  // saved for the backward pass
}

__global__ void __launch_bounds__(256)
softargmin_bwd_kernel(float *__restrict__ glogits, const float *__restrict__ gout,
                      const float *__restrict__ logits, const float2 *__restrict__ stats,
                      const float *__restrict__ fwd_out, int d, int h, int w, int tiles_x) {
    extern __shared__ float smem[];
    float *tile = smem;
    // gradient of the low-res tile: one PRIVATE image per wave (= per output row of the block), [wave][k][2 cell
    // rows][SA_LX], written with plain stores -- within a wave every cell has exactly one writing lane -- and summed
    // by the flush loop below.  (The four waves used to add into one shared image with ds_add_f32: 54 % of the
    // kernel's wave cycles were SQ_WAIT_INST_LDS, profiles/r02_sq_softargmin_patch_reproj.md.)
    float *gtile = smem + d * SA_LY * SA_LX;
    float *gmine = gtile + (threadIdx.x >> 6) * (d * 2 * SA_LX);
    const int H = 4 * h, W = 4 * w;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int b = blockIdx.y;
    const int ybase = ty * (SA_TY / 4) - 1, xbase = tx * (SA_TX / 4) - 1;
    const int ncell = d * SA_LY * SA_LX;
    sa_load_tile(tile, logits, b, d, h, w, ybase, xbase);
    __syncthreads();
    const int Y = ty * SA_TY + (threadIdx.x >> 6), X = tx * SA_TX + (threadIdx.x & 63);
    const bool live = (X < W) && (Y < H);
    SaPix p;
    sa_axis(Y, ybase, p.ly0, p.wy1);  // indices depend on the thread id only: always in-tile
    sa_axis(X, xbase, p.lx0, p.wx1);
    // softmax shift M, normaliser s and the prediction: read back when the forward saved them
    // (training), recomputed otherwise.  p_D = exp(u_D - M) / s below uses the hardware exp2
    // (v_exp_f32, 1 ulp): gradients do not need expf's range reduction, the forward keeps it.
    float M, s, pred;
    if (stats) {
        const size_t pix = live ? ((size_t)b * H + Y) * W + X : 0;
        const float2 ms = stats[pix];
        M = ms.x; s = ms.y; pred = fwd_out[pix];
    } else {
        float t;
        sa_stats(tile, d, p, M, s, t);
        pred = t / s;
    }
    const float g = live ? gout[((size_t)b * H + Y) * W + X] / s : 0.f;  // g * (1/s)

    // x-quad roles: the 4 pixels X = 4q..4q+3 touch cells q-1, q, q+1.
    const int r = threadIdx.x & 3, wl = threadIdx.x & 63;  // lane in the quad / in the wave (= X - 64 tx)
    const int cq = (X >> 2) - xbase;  // tile-local column of cell q, in [1,16]
    const float wx0 = 1.f - p.wx1;
    const float f_m1 = (r < 2) ? wx0 : 0.f;   // share going to cell q-1
    const float f_0 = (r < 2) ? p.wx1 : wx0;  // ... to cell q
    const float f_p1 = (r < 2) ? 0.f : p.wx1; // ... to cell q+1

    float v0 = sa_plane(tile, 0, p);
    float acc0;  // gradient being collected for plane k
    {
        const float e = sa_exp_fast(v0 - M);
        acc0 = g * e * ((0.f - pred) + (1.f - pred));
    }
    for (int k = 0; k < d; ++k) {
        float acc1 = 0.f;
        if (k + 1 < d) {
            const float v1 = sa_plane(tile, k + 1, p);
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float w1 = 0.125f + 0.25f * m;
                const float u = (1.f - w1) * v0 + w1 * v1;
                const float gu = g * sa_exp_fast(u - M) * ((float)(4 * k + 2 + m) - pred);
                acc0 += (1.f - w1) * gu;
                acc1 += w1 * gu;
            }
            v0 = v1;
        } else {
            const float e = sa_exp_fast(v0 - M);
            acc0 += g * e * (((float)(4 * d - 2) - pred) + ((float)(4 * d - 1) - pred));
        }
        // plane k complete: quad-reduce the three x-cell shares, then finish the reduction across
        // quads in registers -- cell q also receives the right share of quad q-1 and the left share
        // of quad q+1 -- so that every LDS cell gets ONE ds_add_f32 lane per wave.  (Three lanes per
        // quad adding their own shares made each instruction a 3-way same-address conflict; those
        // serialised atomics were 0.6 ms of this kernel's 0.87 ms.)
        float s_m1 = f_m1 * acc0, s_0 = f_0 * acc0, s_p1 = f_p1 * acc0;
        s_m1 += __shfl_xor(s_m1, 1); s_m1 += __shfl_xor(s_m1, 2);
        s_0 += __shfl_xor(s_0, 1);   s_0 += __shfl_xor(s_0, 2);
        s_p1 += __shfl_xor(s_p1, 1); s_p1 += __shfl_xor(s_p1, 2);
        const float from_left = __shfl_up(s_p1, 4), from_right = __shfl_down(s_m1, 4);
        const float cell = s_0 + (wl >= 4 ? from_left : 0.f) + (wl < 60 ? from_right : 0.f);
        // lane 0 of every quad owns cell q; lane 1 of the first / last quad owns the halo cells
        const bool edge_l = (wl == 1), edge_r = (wl == 61);
        if (r == 0 || edge_l || edge_r) {
            const float val = (r == 0) ? cell : (edge_l ? s_m1 : s_p1);
            const int col = (r == 0) ? cq : (edge_l ? cq - 1 : cq + 1);
            float *gt = gmine + k * (2 * SA_LX) + col;  // cell rows ly0 (slot 0) and ly0 + 1 (slot 1) of this wave
            gt[0] = (1.f - p.wy1) * val;
            gt[SA_LX] = p.wy1 * val;
        }
        acc0 = acc1;
    }
    __syncthreads();
    // tile-local base cell row of each wave's output row (Y = SA_TY ty + wave; ybase = the tile's first cell row)
    int ly0_of[SA_TY];
#pragma unroll
    for (int wv = 0; wv < SA_TY; ++wv) {
        SaPix q;
        sa_axis(ty * SA_TY + wv, ybase, q.ly0, q.wy1);
        ly0_of[wv] = q.ly0;
    }
    for (int e = threadIdx.x; e < ncell; e += blockDim.x) {
        const int lx = e % SA_LX;
        const int rr = e / SA_LX;
        const int ly = rr % SA_LY, k = rr / SA_LY;
        float v = 0.f;
#pragma unroll
        for (int wv = 0; wv < SA_TY; ++wv) {  // fixed order: the block's sum is deterministic
            const int slot = ly - ly0_of[wv];
            if (slot == 0 || slot == 1) v += gtile[(wv * d + k) * (2 * SA_LX) + slot * SA_LX + lx];
        }
        if (v == 0.f) continue;
        const int gy = min(max(ybase + ly, 0), h - 1);
        const int gx = min(max(xbase + lx, 0), w - 1);
#if defined(SA_DIAG) && (SA_DIAG & 2)
        glogits[(((size_t)b * d + k) * h + gy) * w + gx] = v;
#else
        atomicAdd(&glogits[(((size_t)b * d + k) * h + gy) * w + gx], v);
#endif
    }
}

static int sa_check(int B, int d, int h, int w) {
    if (!(B > 0 && d > 0 && h > 0 && w > 0)) return AZ_EINVAL;
    if (B > 65535) return AZ_EUNSUPPORTED;
    if (((size_t)d * SA_LY * SA_LX + (size_t)SA_TY * d * 2 * SA_LX) * sizeof(float) > 64 * 1024) return AZ_EUNSUPPORTED;
    return AZ_OK;
}

extern "C" int az_softargmin_fwd(float *disp_out, float *stats_out, const float *logits, int B, int d,
                                 int h, int w, void *stream) {
    AZ_REQUIRE_PTR(disp_out); AZ_REQUIRE_PTR(logits);
    if (int e = sa_check(B, d, h, w)) return e;
    const int tiles_x = (4 * w + SA_TX - 1) / SA_TX, tiles_y = (4 * h) / SA_TY;
    hipLaunchKernelGGL(softargmin_fwd_kernel, dim3(tiles_x * tiles_y, B), dim3(256),
                       (size_t)d * SA_LY * SA_LX * sizeof(float), az_stream(stream), disp_out,
                       reinterpret_cast<float2 *>(stats_out), logits, d, h, w, tiles_x);
    return az_launch_status();
}

extern "C" int az_softargmin_bwd(float *grad_logits, const float *grad_disp,
                                 const float *logits, const float *stats, const float *disp_fwd,
                                 int B, int d, int h, int w, void *stream) {
    AZ_REQUIRE_PTR(grad_logits); AZ_REQUIRE_PTR(grad_disp); AZ_REQUIRE_PTR(logits);
    if ((stats == nullptr) != (disp_fwd == nullptr)) return AZ_EINVAL;
    if (int e = sa_check(B, d, h, w)) return e;
    if (hipMemsetAsync(grad_logits, 0, (size_t)B * d * h * w * sizeof(float),
                       az_stream(stream)) != hipSuccess)
        return AZ_ELAUNCH;
    const int tiles_x = (4 * w + SA_TX - 1) / SA_TX, tiles_y = (4 * h) / SA_TY;
    hipLaunchKernelGGL(softargmin_bwd_kernel, dim3(tiles_x * tiles_y, B), dim3(256),
                       ((size_t)d * SA_LY * SA_LX + (size_t)SA_TY * d * 2 * SA_LX) * sizeof(float), az_stream(stream),
                       grad_logits, grad_disp, logits, reinterpret_cast<const float2 *>(stats), disp_fwd,
                       d, h, w, tiles_x);
    return az_launch_status();
}
