// K3+K4 fused algebraically: the first 3-D convolution of the cost aggregation applied to the
// concat cost volume WITHOUT the volume (reference nets/psmnet/psmnet_3.py:149-166: the volume
// V[d,y,x,:32] = L[y,x] (x >= d), V[d,y,x,32:] = R[y,x-d] (x >= d), then dres0[0] = Conv3d(64,32,3,pad 1)).
//
// V is constant along d in its left half and a pure x-shift in its right half, so
//   conv3d(V)[d,y,x] = F_{c(d), min(x-d,2)}[y,x] + G_{c(d), [x = W-1]}[y, x-d]          (x - d >= -2, else 0)
// (the F variants with x - d < 2 are only ever read at x = d + delta <= D: they are computed on the
//  first D+1 columns only -- "edge" maps -- and the four x - d >= 2 maps -- "bulk" -- on the full width)
// where the F are 2-D 3x3 convolutions of L with the 3-D kernel summed over the depth taps that exist
// for depth class c (first / middle / last plane) and that the staircase mask x' >= d' admits at offset
// delta = x - d (only delta < 2 cuts taps), and the G are 2-D 3x5 convolutions of R whose horizontal
// tap index is kw - kd (the mask is R's own left zero padding; only the image's right edge needs its own
// variant).  The 2-D convolutions (32 -> 96 full width + 384 on D+1 columns, and 32 -> 192: ~45 GFLOP at B=4
// instead of 693) run on this library's 2-D convolution kernels through autograd (activezero_amd/costconv.py
// builds the merged kernels with differentiable tensor ops); this file holds the memory-bound ends:
//   assemble_fwd : out[b,d,y,x,:] = F[...] + G[...]                 (writes the 32-channel V0 tensor once)
//   assemble_bwd : dF, dG = the matching reductions of grad_out over d (reads it twice)
#include "az_common.h"

#define CC_MAXCLS 3
// Depth classes actually present: NC = min(D, 3).  D >= 3: 0 = first plane (kd in {1,2}), 1 = middle
// ({0,1,2}), 2 = last ({0,1}); D = 2: 0 = first, 1 = last; D = 1: 0 = the only plane ({1}).
// Channel counts of the maps: bulk NC*32 (delta >= 2, full width), edge NC*4*32 (delta = -2..1, columns
// x <= D only, x = d + delta), G NC*2*32.
__host__ __device__ __forceinline__ int cc_nclass(int D) { return D < 3 ? D : 3; }
__device__ __forceinline__ int cc_class(int d, int D) {
    return (D == 1 || d == 0) ? 0 : (d == D - 1 ? (D == 2 ? 1 : 2) : 1);
}

// one thread = 4 channels of one output voxel
__global__ void __launch_bounds__(256)
costconv_assemble_fwd_kernel(float4 *__restrict__ out, const float4 *__restrict__ Fb,
                             const float4 *__restrict__ Fe, const float4 *__restrict__ G, int D, int H, int W,
                             int XE, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cq = (int)(i & 7);
        long long v = i >> 3;
        const int x = (int)(v % W); v /= W;
        const int y = (int)(v % H); v /= H;
        const int d = (int)(v % D);
        const long long b = v / D;
        const int delta = x - d;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (delta >= -2) {
            const int cls = cc_class(d, D), nc8 = cc_nclass(D) * 8;  // float4 per 32 channels = 8
            const float4 f = (delta >= 2)
                ? Fb[(((b * H + y) * W + x) * nc8) + cls * 8 + cq]
                : Fe[(((b * H + y) * XE + x) * (nc8 * 4)) + (cls * 4 + (delta + 2)) * 8 + cq];  // x = d + delta < XE
            const int xb = (x == W - 1) ? 1 : 0;
            const float4 g = G[(((b * H + y) * (W + 2) + (delta + 2)) * (nc8 * 2)) + (cls * 2 + xb) * 8 + cq];
            o = make_float4(f.x + g.x, f.y + g.y, f.z + g.z, f.w + g.w);
        }
        out[i] = o;
    }
}

// dF[b,y,x,(cls,dl),:]: dl = 4 (delta >= 2): sum over d <= x-2 of class cls; dl < 4: the single plane d = x - (dl-2)
__global__ void __launch_bounds__(256)
costconv_grad_f_kernel(float4 *__restrict__ dFb, float4 *__restrict__ dFe, const float4 *__restrict__ gy,
                       int D, int H, int W, int XE, long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cq = (int)(i & 7);
        long long v = i >> 3;
        const int x = (int)(v % W); v /= W;
        const int y = (int)(v % H);
        const long long b = v / H;
        const size_t plane = (size_t)H * W * 8;
        const float4 *g0 = gy + (size_t)b * D * plane + ((size_t)y * W + x) * 8 + cq;
        const int nc = cc_nclass(D);
        float4 bulk[CC_MAXCLS];
#pragma unroll
        for (int c = 0; c < CC_MAXCLS; ++c) bulk[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int dmax = min(D - 1, x - 2);
        for (int d = 0; d <= dmax; ++d) {
            const float4 g = g0[(size_t)d * plane];
            const int cls = cc_class(d, D);
#pragma unroll
            for (int c = 0; c < CC_MAXCLS; ++c)
                if (c == cls) { bulk[c].x += g.x; bulk[c].y += g.y; bulk[c].z += g.z; bulk[c].w += g.w; }
        }
        float4 *ob = dFb + (((size_t)b * H + y) * W + x) * (nc * 8) + cq;
#pragma unroll
        for (int c = 0; c < CC_MAXCLS; ++c)
            if (c < nc) ob[c * 8] = bulk[c];
        if (x < XE) {
            float4 *oe = dFe + (((size_t)b * H + y) * XE + x) * (nc * 32) + cq;
#pragma unroll
            for (int c = 0; c < CC_MAXCLS; ++c)
#pragma unroll
                for (int dl = 0; dl < 4; ++dl) {  // delta = dl - 2 in {-2,-1,0,1}: plane d = x - delta
                    const int d = x - (dl - 2);
                    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (d >= 0 && d < D && cc_class(d, D) == c) g = g0[(size_t)d * plane];
                    if (c < nc) oe[(c * 4 + dl) * 8] = g;
                }
        }
    }
}

// dG[b,y,t,(cls,xb),:], t = u + 2, u = x - d in [-2, W): sum over d of grad_out[b,d,y,u+d,:]
__global__ void __launch_bounds__(256)
costconv_grad_g_kernel(float4 *__restrict__ dG, const float4 *__restrict__ gy, int D, int H, int W,
                       long long total) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cq = (int)(i & 7);
        long long v = i >> 3;
        const int t = (int)(v % (W + 2)); v /= (W + 2);
        const int y = (int)(v % H);
        const long long b = v / H;
        const int u = t - 2;
        const size_t plane = (size_t)H * W * 8;
        const float4 *g0 = gy + (size_t)b * D * plane + (size_t)y * W * 8 + cq;
        const int nc = cc_nclass(D);
        float4 acc[CC_MAXCLS][2];
#pragma unroll
        for (int c = 0; c < CC_MAXCLS; ++c) { acc[c][0] = make_float4(0.f, 0.f, 0.f, 0.f); acc[c][1] = acc[c][0]; }
        const int d_lo = max(0, -u), d_hi = min(D - 1, W - 1 - u);  // 0 <= x = u + d <= W-1
        for (int d = d_lo; d <= d_hi; ++d) {
            const int x = u + d;
            const float4 g = g0[(size_t)d * plane + (size_t)x * 8];
            const int cls = cc_class(d, D), xb = (x == W - 1) ? 1 : 0;
#pragma unroll
            for (int c = 0; c < CC_MAXCLS; ++c)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    if (c == cls && e == xb) { acc[c][e].x += g.x; acc[c][e].y += g.y; acc[c][e].z += g.z; acc[c][e].w += g.w; }
        }
        float4 *o = dG + (((size_t)b * H + y) * (W + 2) + t) * (nc * 16) + cq;
#pragma unroll
        for (int c = 0; c < CC_MAXCLS; ++c)
            if (c < nc) { o[(c * 2 + 0) * 8] = acc[c][0]; o[(c * 2 + 1) * 8] = acc[c][1]; }
    }
}

extern "C" int az_costconv_edge_width(int D, int W) { return D + 1 < W ? D + 1 : W; }
extern "C" int az_costconv_num_classes(int D) { return D > 0 ? cc_nclass(D) : AZ_EINVAL; }

extern "C" int az_costconv_assemble_fwd(float *out, const float *F_bulk, const float *F_edge, const float *G, int B,
                                        int D, int H, int W, void *stream) {
    AZ_REQUIRE_PTR(out); AZ_REQUIRE_PTR(F_bulk); AZ_REQUIRE_PTR(F_edge); AZ_REQUIRE_PTR(G);
    AZ_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0);
    const long long total = (long long)B * D * H * W * 8;
    hipLaunchKernelGGL(costconv_assemble_fwd_kernel, dim3(az_grid_for(total, 256)), dim3(256), 0, az_stream(stream),
                       (float4 *)out, (const float4 *)F_bulk, (const float4 *)F_edge, (const float4 *)G, D, H, W,
                       az_costconv_edge_width(D, W), total);
    return az_launch_status();
}

extern "C" int az_costconv_assemble_bwd(float *dF_bulk, float *dF_edge, float *dG, const float *grad_out, int B,
                                        int D, int H, int W, void *stream) {
    AZ_REQUIRE_PTR(dF_bulk); AZ_REQUIRE_PTR(dF_edge); AZ_REQUIRE_PTR(dG); AZ_REQUIRE_PTR(grad_out);
    AZ_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0);
    const long long tf = (long long)B * H * W * 8, tg = (long long)B * H * (W + 2) * 8;
    hipLaunchKernelGGL(costconv_grad_f_kernel, dim3(az_grid_for(tf, 256)), dim3(256), 0, az_stream(stream),
                       (float4 *)dF_bulk, (float4 *)dF_edge, (const float4 *)grad_out, D, H, W,
                       az_costconv_edge_width(D, W), tf);
    hipLaunchKernelGGL(costconv_grad_g_kernel, dim3(az_grid_for(tg, 256)), dim3(256), 0, az_stream(stream),
                       (float4 *)dG, (const float4 *)grad_out, D, H, W, tg);
    return az_launch_status();
}

// ---- merged kernels (costconv.py: K_L, K_R) as own kernels ------------------------------------------------------------
// The depth-summed kernels are 0/1-masked sums of the Conv3d weight over kd (and, for K_R, over kw into the shifted column
// j): weight-space einsums of 55 K elements that rounds 1-4 left to torch.einsum -- two tiny rocBLAS GEMMs forward and two
// backward, the last vendor-library kernels of the step.
//   J = 0:  out[c][e][o][i][h][w] = sum_d      W[o][i][d][h][w] * M[c][e][d][w]          (K_L;  M = ML [nc][ne][3][3])
//   J = 5:  out[c][e][o][i][h][j] = sum_d,w    W[o][i][d][h][w] * M[c][e][d][w][j]       (K_R;  M = MR [nc][ne][3][3][5])
// W: 32 x 32 channel slice of the [32][64][3][3][3] weight (so = 64 * 27 floats between o, 27 between i); the adjoint gives
// the gradient of that slice (every element written).
template <int J>
__global__ void __launch_bounds__(256)
costconv_merge_fwd_kernel(float *__restrict__ out, const float *__restrict__ w, const float *__restrict__ m, int nce, long long so) {
    constexpr int L = J ? J : 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = nce * 32 * 32 * 3 * L;
    if (idx >= total) return;
    const int l = idx % L;
    int r = idx / L;
    const int h = r % 3; r /= 3;
    const int i = r % 32; r /= 32;
    const int o = r % 32;
    const int ce = r / 32;
    const float *wp = w + o * so + i * 27 + h * 3;
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (J) {
#pragma unroll
            for (int x = 0; x < 3; ++x) acc += wp[d * 9 + x] * m[((ce * 3 + d) * 3 + x) * J + l];
        } else {
            acc += wp[d * 9 + l] * m[(ce * 3 + d) * 3 + l];
        }
    }
    out[idx] = acc;
}

template <int J>
__global__ void __launch_bounds__(256)
costconv_merge_bwd_kernel(float *__restrict__ gw, const float *__restrict__ gout, const float *__restrict__ m, int nce, long long so) {
    constexpr int L = J ? J : 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;  // over [o][i][d][h][x] of the slice
    if (idx >= 32 * 32 * 27) return;
    const int x = idx % 3;
    int r = idx / 3;
    const int h = r % 3; r /= 3;
    const int d = r % 3; r /= 3;
    const int i = r % 32;
    const int o = r / 32;
    float acc = 0.f;
    for (int ce = 0; ce < nce; ++ce) {
        const float *g = gout + ((((size_t)ce * 32 + o) * 32 + i) * 3 + h) * L;
        if (J) {
#pragma unroll
            for (int l = 0; l < J; ++l) acc += g[l] * m[((ce * 3 + d) * 3 + x) * J + l];
        } else {
            acc += g[x] * m[(ce * 3 + d) * 3 + x];
        }
    }
    gw[o * so + i * 27 + d * 9 + h * 3 + x] = acc;
}

// kl [nc][5][32][32][3][3], kr [nc][2][32][32][3][5] from weight [32][64][3][3][3]; ml [nc][5][3][3], mr [nc][2][3][3][5]
extern "C" int az_costconv_merge_fwd(float *kl, float *kr, const float *weight, const float *ml, const float *mr, int ncls,
                                     void *stream) {
    AZ_REQUIRE_PTR(kl); AZ_REQUIRE_PTR(kr); AZ_REQUIRE_PTR(weight); AZ_REQUIRE_PTR(ml); AZ_REQUIRE_PTR(mr);
    AZ_REQUIRE(ncls >= 1 && ncls <= 3);
    hipStream_t s = az_stream(stream);
    const int tl = ncls * 5 * 32 * 32 * 9, tr = ncls * 2 * 32 * 32 * 15;
    hipLaunchKernelGGL(costconv_merge_fwd_kernel<0>, dim3((tl + 255) / 256), dim3(256), 0, s, kl, weight, ml, ncls * 5, 64LL * 27);
    hipLaunchKernelGGL(costconv_merge_fwd_kernel<5>, dim3((tr + 255) / 256), dim3(256), 0, s, kr, weight + 32 * 27, mr, ncls * 2, 64LL * 27);
    return az_launch_status();
}

// grad_weight [32][64][3][3][3] (fully written) from the gradients of kl and kr
extern "C" int az_costconv_merge_bwd(float *grad_weight, const float *gkl, const float *gkr, const float *ml, const float *mr,
                                     int ncls, void *stream) {
    AZ_REQUIRE_PTR(grad_weight); AZ_REQUIRE_PTR(gkl); AZ_REQUIRE_PTR(gkr); AZ_REQUIRE_PTR(ml); AZ_REQUIRE_PTR(mr);
    AZ_REQUIRE(ncls >= 1 && ncls <= 3);
    hipStream_t s = az_stream(stream);
    const int t = 32 * 32 * 27;
    hipLaunchKernelGGL(costconv_merge_bwd_kernel<0>, dim3((t + 255) / 256), dim3(256), 0, s, grad_weight, gkl, ml, ncls * 5, 64LL * 27);
    hipLaunchKernelGGL(costconv_merge_bwd_kernel<5>, dim3((t + 255) / 256), dim3(256), 0, s, grad_weight + 32 * 27, gkr, mr, ncls * 2, 64LL * 27);
    return az_launch_status();
}
