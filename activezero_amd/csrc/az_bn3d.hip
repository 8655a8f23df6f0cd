// BatchNorm3d around the MFMA convolutions (reference
// nets/psmnet/psmnet_submodule_3.py:44-56: Conv3d -> BatchNorm3d, train-mode batch
// statistics per GPU, no SyncBN; psmnet_3.py adds ReLU / residual sums after it).
//
// Train forward:  conv epilogue writes raw x and per-tile (count, sum, centred M2)
//   -> az_bn3d_finalize merges the partials per channel with Chan's parallel
//      formula in fp64 (deterministic, no cancellation), updates the running stats
//      and emits scale = gamma*invstd, shift = beta - mean*scale
//   -> az_bn3d_apply: y = relu?(x*scale + shift (+residual)), one HBM pass.
// Eval forward: az_bn3d_eval_affine folds running stats into (scale, shift), which
//   the conv epilogue applies directly (no extra pass).
// Train backward (dy -> dx_raw, dgamma, dbeta, dresidual):
//   az_bn3d_bwd_reduce: per-block partials of sum(dz), sum(dz*xhat), dz = dy*[y>0]
//   az_bn3d_bwd_finalize: fp64 merge -> dgamma, dbeta, and the two per-channel
//      coefficients of the apply pass
//   az_bn3d_bwd_apply: dx = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)); optional dz out.
#include <stdlib.h>

#include "az_roll_common.h"
#include "az_options.h"

// Block-wide fp64 sum in two barriers: DPP/shuffle inside each wave, one LDS slot per wave, then every thread
// adds the (at most 16) wave sums.  The finalize kernels below are latency bound -- one block per channel, a
// few thousand partials -- and the 10-step LDS tree they used cost 10 barriers per reduction (18 us per launch,
// 170 launches per step).
__device__ __forceinline__ double bn_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int NT>
__device__ __forceinline__ double bn_block_sum(double v, double *slots /* [NT/64] */) {
    v = bn_wave_sum(v);
    __syncthreads();  // the slots may still be read from a previous call
    if ((threadIdx.x & 63) == 0) slots[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += slots[w];
    return t;
}

// First stage for layers with very many partials (V0: 97 920 tiles per channel = 25 MB over 32 channels, which 32
// blocks on 32 CUs pulled in 105 us): grid (C, S), block (c, s) merges slice s of channel c's partials into ONE
// partial of the same format -- (sum, M2 about the slice mean) and the slice's count -- which bn_finalize_kernel
// then merges as usual (Chan's formula is associative).
#define BN_PRE_SLICES 32
__global__ void __launch_bounds__(256)
bn_premerge_kernel(float *__restrict__ part_out, float *__restrict__ cnt_out, const float *__restrict__ part,
                   const float *__restrict__ cnt, long long ntiles) {
    const int c = blockIdx.x, sl = blockIdx.y;
    const long long per = (ntiles + BN_PRE_SLICES - 1) / BN_PRE_SLICES;
    const long long t0 = sl * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    __shared__ double slots[256 / 64];
    const float2 *pc = reinterpret_cast<const float2 *>(part) + (long long)c * ntiles;
    double n = 0.0, sum = 0.0;
    for (long long t = t0 + threadIdx.x; t < t1; t += 256) {
        n += (double)cnt[t];
        sum += (double)pc[t].x;
    }
    const double N = bn_block_sum<256>(n, slots);
    const double S = bn_block_sum<256>(sum, slots);
    const double mean = N > 0.0 ? S / N : 0.0;
    double m2 = 0.0;
    for (long long t = t0 + threadIdx.x; t < t1; t += 256) {
        const float nt = cnt[t];
        if (nt <= 0.f) continue;
        const float2 pr = pc[t];
        const double dlt = (double)pr.x / (double)nt - mean;
        m2 += (double)pr.y + (double)nt * dlt * dlt;
    }
    const double M2 = bn_block_sum<256>(m2, slots);
    if (threadIdx.x == 0) {
        reinterpret_cast<float2 *>(part_out)[c * BN_PRE_SLICES + sl] = make_float2((float)S, (float)M2);
        if (c == 0) cnt_out[sl] = (float)N;
    }
}

// one block per channel; 1024 threads: the V0 layers merge 97 920 tile partials per channel and a
// 256-thread block took 130 us per layer (25 layers per step) on two dependent fp64 passes
#define BN_FIN_THREADS 1024
__global__ void __launch_bounds__(BN_FIN_THREADS)
bn_finalize_kernel(float *__restrict__ mean_out, float *__restrict__ invstd_out,
                   float *__restrict__ scale, float *__restrict__ shift,
                   float *__restrict__ running_mean, float *__restrict__ running_var,
                   const float *__restrict__ part, const float *__restrict__ cnt,
                   const float *__restrict__ gamma, const float *__restrict__ beta,
                   long long ntiles, int C, float eps, float momentum, int groups,
                   long long *__restrict__ num_batches_tracked) {
    const int c = blockIdx.x;
    // BatchNorm's call counter (nn.BatchNorm*.num_batches_tracked, one count per statistic group): bumped here
    // instead of by a separate elementwise launch per layer (85 per step)
    if (num_batches_tracked != nullptr && c == 0 && threadIdx.x == 0) *num_batches_tracked += groups;
    // statistic groups (bn2d: the left and the right image set) are finalised one after the other
    // by the same block, so the running statistics see them in order, as two module calls would
    for (int grp = 0; grp < groups; ++grp, part += (long long)C * ntiles * 2, cnt += ntiles,
             mean_out += C, invstd_out += C, scale += C, shift += C) {
    // pass 1: N = sum n_t, S = sum s_t  ->  mean
    // pass 2: M2 = sum [ M2_t + n_t (s_t/n_t - mean)^2 ]   (Chan's merge with the final mean)
    __shared__ double slots[BN_FIN_THREADS / 64];
    const float2 *pc = reinterpret_cast<const float2 *>(part) + (long long)c * ntiles;
    double n = 0.0, sum = 0.0;
    for (long long t = threadIdx.x; t < ntiles; t += BN_FIN_THREADS) {
        n += (double)cnt[t];
        sum += (double)pc[t].x;
    }
    const double Ntot = bn_block_sum<BN_FIN_THREADS>(n, slots);
    const double mean_all = bn_block_sum<BN_FIN_THREADS>(sum, slots) / Ntot;
    double m2 = 0.0;
    for (long long t = threadIdx.x; t < ntiles; t += BN_FIN_THREADS) {
        const float nt = cnt[t];
        if (nt <= 0.f) continue;
        const float2 pr = pc[t];
        const double dlt = (double)pr.x / (double)nt - mean_all;
        m2 += (double)pr.y + (double)nt * dlt * dlt;
    }
    const double M2 = bn_block_sum<BN_FIN_THREADS>(m2, slots);
    if (threadIdx.x == 0) {
        const double N = Ntot, mu = mean_all, var = M2 / N;
        const double istd = 1.0 / sqrt(var + (double)eps);
        mean_out[c] = (float)mu;
        invstd_out[c] = (float)istd;
        const float sc = gamma[c] * (float)istd;
        scale[c] = sc;
        shift[c] = beta[c] - (float)mu * sc;
        if (running_mean) {
            const double unbiased = N > 1.0 ? M2 / (N - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
    }  // groups
}

__global__ void bn_eval_affine_kernel(float *scale, float *shift, const float *gamma,
                                      const float *beta, const float *rm, const float *rv,
                                      float eps, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

// Streaming access for tensors larger than the 256 MB Infinity Cache (the V0 volumes are 802 MB): nontemporal
// loads and stores keep lines that will not be reused out of L2 / MALL (measured on bn_apply over a V0 tensor:
// 5.32 -> 5.71 TB/s; torch's copy kernel on the same box 5.55).  Smaller tensors keep the default policy: their
// consumer finds them in the cache.
typedef float az_v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 bn_ld4(const float4 *p) {
    if (NT) {
        const az_v4f v = __builtin_nontemporal_load(reinterpret_cast<const az_v4f *>(p));
        return make_float4(v[0], v[1], v[2], v[3]);
    }
    return *p;
}
template <bool NT>
__device__ __forceinline__ void bn_st4(float4 *p, const float4 &o) {
    if (NT) {
        const az_v4f v = {o.x, o.y, o.z, o.w};
        __builtin_nontemporal_store(v, reinterpret_cast<az_v4f *>(p));
    } else {
        *p = o;
    }
}
#define BN_NT_BYTES (256LL << 20)

template <int C, bool NT>
__global__ void __launch_bounds__(256)
bn_apply_kernel(float4 *__restrict__ y, const float4 *__restrict__ x,
                const float *__restrict__ scale, const float *__restrict__ shift,
                const float4 *__restrict__ res, int relu, long long total4, unsigned *__restrict__ amax = nullptr) {
    constexpr int C4 = C / 4;
    unsigned am = 0;
    __shared__ float4 ssc[C4], ssh[C4];
    {   // blockIdx.y = statistic group: its own slice of the tensors and its own (scale, shift)
        const long long go = (long long)blockIdx.y * total4;
        y += go; x += go; if (res) res += go;
        scale += blockIdx.y * C; shift += blockIdx.y * C;
    }
    if (threadIdx.x < C4) {
        ssc[threadIdx.x] = reinterpret_cast<const float4 *>(scale)[threadIdx.x];
        ssh[threadIdx.x] = reinterpret_cast<const float4 *>(shift)[threadIdx.x];
    }
    __syncthreads();
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += gridDim.x * 256LL) {
        const int c4 = (int)(i % C4);
        const float4 v = bn_ld4<NT>(&x[i]), sc = ssc[c4], sh = ssh[c4];
        // (fmaf spelled out: the backward kernels recompute this value for the ReLU mask and must
        //  round identically)
        float4 o = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z),
                               fmaf(v.w, sc.w, sh.w));
        if (res) {
            const float4 r = bn_ld4<NT>(&res[i]);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        bn_st4<NT>(&y[i], o);
        az_amax_acc(am, o);
    }
    if (amax) az_amax_flush(amax, am);
}

__device__ __forceinline__ int c4_of(int tid, int C4) { return tid % C4; }

// ---- backward ---------------------------------------------------------------------------
// partial[block][C][2]: sum(dz), sum(dz * xhat) over the block's voxels
template <int C, bool NT>
__global__ void __launch_bounds__(256)
bn_bwd_reduce_kernel(float *__restrict__ partial, const float *__restrict__ dy,
                     const float *__restrict__ y, const float *__restrict__ x,
                     const float *__restrict__ mean, const float *__restrict__ invstd,
                     const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                     long long nvox, unsigned *__restrict__ amax_zero = nullptr, unsigned *__restrict__ pmax = nullptr) {
    // (the apply kernel that follows in the stream takes max |dx| into this word with atomicMax: start it at zero)
    if (amax_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < AZ_AMAX_SLOTS) amax_zero[threadIdx.x * AZ_AMAX_STRIDE] = 0u;
    // thread t owns channel quad (t % C4) and voxel lane (t / C4); C4 divides 256
    constexpr int C4 = C / 4, VPB = 256 / C4;
    // ReLU mask: from the saved output y, or -- when the forward's (scale, shift) are given and the
    // layer had no residual -- recomputed from x with the apply kernel's own expression
    // (x*scale + shift > 0), which saves reading y (one tensor pass here and one in the apply kernel)
    const bool remask = relu && scale != nullptr;
    {   // blockIdx.y = statistic group
        const long long go = (long long)blockIdx.y * nvox * C;
        dy += go; x += go; if (relu && !remask) y += go;
        mean += blockIdx.y * C; invstd += blockIdx.y * C;
        if (remask) { scale += blockIdx.y * C; shift += blockIdx.y * C; }
        partial += (size_t)blockIdx.y * gridDim.x * C * 2;
    }
#if defined(BN_ABL) && (BN_ABL & 4)
    // timing-only build (-DBN_ABL=4, run with AZ_PRESPLIT=0; tools/abl_bn_reduce.sh): the pass reads nothing and reports zero
    // sums, so that the apply pass writes finite gradients of the usual size (dx = k0 dz) and every later kernel works on
    // ordinary numbers -- what the step would gain if this pass were free (profiles/r05y_bn_reduce_ablation.txt: 4.4 ms)
    if (threadIdx.x < 2 * C) partial[(size_t)blockIdx.x * C * 2 + threadIdx.x] = 0.f;
    return;
#endif
    float4 sc = make_float4(0, 0, 0, 0), sh = sc;
    if (remask) { sc = reinterpret_cast<const float4 *>(scale)[c4_of(threadIdx.x, C4)]; sh = reinterpret_cast<const float4 *>(shift)[c4_of(threadIdx.x, C4)]; }
    const int c4 = threadIdx.x % C4, vl = threadIdx.x / C4;
    const float4 mu = reinterpret_cast<const float4 *>(mean)[c4];
    const float4 is = reinterpret_cast<const float4 *>(invstd)[c4];
    float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
    // `pmax` given (the apply pass will write dx PRE-SPLIT, az_roll_common.h): also the largest finite |dz| and |xhat| per
    // channel, as bit patterns -- what the apply pass bounds max |dx| with before it writes its first element
    uint4 mg = make_uint4(0, 0, 0, 0), mx = make_uint4(0, 0, 0, 0);
    auto body = [&](const float4 gin, const float4 xx, const float4 yin, float4 &t1, float4 &t2) {
        float4 g = gin;
        if (relu) {
            float4 yy = yin;
            if (remask) yy = make_float4(fmaf(xx.x, sc.x, sh.x), fmaf(xx.y, sc.y, sh.y), fmaf(xx.z, sc.z, sh.z), fmaf(xx.w, sc.w, sh.w));
            g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
            g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        }
        const float4 xh = make_float4((xx.x - mu.x) * is.x, (xx.y - mu.y) * is.y, (xx.z - mu.z) * is.z, (xx.w - mu.w) * is.w);
        t1.x += g.x; t1.y += g.y; t1.z += g.z; t1.w += g.w;
        t2.x += g.x * xh.x; t2.y += g.y * xh.y; t2.z += g.z * xh.z; t2.w += g.w * xh.w;
        if (pmax) {
            mg.x = max(mg.x, az_finite_abs_bits(g.x)); mg.y = max(mg.y, az_finite_abs_bits(g.y));
            mg.z = max(mg.z, az_finite_abs_bits(g.z)); mg.w = max(mg.w, az_finite_abs_bits(g.w));
            mx.x = max(mx.x, az_finite_abs_bits(xh.x)); mx.y = max(mx.y, az_finite_abs_bits(xh.y));
            mx.z = max(mx.z, az_finite_abs_bits(xh.z)); mx.w = max(mx.w, az_finite_abs_bits(xh.w));
        }
    };
    // two voxels per thread in flight (the grid is capped so that the partials stay L2-sized: bn_bwd_apply_kernel)
    const long long stride = (long long)gridDim.x * VPB;
    float4 u1 = make_float4(0, 0, 0, 0), u2 = make_float4(0, 0, 0, 0);
    const float4 zero4 = make_float4(0, 0, 0, 0);
    long long v = (long long)blockIdx.x * VPB + vl;
    for (; v + stride < nvox; v += 2 * stride) {
        const size_t i = (size_t)v * C4 + c4, i2 = (size_t)(v + stride) * C4 + c4;
        const float4 ga = bn_ld4<NT>(reinterpret_cast<const float4 *>(dy) + i), gb = bn_ld4<NT>(reinterpret_cast<const float4 *>(dy) + i2);
        const float4 xa = bn_ld4<NT>(reinterpret_cast<const float4 *>(x) + i), xb = bn_ld4<NT>(reinterpret_cast<const float4 *>(x) + i2);
        float4 ya = zero4, yb = zero4;
        if (relu && !remask) { ya = bn_ld4<NT>(reinterpret_cast<const float4 *>(y) + i); yb = bn_ld4<NT>(reinterpret_cast<const float4 *>(y) + i2); }
        body(ga, xa, ya, s1, s2);
        body(gb, xb, yb, u1, u2);
    }
    if (v < nvox) {
        const size_t i = (size_t)v * C4 + c4;
        const float4 ga = bn_ld4<NT>(reinterpret_cast<const float4 *>(dy) + i), xa = bn_ld4<NT>(reinterpret_cast<const float4 *>(x) + i);
        float4 ya = zero4;
        if (relu && !remask) ya = bn_ld4<NT>(reinterpret_cast<const float4 *>(y) + i);
        body(ga, xa, ya, s1, s2);
    }
    s1.x += u1.x; s1.y += u1.y; s1.z += u1.z; s1.w += u1.w;
    s2.x += u2.x; s2.y += u2.y; s2.z += u2.z; s2.w += u2.w;
    // block sum per channel quad: lanes with equal (lane % C4) through cross-lane shuffles, then the four waves through
    // 2 x 4 x C4 float4 of LDS.  (The first version staged all 256 threads' sums: 8 KB per block -- which does not fit
    // beside two workgroups of the 69-79 KB matrix kernels of the other stream, so this kernel, first in line after
    // every input-gradient convolution, waited for them to drain: 0.91 ms in the step against 0.27 ms alone.)
    static_assert(C4 <= 64 && 64 % C4 == 0, "channel quads must tile a wave");
#pragma unroll
    for (int off = C4; off < 64; off <<= 1) {
        s1.x += __shfl_xor(s1.x, off); s1.y += __shfl_xor(s1.y, off); s1.z += __shfl_xor(s1.z, off); s1.w += __shfl_xor(s1.w, off);
        s2.x += __shfl_xor(s2.x, off); s2.y += __shfl_xor(s2.y, off); s2.z += __shfl_xor(s2.z, off); s2.w += __shfl_xor(s2.w, off);
    }
    __shared__ float4 r1[4][C4], r2[4][C4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < C4) { r1[wave][lane] = s1; r2[wave][lane] = s2; }
    __syncthreads();
    if (threadIdx.x < C4) {
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const float4 a = r1[k][c4], b = r2[k][c4];
            s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
            s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
        }
        float *p = partial + ((size_t)blockIdx.x * C + c4 * 4) * 2;
        p[0] = s1.x; p[1] = s2.x; p[2] = s1.y; p[3] = s2.y;
        p[4] = s1.z; p[5] = s2.z; p[6] = s1.w; p[7] = s2.w;
    }
    if (pmax) {  // the same reduction with max, rows laid out like the sums' ([block][C][2]: max |dz|, max |xhat|)
#pragma unroll
        for (int off = C4; off < 64; off <<= 1) {
            mg.x = max(mg.x, (unsigned)__shfl_xor((int)mg.x, off)); mg.y = max(mg.y, (unsigned)__shfl_xor((int)mg.y, off));
            mg.z = max(mg.z, (unsigned)__shfl_xor((int)mg.z, off)); mg.w = max(mg.w, (unsigned)__shfl_xor((int)mg.w, off));
            mx.x = max(mx.x, (unsigned)__shfl_xor((int)mx.x, off)); mx.y = max(mx.y, (unsigned)__shfl_xor((int)mx.y, off));
            mx.z = max(mx.z, (unsigned)__shfl_xor((int)mx.z, off)); mx.w = max(mx.w, (unsigned)__shfl_xor((int)mx.w, off));
        }
        __shared__ uint4 q1[4][C4], q2[4][C4];
        if (lane < C4) { q1[wave][lane] = mg; q2[wave][lane] = mx; }
        __syncthreads();
        if (threadIdx.x < C4) {
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const uint4 a = q1[k][c4], b = q2[k][c4];
                mg.x = max(mg.x, a.x); mg.y = max(mg.y, a.y); mg.z = max(mg.z, a.z); mg.w = max(mg.w, a.w);
                mx.x = max(mx.x, b.x); mx.y = max(mx.y, b.y); mx.z = max(mx.z, b.z); mx.w = max(mx.w, b.w);
            }
            unsigned *p = pmax + ((size_t)blockIdx.y * gridDim.x * C + (size_t)blockIdx.x * C + c4 * 4) * 2;
            p[0] = mg.x; p[1] = mx.x; p[2] = mg.y; p[3] = mx.y;
            p[4] = mg.z; p[5] = mx.z; p[6] = mg.w; p[7] = mx.w;
        }
    }
}

__global__ void __launch_bounds__(256)
bn_bwd_finalize_kernel(float *__restrict__ dgamma, float *__restrict__ dbeta,
                       float *__restrict__ coef, const float *__restrict__ partial,
                       const float *__restrict__ gamma, const float *__restrict__ invstd,
                       int nblocks, int C, double nvox, int groups) {
    const int c = blockIdx.x;
    __shared__ double slots[256 / 64];
    double dg_tot = 0.0, db_tot = 0.0;  // parameter gradients: summed over the statistic groups
    for (int grp = 0; grp < groups; ++grp, partial += (size_t)nblocks * C * 2, invstd += C, coef += C * 3) {
        double a = 0.0, b = 0.0;
        for (int t = threadIdx.x; t < nblocks; t += 256) {
            a += partial[((size_t)t * C + c) * 2 + 0];
            b += partial[((size_t)t * C + c) * 2 + 1];
        }
        const double sa = bn_block_sum<256>(a, slots), sb = bn_block_sum<256>(b, slots);
        if (threadIdx.x == 0) {
            db_tot += sa;
            dg_tot += sb;
            // dx = k0 * (dz - k1 - xhat * k2)
            coef[c * 3 + 0] = gamma[c] * invstd[c];
            coef[c * 3 + 1] = (float)(sa / nvox);
            coef[c * 3 + 2] = (float)(sb / nvox);
        }
    }
    if (threadIdx.x == 0) {
        dbeta[c] = (float)db_tot;
        dgamma[c] = (float)dg_tot;
    }
}

// `partial` given: the finalize step runs HERE, in every block's prologue, instead of in a kernel of its own -- the
// reduce kernel then writes at most BN_BWD_PARTIAL_FLOATS floats of partials per statistic group (128 KB, L2-resident),
// each block merges them (fp64 per thread, cross-lane, four waves through 2C floats of LDS) and block (0, g) also writes
// coef / dgamma / dbeta.  One launch less per BatchNorm backward (85 per step), and the one that went away was a
// 32-block kernel that, beside the other stream's matrix kernels, waited ~60 us for a free wave slot each time
// (5 ms per step on the main stream, profiles/r03a_bench_b4_summary.md).
#define BN_BWD_PARTIAL_FLOATS 32768
template <int C, bool NT>
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(float4 *__restrict__ dx, float4 *__restrict__ dz_out,
                    const float4 *__restrict__ dy, const float4 *__restrict__ y,
                    const float4 *__restrict__ x, const float *__restrict__ mean,
                    const float *__restrict__ invstd, float *__restrict__ coef,
                    const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                    long long total4, const float *__restrict__ partial, int nblocks, const float *__restrict__ gamma,
                    float *__restrict__ dgamma, float *__restrict__ dbeta, double nvox,
                    unsigned *__restrict__ amax = nullptr, const unsigned *__restrict__ pmax = nullptr) {
    constexpr int C4 = C / 4;
    unsigned am = 0;  // max |dx| of this thread, as a bit pattern (az_absmax.hip): the f16x3 scale of dx's consumers
    // `pmax` given: dx is written PRE-SPLIT (az_roll_common.h) -- the two fp16 parts of dx * 2^k in place of the float.  k
    // must be known before the first element is written: from  |dx| <= |k0| (max |dz| + |k1| + max |xhat| |k2|)  per channel,
    // the reduce pass's maxima merged like its sums; the bound (not the true maximum) is what `amax` receives, so that the
    // consumers derive the same k.  Every block computes the same value from the same partials.
    float split_scale = 1.f;
    __shared__ float smu[C], sis[C], k0[C], k1[C], k2[C], ssc[C], ssh[C];
    const bool remask = relu && scale != nullptr;  // see bn_bwd_reduce_kernel
    const int grp = blockIdx.y;
    {   // blockIdx.y = statistic group
        const long long go = (long long)grp * total4;
        dx += go; dy += go; x += go; if (relu && !remask) y += go; if (dz_out) dz_out += go;
        mean += grp * C; invstd += grp * C; coef += grp * C * 3;
        if (remask) { scale += grp * C; shift += grp * C; }
    }
    if (remask && threadIdx.x < C) { ssc[threadIdx.x] = scale[threadIdx.x]; ssh[threadIdx.x] = shift[threadIdx.x]; }
    if (threadIdx.x < C) {
        smu[threadIdx.x] = mean[threadIdx.x];
        sis[threadIdx.x] = invstd[threadIdx.x];
    }
    if (partial == nullptr) {
        if (threadIdx.x < C) {
            k0[threadIdx.x] = coef[threadIdx.x * 3 + 0];
            k1[threadIdx.x] = coef[threadIdx.x * 3 + 1];
            k2[threadIdx.x] = coef[threadIdx.x * 3 + 2];
        }
    } else {
        // one partial row = 2C floats = COLS float4; thread (r, j) sums column j of rows r, r + ROWS, ...
        constexpr int COLS = C / 2, ROWS = 256 / COLS;
        static_assert(COLS <= 64 && 64 % COLS == 0, "a partial row must tile a wave");
        __shared__ float wsum[4][2 * C];
        const int j = threadIdx.x % COLS, r = threadIdx.x / COLS, wave = threadIdx.x >> 6;
        const bool writer = blockIdx.x == 0;
        // block (0, 0) also needs the other groups' sums (dgamma / dbeta are summed over the groups)
        const int g_lo = (writer && grp == 0) ? 0 : grp, g_hi = (writer && grp == 0) ? (int)gridDim.y : grp + 1;
        double dg_tot = 0.0, db_tot = 0.0;
        for (int g = g_lo; g < g_hi; ++g) {
            const float4 *p4 = reinterpret_cast<const float4 *>(partial) + (size_t)g * nblocks * COLS;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            for (int blk = r; blk < nblocks; blk += ROWS) {
                const float4 v = p4[(size_t)blk * COLS + j];
                a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
            }
#pragma unroll
            for (int off = COLS; off < 64; off <<= 1) {
                a0 += __shfl_xor(a0, off); a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off); a3 += __shfl_xor(a3, off);
            }
            __syncthreads();  // wsum may still be read (previous group)
            if ((threadIdx.x & 63) < COLS) {
                wsum[wave][4 * j + 0] = (float)a0; wsum[wave][4 * j + 1] = (float)a1;
                wsum[wave][4 * j + 2] = (float)a2; wsum[wave][4 * j + 3] = (float)a3;
            }
            __syncthreads();
            if (threadIdx.x < C) {
                const int c = threadIdx.x;
                const double sa = ((double)wsum[0][2 * c] + (double)wsum[1][2 * c]) + ((double)wsum[2][2 * c] + (double)wsum[3][2 * c]);
                const double sb = ((double)wsum[0][2 * c + 1] + (double)wsum[1][2 * c + 1]) + ((double)wsum[2][2 * c + 1] + (double)wsum[3][2 * c + 1]);
                db_tot += sa; dg_tot += sb;
                if (g == grp) {  // dx = k0 * (dz - k1 - xhat * k2)
                    k0[c] = gamma[c] * sis[c];
                    k1[c] = (float)(sa / nvox);
                    k2[c] = (float)(sb / nvox);
                    if (writer) { coef[c * 3 + 0] = k0[c]; coef[c * 3 + 1] = k1[c]; coef[c * 3 + 2] = k2[c]; }
                }
            }
        }
        if (writer && grp == 0 && threadIdx.x < C) { dbeta[threadIdx.x] = (float)db_tot; dgamma[threadIdx.x] = (float)dg_tot; }
        if (pmax) {  // (one statistic group only: az_bn3d_bwd rejects the combination otherwise)
            const uint4 *m4 = reinterpret_cast<const uint4 *>(pmax);
            uint4 m = make_uint4(0, 0, 0, 0);
            for (int blk = r; blk < nblocks; blk += ROWS) {
                const uint4 v = m4[(size_t)blk * COLS + j];
                m.x = max(m.x, v.x); m.y = max(m.y, v.y); m.z = max(m.z, v.z); m.w = max(m.w, v.w);
            }
#pragma unroll
            for (int off = COLS; off < 64; off <<= 1) {
                m.x = max(m.x, (unsigned)__shfl_xor((int)m.x, off)); m.y = max(m.y, (unsigned)__shfl_xor((int)m.y, off));
                m.z = max(m.z, (unsigned)__shfl_xor((int)m.z, off)); m.w = max(m.w, (unsigned)__shfl_xor((int)m.w, off));
            }
            __syncthreads();  // (k0 / k1 / k2 of every channel written; wsum free again)
            if ((threadIdx.x & 63) < COLS) {
                wsum[wave][4 * j + 0] = __uint_as_float(m.x); wsum[wave][4 * j + 1] = __uint_as_float(m.y);
                wsum[wave][4 * j + 2] = __uint_as_float(m.z); wsum[wave][4 * j + 3] = __uint_as_float(m.w);
            }
            __syncthreads();
            float bnd = 0.f;
            if (threadIdx.x < C) {
                const int c = threadIdx.x;
                const float gm = fmaxf(fmaxf(wsum[0][2 * c], wsum[1][2 * c]), fmaxf(wsum[2][2 * c], wsum[3][2 * c]));
                const float xm = fmaxf(fmaxf(wsum[0][2 * c + 1], wsum[1][2 * c + 1]), fmaxf(wsum[2][2 * c + 1], wsum[3][2 * c + 1]));
                // (1 + 2^-20: the rounding of the apply expression itself)
                bnd = fabsf(k0[c]) * (gm + fabsf(k1[c]) + xm * fabsf(k2[c])) * 1.000001f;
                bnd = bnd < __builtin_inff() ? bnd : 0.f;  // (a non-finite coefficient: every element of that channel is, too)
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) bnd = fmaxf(bnd, __shfl_xor(bnd, off));
            __shared__ float bwave[4];
            if ((threadIdx.x & 63) == 0) bwave[wave] = bnd;
            __syncthreads();
            bnd = fmaxf(fmaxf(bwave[0], bwave[1]), fmaxf(bwave[2], bwave[3]));
            split_scale = az_pow2(az_f16_scale_exp(bnd));
            // slot 0 of the amax array (the reduce kernel zeroed all sixteen): the bound
            if (writer && grp == 0 && threadIdx.x == 0 && amax) amax[0] = __float_as_uint(bnd);
        }
    }
    __syncthreads();
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += gridDim.x * 256LL) {
        const int c = (int)(i % C4) * 4;
        float4 g = bn_ld4<NT>(&dy[i]);
        const float4 xx = bn_ld4<NT>(&x[i]);
        if (relu) {
            float4 yy;
            if (remask) yy = make_float4(fmaf(xx.x, ssc[c + 0], ssh[c + 0]), fmaf(xx.y, ssc[c + 1], ssh[c + 1]),
                                         fmaf(xx.z, ssc[c + 2], ssh[c + 2]), fmaf(xx.w, ssc[c + 3], ssh[c + 3]));
            else yy = bn_ld4<NT>(&y[i]);
            g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
            g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        }
        if (dz_out) bn_st4<NT>(&dz_out[i], g);
        float4 o;
        o.x = k0[c + 0] * (g.x - k1[c + 0] - (xx.x - smu[c + 0]) * sis[c + 0] * k2[c + 0]);
        o.y = k0[c + 1] * (g.y - k1[c + 1] - (xx.y - smu[c + 1]) * sis[c + 1] * k2[c + 1]);
        o.z = k0[c + 2] * (g.z - k1[c + 2] - (xx.z - smu[c + 2]) * sis[c + 2] * k2[c + 2]);
        o.w = k0[c + 3] * (g.w - k1[c + 3] - (xx.w - smu[c + 3]) * sis[c + 3] * k2[c + 3]);
        if (pmax) {
            bn_st4<NT>(&dx[i], az_presplit_f16x4(make_float4(o.x * split_scale, o.y * split_scale, o.z * split_scale, o.w * split_scale)));
        } else {
            bn_st4<NT>(&dx[i], o);
            az_amax_acc(am, o);
        }
    }
    if (amax && !pmax) az_amax_flush(amax, am);
}

// y = relu?(a + b) and its masked backward, for the plain residual sums of psmnet_3.py
__global__ void __launch_bounds__(256)
add_relu_kernel(float4 *__restrict__ y, const float4 *__restrict__ a,
                const float4 *__restrict__ b, int relu, long long total4, unsigned *__restrict__ amax) {
    unsigned am = 0;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += gridDim.x * 256LL) {
        const float4 p = a[i], q = b[i];
        float4 o = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
        if (relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        y[i] = o;
        az_amax_acc(am, o);
    }
    if (amax) az_amax_flush(amax, am);
}


// Standalone batch statistics of a channels-last tensor x[nvox][C] (used where the producing kernel does not
// emit BatchNorm partials in its epilogue: the 2-D convolution kernels of the feature extractor).  Block `t` reduces voxels
// t*VPB + k*gridDim*VPB ... to one (sum, centred M2, count) partial per channel, in the layout
// az_bn3d_finalize consumes ([C][tiles][2], counts[tiles]).  Sums are taken about the block's first
// voxel (shifted data) so M2 = S2 - S1^2/n does not cancel.
template <int C>
__global__ void __launch_bounds__(256)
bn_stats_kernel(float *__restrict__ part, float *__restrict__ cnt, const float *__restrict__ x,
                long long nvox, long long ntiles) {
    constexpr int C4 = C / 4, VPB = 256 / C4;
    const int c4 = threadIdx.x % C4, vl = threadIdx.x / C4;
    x += (size_t)blockIdx.y * nvox * C;  // blockIdx.y = statistic group
    part += (size_t)blockIdx.y * C * ntiles * 2;
    cnt += (size_t)blockIdx.y * ntiles;
    const long long v0 = (long long)blockIdx.x * VPB;  // first voxel of this block: always < nvox
    const float4 K = reinterpret_cast<const float4 *>(x)[(size_t)v0 * C4 + c4];
    float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
    int n = 0;
    for (long long v = v0 + vl; v < nvox; v += (long long)gridDim.x * VPB) {
        const float4 a = reinterpret_cast<const float4 *>(x)[(size_t)v * C4 + c4];
        const float dx = a.x - K.x, dy = a.y - K.y, dz = a.z - K.z, dw = a.w - K.w;
        s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
        s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
        ++n;
    }
    __shared__ float4 r1[256], r2[256];
    __shared__ int rn[256];
    r1[threadIdx.x] = s1; r2[threadIdx.x] = s2; rn[threadIdx.x] = n;
    __syncthreads();
    if (vl == 0) {
        for (int k = 1; k < VPB; ++k) {
            const float4 a = r1[k * C4 + c4], b = r2[k * C4 + c4];
            s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
            s2.x += b.x; s2.y += b.y; s2.z += b.z; s2.w += b.w;
            n += rn[k * C4 + c4];
        }
        const float fn = (float)n;
        const float s[4] = {s1.x, s1.y, s1.z, s1.w}, q[4] = {s2.x, s2.y, s2.z, s2.w};
        const float k4[4] = {K.x, K.y, K.z, K.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float2 o;
            o.x = s[e] + fn * k4[e];                       // sum of x
            o.y = fmaxf(q[e] - s[e] * s[e] / fn, 0.f);     // centred second moment
            reinterpret_cast<float2 *>(part)[(size_t)(c4 * 4 + e) * ntiles + blockIdx.x] = o;
        }
        if (c4 == 0) cnt[blockIdx.x] = fn;
    }
}

// y = a + b (+ c) (+ d): gradients of a tensor with several consumers, summed in ONE pass
__global__ void __launch_bounds__(256)
sum4_kernel(float4 *__restrict__ y, const float4 *__restrict__ a, const float4 *__restrict__ b,
            const float4 *__restrict__ c, const float4 *__restrict__ d, long long total4) {
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total4; i += gridDim.x * 256LL) {
        const float4 p = a[i], q = b[i];
        float4 o = make_float4(p.x + q.x, p.y + q.y, p.z + q.z, p.w + q.w);
        if (c) { const float4 r = c[i]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
        if (d) { const float4 r = d[i]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
        y[i] = o;
    }
}

#define BN_GRID(total) az_grid_for((total), 256)
// finalize folded into the apply kernel's prologue (AZ_BN_BWD_FUSED=0: the three-kernel sequence, for A/B runs)
static bool bn_bwd_fused() {
    const int on = az_options().bn_bwd_fused;
    return on != 0;
}
static int bn_bwd_fused_blocks(int blocks, int C) {
    const int cap = BN_BWD_PARTIAL_FLOATS / (2 * C);  // 512 / 256 / 128 reduce blocks for 32 / 64 / 128 channels
    return blocks < cap ? blocks : cap;
}
static unsigned bn_bwd_apply_grid(long long total4) {
    const unsigned g = az_grid_for(total4, 256);
    return g < 1024u ? g : 1024u;  // every block re-reads the partials: 1024 x 128 KB of L2 traffic at most
}
// KERNEL<C, NT> for the three channel counts, NT chosen at run time
#define BN_LAUNCH(KERNEL, C, NT, GRID, STREAM, ...)                                                             \
    do {                                                                                                        \
        if (NT) {                                                                                               \
            if ((C) == 32) hipLaunchKernelGGL((KERNEL<32, true>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);     \
            else if ((C) == 64) hipLaunchKernelGGL((KERNEL<64, true>), GRID, dim3(256), 0, STREAM, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<128, true>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);              \
        } else {                                                                                                \
            if ((C) == 32) hipLaunchKernelGGL((KERNEL<32, false>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);    \
            else if ((C) == 64) hipLaunchKernelGGL((KERNEL<64, false>), GRID, dim3(256), 0, STREAM, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<128, false>), GRID, dim3(256), 0, STREAM, __VA_ARGS__);             \
        }                                                                                                       \
    } while (0)

extern "C" long long az_bn3d_finalize_scratch(int C) { return (long long)C * BN_PRE_SLICES * 2 + BN_PRE_SLICES; }

extern "C" int az_bn3d_finalize(float *mean, float *invstd, float *scale, float *shift,
                                float *running_mean, float *running_var, const float *partials,
                                const float *counts, const float *gamma, const float *beta,
                                long long ntiles, int C, float eps, float momentum,
                                long long *num_batches_tracked, float *scratch,
                                long long scratch_floats, void *stream) {
    AZ_REQUIRE_PTR(mean); AZ_REQUIRE_PTR(invstd); AZ_REQUIRE_PTR(scale); AZ_REQUIRE_PTR(shift);
    AZ_REQUIRE_PTR(partials); AZ_REQUIRE_PTR(counts); AZ_REQUIRE_PTR(gamma); AZ_REQUIRE_PTR(beta);
    AZ_REQUIRE(ntiles > 0 && C > 0);
    if ((running_mean == nullptr) != (running_var == nullptr)) return AZ_EINVAL;
    if (scratch != nullptr && ntiles >= 4096) {  // two stages (see bn_premerge_kernel)
        if (scratch_floats < az_bn3d_finalize_scratch(C)) return AZ_EWORKSPACE;
        float *p2 = scratch, *c2 = scratch + (size_t)C * BN_PRE_SLICES * 2;
        hipLaunchKernelGGL(bn_premerge_kernel, dim3(C, BN_PRE_SLICES), dim3(256), 0, az_stream(stream), p2, c2, partials,
                           counts, ntiles);
        partials = p2; counts = c2; ntiles = BN_PRE_SLICES;
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(BN_FIN_THREADS), 0, az_stream(stream), mean, invstd,
                       scale, shift, running_mean, running_var, partials, counts, gamma, beta,
                       ntiles, C, eps, momentum, 1, num_batches_tracked);
    return az_launch_status();
}

extern "C" int az_bn3d_eval_affine(float *scale, float *shift, const float *gamma,
                                   const float *beta, const float *running_mean,
                                   const float *running_var, float eps, int C, void *stream) {
    AZ_REQUIRE_PTR(scale); AZ_REQUIRE_PTR(shift); AZ_REQUIRE_PTR(gamma); AZ_REQUIRE_PTR(beta);
    AZ_REQUIRE_PTR(running_mean); AZ_REQUIRE_PTR(running_var);
    AZ_REQUIRE(C > 0);
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 63) / 64), dim3(64), 0, az_stream(stream),
                       scale, shift, gamma, beta, running_mean, running_var, eps, C);
    return az_launch_status();
}

extern "C" int az_bn3d_apply(float *y, const float *x, const float *scale, const float *shift,
                             const float *residual, int relu, long long nvox, int C, float *y_amax,
                             void *stream) {
    AZ_REQUIRE_PTR(y); AZ_REQUIRE_PTR(x); AZ_REQUIRE_PTR(scale); AZ_REQUIRE_PTR(shift);
    AZ_REQUIRE(nvox > 0);
    const long long total4 = nvox * C / 4;
    const float4 *r4 = reinterpret_cast<const float4 *>(residual);
    if (C != 32 && C != 64 && C != 128) return AZ_EUNSUPPORTED;
    const bool nt = total4 * 16 >= BN_NT_BYTES;
    BN_LAUNCH(bn_apply_kernel, C, nt, dim3(BN_GRID(total4)), az_stream(stream), (float4 *)y, (const float4 *)x, scale,
              shift, r4, relu, total4, reinterpret_cast<unsigned *>(y_amax));
    return az_launch_status();
}

static long long bn3d_bwd_blocks(long long nvox, int C) {
    const int vpb = 256 / (C / 4);
    return az_grid_for((nvox + vpb - 1) / vpb * 256, 256);
}
// (the sums' rows, then -- split_out launches -- as many rows of per-channel maxima)
extern "C" long long az_bn3d_bwd_workspace(long long nvox, int C) {
    if (nvox <= 0 || (C != 32 && C != 64 && C != 128)) return AZ_EINVAL;
    return 2 * bn3d_bwd_blocks(nvox, C) * C * 2 * (long long)sizeof(float);
}

extern "C" int az_bn3d_bwd(float *dx, float *dz_out, float *dgamma, float *dbeta, float *coef,
                           float *workspace, long long workspace_bytes, const float *dy,
                           const float *y, const float *x, const float *mean,
                           const float *invstd, const float *gamma, const float *scale,
                           const float *shift, int relu, long long nvox, int C, float *dx_amax, int split_out,
                           void *stream) {
    unsigned *const am = reinterpret_cast<unsigned *>(dx_amax);
    AZ_REQUIRE_PTR(dx); AZ_REQUIRE_PTR(dgamma); AZ_REQUIRE_PTR(dbeta); AZ_REQUIRE_PTR(coef);
    AZ_REQUIRE_PTR(workspace); AZ_REQUIRE_PTR(dy); AZ_REQUIRE_PTR(x); AZ_REQUIRE_PTR(mean);
    AZ_REQUIRE_PTR(invstd); AZ_REQUIRE_PTR(gamma);
    if ((scale == nullptr) != (shift == nullptr)) return AZ_EINVAL;
    if (relu && !scale) AZ_REQUIRE_PTR(y);
    const long long need = az_bn3d_bwd_workspace(nvox, C);
    if (need < 0) return (int)need;
    if (workspace_bytes < need) return AZ_EWORKSPACE;
    int blocks = (int)bn3d_bwd_blocks(nvox, C);
    const long long total4 = nvox * C / 4;
    hipStream_t s = az_stream(stream);
    const bool nt = total4 * 16 >= BN_NT_BYTES;
    if (split_out && (!am || !bn_bwd_fused())) return split_out && !am ? AZ_ENULL : AZ_EUNSUPPORTED;  // (the fused two-launch form only)
    if (bn_bwd_fused()) {
        blocks = bn_bwd_fused_blocks(blocks, C);
        unsigned *const pmax = split_out ? reinterpret_cast<unsigned *>(workspace) + (size_t)blocks * C * 2 : nullptr;
        BN_LAUNCH(bn_bwd_reduce_kernel, C, nt, dim3(blocks), s, workspace, dy, y, x, mean, invstd, scale, shift, relu, nvox, am, pmax);
        BN_LAUNCH(bn_bwd_apply_kernel, C, nt, dim3(bn_bwd_apply_grid(total4)), s, (float4 *)dx, (float4 *)dz_out, (const float4 *)dy,
                  (const float4 *)y, (const float4 *)x, mean, invstd, coef, scale, shift, relu, total4,
                  (const float *)workspace, blocks, gamma, dgamma, dbeta, (double)nvox, am, (const unsigned *)pmax);
        return az_launch_status();
    }
    BN_LAUNCH(bn_bwd_reduce_kernel, C, nt, dim3(blocks), s, workspace, dy, y, x, mean, invstd, scale, shift, relu, nvox, am);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, dgamma, dbeta, coef,
                       workspace, gamma, invstd, blocks, C, (double)nvox, 1);
    BN_LAUNCH(bn_bwd_apply_kernel, C, nt, dim3(BN_GRID(total4)), s, (float4 *)dx, (float4 *)dz_out, (const float4 *)dy,
              (const float4 *)y, (const float4 *)x, mean, invstd, coef, scale, shift, relu, total4,
              (const float *)nullptr, 0, gamma, dgamma, dbeta, (double)nvox, am);
    return az_launch_status();
}

extern "C" int az_add_relu(float *y, const float *a, const float *b, int relu, long long n, float *y_amax,
                           void *stream) {
    AZ_REQUIRE_PTR(y); AZ_REQUIRE_PTR(a); AZ_REQUIRE_PTR(b);
    AZ_REQUIRE(n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(add_relu_kernel, dim3(BN_GRID(n / 4)), dim3(256), 0, az_stream(stream),
                       (float4 *)y, (const float4 *)a, (const float4 *)b, relu, n / 4, reinterpret_cast<unsigned *>(y_amax));
    return az_launch_status();
}

extern "C" int az_sum4(float *y, const float *a, const float *b, const float *c, const float *d, long long n,
                       void *stream) {
    AZ_REQUIRE_PTR(y); AZ_REQUIRE_PTR(a); AZ_REQUIRE_PTR(b);
    if (d && !c) return AZ_EINVAL;
    AZ_REQUIRE(n > 0 && n % 4 == 0);
    hipLaunchKernelGGL(sum4_kernel, dim3(BN_GRID(n / 4)), dim3(256), 0, az_stream(stream), (float4 *)y,
                       (const float4 *)a, (const float4 *)b, (const float4 *)c, (const float4 *)d, n / 4);
    return az_launch_status();
}

// number of partial tiles az_bn3d_stats writes for a tensor of nvox voxels
extern "C" long long az_bn3d_stats_tiles(long long nvox, int C) {
    if (nvox <= 0 || (C != 32 && C != 64 && C != 128)) return AZ_EINVAL;
    const int vpb = 256 / (C / 4);
    long long t = (nvox + (long long)vpb * 16 - 1) / ((long long)vpb * 16);  // >= 16 voxels per thread
    if (t > 2048) t = 2048;
    if (t < 1) t = 1;
    return t;
}

extern "C" int az_bn3d_stats(float *partials, float *counts, const float *x, long long nvox, int C,
                             void *stream) {
    AZ_REQUIRE_PTR(partials); AZ_REQUIRE_PTR(counts); AZ_REQUIRE_PTR(x);
    const long long tiles = az_bn3d_stats_tiles(nvox, C);
    if (tiles < 0) return (int)tiles;
    hipStream_t s = az_stream(stream);
    if (C == 32)
        hipLaunchKernelGGL(bn_stats_kernel<32>, dim3((unsigned)tiles), dim3(256), 0, s, partials, counts, x, nvox, tiles);
    else if (C == 64)
        hipLaunchKernelGGL(bn_stats_kernel<64>, dim3((unsigned)tiles), dim3(256), 0, s, partials, counts, x, nvox, tiles);
    else
        hipLaunchKernelGGL(bn_stats_kernel<128>, dim3((unsigned)tiles), dim3(256), 0, s, partials, counts, x, nvox, tiles);
    return az_launch_status();
}

// ---- grouped BatchNorm of a channels-last tensor [groups][nvox][C] (the extractor's layers) ------
// One call = statistics + finalize + apply (forward) or reduce + finalize + apply (backward) for ALL
// statistic groups: three launches per layer instead of six per group.
static int bn2d_blocks(long long nvox, int C) {
    const int vpb = 256 / (C / 4);
    return (int)az_grid_for((nvox + vpb - 1) / vpb * 256, 256);
}

// floats of scratch: forward {partials G*C*T*2, counts G*T}, backward {partials G*blocks*C*2, coef G*C*3}
extern "C" long long az_bn2d_workspace(int groups, long long nvox, int C) {
    const long long tiles = az_bn3d_stats_tiles(nvox, C);
    if (tiles < 0 || groups <= 0) return AZ_EINVAL;
    const long long fwd = (long long)groups * (C * tiles * 2 + tiles);
    const long long bwd = (long long)groups * ((long long)bn2d_blocks(nvox, C) * C * 2 + C * 3);
    return (fwd > bwd ? fwd : bwd) * (long long)sizeof(float);
}

template <int C>
static void bn2d_fwd_launch(float *y, float *mean, float *invstd, float *scale, float *shift, float *rm,
                            float *rv, const float *x, const float *res, const float *gamma,
                            const float *beta, float *ws, int relu, int groups, long long nvox, float eps,
                            float momentum, long long *nbt, const float *pre_part, const float *pre_cnt,
                            long long pre_tiles, hipStream_t s, unsigned *amax) {
    long long tiles = az_bn3d_stats_tiles(nvox, C);
    const float *part = ws, *cnt = ws + (size_t)groups * C * tiles * 2;
    if (pre_part != nullptr) {  // the producing convolution already reduced its patches (az_conv2d_fwd_stats)
        part = pre_part; cnt = pre_cnt; tiles = pre_tiles;
    } else {
        hipLaunchKernelGGL(bn_stats_kernel<C>, dim3((unsigned)tiles, groups), dim3(256), 0, s, ws,
                           ws + (size_t)groups * C * tiles * 2, x, nvox, tiles);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(BN_FIN_THREADS), 0, s, mean, invstd, scale, shift, rm, rv,
                       part, cnt, gamma, beta, tiles, C, eps, momentum, groups, nbt);
    const long long total4 = nvox * C / 4;
    hipLaunchKernelGGL((bn_apply_kernel<C, false>), dim3(BN_GRID(total4), groups), dim3(256), 0, s, (float4 *)y,
                       (const float4 *)x, scale, shift, (const float4 *)res, relu, total4, amax);
}

/* y = relu?(bn(x) + residual) with batch statistics per group; mean/invstd/scale/shift: [groups][C] outputs */
extern "C" int az_bn2d_fwd(float *y, float *mean, float *invstd, float *scale, float *shift,
                           float *running_mean, float *running_var, const float *x, const float *residual,
                           const float *gamma, const float *beta, float *workspace, long long workspace_bytes,
                           int relu, int groups, long long nvox, int C, float eps, float momentum,
                           long long *num_batches_tracked, const float *partials,
                           const float *counts, long long partial_tiles, float *y_amax, void *stream) {
    unsigned *const am = reinterpret_cast<unsigned *>(y_amax);
    AZ_REQUIRE_PTR(y); AZ_REQUIRE_PTR(mean); AZ_REQUIRE_PTR(invstd); AZ_REQUIRE_PTR(scale); AZ_REQUIRE_PTR(shift);
    AZ_REQUIRE_PTR(x); AZ_REQUIRE_PTR(gamma); AZ_REQUIRE_PTR(beta); AZ_REQUIRE_PTR(workspace);
    if ((running_mean == nullptr) != (running_var == nullptr)) return AZ_EINVAL;
    if ((partials == nullptr) != (counts == nullptr) || (partials != nullptr && partial_tiles <= 0)) return AZ_EINVAL;
    const long long need = az_bn2d_workspace(groups, nvox, C);
    if (need < 0) return (int)need;
    if (workspace_bytes < need) return AZ_EWORKSPACE;
    if (groups > 65535) return AZ_EUNSUPPORTED;
    hipStream_t s = az_stream(stream);
    if (C == 32) bn2d_fwd_launch<32>(y, mean, invstd, scale, shift, running_mean, running_var, x, residual, gamma, beta, workspace, relu, groups, nvox, eps, momentum, num_batches_tracked, partials, counts, partial_tiles, s, am);
    else if (C == 64) bn2d_fwd_launch<64>(y, mean, invstd, scale, shift, running_mean, running_var, x, residual, gamma, beta, workspace, relu, groups, nvox, eps, momentum, num_batches_tracked, partials, counts, partial_tiles, s, am);
    else bn2d_fwd_launch<128>(y, mean, invstd, scale, shift, running_mean, running_var, x, residual, gamma, beta, workspace, relu, groups, nvox, eps, momentum, num_batches_tracked, partials, counts, partial_tiles, s, am);
    return az_launch_status();
}

template <int C>
static void bn2d_bwd_launch(float *dx, float *dz, float *dgamma, float *dbeta, float *ws, const float *dy,
                            const float *y, const float *x, const float *mean, const float *invstd,
                            const float *gamma, const float *scale, const float *shift, int relu, int groups,
                            long long nvox, hipStream_t s, unsigned *amax) {
    int blocks = bn2d_blocks(nvox, C);
    float *partial = ws, *coef = ws + (size_t)groups * blocks * C * 2;  // (coef behind the UNCAPPED partial area)
    const long long total4 = nvox * C / 4;
    if (bn_bwd_fused()) {
        blocks = bn_bwd_fused_blocks(blocks, C);
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<C, false>), dim3(blocks, groups), dim3(256), 0, s, partial, dy, y, x, mean,
                           invstd, scale, shift, relu, nvox);
        hipLaunchKernelGGL((bn_bwd_apply_kernel<C, false>), dim3(bn_bwd_apply_grid(total4), groups), dim3(256), 0, s, (float4 *)dx,
                           (float4 *)dz, (const float4 *)dy, (const float4 *)y, (const float4 *)x, mean, invstd,
                           coef, scale, shift, relu, total4, (const float *)partial, blocks, gamma, dgamma, dbeta, (double)nvox, amax);
        return;
    }
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<C, false>), dim3(blocks, groups), dim3(256), 0, s, partial, dy, y, x, mean,
                       invstd, scale, shift, relu, nvox);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, dgamma, dbeta, coef, partial, gamma,
                       invstd, blocks, C, (double)nvox, groups);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<C, false>), dim3(BN_GRID(total4), groups), dim3(256), 0, s, (float4 *)dx,
                       (float4 *)dz, (const float4 *)dy, (const float4 *)y, (const float4 *)x, mean, invstd,
                       coef, scale, shift, relu, total4, (const float *)nullptr, 0, gamma, dgamma, dbeta, (double)nvox);
}

/* backward of az_bn2d_fwd; dgamma/dbeta [C] are summed over the groups; dz_out may be NULL */
extern "C" int az_bn2d_bwd(float *dx, float *dz_out, float *dgamma, float *dbeta, float *workspace,
                           long long workspace_bytes, const float *dy, const float *y, const float *x,
                           const float *mean, const float *invstd, const float *gamma, const float *scale,
                           const float *shift, int relu, int groups, long long nvox, int C, float *dx_amax,
                           void *stream) {
    AZ_REQUIRE_PTR(dx); AZ_REQUIRE_PTR(dgamma); AZ_REQUIRE_PTR(dbeta); AZ_REQUIRE_PTR(workspace);
    AZ_REQUIRE_PTR(dy); AZ_REQUIRE_PTR(x); AZ_REQUIRE_PTR(mean); AZ_REQUIRE_PTR(invstd); AZ_REQUIRE_PTR(gamma);
    if ((scale == nullptr) != (shift == nullptr)) return AZ_EINVAL;
    if (relu && !scale) AZ_REQUIRE_PTR(y);
    unsigned *const am = reinterpret_cast<unsigned *>(dx_amax);
    const long long need = az_bn2d_workspace(groups, nvox, C);
    if (need < 0) return (int)need;
    if (workspace_bytes < need) return AZ_EWORKSPACE;
    if (groups > 65535) return AZ_EUNSUPPORTED;
    hipStream_t s = az_stream(stream);
    if (C == 32) bn2d_bwd_launch<32>(dx, dz_out, dgamma, dbeta, workspace, dy, y, x, mean, invstd, gamma, scale, shift, relu, groups, nvox, s, am);
    else if (C == 64) bn2d_bwd_launch<64>(dx, dz_out, dgamma, dbeta, workspace, dy, y, x, mean, invstd, gamma, scale, shift, relu, groups, nvox, s, am);
    else bn2d_bwd_launch<128>(dx, dz_out, dgamma, dbeta, workspace, dy, y, x, mean, invstd, gamma, scale, shift, relu, groups, nvox, s, am);
    return az_launch_status();
}
