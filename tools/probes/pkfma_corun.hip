// Does v_pk_fma_f32 keep its results when a wave of ANOTHER kernel issues MFMAs on the same SIMD?
// (found through the 32->1 weight-gradient kernel: lanes 48..63 of some v_pk_fma_f32 results differed when the
//  kernel ran beside the depth-rolling convolution on a second stream; profiles/r03_pkfma_corun.md)
//   victim<MODE>: per lane 16 accumulators updated `iters` times, results written out; run alone -> reference,
//                 then beside an aggressor kernel on a second stream; results must be bit-identical.
//     MODE 0: acc = v_pk_fma_f32(x, {g,g}, acc), g from LDS (op_sel broadcast)      1: the same with two v_fma_f32
//     MODE 2: v_pk_fma_f32, g from registers (no LDS in the loop)                   3: v_pk_fma_f32, {g0,g1} a real pair
//     MODE 4: v_pk_mul_f32 + v_pk_add_f32                                           5: v_pk_add_f32 only
//   aggressor: one wave per SIMD (256 threads x 256 workgroups) of back-to-back v_mfma_f32_16x16x32_bf16 (hipcc adds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__device__ __forceinline__ void victim_body(float *o, const float *gt, int iters, int tid) {
    f2 acc[16];
    f2 x[4];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = f2{0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = f2{1.f + 0.001f * (tid + j), 1.f - 0.002f * (tid + 3 * j)};
    float4 rreg = make_float4(0.1f + 1e-3f * tid, -0.2f + 1e-3f * tid, 0.3f - 1e-3f * tid, 0.05f);
    for (int it = 0; it < iters; ++it) {
        float4 r;
        if (MODE == 2) { rreg.x = -rreg.x; asm volatile("" : "+v"(rreg.x), "+v"(rreg.y), "+v"(rreg.z), "+v"(rreg.w)); r = rreg; }
        else r = *reinterpret_cast<const float4 *>(&gt[((tid + it) & 255) * 4]);
        const float rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f2 g2 = (MODE == 3) ? f2{rr[k & 3], rr[(k + 1) & 3]} : f2{rr[k & 3], rr[k & 3]};
            if (MODE == 1) {
                float lo, hi;
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(x[k >> 2][0]), "v"(g2[0]), "v"(acc[k][0]));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(x[k >> 2][1]), "v"(g2[1]), "v"(acc[k][1]));
                acc[k] = f2{lo, hi};
            } else if (MODE == 4) {
                f2 m;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m) : "v"(x[k >> 2]), "v"(g2));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[k]) : "v"(m), "v"(acc[k]));
            } else if (MODE == 5) {
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[k]) : "v"(g2), "v"(acc[k]));
            } else {
                acc[k] = __builtin_elementwise_fma(x[k >> 2], g2, acc[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) { o[2 * k] = acc[k][0]; o[2 * k + 1] = acc[k][1]; }
}

__global__ void __launch_bounds__(256) aggressor(float *sink, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[threadIdx.x] = c0[0];
}


template <int MODE>
__global__ void __launch_bounds__(256) victim(float *out, const float *gsrc, int iters) {
    __shared__ __attribute__((aligned(16))) float gt[1024];
    for (int q = threadIdx.x; q < 1024; q += 256) gt[q] = gsrc[q];
    __syncthreads();
    victim_body<MODE>(out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 32, gt, iters, threadIdx.x);
}


// Hand-scheduled victims (one asm block, fixed registers v[100:101]) that separate the two candidate hazards:
//   T 1  write after read: v_pk_fma_f32 reads v[100:101]; the NEXT instruction is a ds_read_b64 into v[100:101]
//        (its address was computed earlier); 15 unrelated v_pk_fma_f32; s_waitcnt lgkmcnt(0); repeat.
//   T 2  read after write only: ds_read_b64 v[100:101]; s_waitcnt lgkmcnt(0); v_pk_fma_f32 reads them at once;
//        30 unrelated v_pk_fma_f32 before the next load.
//   T 3  T 1 with one VALU instruction (v_mov of the address register) between the v_pk_fma_f32 and the ds_read_b64:
//        the load's address then leaves the in-order VALU pipe BEHIND the v_pk_fma_f32.
//   T 4  T 1 with v_fma_f32 (x2) in place of v_pk_fma_f32.
#define F1 " v_pk_fma_f32 %[c1], %[y], %[y], %[c1]\n"
#define F5 F1 F1 F1 F1 F1
#define LOOP_TAIL " v_mov_b32 %[a], %[a2]\n s_waitcnt lgkmcnt(0)\n s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
#define LOOP_HEAD " ds_read_b64 v[100:101], %[a]\n s_waitcnt lgkmcnt(0)\n 1:\n v_add_u32 %[a2], 8, %[a]\n v_and_b32 %[a2], 0xff8, %[a2]\n s_nop 7\n s_nop 7\n"
#define OPERANDS : [c0] "+v"(c0), [c1] "+v"(c1), [a] "+v"(a), [a2] "+v"(a2), [n] "+s"(n) : [x] "v"(x), [y] "v"(y) : "v100", "v101", "scc", "memory"
template <int T>
__global__ void __launch_bounds__(256) victim_asm(float *out, const float *gsrc, int iters) {
    __shared__ __attribute__((aligned(16))) float gt[1024];
    for (int q = threadIdx.x; q < 1024; q += 256) gt[q] = gsrc[q];
    __syncthreads();
    f2 c0 = {0.f, 0.f}, c1 = {0.f, 0.f};
    const f2 x = {1.f + 0.001f * threadIdx.x, 1.f - 0.002f * threadIdx.x}, y = {0.5f, 0.25f};
    unsigned a = (threadIdx.x * 8u) & 0xff8u, a2 = 0;
    int n = iters;
    if (T == 1)
        asm volatile(LOOP_HEAD " v_pk_fma_f32 %[c0], %[x], v[100:101], %[c0] op_sel_hi:[1,0,1]\n"
                     " ds_read_b64 v[100:101], %[a2]\n" F5 F5 F5 LOOP_TAIL OPERANDS);
    if (T == 3)
        asm volatile(LOOP_HEAD " v_pk_fma_f32 %[c0], %[x], v[100:101], %[c0] op_sel_hi:[1,0,1]\n"
                     " v_mov_b32 %[a2], %[a2]\n"
                     " ds_read_b64 v[100:101], %[a2]\n" F5 F5 F5 LOOP_TAIL OPERANDS);
    if (T == 4) {
        float lo = 0.f, hi = 0.f;
        const float xl = x[0], xh = x[1];
        asm volatile(LOOP_HEAD " v_fma_f32 %[lo], %[xl], v100, %[lo]\n v_fma_f32 %[hi], %[xh], v100, %[hi]\n"
                     " ds_read_b64 v[100:101], %[a2]\n" F5 F5 F5 LOOP_TAIL
                     : [lo] "+v"(lo), [hi] "+v"(hi), [c1] "+v"(c1), [a] "+v"(a), [a2] "+v"(a2), [n] "+s"(n)
                     : [xl] "v"(xl), [xh] "v"(xh), [y] "v"(y) : "v100", "v101", "scc", "memory");
        c0 = f2{lo, hi};
    }
    if (T == 2)
        asm volatile(" 1:\n v_add_u32 %[a2], 8, %[a]\n v_and_b32 %[a2], 0xff8, %[a2]\n"
                     " ds_read_b64 v[100:101], %[a2]\n s_waitcnt lgkmcnt(0)\n"
                     " v_pk_fma_f32 %[c0], %[x], v[100:101], %[c0] op_sel_hi:[1,0,1]\n" F5 F5 F5 F5 F5 F5
                     " v_mov_b32 %[a], %[a2]\n s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n" OPERANDS);
    if (T >= 5 && T <= 10) {  // two ds_read_b128 in flight, s_waitcnt lgkmcnt(1), use of the FIRST one's data at once (T 6: lgkmcnt(0))
        f2 c2 = {0.f, 0.f}, c3 = {0.f, 0.f};
        a = (threadIdx.x * 16u) & 0xff0u;
#define T5_LOADS128 " ds_read_b128 v[100:103], %[a]\n ds_read_b128 v[104:107], %[a2]\n"
#define T5_LOADS64 " ds_read_b64 v[100:101], %[a]\n ds_read_b64 v[102:103], %[a] offset:8\n ds_read_b64 v[104:105], %[a2]\n ds_read_b64 v[106:107], %[a2] offset:8\n"
#define T5_BODYX(PRE, LOADS, WAIT, ODD) " 1:\n v_add_u32 %[a2], 16, %[a]\n v_and_b32 %[a2], 0xff0, %[a2]\n" PRE LOADS WAIT \
                     " v_pk_fma_f32 %[c0], %[x], v[100:101], %[c0] op_sel_hi:[1,0,1]\n" \
                     " v_pk_fma_f32 %[c1], %[x], v[100:101], %[c1] " ODD "\n" \
                     " v_pk_fma_f32 %[c2], %[x], v[102:103], %[c2] op_sel_hi:[1,0,1]\n" \
                     " v_pk_fma_f32 %[c3], %[x], v[102:103], %[c3] " ODD "\n" \
                     " s_waitcnt lgkmcnt(0)\n" \
                     " v_pk_fma_f32 %[c0], %[x], v[104:105], %[c0] op_sel_hi:[1,0,1]\n" \
                     " v_pk_fma_f32 %[c1], %[x], v[104:105], %[c1] " ODD "\n" \
                     " v_pk_fma_f32 %[c2], %[x], v[106:107], %[c2] op_sel_hi:[1,0,1]\n" \
                     " v_pk_fma_f32 %[c3], %[x], v[106:107], %[c3] " ODD "\n" \
                     " v_mov_b32 %[a], %[a2]\n s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
#define T5_BODY(WAIT) T5_BODYX("", T5_LOADS128, WAIT, "op_sel:[0,1,0]")
#define NOP64 " s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n"
#define T5_OPS : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [a] "+v"(a), [a2] "+v"(a2), [n] "+s"(n) : [x] "v"(x) \
               : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "scc", "memory"
        if (T == 7) asm volatile(T5_BODYX(NOP64 NOP64, T5_LOADS128, " s_waitcnt lgkmcnt(0)\n", "op_sel:[0,1,0]") T5_OPS);
        else if (T == 8) asm volatile(T5_BODYX("", T5_LOADS128, " s_waitcnt lgkmcnt(0)\n", "op_sel_hi:[1,0,1]") T5_OPS);
        else if (T == 9) asm volatile(T5_BODYX("", T5_LOADS64, " s_waitcnt lgkmcnt(0)\n", "op_sel:[0,1,0]") T5_OPS);
        else if (T == 10) asm volatile(T5_BODYX("", T5_LOADS128, " s_waitcnt lgkmcnt(0)\n" NOP64, "op_sel:[0,1,0]") T5_OPS);
        else if (T == 5) asm volatile(T5_BODY(" s_waitcnt lgkmcnt(1)\n") T5_OPS);
        else asm volatile(T5_BODY(" s_waitcnt lgkmcnt(0)\n") T5_OPS);
        c0 += c2; c1 += c3;
    }
    if (T >= 11 && T <= 15) {  // no LDS in the loop: v[100:103] are loaded once; only the operand selection differs
        f2 c2 = {0.f, 0.f}, c3 = {0.f, 0.f};
        a = (threadIdx.x * 16u) & 0xff0u;
#define T11_BODY(I0, I1) " ds_read_b128 v[100:103], %[a]\n s_waitcnt lgkmcnt(0)\n 1:\n" \
                     I0 " %[c0], %[x], v[100:101], %[c0] op_sel_hi:[1,0,1]\n" I1 "\n" \
                     I0 " %[c2], %[x], v[102:103], %[c2] op_sel_hi:[1,0,1]\n" I1 "\n" \
                     " s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
#define T11_OPS : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [a] "+v"(a), [n] "+s"(n) : [x] "v"(x) \
               : "v100", "v101", "v102", "v103", "scc", "memory"
        if (T == 11) asm volatile(T11_BODY(" v_pk_fma_f32", " v_pk_fma_f32 %[c1], %[x], v[100:101], %[c1] op_sel:[0,1,0]") T11_OPS);
        if (T == 12) asm volatile(T11_BODY(" v_pk_fma_f32", " v_pk_fma_f32 %[c1], v[100:101], %[x], %[c1] op_sel:[1,0,0]") T11_OPS);
        if (T == 13) asm volatile(T11_BODY(" v_pk_fma_f32", " v_pk_mul_f32 %[c3], %[x], v[100:101] op_sel:[0,1]\n v_pk_add_f32 %[c1], %[c3], %[c1]") T11_OPS);
        if (T == 15) {
            const f2 y2 = {0.5f, 0.25f};
            asm volatile(" ds_read_b128 v[100:103], %[a]\n s_waitcnt lgkmcnt(0)\n 1:\n"
                         " v_pk_fma_f32 %[c0], %[y], %[c0], v[100:101]\n"
                         " v_pk_fma_f32 %[c1], %[y], %[c1], v[100:101] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n"
                         " v_pk_fma_f32 %[c2], %[y], %[c2], v[102:103]\n"
                         " v_pk_fma_f32 %[c3], %[y], %[c3], v[102:103] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n"
                         " s_sub_u32 %[n], %[n], 1\n s_cmp_lg_u32 %[n], 0\n s_cbranch_scc1 1b\n"
                         : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [a] "+v"(a), [n] "+s"(n) : [y] "v"(y2)
                         : "v100", "v101", "v102", "v103", "scc", "memory");
        }
        if (T == 14) asm volatile(T11_BODY(" v_pk_fma_f32", " v_pk_fma_f32 %[c1], %[x], v[100:101], %[c1] op_sel:[0,1,0] op_sel_hi:[1,0,1]") T11_OPS);
        c0 += c2; c1 += c3;
    }
    float *o = out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 32;
    for (int k = 0; k < 32; ++k) o[k] = 0.f;
    o[0] = c0[0]; o[1] = c0[1]; o[2] = c1[0]; o[3] = c1[1];
}

static float *g_dev, *sink_dev;
static hipStream_t s0, s1;
static long compare(const char *name, const float *ref, const float *co, size_t n) {
    std::vector<float> a(n), b(n);
    CHECK(hipMemcpy(a.data(), ref, n * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(b.data(), co, n * 4, hipMemcpyDeviceToHost));
    long bad = 0, q[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; ++i) if (a[i] != b[i]) { ++bad; ++q[((i / 32) & 63) >> 4]; }
    printf("%-52s %9ld of %zu results differ; lanes 0-15 / 16-31 / 32-47 / 48-63: %ld %ld %ld %ld\n", name, bad, n, q[0], q[1], q[2], q[3]);
    return bad;
}
typedef void (*vk_t)(float *, const float *, int);
static void run_k(vk_t k, const char *name, int vic_iters, int agg_iters) {
    const int blocks = 2048;
    const size_t n = (size_t)blocks * 256 * 32;
    float *out_ref, *out_co;
    CHECK(hipMalloc(&out_ref, n * 4)); CHECK(hipMalloc(&out_co, n * 4));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, s0, out_ref, g_dev, vic_iters);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, s0, out_co, g_dev, vic_iters);
    CHECK(hipDeviceSynchronize());
    compare((std::string(name) + " | alone, 2nd run").c_str(), out_ref, out_co, n);
    for (int trial = 0; trial < 2; ++trial) {
        CHECK(hipMemsetAsync(out_co, 0, n * 4, s0));
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(aggressor, dim3(256), dim3(256), 0, s1, sink_dev, agg_iters);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, s0, out_co, g_dev, vic_iters);
        CHECK(hipDeviceSynchronize());
        compare((std::string(name) + " | beside the MFMA kernel").c_str(), out_ref, out_co, n);
    }
    CHECK(hipFree(out_ref)); CHECK(hipFree(out_co));
}

int main(int argc, char **argv) {
    const int vi = argc > 1 ? atoi(argv[1]) : 20000, ai = argc > 2 ? atoi(argv[2]) : 300000;
    CHECK(hipMalloc(&g_dev, 4096)); CHECK(hipMalloc(&sink_dev, 4096));
    std::vector<float> hg(1024);
    for (int i = 0; i < 1024; ++i) hg[i] = 1e-3f * (float)((i * 2654435761u) % 1000) - 0.4f;
    CHECK(hipMemcpy(g_dev, hg.data(), 4096, hipMemcpyHostToDevice));
    CHECK(hipStreamCreate(&s0)); CHECK(hipStreamCreate(&s1));
    run_k(victim<0>, "v_pk_fma_f32 (g from LDS, broadcast)", vi, ai);
    run_k(victim<1>, "2 x v_fma_f32", vi, ai);
    run_k(victim<2>, "v_pk_fma_f32 (g in registers)", vi, ai);
    run_k(victim<3>, "v_pk_fma_f32 (real pair)", vi, ai);
    run_k(victim<4>, "v_pk_mul_f32 + v_pk_add_f32", vi, ai);
    run_k(victim<5>, "v_pk_add_f32", vi, ai);
    run_k(victim_asm<1>, "asm T1: pk_fma reads v[100:101]; ds_read_b64 into them next", vi * 4, ai);
    run_k(victim_asm<2>, "asm T2: ds_read; wait; pk_fma reads at once; no early reuse", vi * 4, ai);
    run_k(victim_asm<3>, "asm T3: T1 + v_mov of the address between", vi * 4, ai);
    run_k(victim_asm<4>, "asm T4: T1 with 2 x v_fma_f32", vi * 4, ai);
    run_k(victim_asm<5>, "asm T5: 2 ds_read_b128 in flight, lgkmcnt(1), use 1st", vi * 4, ai);
    run_k(victim_asm<6>, "asm T6: the same with lgkmcnt(0)", vi * 4, ai);
    run_k(victim_asm<7>, "asm T7: T6 + 128 idle cycles BEFORE the loads", vi * 4, ai);
    run_k(victim_asm<10>, "asm T10: T6 + 64 idle cycles AFTER the wait", vi * 4, ai);
    run_k(victim_asm<8>, "asm T8: T6, only low-element broadcasts", vi * 4, ai);
    run_k(victim_asm<11>, "asm T11: registers only, op_sel:[0,1,0] (high element to both)", vi * 8, ai);
    run_k(victim_asm<12>, "asm T12: registers only, op_sel:[1,0,0] on src0", vi * 8, ai);
    run_k(victim_asm<13>, "asm T13: registers only, v_pk_mul_f32 op_sel:[0,1]", vi * 8, ai);
    run_k(victim_asm<14>, "asm T14: registers only, op_sel:[0,1,0] op_sel_hi:[1,0,1] (swap)", vi * 8, ai);
    run_k(victim_asm<15>, "asm T15: registers only, src2 swapped (op_sel:[0,0,1] op_sel_hi:[1,1,0])", vi * 8, ai);
    run_k(victim_asm<9>, "asm T9: T6 with 4 x ds_read_b64", vi * 4, ai);
    return 0;
}
