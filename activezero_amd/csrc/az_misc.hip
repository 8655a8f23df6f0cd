#include <stdlib.h>

#include "az_pack_f16.h"
#include "az_options.h"

extern "C" const char *az_strerror(int code) {
    switch (code) {
        case AZ_OK: return "AZ_OK";
        case AZ_EINVAL: return "AZ_EINVAL";
        case AZ_ENULL: return "AZ_ENULL";
        case AZ_ELAUNCH: return "AZ_ELAUNCH";
        case AZ_EUNSUPPORTED: return "AZ_EUNSUPPORTED";
        case AZ_EWORKSPACE: return "AZ_EWORKSPACE";
        default: return "AZ_E?";
    }
}

extern "C" int az_abi_version(void) { return AZ_ABI_VERSION; }

// Measurement only (bench.py roofline.measured_hbm; SURVEY.md 8d's second denominator beside the 8 TB/s spec figure): a
// float4 grid-stride copy, 16 bytes per lane per access -- the stream MI355X_MICROARCH.md quotes 6.29 TB/s for.
typedef float az_v4 __attribute__((ext_vector_type(4)));
// variant 0: one float4 per thread, one block per 4 KB (no loop: the dispatcher keeps every CU supplied with short blocks);
// variant 1: 4096 blocks, grid-stride loop (the form of this library's streaming kernels)
template <int LOOP>
__global__ void __launch_bounds__(256)
hbm_copy_kernel(az_v4 *__restrict__ dst, const az_v4 *__restrict__ src, long long n4) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (!LOOP) {
        if (i < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
        return;
    }
    for (; i < n4; i += (long long)gridDim.x * 256) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
extern "C" int az_hbm_copy_probe(float *dst, const float *src, long long n, void *stream) {
    AZ_REQUIRE_PTR(dst); AZ_REQUIRE_PTR(src);
    AZ_REQUIRE(n > 0 && n % 4 == 0);
    if ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) return AZ_EINVAL;
    const long long n4 = n / 4, blocks = (n4 + 255) / 256;
    if (blocks <= 0x7fffffffLL)
        hipLaunchKernelGGL(hbm_copy_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, az_stream(stream), reinterpret_cast<az_v4 *>(dst),
                           reinterpret_cast<const az_v4 *>(src), n4);
    else
        hipLaunchKernelGGL(hbm_copy_kernel<1>, dim3(4096), dim3(256), 0, az_stream(stream), reinterpret_cast<az_v4 *>(dst),
                           reinterpret_cast<const az_v4 *>(src), n4);
    return az_launch_status();
}

static int az_env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
const AzOptions &az_options() {
    static const AzOptions o = [] {
        AzOptions v;
        v.bn_bwd_fused = az_env_int("AZ_BN_BWD_FUSED", 1);
        v.conv2d_roll_nt4 = az_env_int("AZ_CONV2D_ROLL_NT4", 1);
        v.conv2d_wgrad_r16 = az_env_int("AZ_CONV2D_WGRAD_R16", 1);
        v.conv2d_wgrad_w64 = az_env_int("AZ_CONV2D_WGRAD_W64", 1);
        v.conv_m128 = az_env_int("AZ_CONV_M128", 1);
        v.conv_map = az_env_int("AZ_CONV_MAP", 2);
        if (v.conv_map < 0 || v.conv_map > 3) v.conv_map = 2;
        v.roll_seglen = az_env_int("AZ_ROLL_SEGLEN", 0);
        v.wgrad_slots = az_env_int("AZ_WGRAD_SLOTS", 256 * 8);
        v.wgrad_order = az_env_int("AZ_WGRAD_ORDER", 1);
        v.wgrad_fw = az_env_int("AZ_WGRAD_FW", 1);
        v.wgrad_r16 = az_env_int("AZ_WGRAD_R16", 2);
        v.wgrad_r16_wgs = az_env_int("AZ_WGRAD_R16_WGS", 0);
        v.wgrad_s2r16 = az_env_int("AZ_WGRAD_S2R16", 1);
        v.wgrad_r16_xcd = az_env_int("AZ_WGRAD_R16_XCD", 1);
        v.wgrad_r16_wide = az_env_int("AZ_WGRAD_R16_WIDE", 0);
        v.conv_t2roll = az_env_int("AZ_CONV_T2ROLL", 1);
        v.conv_s2roll = az_env_int("AZ_CONV_S2ROLL", 1);
        v.conv_roll64 = az_env_int("AZ_CONV_ROLL64", 1);
        v.s2roll_seglen = az_env_int("AZ_S2ROLL_SEGLEN", 0);
        v.conv2d_roll_h = az_env_int("AZ_CONV2D_ROLL_H", 1);
        v.corr_fp32 = az_env_int("AZ_CORR_FP32", 0);
        v.patch_tiled = az_env_int("AZ_PATCH_TILED", 1);
        v.patch_k = az_env_int("AZ_PATCH_K", 4);
        return v;
    }();
    return o;
}
// the value of one switch as the library reads it (tests; name = the environment variable); AZ_EINVAL: no such switch
extern "C" int az_option(const char *name) {
    AZ_REQUIRE_PTR(name);
    const AzOptions &o = az_options();
    struct { const char *n; int v; } t[] = {
        {"AZ_BN_BWD_FUSED", o.bn_bwd_fused}, {"AZ_CONV2D_ROLL_NT4", o.conv2d_roll_nt4}, {"AZ_CONV2D_WGRAD_R16", o.conv2d_wgrad_r16}, {"AZ_CONV2D_WGRAD_W64", o.conv2d_wgrad_w64},
        {"AZ_CONV_M128", o.conv_m128}, {"AZ_CONV_MAP", o.conv_map}, {"AZ_ROLL_SEGLEN", o.roll_seglen},
        {"AZ_WGRAD_SLOTS", o.wgrad_slots}, {"AZ_WGRAD_ORDER", o.wgrad_order}, {"AZ_WGRAD_FW", o.wgrad_fw},
        {"AZ_WGRAD_R16", o.wgrad_r16}, {"AZ_WGRAD_R16_WGS", o.wgrad_r16_wgs}, {"AZ_WGRAD_S2R16", o.wgrad_s2r16}, {"AZ_WGRAD_R16_XCD", o.wgrad_r16_xcd}, {"AZ_WGRAD_R16_WIDE", o.wgrad_r16_wide}, {"AZ_CONV_T2ROLL", o.conv_t2roll}, {"AZ_CONV_S2ROLL", o.conv_s2roll}, {"AZ_CONV_ROLL64", o.conv_roll64}, {"AZ_S2ROLL_SEGLEN", o.s2roll_seglen}, {"AZ_CONV2D_ROLL_H", o.conv2d_roll_h}, {"AZ_CORR_FP32", o.corr_fp32},
        {"AZ_PATCH_TILED", o.patch_tiled}, {"AZ_PATCH_K", o.patch_k}};
    for (const auto &e : t) {
        const char *a = e.n, *b = name;
        while (*a && *a == *b) { ++a; ++b; }
        if (!*a && !*b) return e.v;
    }
    return AZ_EINVAL;
}

// ---- every f16x3 weight image of a model in one launch (include/azhip.h, az_pack_f16.h) -----------------------------------
__global__ void __launch_bounds__(256)
pack_f16_multi_kernel(const AzPackDesc *__restrict__ descs, const int *__restrict__ block_desc, const int *__restrict__ first_block) {
    const int di = block_desc[blockIdx.x];
    const AzPackDesc d = descs[di];  // (block-uniform: scalar loads)
    const float scale = az_pow2(az_f16_scale_exp(az_amax_read(d.amax)));
    const long long idx = (long long)(blockIdx.x - first_block[di]) * 256 + threadIdx.x;
    if (idx >= az_pack_f16_total(d)) return;
    reinterpret_cast<unsigned short *>(d.dst)[idx] = az_pack_f16_elem(d, idx, scale);
}

extern "C" int az_pack_f16_multi(const AzPackDesc *descs, const int *block_desc, const int *first_block, int nd, int nblocks,
                                 void *stream) {
    AZ_REQUIRE_PTR(descs); AZ_REQUIRE_PTR(block_desc); AZ_REQUIRE_PTR(first_block);
    AZ_REQUIRE(nd > 0 && nblocks > 0);
    hipLaunchKernelGGL(pack_f16_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, az_stream(stream), descs, block_desc, first_block);
    return az_launch_status();
}

// ---- the weight-gradient workspaces of a whole backward pass -> PyTorch-layout gradients, one launch (include/azhip.h) ------
// ws [taps][cm][cn] -> dst[(co * cn_real + ci) * taps + t], co < cm_real, ci < cn_real (wgrad_unpack_kernel / wgrad2d_unpack_kernel)
__global__ void __launch_bounds__(256)
wgrad_unpack_multi_kernel(const AzUnpackDesc *__restrict__ descs, const int *__restrict__ block_desc, const int *__restrict__ first_block) {
    const int di = block_desc[blockIdx.x];
    const AzUnpackDesc d = descs[di];
    const int idx = (blockIdx.x - first_block[di]) * 256 + threadIdx.x;
    if (idx >= d.cm_real * d.cn_real * d.taps) return;
    const int t = idx % d.taps, mn = idx / d.taps;
    const int ci = mn % d.cn_real, co = mn / d.cn_real;
    d.dst[idx] = d.ws[((size_t)t * d.cm + co) * d.cn + ci];
}

extern "C" int az_wgrad_unpack_multi(const AzUnpackDesc *descs, const int *block_desc, const int *first_block, int nd, int nblocks,
                                     void *stream) {
    AZ_REQUIRE_PTR(descs); AZ_REQUIRE_PTR(block_desc); AZ_REQUIRE_PTR(first_block);
    AZ_REQUIRE(nd > 0 && nblocks > 0);
    hipLaunchKernelGGL(wgrad_unpack_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, az_stream(stream), descs, block_desc, first_block);
    return az_launch_status();
}
