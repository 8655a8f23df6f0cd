// The four packed f16x3 weight images of the library as ONE index function (round 5: az_pack_f16_multi packs every weight
// of a model -- forward and input-gradient images -- in one launch; the per-tensor entry points call the same function):
//   AZ_PACK_2D_SAME   az_conv2d.hip        [tap][cin/16][cout/32][part 2][lane 64][8]               (zero-padded channels)
//   AZ_PACK_2D_ROLL   az_conv2d_roll.hip   [cout/32][tap 9][cin/32][n16 2][part 2][lane 64][8]
//   AZ_PACK_3D_GATHER az_conv3d.hip        [tap 27][cin/32][cout/32][part 2][kb 2][lane 64][8]
//   AZ_PACK_3D_ROLL   az_conv3d_roll.hip   [tap 27][cin/32][cout/16][part 2][lane 64][8]
//   AZ_PACK_3D_ROLL2  az_conv3d_roll.hip   [cout/32][tap 27][cin/32][n16 2][part 2][lane 64][8]   (64 output channels: one image per half)
// every element = part p (0: hi, 1: lo) of  w[co * s_co + ci * s_ci + (flip ? taps - 1 - tap : tap)] * 2^k  as fp16,
// k = az_f16_scale_exp(max |w|).
#pragma once
#include "az_roll_common.h"

__device__ __forceinline__ unsigned short az_pack_f16_elem(const AzPackDesc &d, long long idx, float scale) {
    const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    long long r = idx >> 9;
    int p, co, ci, tap;
    if (d.kind == AZ_PACK_2D_SAME) {
        p = (int)(r & 1); r >>= 1;
        const int nt = d.cout / 32, nch = d.cin / 16;
        const int n = (int)(r % nt); r /= nt;
        const int cc = (int)(r % nch);
        tap = (int)(r / nch);
        co = n * 32 + (lane & 31); ci = cc * 16 + 8 * (lane >> 5) + j;
    } else if (d.kind == AZ_PACK_2D_ROLL) {
        p = (int)(r & 1); r >>= 1;
        const int n = (int)(r & 1); r >>= 1;
        const int nch = d.cin / 32;
        const int cc = (int)(r % nch); r /= nch;
        tap = (int)(r % 9);
        const int cg = (int)(r / 9);
        co = cg * 32 + n * 16 + (lane & 15); ci = cc * 32 + 8 * (lane >> 4) + j;
    } else if (d.kind == AZ_PACK_3D_GATHER) {
        const int kb = (int)(r & 1); r >>= 1;
        p = (int)(r & 1); r >>= 1;
        const int nr = d.cout / 32, nch = d.cin / 32;
        const int n = (int)(r % nr); r /= nr;
        const int cc = (int)(r % nch);
        tap = (int)(r / nch);
        co = n * 32 + (lane & 31); ci = cc * 32 + 16 * kb + 8 * (lane >> 5) + j;
    } else if (d.kind == AZ_PACK_3D_ROLL2) {
        p = (int)(r & 1); r >>= 1;
        const int n = (int)(r & 1); r >>= 1;
        const int nch = d.cin / 32;
        const int cc = (int)(r % nch); r /= nch;
        tap = (int)(r % d.taps);
        const int half = (int)(r / d.taps);
        co = half * 32 + n * 16 + (lane & 15); ci = cc * 32 + 8 * (lane >> 4) + j;
    } else {  // AZ_PACK_3D_ROLL
        p = (int)(r & 1); r >>= 1;
        const int nn = d.cout / 16, nch = d.cin / 32;
        const int n = (int)(r % nn); r /= nn;
        const int cc = (int)(r % nch);
        tap = (int)(r / nch);
        co = n * 16 + (lane & 15); ci = cc * 32 + 8 * (lane >> 4) + j;
    }
    float x = 0.f;
    if (co < d.co_real && ci < d.ci_real) x = d.src[co * d.s_co + ci * d.s_ci + (d.flip ? d.taps - 1 - tap : tap)];
    return az_split2_f16_part(x * scale, p);
}

// elements of the image of a descriptor (two fp16 parts per weight of the PADDED operation)
__host__ __device__ static inline long long az_pack_f16_total(const AzPackDesc &d) { return 2LL * d.taps * d.cin * d.cout; }
