// Launch arguments shared by the 3x3x3 convolution kernels (az_conv3d.hip, az_conv3d_m128.hip).
#pragma once
#include "az_common.h"

struct ConvArgs {
    const float *in;    // NDHWC input (coarse tensor for MODE 2); SRC 1: left features NHWC
    const float *in2;   // SRC 1: right features NHWC
    const float *wp;    // packed weights
    float *out;         // NDHWC output
    const float *scale, *shift, *res;  // EPI 0 (any may be null)
    float *part;        // EPI 1: [COUT][tiles][2] (channel-major: finalize reads it coalesced)
    long long ntiles;
    float *cnt;         // EPI 1: [tiles]
    int B, Di, Hi, Wi;  // input dims
    int Do, Ho, Wo;     // output dims
    int Dt, tiles_y, tiles_x;  // index-space extents (MODE 2: coarse dims / phase tiles)
    int relu;
    int map_mode;  // block->tile map: 0 linear, 1 XCD-chunked linear, 2 XCD-chunked + banded
    int nseg, seg_len;  // az_conv3d_roll.hip: depth segments per patch, output depths per segment
    // f16x3 launches (az_conv3d_bwd_f16): device scalars holding max |in| and max |w| (az_absmax); the kernels derive
    // the power-of-two scales from them (az_f16_scale_exp) -- the packed weights are already scaled
    const float *in_amax, *w_amax;
    int in_split;  // f16x3: `in` is a pre-split tensor (az_roll_common.h, include/azhip.h "S2 format"); in_amax = its producer's bound
};

// bf16x6, stride-1, 32 output channels, 8x16-voxel tile per wave (az_conv3d_m128.hip)
int az_conv3d_m128_launch(const ConvArgs &a, int cin, int epi, int src, hipStream_t s);

// bf16x6 on the 16x16x32 MFMA, stride 1, 32 output channels, depth-rolling workgroups (az_conv3d_roll.hip);
// weights in that kernel's own packed layout (az_conv3d_pack_r16 = az_conv3d_pack_weights precision 2)
int az_conv3d_roll_launch(const ConvArgs &a, int cin, int epi, hipStream_t s);
// the same kernel on the f16x3 arithmetic (az_roll_common.h): input gradients; a.in_amax / a.w_amax set, weights packed
// by az_conv3d_pack_r16_f16
int az_conv3d_roll_launch_f16(const ConvArgs &a, int cin, int epi, hipStream_t s, int cout = 32);
int az_conv3d_pack_r16_f16(float *packed, const float *w, const float *w_amax, int cin, int cout, long long stride_out,
                           long long stride_in, int flip, hipStream_t s, int halves = 1);
long long az_conv3d_roll_stats_tiles(const ConvArgs &a, int cout = 32);  // rows of the BatchNorm partial buffers of an EPI-1 launch
int az_conv3d_pack_r16(float *packed, const float *w, int cin, int cout, long long stride_out, long long stride_in,
                       int flip, hipStream_t s);

// f16x3, stride-2 transposed, 64 -> 32 channels, depth-rolling workgroups of eight waves (az_conv3d_t2roll.hip); weights
// packed by az_conv3d_pack_r16_f16(cin = 64, cout = 32)
int az_conv3d_t2roll_launch(ConvArgs a, int epi, hipStream_t s);
long long az_conv3d_t2roll_stats_tiles(const ConvArgs &a);

// f16x3, stride 2, 32 -> 64 channels, depth-rolling workgroups of four waves walking the FINE depth (az_conv3d_s2roll.hip);
// weights packed by az_conv3d_pack_r16_f16(cin = 32, cout = 64)
int az_conv3d_s2roll_launch(ConvArgs a, int epi, hipStream_t s);
long long az_conv3d_s2roll_stats_tiles(ConvArgs a);
bool az_conv3d_s2roll_fits(const ConvArgs &a);

// bf16x6, stride-2 transposed, 32 output channels: one workgroup owns all 8 output-parity phases of a coarse
// patch (az_conv3d_t2.hip)
int az_conv3d_t2_launch(const ConvArgs &a, int cin, int epi, hipStream_t s);
