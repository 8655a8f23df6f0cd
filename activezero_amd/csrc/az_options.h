// The library's A/B switches (DESIGN.md section 4, table "A/B switches"): environment variables read ONCE, when the first
// call needs one, into one immutable struct -- no getenv in a launch path, no unsynchronised first-write caches
// (C++11 guarantees the initialisation of the function-local static in az_options() happens exactly once, under a lock).
#pragma once

struct AzOptions {
    int bn_bwd_fused;      // AZ_BN_BWD_FUSED      1: BatchNorm backward in two launches (reduce, apply + merge); 0: three
    int conv2d_roll_nt4;   // AZ_CONV2D_ROLL_NT4   1: 64 output channels as four N tiles per wave in az_conv2d_roll.hip
    int conv2d_wgrad_r16;  // AZ_CONV2D_WGRAD_R16  1: 3x3 32/64-channel 2-D weight gradients on az_conv2d_wgrad16.hip
    int conv2d_wgrad_w64;  // AZ_CONV2D_WGRAD_W64  1: its f16x3 64 -> 64 layers as one 64 x 64 tile per eight-wave workgroup (round 5)
    int conv_m128;         // AZ_CONV_M128         1: bf16x6 (precision 1) stride-1 32-output layers on az_conv3d_m128.hip
    int conv_map;          // AZ_CONV_MAP          block -> tile map of the 3-D kernels: 0 linear, 1 XCD-chunked, 2 + banded
    int roll_seglen;       // AZ_ROLL_SEGLEN       > 0: depth-segment length of az_conv3d_roll.hip (0: chosen per launch)
    int wgrad_slots;       // AZ_WGRAD_SLOTS       resident waves a 3-D one-kd-per-wave weight gradient may take
    int wgrad_order;       // AZ_WGRAD_ORDER       work-list order of those kernels (1: XCD-chunked, depth fastest)
    int wgrad_fw;          // AZ_WGRAD_FW          1: stride-1 bf16x6 fallback kernel walks fine rows
    int wgrad_r16;         // AZ_WGRAD_R16         0 / 1 / 2: stride-1 3-D weight gradients on az_conv3d_wgrad16.hip (none / 32x32 / all)
    int wgrad_r16_wgs;     // AZ_WGRAD_R16_WGS     > 0: cap on that kernel's persistent workgroups
    int conv2d_roll_h;     // AZ_CONV2D_ROLL_H     1: f16x3 2-D layers with 64 output channels on conv2d_roll64_kernel (half channels x half patch per wave)
    int conv_t2roll;       // AZ_CONV_T2ROLL       1: f16x3 transposed 64 -> 32 layers on az_conv3d_t2roll.hip
    int s2roll_seglen;     // AZ_S2ROLL_SEGLEN     > 0: output planes per depth segment of az_conv3d_s2roll.hip (0: chosen per launch)
    int conv_roll64;       // AZ_CONV_ROLL64       1: f16x3 stride-1 64 -> 64 layers on az_conv3d_roll.hip, two workgroups per patch (0: az_conv3d.hip's gather kernel)
    int conv_s2roll;       // AZ_CONV_S2ROLL       1: f16x3 stride-2 32 -> 64 layers on az_conv3d_s2roll.hip (0: az_conv3d.hip's gather kernel)
    int wgrad_r16_xcd;     // AZ_WGRAD_R16_XCD     1: that kernel's columns in XCD-contiguous runs
    int wgrad_r16_wide;    // AZ_WGRAD_R16_WIDE    1: its f16x3 form with the taps split over the waves on 32x32x16 tiles (round 5; default 0: not faster, az_conv3d_wgrad16.hip)
    int wgrad_s2r16;       // AZ_WGRAD_S2R16       1: f16x3 stride-2 3-D weight gradients on az_conv3d_wgrad16s2.hip
    int corr_fp32;         // AZ_CORR_FP32         1: RAFT correlation GEMMs on the fp32-MFMA kernel
    int patch_tiled;       // AZ_PATCH_TILED       1: band-tiled patch-reprojection kernel
    int patch_k;           // AZ_PATCH_K           pixels per thread of that kernel
};

const AzOptions &az_options();
