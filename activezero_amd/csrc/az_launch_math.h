// Host-side (and host/device) launch arithmetic of the tiled kernels, free of HIP types so that the SAME functions the
// launches use are also compiled with g++ -fsanitize=address,undefined and swept over shapes on the CPU
// (tests/host/launch_math_test.cpp, tests/test_launch_math_cpu.py; SURVEY.md 5.2: "a host check of the launch
// arithmetic").  Device code includes it too: AZ_HD marks what a kernel calls.
#pragma once
#if defined(__HIPCC__)
#define AZ_HD __host__ __device__ __forceinline__
#else
#define AZ_HD inline
#endif

// Workgroup -> work-item map shared by every tiled kernel: blocks b and b + 8 share an XCD (and its L2), so each XCD gets
// ONE contiguous chunk of the linear order.  A bijection of [0, nblk) for every nblk >= 1.
AZ_HD int az_xcd_map(int bid, int nblk) {
    const int xcd = bid & 7, q8 = nblk >> 3, r8 = nblk & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
}

// az_conv3d_roll.hip: depth segments of a launch -- one round of workgroups over the chip's 512 slots (256 CUs x 2) if the
// patches allow it, otherwise the split that minimises rounds x (planes walked per workgroup).  patches = B * 8-row patch
// rows * patch columns; forced > 0: that segment length (AZ_ROLL_SEGLEN).  Postconditions: nseg * seg_len >= Do,
// (nseg - 1) * seg_len < Do, 1 <= seg_len <= Do.
inline void az_roll_segments(long long patches, int Do, int forced, int &nseg, int &seg_len) {
    if (forced > 0) {
        seg_len = forced < Do ? forced : Do;
        nseg = (Do + seg_len - 1) / seg_len;
        return;
    }
    long long best = -1;
    nseg = 1; seg_len = Do;
    for (int n = 1; n <= Do; ++n) {
        const int len = (Do + n - 1) / n;
        const int nn = (Do + len - 1) / len;
        if (nn != n) continue;
        const long long rounds = (patches * n + 511) / 512;
        const long long cost = rounds * (len + 2) * 3 + 2;  // +: fixed cost per workgroup
        if (best < 0 || cost < best) { best = cost; nseg = n; seg_len = len; }
    }
}

// az_conv3d_t2roll.hip: coarse-depth segments of a launch with ONE workgroup per CU (256 slots); a workgroup runs seg_len full
// stages and a closing one of nine of the 27 taps.  Postconditions as az_roll_segments.
inline void az_t2roll_segments(long long patches, int Di, int &nseg, int &seg_len) {
    long long best = -1;
    nseg = 1; seg_len = Di;
    for (int n = 1; n <= Di; ++n) {
        const int len = (Di + n - 1) / n;
        if ((Di + len - 1) / len != n) continue;
        const long long rounds = (patches * n + 255) / 256;
        const long long cost = rounds * (len * 4 + 2) + 1;
        if (best < 0 || cost < best) { best = cost; nseg = n; seg_len = len; }
    }
}

// az_conv3d_s2roll.hip: coarse-depth segments of a launch with two workgroups per CU (512 slots); a workgroup of a segment
// of len output planes stages 2 len + 1 fine planes and multiplies 27 len tap positions; a staged plane is priced at four
// tap positions (measured at B = 4, V0 -> V1, 288 patches: 0.34 ms for len 4 .. 6, 0.36 at 3, 0.38 at 1 and 12, 0.46 at 24).  forced > 0: that segment length (AZ_S2ROLL_SEGLEN).  Postconditions as az_roll_segments.
inline void az_s2roll_segments(long long patches, int Do, int forced, int &nseg, int &seg_len, int slots = 512) {
    if (forced > 0) {
        seg_len = forced < Do ? forced : Do;
        nseg = (Do + seg_len - 1) / seg_len;
        return;
    }
    long long best = -1;
    nseg = 1; seg_len = Do;
    for (int n = 1; n <= Do; ++n) {
        const int len = (Do + n - 1) / n;
        if ((Do + len - 1) / len != n) continue;
        const long long rounds = (patches * n + slots - 1) / slots;
        const long long cost = rounds * (27LL * len + 4 * (2 * len + 1) + 4);
        if (best < 0 || cost < best) { best = cost; nseg = n; seg_len = len; }
    }
}

// az_conv2d_roll.hip: image segments per statistic group (patches = groups * patch rows * patch columns; N images)
inline void az_c2r_segments(long long patches, int N, int &nseg, int &seg_len) {
    long long best = -1;
    nseg = 1; seg_len = N;
    for (int n = 1; n <= N; ++n) {
        const int len = (N + n - 1) / n;
        if ((N + len - 1) / len != n) continue;
        const long long rounds = (patches * n + 511) / 512;
        const long long cost = rounds * (len * 3 + 2);  // stages per workgroup + its fixed cost
        if (best < 0 || cost < best) { best = cost; nseg = n; seg_len = len; }
    }
}

// linear index -> (patch column, 8-row patch row, depth segment, batch) of az_conv3d_roll.hip
AZ_HD void az_roll_decode(int lin, int tiles_x, int tyb, int nseg, int &tix, int &tiy, int &seg, int &b) {
    tix = lin % tiles_x; lin /= tiles_x;
    tiy = lin % tyb; lin /= tyb;
    seg = lin % nseg;
    b = lin / nseg;
}

// Slab staging of the rolling kernels: 10 x 18 voxels x 8 sixteen-byte pieces, 256 threads, piece q = tid + 256 it; thread
// tid's it-th piece is voxel (tid >> 3) + 32 it.  The kernels step (sy, sx) incrementally (32 voxels = one 18-voxel row +
// 14); this is the closed form the CPU test compares the stepping against.
#define AZ_R_SY 10
#define AZ_R_SX 18
#define AZ_R_NQ (AZ_R_SY * AZ_R_SX * 8)
AZ_HD bool az_roll_piece(int tid, int it, int &sy, int &sx) {
    const int vox = (tid >> 3) + 32 * it;
    sy = vox / AZ_R_SX;
    sx = vox - sy * AZ_R_SX;
    return tid + 256 * it < AZ_R_NQ;
}

// az_conv3d_wgrad16.hip: persistent workgroups per (m, n) tile -- at most `slots` resident (two per CU over the tiles), the
// count that balances the columns best; cap > 0: AZ_WGRAD_R16_WGS.  1 <= result <= max(1, min(slots, ncols)).
inline int az_wgrad16_workgroups(long long ncols, int slots, int ntiles, int cap) {
    int best = 1;
    double best_score = -1.0;
    const int step = 8 / (ntiles > 2 ? 4 : ntiles);
    for (int w = slots; w >= slots / 4 && w >= 1; w -= step) {
        if (w > ncols) continue;
        const long long per = (ncols + w - 1) / w;
        const double score = (double)ncols / (double)(per * slots);  // useful fraction of the chip-time taken
        if (score > best_score + 1e-9) { best_score = score; best = w; }
    }
    if (ncols < slots / 4) best = (int)(ncols > 0 ? ncols : 1);
    if (cap > 0 && cap < best) best = cap;
    return best;
}

// az_conv3d_wgrad16s2.hip (stride-2 weight gradient): a step stages 3 planes x 8 new fine rows x 17 positions x 8 float4
// pieces.  Piece f -> plane pk, row j, position pp (fine x = 2 cw0 - 1 + pp) and the LDS row of the position inside a
// staged row: the ODD fine positions (pp even) are rows 0..8, the EVEN ones rows 9..16, so that tap kw of coarse position x
// reads row x (kw = 0), 9 + x (kw = 1), 1 + x (kw = 2).
#define AZ_S2W_FPOS 17
#define AZ_S2W_FROWQ (AZ_S2W_FPOS * 8)
#define AZ_S2W_NFQ (3 * 8 * AZ_S2W_FROWQ)
#define AZ_S2W_RING 17
AZ_HD bool az_s2w_fine_piece(int f, int &pk, int &j, int &pp, int &lrow) {
    pk = f / (8 * AZ_S2W_FROWQ);
    const int g = f - pk * (8 * AZ_S2W_FROWQ);
    j = g / AZ_S2W_FROWQ;
    pp = (g - j * AZ_S2W_FROWQ) >> 3;
    lrow = (pp & 1) ? 9 + (pp >> 1) : (pp >> 1);
    return f < AZ_S2W_NFQ;
}
AZ_HD int az_s2w_tap_row(int kw) { return kw == 0 ? 0 : kw == 1 ? 9 : 1; }
// ring slot of fine row fr >= -1 (17 slots: the 9 rows a step reads + the 8 it writes for the next one)
AZ_HD int az_s2w_ring_slot(int fr) { return (fr + 1) % AZ_S2W_RING; }
// column -> (coarse depth, 8-position chunk, batch); consecutive columns = consecutive depths of one chunk
AZ_HD void az_s2w_col_decode(long long col, int Dc, int nwchunk, int &cd, int &wc, int &b) {
    cd = (int)(col % Dc); col /= Dc;
    wc = (int)(col % nwchunk);
    b = (int)(col / nwchunk);
}

// a tensor slice addressed through one 32-bit buffer offset must stay below the out-of-range marker the kernels use
inline bool az_fits_buffer_offset(long long bytes) { return bytes > 0 && bytes < 0xffffff00LL; }
