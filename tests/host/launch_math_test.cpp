// CPU sweep of the launch arithmetic the HIP launches use (activezero_amd/csrc/az_launch_math.h), built with
// g++ -fsanitize=address,undefined by tests/test_launch_math_cpu.py.  Exit code 0 = every property held.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../activezero_amd/csrc/az_launch_math.h"

static int fails = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (++fails < 20) { std::printf("FAIL %s:%d: %s  ", __FILE__, __LINE__, #cond); std::printf(__VA_ARGS__); std::printf("\n"); } } } while (0)

int main() {
    // 1. the XCD map is a bijection of [0, nblk)
    for (int nblk = 1; nblk <= 5000; nblk += (nblk < 600 ? 1 : 37)) {
        std::vector<char> seen(nblk, 0);
        for (int b = 0; b < nblk; ++b) {
            const int l = az_xcd_map(b, nblk);
            CHECK(l >= 0 && l < nblk, "nblk %d bid %d -> %d", nblk, b, l);
            if (l >= 0 && l < nblk) { CHECK(!seen[l], "nblk %d: %d hit twice", nblk, l); seen[l] = 1; }
        }
    }
    // 2. depth segments cover [0, Do) exactly; every workgroup decodes to a valid (patch, segment, batch), all distinct
    const int dims[][4] = {{1, 1, 1, 1}, {4, 48, 136, 240}, {4, 24, 68, 120}, {1, 3, 9, 33}, {2, 5, 7, 19}, {8, 1, 272, 480},
                           {1, 48, 136, 240}, {3, 47, 131, 239}, {6, 12, 34, 60}, {1, 192, 8, 16}, {16, 2, 300, 17}};
    for (const auto &d : dims)
        for (int forced = 0; forced <= 7; forced += (forced ? 3 : 1)) {
            const int B = d[0], Do = d[1], H = d[2], W = d[3];
            const int tiles_y = (H + 3) / 4, tiles_x = (W + 15) / 16, tyb = (tiles_y + 1) / 2;
            int nseg = -1, seg_len = -1;
            az_roll_segments((long long)B * tyb * tiles_x, Do, forced, nseg, seg_len);
            CHECK(nseg >= 1 && seg_len >= 1 && seg_len <= Do, "B%d Do%d: nseg %d len %d", B, Do, nseg, seg_len);
            CHECK((long long)nseg * seg_len >= Do && (long long)(nseg - 1) * seg_len < Do, "Do %d nseg %d len %d", Do, nseg, seg_len);
            const long long blocks = (long long)B * nseg * tyb * tiles_x;
            if (blocks > 200000) continue;
            std::vector<char> seen((size_t)blocks, 0);
            for (int bid = 0; bid < (int)blocks; ++bid) {
                int tix, tiy, seg, b;
                az_roll_decode(az_xcd_map(bid, (int)blocks), tiles_x, tyb, nseg, tix, tiy, seg, b);
                CHECK(tix >= 0 && tix < tiles_x && tiy >= 0 && tiy < tyb && seg >= 0 && seg < nseg && b >= 0 && b < B,
                      "decode (%d,%d,%d,%d) of (%d,%d,%d,%d)", tix, tiy, seg, b, tiles_x, tyb, nseg, B);
                const long long id = (((long long)b * nseg + seg) * tyb + tiy) * tiles_x + tix;
                if (id >= 0 && id < blocks) { CHECK(!seen[(size_t)id], "work item %lld twice", id); seen[(size_t)id] = 1; }
                const int d0 = seg * seg_len, d1 = d0 + seg_len < Do ? d0 + seg_len : Do;
                CHECK(d0 < d1 && d1 <= Do, "segment [%d, %d) of %d", d0, d1, Do);
            }
        }
    // 3. image segments of the 2-D batch-walking kernel
    for (int N = 1; N <= 64; ++N)
        for (long long patches = 1; patches <= 4096; patches *= 3) {
            int nseg = -1, len = -1;
            az_c2r_segments(patches, N, nseg, len);
            CHECK(nseg >= 1 && len >= 1 && (long long)nseg * len >= N && (long long)(nseg - 1) * len < N, "N %d: %d x %d", N, nseg, len);
        }
    // 4. slab staging: the kernels' incremental (sy, sx) stepping equals the closed form and stays inside the slab
    for (int tid = 0; tid < 256; ++tid) {
        int sy = 0, sx = tid >> 3;
        if (sx >= AZ_R_SX) { sx -= AZ_R_SX; ++sy; }
        for (int it = 0; it < (AZ_R_NQ + 255) / 256; ++it) {
            int cy, cx;
            const bool live = az_roll_piece(tid, it, cy, cx);
            if (live) {
                CHECK(cy == sy && cx == sx, "tid %d it %d: stepped (%d,%d) closed form (%d,%d)", tid, it, sy, sx, cy, cx);
                CHECK(cy >= 0 && cy < AZ_R_SY && cx >= 0 && cx < AZ_R_SX, "tid %d it %d outside the slab", tid, it);
            }
            sx += 14; ++sy;
            if (sx >= AZ_R_SX) { sx -= AZ_R_SX; ++sy; }
        }
    }
    // every live piece index is hit exactly once
    {
        std::vector<char> seen(AZ_R_NQ, 0);
        for (int tid = 0; tid < 256; ++tid)
            for (int it = 0; it < (AZ_R_NQ + 255) / 256; ++it) {
                int cy, cx;
                if (!az_roll_piece(tid, it, cy, cx)) continue;
                const int q = (cy * AZ_R_SX + cx) * 8 + (tid & 7);
                CHECK(q >= 0 && q < AZ_R_NQ && !seen[q], "piece %d twice or outside", q);
                if (q >= 0 && q < AZ_R_NQ) seen[q] = 1;
            }
        for (int q = 0; q < AZ_R_NQ; ++q) CHECK(seen[q], "piece %d never staged", q);
    }
    // 5. persistent workgroups of the weight-gradient kernel
    for (long long ncols = 1; ncols <= 100000; ncols = ncols * 5 / 3 + 1)
        for (int ntiles = 1; ntiles <= 4; ntiles *= 2)
            for (int cap = 0; cap <= 300; cap += 150) {
                const int slots = 512 / ntiles;
                const int w = az_wgrad16_workgroups(ncols, slots, ntiles, cap);
                CHECK(w >= 1 && w <= slots && (w <= ncols || ncols < 1), "ncols %lld slots %d -> %d", ncols, slots, w);
                if (cap > 0) CHECK(w <= cap, "cap %d -> %d", cap, w);
            }
    // 6. the 32-bit buffer-offset guard
    CHECK(az_fits_buffer_offset(1) && az_fits_buffer_offset(0xfffffeffLL) && !az_fits_buffer_offset(0xffffff00LL) &&
          !az_fits_buffer_offset(1LL << 33) && !az_fits_buffer_offset(0), "buffer offset guard");
    std::printf("launch math: %d failures\n", fails);
    return fails ? 1 : 0;
}
