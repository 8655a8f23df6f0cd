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
    // 2b. coarse-depth segments of the transposed depth-rolling kernel; its workgroups decode like the stride-1 kernel's
    for (const auto &d : dims) {
        const int B = d[0], Di = d[1], H = d[2], W = d[3];
        const int tiles_y = (H + 3) / 4, tiles_x = (W + 15) / 16, tyb = (tiles_y + 1) / 2;
        int nseg = -1, seg_len = -1;
        az_t2roll_segments((long long)B * tyb * tiles_x, Di, nseg, seg_len);
        CHECK(nseg >= 1 && seg_len >= 1 && seg_len <= Di && (long long)nseg * seg_len >= Di && (long long)(nseg - 1) * seg_len < Di,
              "t2roll B%d Di%d: nseg %d len %d", B, Di, nseg, seg_len);
        for (int seg = 0; seg < nseg; ++seg) {  // every segment owns at least one coarse plane (the kernel's c0 < c1)
            const int c0 = seg * seg_len, c1 = c0 + seg_len < Di ? c0 + seg_len : Di;
            CHECK(c0 < c1, "t2roll empty segment %d of %d (len %d, Di %d)", seg, nseg, seg_len, Di);
        }
    }
    // 2c. output-depth segments of the stride-2 depth-rolling kernel (8 x 16 coarse patches); its workgroups decode with the
    //     same function, every (patch, segment, batch) once
    for (const auto &d : dims)
        for (int forced = 0; forced <= 7; forced += (forced ? 3 : 1)) {
            const int B = d[0], Do = (d[1] - 1) / 2 + 1, Ho = (d[2] - 1) / 2 + 1, Wo = (d[3] - 1) / 2 + 1;
            const int ty = (Ho + 7) / 8, tx = (Wo + 15) / 16;
            int nseg = -1, seg_len = -1;
            az_s2roll_segments((long long)B * ty * tx, Do, forced, nseg, seg_len);
            CHECK(nseg >= 1 && seg_len >= 1 && seg_len <= Do && (long long)nseg * seg_len >= Do && (long long)(nseg - 1) * seg_len < Do,
                  "s2roll B%d Do%d: nseg %d len %d", B, Do, nseg, seg_len);
            const long long blocks = (long long)B * nseg * ty * tx;
            std::vector<char> seen((size_t)blocks, 0);
            for (int bid = 0; bid < (int)blocks; ++bid) {
                int tix, tiy, seg, b;
                az_roll_decode(az_xcd_map(bid, (int)blocks), tx, ty, nseg, tix, tiy, seg, b);
                CHECK(tix >= 0 && tix < tx && tiy >= 0 && tiy < ty && seg >= 0 && seg < nseg && b >= 0 && b < B, "s2roll decode");
                const long long id = (((long long)b * nseg + seg) * ty + tiy) * tx + tix;
                if (id >= 0 && id < blocks) { CHECK(!seen[(size_t)id], "s2roll work item %lld twice", id); seen[(size_t)id] = 1; }
                CHECK(seg * seg_len < Do, "s2roll empty segment %d", seg);
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
    // 7. stride-2 weight gradient (az_conv3d_wgrad16s2.hip): staging pieces, tap rows, ring slots, column walk, offsets
    {
        // every (plane, row, position, float4) of a step's fine set is one piece; the LDS rows of a staged row are a permutation
        std::vector<char> seen(AZ_S2W_NFQ, 0);
        for (int f = 0; f < AZ_S2W_NFQ + 600; ++f) {
            int pk, j, pp, lrow;
            const bool live = az_s2w_fine_piece(f, pk, j, pp, lrow);
            CHECK(live == (f < AZ_S2W_NFQ), "piece %d liveness", f);
            if (!live) continue;
            CHECK(pk >= 0 && pk < 3 && j >= 0 && j < 8 && pp >= 0 && pp < AZ_S2W_FPOS && lrow >= 0 && lrow < AZ_S2W_FPOS, "piece %d -> %d %d %d %d", f, pk, j, pp, lrow);
            const int id = ((pk * 8 + j) * AZ_S2W_FPOS + pp) * 8 + (f & 7);
            CHECK(id == f && !seen[f], "piece %d is not its own index (%d)", f, id);
            seen[f] = 1;
        }
        char rows[AZ_S2W_FPOS] = {0};
        for (int pp = 0; pp < AZ_S2W_FPOS; ++pp) {
            int pk, j, p2, lrow;
            az_s2w_fine_piece(pp * 8, pk, j, p2, lrow);
            CHECK(p2 == pp && !rows[lrow], "LDS row %d of position %d taken", lrow, pp);
            rows[lrow] = 1;
            // tap kw of coarse position x reads fine position offset pp = 2x + kw: its row must be tap_row(kw) + x
            for (int kw = 0; kw < 3; ++kw)
                if ((pp - kw) >= 0 && (pp - kw) % 2 == 0 && (pp - kw) / 2 < 8)
                    CHECK(lrow == az_s2w_tap_row(kw) + (pp - kw) / 2, "pp %d kw %d: row %d", pp, kw, lrow);
        }
        // ring: the 9 rows step s reads and the 8 rows it writes for step s+1 sit in 17 distinct slots
        for (int s2 = 0; s2 < 200; ++s2) {
            char slot[AZ_S2W_RING] = {0};
            for (int fr = 8 * s2 - 1; fr <= 8 * s2 + 15; ++fr) {
                const int sl = az_s2w_ring_slot(fr);
                CHECK(sl >= 0 && sl < AZ_S2W_RING && !slot[sl], "step %d row %d: slot %d", s2, fr, sl);
                if (sl >= 0 && sl < AZ_S2W_RING) slot[sl] = 1;
            }
            // the kernel's per-lane form: (8s mod 17 + 2 oct + kh), minus 17 once
            for (int oct = 0; oct < 4; ++oct)
                for (int kh = 0; kh < 3; ++kh) {
                    int v = (8 * s2) % AZ_S2W_RING + 2 * oct + kh;
                    v = v >= AZ_S2W_RING ? v - AZ_S2W_RING : v;
                    CHECK(v == az_s2w_ring_slot(8 * s2 + 2 * oct - 1 + kh), "lane slot s %d oct %d kh %d", s2, oct, kh);
                }
        }
        // columns: the persistent workgroups (with the XCD map) visit every column once; global offsets of valid pieces,
        // formed the way the kernel forms them (32-bit wrap-around base + per-piece relative part), stay inside the tensor
        const int shapes[][7] = {{4, 24, 68, 120, 48, 136, 240}, {4, 12, 34, 60, 24, 68, 120}, {1, 2, 4, 7, 3, 7, 13}, {2, 3, 5, 18, 6, 10, 36},
                                 {8, 1, 136, 240, 1, 272, 480}, {1, 1, 1, 25, 2, 2, 50}};
        for (const auto &sh : shapes) {
            const int B = sh[0], Dc = sh[1], Wc = sh[3], Df = sh[4], Hf = sh[5], Wf = sh[6], CN = 32;
            const int nwchunk = (Wc + 7) / 8;
            const long long ncols = (long long)B * Dc * nwchunk;
            const int wgs = az_wgrad16_workgroups(ncols, 256, 1, 0);
            std::vector<char> seen_col((size_t)ncols, 0);
            for (int wgl = 0; wgl < wgs; ++wgl) {
                const int wg0 = (wgs & 7) ? wgl : az_xcd_map(wgl, wgs);
                for (long long col = wg0; col < ncols; col += wgs) {
                    CHECK(!seen_col[(size_t)col], "column %lld twice", col);
                    seen_col[(size_t)col] = 1;
                    int cd, wc, b;
                    az_s2w_col_decode(col, Dc, nwchunk, cd, wc, b);
                    CHECK(cd >= 0 && cd < Dc && wc >= 0 && wc < nwchunk && b >= 0 && b < B, "column %lld -> %d %d %d", col, cd, wc, b);
                    if (col % 7) continue;  // (offsets: a sample of the columns)
                    const unsigned vb_f = CN * 4u, plane_f = (unsigned)Hf * Wf * vb_f;
                    const long long vol_f = (long long)Df * plane_f;
                    const int cw0 = wc * 8;
                    const unsigned gbase = (unsigned)(2 * cd - 1) * plane_f + (unsigned)(2 * cw0 - 1) * vb_f;
                    for (int frow0 = 0; frow0 < Hf + 8; frow0 += 8)
                        for (int f = 0; f < AZ_S2W_NFQ; f += 5) {
                            int pk, j, pp, lrow;
                            az_s2w_fine_piece(f, pk, j, pp, lrow);
                            const int fd = 2 * cd - 1 + pk, fr = frow0 + j, fw = 2 * cw0 - 1 + pp;
                            if (fd < 0 || fd >= Df || fr >= Hf || fw < 0 || fw >= Wf) continue;
                            const unsigned rel = (unsigned)pk * plane_f + (unsigned)(j * Wf + pp) * vb_f + (unsigned)(f & 7) * 16u;
                            const unsigned off = gbase + (unsigned)frow0 * (unsigned)Wf * vb_f + rel;
                            const long long want = (long long)fd * plane_f + ((long long)fr * Wf + fw) * vb_f + (f & 7) * 16;
                            CHECK((long long)off == want && want + 16 <= vol_f, "offset %u vs %lld (volume %lld)", off, want, vol_f);
                        }
                }
            }
            for (long long c = 0; c < ncols; ++c) CHECK(seen_col[(size_t)c], "column %lld never walked", c);
        }
    }
    std::printf("launch math: %d failures\n", fails);
    return fails ? 1 : 0;
}
