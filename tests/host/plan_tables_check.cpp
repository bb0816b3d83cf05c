// CPU replay of the head-tower kernels' index arithmetic against the host-built tables (bayes-od-rc_amd/csrc/plan_tables.h), compiled
// with -fsanitize=address,undefined by tests/test_host_sanitizers.py.  For every tile, slot and 3x3 tap the activation element the
// row-reuse loop stages -- extended row (pad1 + kx) of the tile, advanced by ky input rows -- must be the element the generic
// im2col gather reads: in_off + ky * in_pitch + kx (conv_igemm.hip: XR loop vs the row-table loop).
#include "plan_tables.h"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (++failures <= 20) { std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } } } while (0)

static PyramidGeometry geometry(int H, int W) {
    // feature_extractor.py / feature_decoder.py: C3..C5 at strides 8, 16, 32 (SAME-padded stride-2 stages: ceil), P6 and P7 by stride-2 3x3 convs
    auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
    int lh[5], lw[5];
    int h = cdiv(cdiv(H, 2), 2), w = cdiv(cdiv(W, 2), 2);          // stem conv s2 + max-pool s2
    for (int l = 0; l < 3; ++l) { h = cdiv(h, 2); w = cdiv(w, 2); lh[l] = h; lw[l] = w; }
    lh[3] = (lh[2] + 1) / 2; lw[3] = (lw[2] + 1) / 2;
    lh[4] = (lh[3] + 1) / 2; lw[4] = (lw[3] + 1) / 2;
    return pyramid_geometry(lh, lw);
}

// the kernel's view of one tiling: tiles of 256 slots + XR_EXT_ROWS extended rows each
static void check_tiling(const char* what, const std::vector<RowEnt>& src, const std::vector<RowEnt>& tiled, const std::vector<ExtRow>& ext,
                         int64_t in_pixels, int group_n /* 0: plain tiles; N: sample-complete tiles */) {
    CHECK(tiled.size() % 256 == 0, "%s: %zu rows are not whole tiles", what, tiled.size());
    const size_t tiles = tiled.size() / 256;
    CHECK(ext.size() == tiles * XR_EXT_ROWS, "%s: %zu extended rows for %zu tiles", what, ext.size(), tiles);
    std::map<int32_t, int> seen;                                     // out_off -> count
    for (size_t t = 0; t < tiles; ++t) {
        const ExtRow* e = &ext[t * XR_EXT_ROWS];
        for (int q = 0; q < XR_EXT_ROWS; ++q) {
            CHECK(e[q].x >= e[0].x, "%s: tile %zu ext row %d below the tile's first", what, t, q);
            // 32-bit byte offsets against the tile's first extended row, 512-byte pixel rows (conv_igemm.hip: xo[])
            CHECK(((int64_t)e[q].x - e[0].x) * 512 + 2 * (int64_t)e[q].y * 512 < (int64_t)1 << 32, "%s: tile %zu ext row %d beyond 32-bit offsets", what, t, q);
            // an extended row is ONE pixel row of the input, read at ky = 0, 1, 2 input rows below its first tap
            CHECK(e[q].x >= 0 && (int64_t)e[q].x + 2 * (int64_t)e[q].y < in_pixels, "%s: tile %zu ext row %d reads outside the input planes", what, t, q);
        }
        for (int s = 0; s < 256; ++s) {
            const RowEnt& r = tiled[t * 256 + s];
            if (r.out_off < 0) continue;
            ++seen[r.out_off];
            CHECK(r.pad1 >= 0 && r.pad1 + 2 < XR_EXT_ROWS, "%s: tile %zu slot %d extended row %d", what, t, s, r.pad1);
            if (r.pad1 < 0 || r.pad1 + 2 >= XR_EXT_ROWS) continue;
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const int64_t staged = (int64_t)e[r.pad1 + kx].x + (int64_t)ky * e[r.pad1 + kx].y;     // row-reuse loop
                    const int64_t gathered = (int64_t)r.in_off + (int64_t)ky * r.in_pitch + kx;             // generic loop
                    CHECK(staged == gathered, "%s: tile %zu slot %d tap (%d,%d): staged %lld, gathered %lld", what, t, s, ky, kx, (long long)staged, (long long)gathered);
                }
        }
        if (group_n > 0) {
            for (int q = 0; q + 1 <= 256 / group_n; ++q) {
                const RowEnt& first = tiled[t * 256 + (size_t)q * group_n];
                for (int n = 0; n < group_n; ++n) {
                    const RowEnt& r = tiled[t * 256 + (size_t)q * group_n + n];
                    CHECK((r.out_off < 0) == (first.out_off < 0), "%s: tile %zu slot %d: samples partly valid", what, t, q);
                    if (r.out_off < 0) continue;
                    CHECK((r.rng_zs & 0xFFFF) == n, "%s: tile %zu slot %d row %d holds sample %d", what, t, q, n, r.rng_zs & 0xFFFF);
                    CHECK(r.rng_p == first.rng_p && (r.rng_zs >> 16) == (first.rng_zs >> 16), "%s: tile %zu slot %d mixes pixels", what, t, q);
                }
            }
            for (int s = (256 / group_n) * group_n; s < 256; ++s) CHECK(tiled[t * 256 + s].out_off < 0, "%s: tile %zu tail slot %d in use", what, t, s);
        }
    }
    CHECK(seen.size() == src.size(), "%s: %zu of %zu rows tiled", what, seen.size(), src.size());
    for (const RowEnt& r : src) {
        auto it = seen.find(r.out_off);
        CHECK(it != seen.end() && it->second == 1, "%s: output pixel %d tiled %d times", what, r.out_off, it == seen.end() ? 0 : it->second);
    }
}

int main() {
    const int sizes[][2] = {{512, 512}, {384, 1248}, {96, 160}, {160, 160}, {100, 75}, {720, 1280}, {64, 64}};
    const int batches[] = {1, 2, 3};
    const int samples[] = {1, 2, 10, 30};
    long checked = 0;
    for (const auto& hw : sizes)
        for (int B : batches)
            for (int N : samples) {
                if ((int64_t)hw[0] * hw[1] * B * N > (int64_t)512 * 512 * 3 * 10) continue;        // keep the sanitizer run in seconds
                const PyramidGeometry g = geometry(hw[0], hw[1]);
                std::vector<RowEnt> t1, t2, t3;
                head_row_tables(g, B, N, t1, t2, t3);
                CHECK((int64_t)t2.size() == (int64_t)B * N * g.P, "t2 size");
                std::set<int32_t> dense;
                for (const RowEnt& r : t2) {
                    CHECK(r.out_off == r.in_off + r.in_pitch + 1, "t2: output pixel is not the window centre");
                    CHECK(r.in_off >= 0 && r.in_off + 2 * r.in_pitch + 2 < (int64_t)B * N * g.Ppad, "t2: window outside the planes");
                    dense.insert(r.pad0);
                }
                CHECK(dense.size() == t2.size() && *dense.begin() == 0 && *dense.rbegin() == (int32_t)t2.size() - 1, "t2: fused 1x1 rows not dense");
                for (size_t i = 0; i < t3.size(); ++i) CHECK(t3[i].in_off == t2[i].out_off && t3[i].out_off == t2[i].pad0, "t3 row %zu", i);
                std::vector<RowEnt> tiled; std::vector<ExtRow> ext;
                CHECK(xr_tile_rows(t2, tiled, ext), "xr_tile_rows(t2) out of order");
                check_tiling("per-sample layers", t2, tiled, ext, (int64_t)B * N * g.Ppad, 0);
                CHECK(xr_tile_rows(t1, tiled, ext), "xr_tile_rows(t1) out of order");
                check_tiling("fan-out layer", t1, tiled, ext, (int64_t)B * g.Ppad, 0);
                if (N >= 2 && 320 / N - 2 >= 1) {
                    const int rc = xr_tile_rows_aggregated(t2, B, N, g.P, tiled, ext);
                    CHECK(rc == 0, "aggregated tiling: rc %d", rc);
                    if (rc == 0) check_tiling("sample-complete tiles", t2, tiled, ext, (int64_t)B * N * g.Ppad, N);
                }
                ++checked;
            }
    {   // a sample count whose runs cannot fit the staged rows is refused, not mis-tiled
        const PyramidGeometry g = geometry(64, 64);
        std::vector<RowEnt> t1, t2, t3, tiled; std::vector<ExtRow> ext;
        head_row_tables(g, 1, 200, t1, t2, t3);
        CHECK(xr_tile_rows_aggregated(t2, 1, 200, g.P, tiled, ext) == 1, "N = 200 must not fit");
    }
    std::printf("plan_tables_check: %ld configurations, %d failures\n", checked, failures);
    return failures ? 1 : 0;
}
