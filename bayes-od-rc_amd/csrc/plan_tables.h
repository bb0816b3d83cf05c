// Host-side index tables of the head towers: plain C++ (no HIP), shared by engine.hip and by the sanitizer build of
// tests/host/plan_tables_check.cpp (g++ -fsanitize=address,undefined; tests/test_host_sanitizers.py), which replays the kernels' index
// arithmetic against these tables on the CPU.
//   head_row_tables        one RowEnt per output pixel of the three head launches' shapes (multitask_headers.py:98-123: the towers
//                          run on the concatenated p3..p7 pyramid; retinanet_model.py:78-81: N MC samples per image)
//   xr_tile_rows           rows -> 256-slot tiles of runs of x-adjacent pixels + each tile's XR_EXT_ROWS extended input rows
//   xr_tile_rows_aggregated  the same with tiles that hold ALL N samples of their pixels (row = slot * N + sample)
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <algorithm>
#include <vector>

struct RowEnt {            // 32 B per output pixel
    int32_t in_off;        // pixel index (in the group's input buffer) of the window origin
    int32_t in_pitch;      // padded row width of that plane, in pixels
    int32_t out_off;       // pixel index in the output buffer
    int32_t res_off;       // pixel index in the residual buffer (unused if no residual)
    int32_t rng_p;         // pixel index in the image's concatenated p3..p7 pyramid (dropout counter x)
    int32_t rng_zs;        // sample | image_in_batch << 16
    int32_t pad0;          // dense output row of the fused 1x1 head conv ((b*N+n)*P + p)
    int32_t pad1;          // extended-row index of this pixel inside its tile (activation row reuse)
};

// Row-reuse staging (conv_igemm.hip): extended rows per 256-pixel tile.
constexpr int XR_EXT_ROWS = 320;
struct ExtRow { int32_t x, y; };          // (pixel index of the row's first tap, padded row width): layout of HIP's int2

struct PyramidGeometry {                  // the five pyramid levels of one image, each in a plane with a one-pixel zero border
    int lh[5], lw[5];
    int64_t lvl_off[5];                   // padded pixel offset of each level
    int64_t lvl_p0[5];                    // dense pixel offset of each level
    int64_t Ppad;                         // padded pixels per image
    int P;                                // pixels per image
};

inline PyramidGeometry pyramid_geometry(const int lh[5], const int lw[5]) {
    PyramidGeometry g{};
    int64_t pp = 0, p = 0;
    for (int l = 0; l < 5; ++l) {
        g.lh[l] = lh[l]; g.lw[l] = lw[l];
        g.lvl_off[l] = pp; g.lvl_p0[l] = p;
        pp += (int64_t)(lh[l] + 2) * (lw[l] + 2);
        p += lh[l] * lw[l];
    }
    g.Ppad = pp; g.P = (int)p;
    return g;
}

// t1: first tower layer (one row per image pixel, fan-out to N samples in the epilogue); t2: per-sample 3x3 layers; t3: the 1x1 outputs
inline void head_row_tables(const PyramidGeometry& g, int B, int N, std::vector<RowEnt>& t1, std::vector<RowEnt>& t2, std::vector<RowEnt>& t3) {
    t1.assign((size_t)B * g.P, RowEnt{});
    t2.assign((size_t)B * N * g.P, RowEnt{});
    t3.assign((size_t)B * N * g.P, RowEnt{});
    size_t r1 = 0, r2 = 0;
    for (int b = 0; b < B; ++b) {
        for (int l = 0; l < 5; ++l)
            for (int y = 0; y < g.lh[l]; ++y)
                for (int xq = 0; xq < g.lw[l]; ++xq) {
                    const int pitch = g.lw[l] + 2;
                    RowEnt e{};
                    e.in_off = (int32_t)((int64_t)b * g.Ppad + g.lvl_off[l] + (int64_t)y * pitch + xq);
                    e.in_pitch = pitch;
                    e.out_off = (int32_t)((int64_t)b * N * g.Ppad + g.lvl_off[l] + (int64_t)(y + 1) * pitch + (xq + 1));
                    e.rng_p = (int32_t)(g.lvl_p0[l] + y * g.lw[l] + xq);
                    e.rng_zs = (b << 16);
                    t1[r1++] = e;
                }
        for (int n = 0; n < N; ++n)
            for (int l = 0; l < 5; ++l)
                for (int y = 0; y < g.lh[l]; ++y)
                    for (int xq = 0; xq < g.lw[l]; ++xq) {
                        const int pitch = g.lw[l] + 2;
                        const int64_t plane0 = ((int64_t)b * N + n) * g.Ppad + g.lvl_off[l];
                        const int32_t dense = (int32_t)(g.lvl_p0[l] + y * g.lw[l] + xq);
                        RowEnt e{};
                        e.in_off = (int32_t)(plane0 + (int64_t)y * pitch + xq);
                        e.in_pitch = pitch;
                        e.out_off = (int32_t)(plane0 + (int64_t)(y + 1) * pitch + (xq + 1));
                        e.rng_p = dense;
                        e.rng_zs = n | (b << 16);
                        e.pad0 = (int32_t)(((int64_t)b * N + n) * g.P + dense);      // row of the fused 1x1 output
                        t2[r2] = e;
                        RowEnt f = e;
                        f.in_off = e.out_off;                 // 1x1 reads the centre pixel
                        f.out_off = (int32_t)(((int64_t)b * N + n) * g.P + dense);
                        t3[r2] = f;
                        ++r2;
                    }
    }
}

// rows -> 256-slot tiles of x-adjacent runs + each tile's extended input rows (kernels.h, ConvArgs::ext).  false: a tile's
// extended rows would not start with its smallest (the kernel takes its 32-bit activation offsets against the first).
inline bool xr_tile_rows(const std::vector<RowEnt>& src, std::vector<RowEnt>& tiled, std::vector<ExtRow>& ext) {
    tiled.clear(); ext.clear();
    if (src.empty()) return true;
    RowEnt invalid = src[0];
    invalid.out_off = -1; invalid.pad0 = 0; invalid.pad1 = 0;
    size_t r = 0;
    while (r < src.size()) {
        const size_t tile0 = tiled.size();
        const size_t ext0 = ext.size();
        int pix = 0, nx = 0;
        while (r < src.size() && pix < 256 && nx + 3 <= XR_EXT_ROWS) {
            // maximal run of x-adjacent pixels starting at row r
            size_t e = r + 1;
            while (e < src.size() && src[e].in_off == src[e - 1].in_off + 1 && src[e].in_pitch == src[r].in_pitch) ++e;
            const int take = (int)std::min<size_t>(e - r, (size_t)std::min(256 - pix, XR_EXT_ROWS - nx - 2));
            for (int k = 0; k < take + 2; ++k) ext.push_back(ExtRow{src[r].in_off + k, src[r].in_pitch});
            for (int k = 0; k < take; ++k) { RowEnt q = src[r + k]; q.pad1 = nx + k; tiled.push_back(q); }
            nx += take + 2; pix += take; r += take;
        }
        while (tiled.size() < tile0 + 256) tiled.push_back(invalid);
        // pad with the tile's first row: never read by a valid pixel, and it keeps the first entry the smallest of
        // the tile (the kernel takes its 32-bit activation offsets against it)
        while (ext.size() < ext0 + XR_EXT_ROWS) ext.push_back(ext[ext0]);
        for (size_t q = ext0; q < ext0 + XR_EXT_ROWS; ++q)
            if (ext[q].x < ext[ext0].x) return false;
    }
    return true;
}

// MC aggregation fused into the last tower layers' epilogues (inference_utils.py:31-60,220-244): a tile must hold ALL N samples
// of its pixels -- Q <= 256 / N pixel slots, row = slot * N + sample, made of runs of x-adjacent pixels whose extended rows
// (run + 2, once per sample) fit the XR_EXT_ROWS staged rows.  t2 = head_row_tables' per-sample table.  0 ok, 1 no pixel fits a
// tile, 2 extended rows out of order.
inline int xr_tile_rows_aggregated(const std::vector<RowEnt>& t2, int B, int N, int P, std::vector<RowEnt>& tiled, std::vector<ExtRow>& ext) {
    tiled.clear(); ext.clear();
    const int Qmax = 256 / N;
    RowEnt invalid = t2[0];
    invalid.out_off = -1; invalid.pad0 = 0; invalid.pad1 = 0;
    for (int b = 0; b < B; ++b) {
        const size_t img0 = (size_t)b * N * P;               // t2 index of (b, sample 0, pixel 0); sample n: + n * P
        int p = 0;
        while (p < P) {
            const size_t tile0 = tiled.size(), ext0 = ext.size();
            tiled.resize(tile0 + 256, invalid);
            int Q = 0, X = 0;
            while (p < P && Q < Qmax) {
                int L = 1;                                      // maximal run of x-adjacent pixels starting at p
                while (p + L < P && t2[img0 + p + L].in_off == t2[img0 + p + L - 1].in_off + 1 &&
                       t2[img0 + p + L].in_pitch == t2[img0 + p].in_pitch) ++L;
                const int take = std::min(std::min(L, Qmax - Q), (XR_EXT_ROWS - X) / N - 2);
                if (take < 1) break;
                for (int n = 0; n < N; ++n) {
                    const RowEnt& first = t2[img0 + (size_t)n * P + p];
                    for (int k = 0; k < take + 2; ++k) ext.push_back(ExtRow{first.in_off + k, first.in_pitch});
                    for (int k = 0; k < take; ++k) {
                        RowEnt q = t2[img0 + (size_t)n * P + p + k];
                        q.pad1 = X + n * (take + 2) + k;
                        tiled[tile0 + (size_t)(Q + k) * N + n] = q;
                    }
                }
                X += N * (take + 2); Q += take; p += take;
            }
            if (Q == 0) return 1;
            while (ext.size() < ext0 + XR_EXT_ROWS) ext.push_back(ext[ext0]);
            for (size_t q = ext0; q < ext0 + XR_EXT_ROWS; ++q)
                if (ext[q].x < ext[ext0].x) return 2;
        }
    }
    return 0;
}
