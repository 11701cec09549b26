// Host side of GR_DISC_MESH: the table the kernels read, built from what crosses the ABI.
//
// The reference's has_intersect (src/geometry/meshes.jl:53-64) walks EVERY triangle at every accepted step inside the bounding
// box and keeps those whose first vertex V1 lies within 3 of the step's end before running jsf_algorithm; the answer is the OR
// over those triangles, so their order does not matter.  On the device the candidates come from a uniform grid over the V1s
// with cells of (just over) 3: a triangle within 3 of the point has its V1 in the point's cell or one of its 26 neighbours.
// Same candidates, same distance test, same predicate -- O(local triangles) per step instead of O(all triangles).
//
// Layout of the device table (doubles):
//   [0..5]   x_min, x_max, y_min, y_max, z_min, z_max           (in_nearby_region, meshes.jl:46-51; as given by the caller)
//   [6..8]   grid origin (the smallest V1 per axis)              [9] 1 / cell size
//   [10..12] cells per axis nx, ny, nz                           [13] offset (in doubles) of the sorted triangles
//   [14..15] reserved
//   [16.. ]  cell_start: nx ny nz + 1 uint32, cell (ix, iy, iz) at (iz ny + iy) nx + ix -- x runs fastest, so the three
//            x-neighbours of a point are ONE contiguous run of triangles
//   [off..]  the triangles sorted by cell: first every V1 (3 n doubles -- all the distance test reads, so neighbouring
//            candidates share cache lines), then every (V2, V3) (6 n doubles), read only by the triangles that pass it
// Included by the host unit (gradus_mi355x.hip) and by tests/host_harness.cpp, which runs the kernel logic on the CPU.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace gr_mesh {

constexpr int kHeader = 16;
constexpr double kCell = 3.0 * (1.0 + 1e-9);      // (a hair over the reference's radius: the neighbour property survives rounding)

inline int64_t cell_of(double x, double origin, double icell, int64_t n)
{
    const double f = std::floor((x - origin) * icell);
    if (!(f > 0.0)) return 0;
    return f >= (double)n ? n - 1 : (int64_t)f;
}

// every vertex finite?  (an infinite one would make the grid's cell search below run for ever; the extents may be infinite)
inline bool vertices_finite(const double* src, int64_t n)
{
    for (int64_t i = 6; i < 6 + 9 * n; ++i)
        if (!std::isfinite(src[i])) return false;
    return true;
}

// src: 6 extents + 9 n doubles (gr_config.disc_table), vertices finite; out: the table above
inline void build_table(const double* src, int64_t n, std::vector<double>& out)
{
    const double* tri = src + 6;
    double lo[3] = { tri[0], tri[1], tri[2] }, hi[3] = { tri[0], tri[1], tri[2] };
    for (int64_t k = 0; k < n; ++k)
        for (int a = 0; a < 3; ++a) {
            const double v = tri[9 * k + a];
            if (v < lo[a]) lo[a] = v;
            if (v > hi[a]) hi[a] = v;
        }
    // cells per axis from the SAME expression that places a vertex (cell_of), so the largest V1 lands in the last cell
    double cell = kCell, icell = 1.0 / kCell;
    int64_t dim[3];
    for (;;) {
        icell = 1.0 / cell;
        double cells = 1.0;
        for (int a = 0; a < 3; ++a) {
            const double c = std::floor((hi[a] - lo[a]) * icell) + 1.0;
            dim[a] = !(c >= 1.0) ? 1 : (c > 4e6 ? 4000000 : (int64_t)c);
            cells *= (double)dim[a];
        }
        if (cells <= (double)(1 << 21)) break;        // at most 2 M cells (8 MB of cell_start): coarser cells beyond that
        cell *= 2.0;
    }
    const int64_t ncell = dim[0] * dim[1] * dim[2];
    std::vector<uint32_t> start((size_t)ncell + 1, 0);
    std::vector<uint32_t> where((size_t)n);
    for (int64_t k = 0; k < n; ++k) {
        const int64_t ix = cell_of(tri[9 * k + 0], lo[0], icell, dim[0]);
        const int64_t iy = cell_of(tri[9 * k + 1], lo[1], icell, dim[1]);
        const int64_t iz = cell_of(tri[9 * k + 2], lo[2], icell, dim[2]);
        where[(size_t)k] = (uint32_t)((iz * dim[1] + iy) * dim[0] + ix);
        start[where[(size_t)k] + 1]++;
    }
    for (int64_t c = 0; c < ncell; ++c) start[(size_t)c + 1] += start[(size_t)c];
    const int64_t words = ncell + 1;
    const int64_t off = kHeader + (words + 1) / 2;
    out.assign((size_t)(off + 9 * n), 0.0);
    std::memcpy(out.data(), src, 6 * sizeof(double));
    out[6] = lo[0]; out[7] = lo[1]; out[8] = lo[2];
    out[9] = icell;
    out[10] = (double)dim[0]; out[11] = (double)dim[1]; out[12] = (double)dim[2];
    out[13] = (double)off;
    std::memcpy(out.data() + kHeader, start.data(), (size_t)words * sizeof(uint32_t));
    std::vector<uint32_t> fill(start.begin(), start.end() - 1);
    for (int64_t k = 0; k < n; ++k) {                // stable: mesh order inside a cell
        const uint32_t slot = fill[where[(size_t)k]]++;
        std::memcpy(out.data() + off + 3 * (int64_t)slot, tri + 9 * k, 3 * sizeof(double));
        std::memcpy(out.data() + off + 3 * n + 6 * (int64_t)slot, tri + 9 * k + 3, 6 * sizeof(double));
    }
}

// FNV-1a over the 64-bit words of the caller's table: the built table is kept per context while the caller passes the same mesh
inline uint64_t fingerprint(const double* src, int64_t n)
{
    const size_t words = (size_t)(6 + 9 * n);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < words; ++i) {
        uint64_t w;
        std::memcpy(&w, src + i, sizeof w);
        h = (h ^ w) * 1099511628211ull;
    }
    return h ^ (uint64_t)n;
}

}  // namespace gr_mesh
