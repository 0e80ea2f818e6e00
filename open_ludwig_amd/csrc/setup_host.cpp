// libludwig_setup.so - host-side case set-up for levels too large for the numpy restatement in
// open_ludwig_amd/preprocess.py: SAT voxelizer, flood fill, wall distance and the Bouzidi q-map, one thread per block like the
// reference's `@threads for b_idx` loops. Float64 host arithmetic in the reference's operation order, built with -ffp-contract=off
// (Julia's CPU code never fuses a*b+c); results are bit-identical to the numpy restatement (tests/test_setup_native.py), which is
// itself pinned by the set-up integers of the reference's logs. Plain C ABI, declared in include/ludwig_setup.h. No GPU code here.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ludwig_setup.h"

namespace {

constexpr int BS = 8;
thread_local std::string g_err;

int fail(const std::string& m) { g_err = m; return -1; }

// block coordinates (1-based, as in BlockLevel.active_block_coords) -> index, dense over the level's bounding box
struct BlockGrid {
    int lo[3], n[3];
    std::vector<int32_t> ptr;                       // -1 = no block
    bool build(const int32_t* coords, int64_t nb) {
        if (nb <= 0) return false;
        int hi[3];
        for (int a = 0; a < 3; ++a) lo[a] = hi[a] = coords[a];
        for (int64_t b = 0; b < nb; ++b)
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], coords[3 * b + a]); hi[a] = std::max(hi[a], coords[3 * b + a]); }
        for (int a = 0; a < 3; ++a) n[a] = hi[a] - lo[a] + 1;
        ptr.assign((size_t)n[0] * n[1] * n[2], -1);
        for (int64_t b = 0; b < nb; ++b) ptr[idx(coords[3 * b], coords[3 * b + 1], coords[3 * b + 2])] = (int32_t)b;
        return true;
    }
    size_t idx(int bx, int by, int bz) const { return ((size_t)(bz - lo[2]) * n[1] + (by - lo[1])) * n[0] + (bx - lo[0]); }
    int32_t find(int bx, int by, int bz) const {
        if (bx < lo[0] || by < lo[1] || bz < lo[2] || bx >= lo[0] + n[0] || by >= lo[1] + n[1] || bz >= lo[2] + n[2]) return -1;
        return ptr[idx(bx, by, bz)];
    }
};

// triangles per block: build_block_triangle_map (src/domain_generation.jl:34-72) / ..._for_bouzidi (src/bouzidi_setup.jl:11-52):
// a triangle is binned to every existing block its bounding box, widened by `margin`, touches. CSR lists, triangle order kept.
struct TriangleBins {
    std::vector<int64_t> start;
    std::vector<int32_t> tri;
    void build(const double* t, int64_t n_tri, double dx, double margin, const BlockGrid& g, int64_t nb) {
        std::vector<int64_t> count(nb + 1, 0);
        const double w = BS * dx;
        auto range = [&](int64_t i, int* lo, int* hi) {
            for (int a = 0; a < 3; ++a) {
                const double p0 = t[9 * i + a], p1 = t[9 * i + 3 + a], p2 = t[9 * i + 6 + a];
                const double mn = std::min(p0, std::min(p1, p2)), mx = std::max(p0, std::max(p1, p2));
                lo[a] = std::max(1, (int)std::floor((mn - margin) / w) + 1);
                hi[a] = (int)std::floor((mx + margin) / w) + 1;
                lo[a] = std::max(lo[a], g.lo[a]);
                hi[a] = std::min(hi[a], g.lo[a] + g.n[a] - 1);
            }
        };
        for (int pass = 0; pass < 2; ++pass) {
            for (int64_t i = 0; i < n_tri; ++i) {
                int lo[3], hi[3];
                range(i, lo, hi);
                for (int bz = lo[2]; bz <= hi[2]; ++bz)
                    for (int by = lo[1]; by <= hi[1]; ++by)
                        for (int bx = lo[0]; bx <= hi[0]; ++bx) {
                            const int32_t b = g.ptr[g.idx(bx, by, bz)];
                            if (b < 0) continue;
                            if (pass == 0) ++count[b + 1];
                            else tri[start[b] + count[b]++] = (int32_t)i;
                        }
            }
            if (pass == 0) {
                start.assign(nb + 1, 0);
                for (int64_t b = 0; b < nb; ++b) start[b + 1] = start[b] + count[b + 1];
                tri.resize((size_t)start[nb]);
                std::fill(count.begin(), count.end(), 0);
            }
        }
    }
};

template <class F>
void for_blocks(int64_t nb, int n_threads, F&& body) {
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    n_threads = (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, nb));
    std::atomic<int64_t> next{0};
    auto work = [&]() {
        for (;;) {
            const int64_t b0 = next.fetch_add(16);
            if (b0 >= nb) return;
            for (int64_t b = b0; b < std::min(nb, b0 + 16); ++b) body(b);
        }
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < n_threads; ++i) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
}

// triangle_intersects_aabb (src/domain_generation.jl:10-32): AABB slabs, then the nine edge-cross axes (no triangle-plane test).
// c = cell centre, h = box half size * 1.001, v = the triangle's three corners.
inline bool triangle_intersects_aabb(const double c[3], double h, const double* v) {
    double t1[3], t2[3], t3[3];
    for (int a = 0; a < 3; ++a) { t1[a] = v[a] - c[a]; t2[a] = v[3 + a] - c[a]; t3[a] = v[6 + a] - c[a]; }
    for (int a = 0; a < 3; ++a) {
        if (std::min(t1[a], std::min(t2[a], t3[a])) > h || std::max(t1[a], std::max(t2[a], t3[a])) < -h) return false;
    }
    double f[3][3];
    for (int a = 0; a < 3; ++a) { f[0][a] = t2[a] - t1[a]; f[1][a] = t3[a] - t2[a]; f[2][a] = t1[a] - t3[a]; }
    for (int i = 0; i < 3; ++i) {
        const double u[3] = {i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0};
        for (int j = 0; j < 3; ++j) {
            const double ax[3] = {u[1] * f[j][2] - u[2] * f[j][1], u[2] * f[j][0] - u[0] * f[j][2], u[0] * f[j][1] - u[1] * f[j][0]};
            if ((ax[0] * ax[0] + ax[1] * ax[1]) + ax[2] * ax[2] < 1e-10) continue;
            const double p1 = (t1[0] * ax[0] + t1[1] * ax[1]) + t1[2] * ax[2];
            const double p2 = (t2[0] * ax[0] + t2[1] * ax[1]) + t2[2] * ax[2];
            const double p3 = (t3[0] * ax[0] + t3[1] * ax[1]) + t3[2] * ax[2];
            const double r = (h * std::fabs(ax[0]) + h * std::fabs(ax[1])) + h * std::fabs(ax[2]);
            if (std::min(p1, std::min(p2, p3)) > r || std::max(p1, std::max(p2, p3)) < -r) return false;
        }
    }
    return true;
}

// Float16(x::Float64): one rounding, to nearest even (src/bouzidi_setup.jl:128). x in (0, 1] here; general for finite x >= 0.
inline uint16_t f64_to_f16(double x) {
    uint64_t u;
    std::memcpy(&u, &x, 8);
    const int e = (int)((u >> 52) & 0x7ff) - 1023;
    const uint64_t m = (u & 0xfffffffffffffULL) | (1ULL << 52);        // 53-bit significand
    if (x == 0.0) return 0;
    if (e > 15) return 0x7c00;
    int shift;                                                        // bits dropped from the 53-bit significand
    int he;
    if (e >= -14) { shift = 42; he = e + 15; }                         // normal half: 11-bit significand
    else { shift = 42 + (-14 - e); he = 0; }                           // subnormal half
    if (shift > 54) return 0;
    uint64_t q = m >> shift;
    const uint64_t rem = m & ((1ULL << shift) - 1), half = 1ULL << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) ++q;
    // q carries the implicit bit for normals (bit 10); adding it to the exponent field handles the carry into the next binade
    return (uint16_t)(he == 0 ? q : ((uint64_t)(he - 1) << 10) + q);
}

}  // namespace

extern "C" {

const char* lws_last_error(void) { return g_err.c_str(); }

uint16_t lws_f64_to_f16(double x) { return f64_to_f16(x); }

int lws_voxelize(const double* tri, int64_t n_tri, double dx, const int32_t* coords, int64_t n_blocks, uint8_t* obstacle,
                 int n_threads) {
    if (!tri || !coords || !obstacle || n_blocks <= 0 || n_tri < 0 || !(dx > 0)) return fail("lws_voxelize: bad argument");
    BlockGrid g;
    g.build(coords, n_blocks);
    TriangleBins bins;
    bins.build(tri, n_tri, dx, dx * 2, g, n_blocks);
    const double half = 0.75 * dx, h = half * 1.001;
    for_blocks(n_blocks, n_threads, [&](int64_t b) {
        const int64_t s = bins.start[b], e = bins.start[b + 1];
        if (s == e) return;
        const int32_t* bc = coords + 3 * b;
        // only triangles whose bounding box comes within h of the cell centre can pass the slab test: skip the others early
        for (int lz = 1; lz <= BS; ++lz)
            for (int ly = 1; ly <= BS; ++ly)
                for (int lx = 1; lx <= BS; ++lx) {
                    const double c[3] = {((bc[0] - 1) * BS + lx - 0.5) * dx, ((bc[1] - 1) * BS + ly - 0.5) * dx,
                                         ((bc[2] - 1) * BS + lz - 0.5) * dx};
                    bool shell = false;
                    for (int64_t i = s; i < e && !shell; ++i) shell = triangle_intersects_aabb(c, h, tri + 9 * (int64_t)bins.tri[i]);
                    if (shell) obstacle[b * 512 + (lz - 1) * 64 + (ly - 1) * 8 + (lx - 1)] = 1;
                }
    });
    return 0;
}

// perform_flood_fill! (src/domain_generation.jl:114-203): breadth-first over the 6-neighbourhood from every fluid cell of the
// blocks with the smallest bx; whatever fluid it does not reach turns solid. Returns the number of cells filled.
int64_t lws_flood_fill(const int32_t* coords, int64_t n_blocks, uint8_t* obstacle) {
    if (!coords || !obstacle || n_blocks <= 0) return fail("lws_flood_fill: bad argument");
    BlockGrid g;
    g.build(coords, n_blocks);
    std::vector<uint8_t> visited((size_t)n_blocks * 512, 0);
    std::vector<int64_t> queue;
    queue.reserve((size_t)n_blocks * 64);
    int min_bx = coords[0];
    for (int64_t b = 0; b < n_blocks; ++b) min_bx = std::min(min_bx, coords[3 * b]);
    for (int64_t b = 0; b < n_blocks; ++b)
        if (coords[3 * b] == min_bx)
            for (int c = 0; c < 512; ++c)
                if (!obstacle[b * 512 + c]) { visited[b * 512 + c] = 1; queue.push_back(b * 512 + c); }
    static const int D[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    for (size_t head = 0; head < queue.size(); ++head) {
        const int64_t cur = queue[head], b = cur >> 9;
        const int c = (int)(cur & 511), lx = c & 7, ly = (c >> 3) & 7, lz = c >> 6;
        for (int i = 0; i < 6; ++i) {
            int nx = lx + D[i][0], ny = ly + D[i][1], nz = lz + D[i][2];
            int64_t nb = b;
            if (nx < 0 || nx >= BS || ny < 0 || ny >= BS || nz < 0 || nz >= BS) {
                nb = g.find(coords[3 * b] + D[i][0], coords[3 * b + 1] + D[i][1], coords[3 * b + 2] + D[i][2]);
                if (nb < 0) continue;
                nx = (nx + BS) % BS; ny = (ny + BS) % BS; nz = (nz + BS) % BS;
            }
            const int64_t n = nb * 512 + nz * 64 + ny * 8 + nx;
            if (!visited[n] && !obstacle[n]) { visited[n] = 1; queue.push_back(n); }
        }
    }
    int64_t filled = 0;
    for (int64_t i = 0; i < n_blocks * 512; ++i)
        if (!obstacle[i] && !visited[i]) { obstacle[i] = 1; ++filled; }
    return filled;
}

// compute_wall_distances! (src/domain_generation.jl:371-434): fluid cells with a solid cell among their 26 neighbours get the
// smallest sqrt(Float32(d^2)) * Float32(dx); a neighbour in a block that does not exist is skipped. `wall` must hold 100.0f.
int64_t lws_wall_distance(const int32_t* coords, int64_t n_blocks, const uint8_t* obstacle, double dx, float* wall, int n_threads) {
    if (!coords || !obstacle || !wall || n_blocks <= 0) return fail("lws_wall_distance: bad argument");
    BlockGrid g;
    g.build(coords, n_blocks);
    const float fdx = (float)dx;
    std::atomic<int64_t> near_total{0};
    for_blocks(n_blocks, n_threads, [&](int64_t b) {
        int32_t nbr[27];
        for (int k = 0; k < 27; ++k) nbr[k] = g.find(coords[3 * b] + k % 3 - 1, coords[3 * b + 1] + (k / 3) % 3 - 1, coords[3 * b + 2] + k / 9 - 1);
        int64_t near = 0;
        for (int lz = 0; lz < BS; ++lz)
            for (int ly = 0; ly < BS; ++ly)
                for (int lx = 0; lx < BS; ++lx) {
                    if (obstacle[b * 512 + lz * 64 + ly * 8 + lx]) continue;
                    bool is_near = false;
                    float best = 100.0f;
                    for (int dz = -1; dz <= 1; ++dz)
                        for (int dy = -1; dy <= 1; ++dy)
                            for (int dxo = -1; dxo <= 1; ++dxo) {
                                if (!dxo && !dy && !dz) continue;
                                int nx = lx + dxo, ny = ly + dy, nz = lz + dz;
                                const int ox = nx < 0 ? -1 : (nx >= BS ? 1 : 0), oy = ny < 0 ? -1 : (ny >= BS ? 1 : 0),
                                          oz = nz < 0 ? -1 : (nz >= BS ? 1 : 0);
                                const int32_t nb = nbr[(ox + 1) + 3 * (oy + 1) + 9 * (oz + 1)];
                                if (nb < 0) continue;
                                nx -= ox * BS; ny -= oy * BS; nz -= oz * BS;
                                if (obstacle[(int64_t)nb * 512 + nz * 64 + ny * 8 + nx]) {
                                    is_near = true;
                                    const float dist = std::sqrt((float)(dxo * dxo + dy * dy + dz * dz)) * fdx;
                                    best = std::min(best, dist);
                                }
                            }
                    if (is_near) { wall[b * 512 + lz * 64 + ly * 8 + lx] = best; ++near; }
                }
        near_total += near;
    });
    return near_total.load();
}

// compute_bouzidi_qmap_sparse (src/bouzidi_setup.jl:64-167) with compute_q_for_cell / ray_triangle_intersection
// (src/bouzidi_math.jl:9-102): from every cell of every block that has triangles binned to it (margin 2.5 dx), 26 rays against the
// binned triangles; q = t_min / (dx |c|) kept if 0 < q <= 1, stored as Float16. q_map is [27][n_blocks][512] (= the reference's
// [8,8,8,n_blocks,27] column-major) and must be zeroed by the caller; boundary[n_blocks * 512] gets 1 where any q was stored.
// A link is at most sqrt(3) dx long, so a triangle whose bounding box is farther than that from the cell centre on some axis cannot
// give q <= 1 and is skipped before any arithmetic (it could never undercut a nearer hit either).
int64_t lws_bouzidi_qmap(const double* tri, int64_t n_tri, double dx, const int32_t* coords, int64_t n_blocks, uint16_t* q_map,
                         uint8_t* boundary, int n_threads) {
    if (!tri || !coords || !q_map || !boundary || n_blocks <= 0 || n_tri < 0 || !(dx > 0)) return fail("lws_bouzidi_qmap: bad argument");
    BlockGrid g;
    g.build(coords, n_blocks);
    TriangleBins bins;
    bins.build(tri, n_tri, dx, dx * 2.5, g, n_blocks);
    // per triangle: bounding box, v1, edge1, edge2
    std::vector<double> tb((size_t)n_tri * 6), te((size_t)n_tri * 6);
    for (int64_t i = 0; i < n_tri; ++i)
        for (int a = 0; a < 3; ++a) {
            const double p0 = tri[9 * i + a], p1 = tri[9 * i + 3 + a], p2 = tri[9 * i + 6 + a];
            tb[6 * i + a] = std::min(p0, std::min(p1, p2));
            tb[6 * i + 3 + a] = std::max(p0, std::max(p1, p2));
            te[6 * i + a] = p1 - p0;
            te[6 * i + 3 + a] = p2 - p0;
        }
    double dn[27][3], cm[27];
    for (int k = 0; k < 27; ++k) {
        const int c[3] = {k % 3 - 1, (k / 3) % 3 - 1, k / 9 - 1};
        cm[k] = std::sqrt((double)(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]));
        for (int a = 0; a < 3; ++a) dn[k][a] = k == 13 ? 0.0 : (double)c[a] / cm[k];
    }
    const double EPS = 1e-9, reach = std::sqrt(3.0) * dx * 1.0001;
    const size_t comp = (size_t)n_blocks * 512;
    std::atomic<int64_t> n_boundary{0};
    for_blocks(n_blocks, n_threads, [&](int64_t b) {
        const int64_t s = bins.start[b], e = bins.start[b + 1];
        if (s == e) return;
        const int32_t* bc = coords + 3 * b;
        int64_t found = 0;
        for (int lz = 1; lz <= BS; ++lz)
            for (int ly = 1; ly <= BS; ++ly)
                for (int lx = 1; lx <= BS; ++lx) {
                    const double o[3] = {((bc[0] - 1) * BS + lx - 0.5) * dx, ((bc[1] - 1) * BS + ly - 0.5) * dx,
                                         ((bc[2] - 1) * BS + lz - 0.5) * dx};
                    double tmin[27];
                    for (int k = 0; k < 27; ++k) tmin[k] = std::numeric_limits<double>::infinity();
                    bool any = false;
                    for (int64_t i = s; i < e; ++i) {
                        const int64_t t = bins.tri[i];
                        const double* bb = &tb[6 * t];
                        if (o[0] < bb[0] - reach || o[0] > bb[3] + reach || o[1] < bb[1] - reach || o[1] > bb[4] + reach ||
                            o[2] < bb[2] - reach || o[2] > bb[5] + reach)
                            continue;
                        const double* v1 = tri + 9 * t;
                        const double* e1 = &te[6 * t];
                        const double* e2 = e1 + 3;
                        const double sv[3] = {o[0] - v1[0], o[1] - v1[1], o[2] - v1[2]};
                        const double qv[3] = {sv[1] * e1[2] - sv[2] * e1[1], sv[2] * e1[0] - sv[0] * e1[2], sv[0] * e1[1] - sv[1] * e1[0]};
                        const double t_num = (e2[0] * qv[0] + e2[1] * qv[1]) + e2[2] * qv[2];
                        for (int k = 0; k < 27; ++k) {
                            if (k == 13) continue;
                            const double* d = dn[k];
                            const double h[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
                            const double a = (e1[0] * h[0] + e1[1] * h[1]) + e1[2] * h[2];
                            if (std::fabs(a) < EPS) continue;
                            const double f = 1.0 / a;
                            const double u = f * ((sv[0] * h[0] + sv[1] * h[1]) + sv[2] * h[2]);
                            if (u < 0.0 || u > 1.0) continue;
                            const double v = f * ((qv[0] * d[0] + qv[1] * d[1]) + qv[2] * d[2]);
                            if (v < 0.0 || u + v > 1.0) continue;
                            const double tt = f * t_num;
                            if (tt > EPS && tt < tmin[k]) { tmin[k] = tt; any = true; }
                        }
                    }
                    if (!any) continue;
                    bool has = false;
                    const size_t cell = (size_t)b * 512 + (lz - 1) * 64 + (ly - 1) * 8 + (lx - 1);
                    for (int k = 0; k < 27; ++k) {
                        if (!(tmin[k] < std::numeric_limits<double>::infinity())) continue;
                        const double q = tmin[k] / (dx * cm[k]);
                        if (q > 0.0 && q <= 1.0) {
                            // the reference tests q_vals[k] > 0.0 on the Float64 value and stores Float16(q) (possibly 0 after rounding)
                            q_map[k * comp + cell] = f64_to_f16(q);
                            has = true;
                        }
                    }
                    if (has) { boundary[cell] = 1; ++found; }
                }
        n_boundary += found;
    });
    return n_boundary.load();
}

}  // extern "C"
