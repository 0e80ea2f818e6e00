// Diagnostic micro-benchmark (not part of the product): which WORK SCHEDULE lets the stream-collide data movement of a
// 256^3 box (27 + 3 streams in, 27 + 3 + 1 streams out, 244 B per cell) run closest to the copy rate of the box?
//   hipcc --offload-arch=gfx950 -O3 tools/marchbench.hip -o tools/marchbench
// Every wave handles a LIST of (block, z-plane) work items one after the other ("marching"); the lists and the per-item flags
// (which halo pieces / velocity planes the design would have to fetch from global memory instead of registers or LDS)
// are built on the host for each candidate design. No LDS exchange, no collision arithmetic: data movement only.
// Layout as the engine: f[k][block][512] floats, blocks of the NB^3 periodic box sorted with bz fastest.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int Q = 27;
constexpr int F_W = 1 << 20, F_E = 1 << 21, F_S = 1 << 22, F_N = 1 << 23;       // halo faces fetched from global memory
constexpr int F_C = 1 << 24, F_B = 1 << 25, F_T = 1 << 26;                      // velocity planes fetched from global memory
constexpr int ID_MASK = (1 << 20) - 1;

__device__ __forceinline__ int wrap(int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); }
__device__ __forceinline__ float ldf(const float *base, uint32_t off) { return *(const float *)((const char *)base + off); }
__device__ __forceinline__ void stf(float *base, uint32_t off, float v) { __builtin_nontemporal_store(v, (float *)((char *)base + off)); }

template <int NW, bool DEP = false, int WPE = 0>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE == 0 ? 1 : WPE, WPE == 0 ? 8 : WPE)))
void k_march(const float *__restrict__ fin, float *__restrict__ fout, const float *__restrict__ vin,
             float *__restrict__ vout, float *__restrict__ rho, const int *__restrict__ items,
             int niter, size_t sk, int NB, int mask, const int *__restrict__ meta = nullptr, int layout = 0)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, x = lane & 7, y = lane >> 3;
    const int *my = items + ((size_t)blockIdx.x * NW + wave) * niter;
    float uc[3] = {0, 0, 0}, ub[3] = {0, 0, 0}, ut[3] = {0, 0, 0};
    for (int it = 0; it < niter; ++it) {
        const int raw = __builtin_amdgcn_readfirstlane(my[it]);
        if (raw < 0) continue;
        const int fl = raw & mask;
        const int id = raw & ID_MASK, b = id >> 3, z = id & 7;
        const bool aosoa = layout == 2;            // block-major: the 27 populations of a block are contiguous (f[block][k][512])
        const int bz = layout == 0 ? b % NB : b / (NB * NB), by = (b / NB) % NB, bx = layout == 0 ? b / (NB * NB) : b % NB;
        int nbm[27];
        if (DEP) {     // as the product: the 27 neighbour ids come from a per-block row, a scalar load that depends on the item
#pragma unroll
            for (int d = 0; d < 27; ++d) nbm[d] = meta[(size_t)b * 32 + d];
        }
        auto blk = [&](int ox, int oy, int oz) {
            if (DEP) return (uint32_t)nbm[(ox + 1) + 3 * (oy + 1) + 9 * (oz + 1)];
            return layout == 0 ? (uint32_t)((wrap(bx + ox, NB) * NB + wrap(by + oy, NB)) * NB + wrap(bz + oz, NB))
                               : (uint32_t)((wrap(bz + oz, NB) * NB + wrap(by + oy, NB)) * NB + wrap(bx + ox, NB));
        };
        float v[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            const int cz = k / 9 - 1;
            const int sz = z - cz;
            const uint32_t nb = blk(0, 0, sz < 0 ? -1 : sz > 7 ? 1 : 0);
            v[k] = aosoa ? ldf(fin, ((nb * 27u + k) * 512u + 64 * (sz & 7) + lane) * 4u) : ldf(fin + sk * k, (nb * 512u + 64 * (sz & 7) + lane) * 4u);
        }
        const uint32_t own = ((uint32_t)b * 512u + z * 64u + lane) * 4u;
        const uint32_t ownA = ((uint32_t)b * 27u * 512u + z * 64u + lane) * 4u, ownV = ((uint32_t)b * 3u * 512u + z * 64u + lane) * 4u;
        // velocity planes: centre / below / above (in a marching design most of them are already in registers)
        if (fl & F_C) {
#pragma unroll
            for (int c = 0; c < 3; ++c) uc[c] = aosoa ? ldf(vin, ownV + c * 2048u) : ldf(vin + sk * c, own);
        }
        if (fl & F_B) {
            const uint32_t o = (blk(0, 0, z == 0 ? -1 : 0) * 512u + 64 * ((z - 1) & 7) + lane) * 4u;
#pragma unroll
            for (int c = 0; c < 3; ++c) ub[c] = ldf(vin + sk * c, o);
        }
        if (fl & F_T) {
            const uint32_t o = (blk(0, 0, z == 7 ? 1 : 0) * 512u + 64 * ((z + 1) & 7) + lane) * 4u;
#pragma unroll
            for (int c = 0; c < 3; ++c) ut[c] = ldf(vin + sk * c, o);
        }
        // halo faces from global memory: x columns (8 lanes, 32-B stride: 2 lines), y rows (8 lanes, one 32-B piece)
        float h = 0.0f;
        if ((fl & F_W) && x == 0) {
#pragma unroll
            for (int k = 0; k < Q; ++k)
                if (k % 3 == 2) { const int sz = z - (k / 9 - 1); h += ldf(fin + sk * k, (blk(-1, 0, sz < 0 ? -1 : sz > 7 ? 1 : 0) * 512u + 64 * (sz & 7) + 8 * y + 7) * 4u); }
#pragma unroll
            for (int c = 0; c < 3; ++c) h += ldf(vin + sk * c, (blk(-1, 0, 0) * 512u + 64 * z + 8 * y + 7) * 4u);
        }
        if ((fl & F_E) && x == 7) {
#pragma unroll
            for (int k = 0; k < Q; ++k)
                if (k % 3 == 0) { const int sz = z - (k / 9 - 1); h += ldf(fin + sk * k, (blk(1, 0, sz < 0 ? -1 : sz > 7 ? 1 : 0) * 512u + 64 * (sz & 7) + 8 * y + 0) * 4u); }
#pragma unroll
            for (int c = 0; c < 3; ++c) h += ldf(vin + sk * c, (blk(1, 0, 0) * 512u + 64 * z + 8 * y + 0) * 4u);
        }
        if ((fl & F_S) && y == 0) {
#pragma unroll
            for (int k = 0; k < Q; ++k)
                if ((k / 3) % 3 == 2) { const int sz = z - (k / 9 - 1); h += ldf(fin + sk * k, (blk(0, -1, sz < 0 ? -1 : sz > 7 ? 1 : 0) * 512u + 64 * (sz & 7) + 8 * 7 + x) * 4u); }
#pragma unroll
            for (int c = 0; c < 3; ++c) h += ldf(vin + sk * c, (blk(0, -1, 0) * 512u + 64 * z + 8 * 7 + x) * 4u);
        }
        if ((fl & F_N) && y == 7) {
#pragma unroll
            for (int k = 0; k < Q; ++k)
                if ((k / 3) % 3 == 0) { const int sz = z - (k / 9 - 1); h += ldf(fin + sk * k, (blk(0, 1, sz < 0 ? -1 : sz > 7 ? 1 : 0) * 512u + 64 * (sz & 7) + 8 * 0 + x) * 4u); }
#pragma unroll
            for (int c = 0; c < 3; ++c) h += ldf(vin + sk * c, (blk(0, 1, 0) * 512u + 64 * z + 8 * 0 + x) * 4u);
        }
#pragma unroll
        for (int k = 0; k < Q; ++k) { if (aosoa) stf(fout, ownA + k * 2048u, v[k] + h); else stf(fout + sk * k, own, v[k] + h); }
#pragma unroll
        for (int c = 0; c < 3; ++c) { if (aosoa) stf(vout, ownV + c * 2048u, uc[c] + ub[c] + ut[c]); else stf(vout + sk * c, own, uc[c] + ub[c] + ut[c]); }
        stf(rho, own, v[0] + uc[0]);
        // march: the plane above becomes the centre, the centre the plane below
#pragma unroll
        for (int c = 0; c < 3; ++c) { ub[c] = uc[c]; uc[c] = ut[c]; }
    }
}

struct Sched {
    std::string name;
    int nw, niter;
    std::vector<int> items;      // [wg][wave][iter]
    int variant = 0;             // 0 plain, 1 dependent neighbour-id load, 2 capped at 5 waves per SIMD, 3 both
    int halo_mask = ~0;          // which of the design's global fetches the "with halos" pass keeps
    int layout = 0;              // memory order of the blocks the items were built for (g_layout at build time)
};

static int NBg = 32;
static int g_layout = 0;     // 0: reference block order (bz fastest in memory); 1: x fastest in memory (an INTERNAL permutation the library could use)
static int bid(int bx, int by, int bz) { return g_layout == 0 ? (bx * NBg + by) * NBg + bz : (bz * NBg + by) * NBg + bx; }   // 1, 2: x fastest

// current product order: x-runs of 4, the 8 planes of a run on 8 consecutive workgroups, plane (x - bz) mod 8 on XCD x
static Sched sched_current()
{
    Sched s{"cur: 4x1 run, plane/XCD rotated, 1 plane per wave", 4, 1, {}};
    const int NB = NBg;
    for (int bz = 0; bz < NB; ++bz)
        for (int bx0 = 0; bx0 < NB; bx0 += 4)
            for (int by = 0; by < NB; ++by)
                for (int x = 0; x < 8; ++x) {
                    const int z = ((x - bz) % 8 + 8) % 8;
                    for (int w = 0; w < 4; ++w) {
                        int fl = F_C | F_B | F_T | F_S | F_N;
                        if (w == 0) fl |= F_W;
                        if (w == 3) fl |= F_E;
                        s.items.push_back(((bid(bx0 + w, by, bz) << 3) | z) | fl);
                    }
                }
    return s;
}

// z-march: workgroup = PX x PY patch of blocks, every wave marches through `layers` block layers (8 planes each).
// xcd_by_layer: all patches of block layer bz run on XCD bz % 8 (workgroup g -> XCD g % 8); else patches round-robin.
static Sched sched_zmarch(int PX, int PY, int layers, bool xcd_by_layer, bool x_fastest)
{
    char nm[160];
    snprintf(nm, sizeof nm, "zmarch %dx%d patch, %d layer(s), %s, %s sweep", PX, PY, layers, xcd_by_layer ? "XCD = layer group" : "round-robin", x_fastest ? "x-fastest" : "y-fastest");
    Sched s{nm, PX * PY, 8 * layers, {}};
    const int NB = NBg, npx = NB / PX, npy = NB / PY, ngl = NB / layers;   // ngl layer groups
    std::vector<std::vector<int>> wg;                                      // per workgroup: items
    auto emit = [&](int ppx, int ppy, int lg) {
        std::vector<int> v;
        for (int wj = 0; wj < PY; ++wj)
            for (int wi = 0; wi < PX; ++wi)
                for (int l = 0; l < layers; ++l)
                    for (int z = 0; z < 8; ++z) {
                        int fl = F_T;
                        if (l == 0 && z == 0) fl |= F_C | F_B;
                        if (wi == 0) fl |= F_W;
                        if (wi == PX - 1) fl |= F_E;
                        if (wj == 0) fl |= F_S;
                        if (wj == PY - 1) fl |= F_N;
                        v.push_back(((bid(ppx * PX + wi, ppy * PY + wj, lg * layers + l) << 3) | z) | fl);
                    }
        wg.push_back(v);
    };
    if (xcd_by_layer) {
        // workgroup g = 8 * i + xcd: layer group lg = 8 * round + xcd, patch index i inside the group
        const int per = npx * npy;
        for (int round = 0; round < (ngl + 7) / 8; ++round)
            for (int i = 0; i < per; ++i)
                for (int xcd = 0; xcd < 8; ++xcd) {
                    const int lg = round * 8 + xcd;
                    if (lg >= ngl) { wg.push_back(std::vector<int>((size_t)PX * PY * 8 * layers, -1)); continue; }
                    const int ppx = x_fastest ? i % npx : i / npy, ppy = x_fastest ? i / npx : i % npy;
                    emit(ppx, ppy, lg);
                }
    } else {
        for (int lg = 0; lg < ngl; ++lg)
            for (int i = 0; i < npx * npy; ++i) emit(x_fastest ? i % npx : i / npy, x_fastest ? i / npx : i % npy, lg);
    }
    for (auto &v : wg) s.items.insert(s.items.end(), v.begin(), v.end());
    return s;
}

// x-march: workgroup = PY blocks in y x 8 planes, marching through `len` blocks along x; XCD = bz % 8
static Sched sched_xmarch(int PY, int len, bool xcd_by_layer)
{
    char nm[160];
    snprintf(nm, sizeof nm, "xmarch %dx8 (y blocks x planes), %d blocks per march, %s", PY, len, xcd_by_layer ? "XCD = bz % 8" : "round-robin");
    Sched s{nm, PY * 8, len, {}};
    const int NB = NBg, npy = NB / PY, nseg = NB / len;
    auto emit = [&](int ppy, int bz, int seg) {
        for (int wj = 0; wj < PY; ++wj)
            for (int z = 0; z < 8; ++z)
                for (int i = 0; i < len; ++i) {
                    int fl = F_C;
                    if (z == 0) fl |= F_B;
                    if (z == 7) fl |= F_T;
                    if (i == 0) fl |= F_W;
                    if (i == len - 1) fl |= F_E;
                    if (wj == 0) fl |= F_S;
                    if (wj == PY - 1) fl |= F_N;
                    s.items.push_back(((bid(seg * len + i, ppy * PY + wj, bz) << 3) | z) | fl);
                }
    };
    if (xcd_by_layer) {
        for (int round = 0; round < NB / 8; ++round)
            for (int i = 0; i < npy * nseg; ++i)
                for (int xcd = 0; xcd < 8; ++xcd) emit(i % npy, round * 8 + xcd, i / npy);
    } else {
        for (int bz = 0; bz < NB; ++bz)
            for (int i = 0; i < npy * nseg; ++i) emit(i % npy, bz, i / npy);
    }
    return s;
}


// x-run workgroups (4 x-adjacent blocks, one plane each, like the product) with a configurable sweep:
// groups (run, by, bz) are ordered with `fast` dimensions first; ty > 0 sweeps by in tiles of ty (bz fastest inside a tile row);
// the 8 planes of a group go to 8 consecutive workgroups, plane (x - bz) mod 8 on XCD x.
static Sched sched_xrun(const char *sweep, int ty, bool halo_y = true)
{
    char nm[160];
    snprintf(nm, sizeof nm, "xrun 4x1, 1 plane per wave, rotated planes, sweep %s, y tile %d", sweep, ty);
    Sched s{nm, 4, 1, {}};
    const int NB = NBg;
    struct G { int bx0, by, bz; };
    std::vector<G> gs;
    for (int bx0 = 0; bx0 < NB; bx0 += 4) for (int by = 0; by < NB; ++by) for (int bz = 0; bz < NB; ++bz) gs.push_back({bx0, by, bz});
    auto key = [&](const G &g, char c) { return c == 'x' ? g.bx0 : c == 'y' ? (ty > 0 ? g.by % ty : g.by) : c == 'z' ? g.bz : /* 'Y' tile index */ (ty > 0 ? g.by / ty : 0); };
    const std::string sw = sweep;      // fastest first, e.g. "zyxY" / "yzYx"
    std::stable_sort(gs.begin(), gs.end(), [&](const G &a, const G &b) {
        for (int i = (int)sw.size() - 1; i >= 0; --i) { const int ka = key(a, sw[i]), kb = key(b, sw[i]); if (ka != kb) return ka < kb; }
        return false;
    });
    for (const G &g : gs)
        for (int x = 0; x < 8; ++x) {
            const int z = ((x - g.bz) % 8 + 8) % 8;
            for (int w = 0; w < 4; ++w) {
                int fl = F_C | F_B | F_T;
                if (halo_y) fl |= F_S | F_N;
                if (w == 0) fl |= F_W;
                if (w == 3) fl |= F_E;
                s.items.push_back(((bid(g.bx0 + w, g.by, g.bz) << 3) | z) | fl);
            }
        }
    return s;
}

// workgroup = all 8 planes of ONE block (velocity planes shared inside the workgroup), blocks in memory order
static Sched sched_block8()
{
    Sched s{"block per workgroup (8 waves = 8 planes), memory order", 8, 1, {}};
    const int n = NBg * NBg * NBg;
    for (int b = 0; b < n; ++b)
        for (int z = 0; z < 8; ++z) {
            int fl = F_C | F_W | F_E | F_S | F_N;
            if (z == 0) fl |= F_B;
            if (z == 7) fl |= F_T;
            s.items.push_back(((b << 3) | z) | fl);
        }
    return s;
}

// 2 x-adjacent blocks x 8 planes = 16 waves, memory order of the pair's first block
static Sched sched_pair16(int PX)
{
    char nm[96];
    snprintf(nm, sizeof nm, "%d x-adjacent blocks x 8 planes per workgroup, bz fastest", PX);
    Sched s{nm, PX * 8, 1, {}};
    const int NB = NBg;
    for (int bx0 = 0; bx0 < NB; bx0 += PX) for (int by = 0; by < NB; ++by) for (int bz = 0; bz < NB; ++bz)
        for (int w = 0; w < PX; ++w)
            for (int z = 0; z < 8; ++z) {
                int fl = F_C | F_S | F_N;
                if (z == 0) fl |= F_B;
                if (z == 7) fl |= F_T;
                if (w == 0) fl |= F_W;
                if (w == PX - 1) fl |= F_E;
                s.items.push_back(((bid(bx0 + w, by, bz) << 3) | z) | fl);
            }
    return s;
}

// z-march over one block with a STAGGERED, cyclic start plane (z0 = (wave + workgroup) % 8): at any instant the chip works on
// all 8 plane indices (address bits 8..10) instead of marching through them in lockstep
static Sched sched_zmarch_staggered(int PX, int PY)
{
    char nm[160];
    snprintf(nm, sizeof nm, "zmarch %dx%d patch, staggered cyclic start, XCD = layer", PX, PY);
    Sched s{nm, PX * PY, 8, {}};
    const int NB = NBg, npx = NB / PX, npy = NB / PY;
    for (int round = 0; round < NB / 8; ++round)
        for (int i = 0; i < npx * npy; ++i)
            for (int xcd = 0; xcd < 8; ++xcd) {
                const int bz = round * 8 + xcd, ppx = i / npy, ppy = i % npy;
                const int z0 = (i + xcd) % 8;                    // the whole patch marches together (LDS exchange per plane)
                for (int wj = 0; wj < PY; ++wj)
                    for (int wi = 0; wi < PX; ++wi)
                        for (int it = 0; it < 8; ++it) {
                            const int z = (z0 + it) % 8;
                            int fl = F_T;
                            if (it == 0 || z == 0) fl |= F_C | F_B;   // start, and the wrap 7 -> 0
                            if (wi == 0) fl |= F_W;
                            if (wi == PX - 1) fl |= F_E;
                            if (wj == 0) fl |= F_S;
                            if (wj == PY - 1) fl |= F_N;
                            s.items.push_back(((bid(ppx * PX + wi, ppy * PY + wj, bz) << 3) | z) | fl);
                        }
            }
    return s;
}


// brick: workgroup = PX x-adjacent blocks x PY y-adjacent blocks x PZ consecutive planes (PX * PY * PZ waves), waves ordered x fastest
// (same-plane x neighbours in neighbouring waves: what the product kernel's LDS column exchange needs). Blocks are visited in
// MEMORY order (bz fastest, then by, then bx); the 8 / PZ plane groups of a brick are consecutive workgroups, their order rotated
// with bz so that an XCD does not always see the same plane indices. Halo flags as the PRODUCT kernel would fetch them today:
// x columns only on the outer x faces of the brick, y rows and z+-1 velocity planes always (L1 / L2 hits inside the brick).
static Sched sched_brick(int PX, int PY, int PZ, bool rotate)
{
    char nm[160];
    snprintf(nm, sizeof nm, "brick %dx%dx%d (x, y blocks, planes), memory order%s", PX, PY, PZ, rotate ? ", plane groups rotated with bz" : "");
    Sched s{nm, PX * PY * PZ, 1, {}};
    const int NB = NBg, ng = 8 / PZ;
    for (int bx0 = 0; bx0 < NB; bx0 += PX)
        for (int by0 = 0; by0 < NB; by0 += PY)
            for (int bz = 0; bz < NB; ++bz)
                for (int gi = 0; gi < ng; ++gi) {
                    const int g = rotate ? (gi + bz) % ng : gi;
                    for (int pz = 0; pz < PZ; ++pz)
                        for (int wy = 0; wy < PY; ++wy)
                            for (int wx = 0; wx < PX; ++wx) {
                                int fl = F_C | F_B | F_T | F_S | F_N;
                                if (wx == 0) fl |= F_W;
                                if (wx == PX - 1) fl |= F_E;
                                s.items.push_back(((bid(bx0 + wx, by0 + wy, bz) << 3) | (g * PZ + pz)) | fl);
                            }
                }
    return s;
}


// x-run groups in memory order (x-fastest layout), but the 8 planes of a group all on ONE XCD (group g -> XCD g % 8), taken in turn
// with a start plane rotated by the XCD: workgroup 64 s + 8 j + x = plane (j + x) % 8 of group 8 s + x. The z+-1 velocity planes
// and the y neighbour group (8 groups on) then meet in the same L2; the x neighbour group does not.
static Sched sched_xrun_group_per_xcd(bool rotate)
{
    Sched s{rotate ? "4x1 runs in memory order, the 8 planes of a run on ONE XCD, start plane rotated" : "4x1 runs in memory order, the 8 planes of a run on ONE XCD", 4, 1, {}};
    const int NB = NBg;
    struct G { int bx0, by, bz; };
    std::vector<G> gs;
    for (int bz = 0; bz < NB; ++bz) for (int by = 0; by < NB; ++by) for (int bx0 = 0; bx0 < NB; bx0 += 4) gs.push_back({bx0, by, bz});
    for (size_t s0 = 0; s0 < gs.size(); s0 += 8)
        for (int j = 0; j < 8; ++j)
            for (int x = 0; x < 8; ++x) {
                const G &g = gs[s0 + x];
                const int z = rotate ? (j + x) % 8 : j;
                for (int w = 0; w < 4; ++w) {
                    int fl = F_C | F_B | F_T | F_S | F_N;
                    if (w == 0) fl |= F_W;
                    if (w == 3) fl |= F_E;
                    s.items.push_back(((bid(g.bx0 + w, g.by, g.bz) << 3) | z) | fl);
                }
            }
    return s;
}

// natural: block order, 4 consecutive planes per workgroup, nothing shared
static Sched sched_natural(int niter)
{
    char nm[96];
    snprintf(nm, sizeof nm, "natural block order, %d plane(s) per wave", niter);
    Sched s{nm, 4, niter, {}};
    const int n = NBg * NBg * NBg * 8;
    for (int i = 0; i < n; i += 4 * niter)
        for (int w = 0; w < 4; ++w)
            for (int it = 0; it < niter; ++it) {
                const int id = i + w * niter + it;
                int fl = F_T | F_W | F_E | F_S | F_N;
                if (it == 0) fl |= F_C | F_B;
                s.items.push_back(id | fl);
            }
    return s;
}

int main(int argc, char **argv)
{
    const int NB = argc > 1 ? atoi(argv[1]) : 32;
    const int reps = argc > 2 ? atoi(argv[2]) : 7;
    NBg = NB;
    const size_t nblk = (size_t)NB * NB * NB, sk = nblk * 512, cells = sk;
    float *fin, *fout, *vin, *vout, *rho;
    CK(hipMalloc(&fin, sk * Q * 4)); CK(hipMalloc(&fout, sk * Q * 4));
    CK(hipMalloc(&vin, sk * 3 * 4)); CK(hipMalloc(&vout, sk * 3 * 4)); CK(hipMalloc(&rho, sk * 4));
    CK(hipMemset(fin, 0, sk * Q * 4)); CK(hipMemset(vin, 0, sk * 3 * 4));
    int *d_meta;
    {
        std::vector<int> m(nblk * 32, 0);
        for (int bx = 0; bx < NB; ++bx) for (int by = 0; by < NB; ++by) for (int bz = 0; bz < NB; ++bz)
            for (int dd = 0; dd < 27; ++dd) {
                const int ox = dd % 3 - 1, oy = (dd / 3) % 3 - 1, oz = dd / 9 - 1;
                m[(size_t)bid(bx, by, bz) * 32 + dd] = bid((bx + ox + NB) % NB, (by + oy + NB) % NB, (bz + oz + NB) % NB);
            }
        CK(hipMalloc(&d_meta, m.size() * 4));
        CK(hipMemcpy(d_meta, m.data(), m.size() * 4, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    std::vector<Sched> all;
    const char *sel = argc > 3 ? argv[3] : "2";
    if (strchr(sel, '1')) {
        all.push_back(sched_current());
        all.push_back(sched_natural(1));
        all.push_back(sched_natural(8));
        all.push_back(sched_zmarch(4, 1, 1, true, false));
        all.push_back(sched_zmarch(4, 2, 1, true, false));
        all.push_back(sched_zmarch(4, 2, 1, true, true));
        all.push_back(sched_zmarch(4, 2, 1, false, false));
        all.push_back(sched_zmarch(4, 4, 1, true, false));
        all.push_back(sched_zmarch(8, 2, 1, true, false));
        all.push_back(sched_zmarch(2, 2, 1, true, false));
        all.push_back(sched_zmarch(4, 2, 4, true, false));
        all.push_back(sched_zmarch(4, 2, 4, false, false));
        all.push_back(sched_xmarch(1, 32, true));
        all.push_back(sched_xmarch(1, 16, true));
        all.push_back(sched_xmarch(2, 32, true));
        all.push_back(sched_xmarch(2, 16, true));
        all.push_back(sched_xmarch(1, 32, false));
    }
    if (strchr(sel, '3')) {
        auto with = [&](const char *nm, int variant, int hm) { Sched c = sched_current(); c.name = nm; c.variant = variant; c.halo_mask = hm; all.push_back(c); };
        with("cur, all halos", 0, ~0);
        with("cur, only T/B velocity planes", 0, F_C | F_B | F_T);
        with("cur, only y rows (S/N)", 0, F_C | F_S | F_N);
        with("cur, only outer x columns (W/E)", 0, F_C | F_W | F_E);
        with("cur, y rows + x columns", 0, F_C | F_S | F_N | F_W | F_E);
        with("cur, all halos, dependent neighbour-id load", 1, ~0);
        with("cur, all halos, 5 waves/SIMD", 2, ~0);
        with("cur, all halos, dependent load + 5 waves/SIMD", 3, ~0);
        with("cur, all halos, 3 waves/SIMD", 4, ~0);
        with("cur, all halos (again)", 0, ~0);
    }
    if (strchr(sel, '9')) {
        g_layout = 1;
        { Sched c = sched_xrun("xyz", 0); c.name = "[x-fastest layout] 4x1 runs swept x, y, z, planes of a run spread over the 8 XCDs (product)"; c.layout = 1; all.push_back(c); }
        { Sched c = sched_xrun_group_per_xcd(true); c.layout = 1; all.push_back(c); }
        { Sched c = sched_xrun_group_per_xcd(false); c.layout = 1; all.push_back(c); }
        { Sched c = sched_natural(1); c.name = "[x-fastest layout] memory order, 4 planes of a block per workgroup"; c.layout = 1; all.push_back(c); }
        { Sched c = sched_xrun("xyz", 0); c.name = "[x-fastest layout] 4x1 runs swept x, y, z, planes of a run spread over the 8 XCDs (product)"; c.layout = 1; all.push_back(c); }
        g_layout = 0;
    }
    if (strchr(sel, '8')) {        // block-major storage (all 27 populations of a block contiguous) on top of the x-fastest block order; floor only
        g_layout = 1;
        { Sched c = sched_xrun("xyz", 0); c.name = "[x-fastest layout, population-major] 4x1 runs swept x, y, z"; c.layout = 1; all.push_back(c); }
        { Sched c = sched_xrun("xyz", 0); c.name = "[x-fastest layout, BLOCK-major f[b][k][512]] 4x1 runs swept x, y, z (floor column only)"; c.layout = 2; c.halo_mask = F_C; all.push_back(c); }
        { Sched c = sched_natural(1); c.name = "[x-fastest layout, population-major] memory order, 4 planes per workgroup"; c.layout = 1; c.halo_mask = F_C; all.push_back(c); }
        { Sched c = sched_natural(1); c.name = "[x-fastest layout, BLOCK-major] memory order, 4 planes per workgroup (floor column only)"; c.layout = 2; c.halo_mask = F_C; all.push_back(c); }
        { Sched c = sched_xrun("xyz", 0); c.name = "[x-fastest layout, population-major] 4x1 runs swept x, y, z"; c.layout = 1; all.push_back(c); }
        g_layout = 0;
    }
    if (strchr(sel, '7')) {        // x-fastest internal block order: x-run schedules become memory-sequential
        all.push_back(sched_current());
        all.push_back(sched_natural(1));
        g_layout = 1;
        auto L1 = [&](Sched c, const char *tag) { c.name = std::string("[x-fastest layout] ") + tag; c.layout = 1; all.push_back(c); };
        L1(sched_current(), "cur sweep (y fastest, x, z), 4x1 runs, rotated planes");
        L1(sched_xrun("xyz", 0), "4x1 runs swept x fastest, then y, then z (memory order)");
        L1(sched_xrun("xzy", 0), "4x1 runs swept x, z, y");
        L1(sched_xrun("yxz", 0), "4x1 runs swept y, x, z");
        L1(sched_xrun("xyYz", 4), "4x1 runs swept x, y inside tiles of 4, z");
        L1(sched_natural(1), "natural (memory) order, 4 planes of a block per workgroup");
        g_layout = 0;
        all.push_back(sched_current());
    }
    if (strchr(sel, '6')) {        // bricks in memory order: orders the product kernel can run as it is (LUDWIG_XRUN = waves per workgroup)
        all.push_back(sched_current());
        all.push_back(sched_pair16(2));
        all.push_back(sched_brick(2, 1, 2, true));
        all.push_back(sched_brick(2, 1, 2, false));
        all.push_back(sched_brick(4, 1, 1, true));
        all.push_back(sched_brick(2, 1, 4, true));
        all.push_back(sched_brick(4, 1, 2, true));
        all.push_back(sched_brick(2, 2, 2, true));
        all.push_back(sched_brick(2, 1, 8, false));
        all.push_back(sched_brick(4, 1, 4, true));
        all.push_back(sched_brick(2, 2, 4, true));
        all.push_back(sched_brick(4, 2, 2, true));
        all.push_back(sched_current());
    }
    if (strchr(sel, '5')) {        // DRAM-order study: which sweeps of x-run workgroups keep the natural order's floor?
        all.push_back(sched_current());
        all.push_back(sched_natural(1));
        all.push_back(sched_block8());
        all.push_back(sched_pair16(2));
        all.push_back(sched_xrun("zyx", 0));
        all.push_back(sched_xrun("zxy", 0));
        all.push_back(sched_xrun("yzx", 0));
        all.push_back(sched_xrun("xyz", 0));
        all.push_back(sched_xrun("zyYx", 2));
        all.push_back(sched_xrun("zyYx", 4));
        all.push_back(sched_xrun("zyYx", 8));
        all.push_back(sched_xrun("yzYx", 4));
        all.push_back(sched_current());
    }
    if (strchr(sel, '4')) {        // the round-end floor lines (tools/final_profile.sh)
        all.push_back(sched_current());
        all.push_back(sched_natural(1));
        all.push_back(sched_block8());
        all.push_back(sched_current());
    }
    if (strchr(sel, '2')) {
        all.push_back(sched_current());
        all.push_back(sched_natural(1));
        all.push_back(sched_xrun("yxz", 0));          // = cur
        all.push_back(sched_xrun("zyx", 0));
        all.push_back(sched_xrun("zxy", 0));
        all.push_back(sched_xrun("xyz", 0));
        all.push_back(sched_xrun("xzy", 0));
        all.push_back(sched_xrun("yzx", 0));
        all.push_back(sched_xrun("yzYx", 2));
        all.push_back(sched_xrun("yzYx", 4));
        all.push_back(sched_xrun("yzYx", 8));
        all.push_back(sched_xrun("zyYx", 4));
        all.push_back(sched_xrun("yzxY", 4));
        all.push_back(sched_block8());
        all.push_back(sched_pair16(2));
        all.push_back(sched_zmarch_staggered(4, 1));
        all.push_back(sched_zmarch_staggered(4, 2));
        all.push_back(sched_current());
    }

    printf("# NB = %d (%zu cells), %d reps each, median; 244 B/cell compulsory (27+3 in, 27+3+1 out)\n", NB, cells, reps);
    printf("# columns: ms without any halo / velocity-plane refetch (mask: only F_C)  |  ms with the design's global halo + plane fetches\n");
    for (Sched &s : all) {
        // sanity: every (block, plane) exactly once
        std::vector<char> seen(nblk * 8, 0);
        size_t real = 0;
        for (int it : s.items) if (it >= 0) { ++real; if (seen[it & ID_MASK]++) { printf("%s: duplicate item\n", s.name.c_str()); return 1; } }
        if (real != nblk * 8 || s.items.size() % ((size_t)s.nw * s.niter)) { printf("%s: bad schedule (%zu items)\n", s.name.c_str(), real); return 1; }
        int *d;
        CK(hipMalloc(&d, s.items.size() * 4));
        CK(hipMemcpy(d, s.items.data(), s.items.size() * 4, hipMemcpyHostToDevice));
        const unsigned grid = (unsigned)(s.items.size() / ((size_t)s.nw * s.niter));
        float res[2];
        for (int pass = 0; pass < 2; ++pass) {
            // pass 0: no halo fetches, velocity: centre plane only (every item) = the 244 B/cell floor of this schedule
            const int mask = pass == 0 ? (ID_MASK | F_C) : (s.halo_mask | ID_MASK);
            std::vector<int> it2;
            if (pass == 0) {   // the floor needs the centre plane on every item
                it2 = s.items;
                for (int &v : it2) if (v >= 0) v |= F_C;
                CK(hipMemcpy(d, it2.data(), it2.size() * 4, hipMemcpyHostToDevice));
            } else CK(hipMemcpy(d, s.items.data(), s.items.size() * 4, hipMemcpyHostToDevice));
            std::vector<float> t;
            for (int r = 0; r < reps + 2; ++r) {
                CK(hipEventRecord(e0));
                switch (s.nw) {
                case 4:
                    if (s.variant == 0) hipLaunchKernelGGL(k_march<4>, dim3(grid), dim3(256), 0, 0, fin, fout, vin, vout, rho, d, s.niter, sk, NB, mask, d_meta, s.layout);
                    else if (s.variant == 1) hipLaunchKernelGGL((k_march<4, true, 0>), dim3(grid), dim3(256), 0, 0, fin, fout, vin, vout, rho, d, s.niter, sk, NB, mask, d_meta, s.layout);
                    else if (s.variant == 2) hipLaunchKernelGGL((k_march<4, false, 5>), dim3(grid), dim3(256), 0, 0, fin, fout, vin, vout, rho, d, s.niter, sk, NB, mask, d_meta, s.layout);
                    else if (s.variant == 3) hipLaunchKernelGGL((k_march<4, true, 5>), dim3(grid), dim3(256), 0, 0, fin, fout, vin, vout, rho, d, s.niter, sk, NB, mask, d_meta, s.layout);
                    else hipLaunchKernelGGL((k_march<4, false, 3>), dim3(grid), dim3(256), 0, 0, fin, fout, vin, vout, rho, d, s.niter, sk, NB, mask, d_meta, s.layout);
                    break;
                case 8: hipLaunchKernelGGL(k_march<8>, dim3(grid), dim3(512), 0, 0, fin, fout, vin, vout, rho, d, s.niter, sk, NB, mask, d_meta, s.layout); break;
                case 16: hipLaunchKernelGGL(k_march<16>, dim3(grid), dim3(1024), 0, 0, fin, fout, vin, vout, rho, d, s.niter, sk, NB, mask, d_meta, s.layout); break;
                default: printf("bad nw\n"); return 1;
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) t.push_back(ms);
            }
            std::sort(t.begin(), t.end());
            res[pass] = t[t.size() / 2];
        }
        printf("%-72s  floor %.4f ms (%5.2f TB/s)   with halos %.4f ms  -> %7.1f MLUPS-eq\n", s.name.c_str(), res[0], 244.0 * cells / res[0] / 1e9,
               res[1], cells / res[1] / 1e3);
        fflush(stdout);
        CK(hipFree(d));
    }
    return 0;
}
