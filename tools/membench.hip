// Diagnostic micro-benchmark (not part of the product): what does the MI355X memory system deliver for the
// access patterns a D3Q27 pull over 8^3 blocks can use?  hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o membench
//
// Layout as the engine: f[k][block][512] floats, 27 populations, n_blocks = NB^3 (periodic box, bz fastest).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>
#include <utility>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int Q = 27;

// D: classic float4 copy, grid-stride
__global__ void k_copy4(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

// B: one wave = one z-plane of one block, 27 aligned dword streams in, 27 out (no shifts)
template <int NPOP>
__global__ __launch_bounds__(256) void k_plane_copy(const float *__restrict__ in, float *__restrict__ out, const int *__restrict__ items, size_t sk)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = items[blockIdx.x * 4 + wave];
    const uint32_t off = ((uint32_t)(item >> 3) * 512u + (uint32_t)(item & 7) * 64u + (threadIdx.x & 63)) * 4u;
    float v[NPOP];
#pragma unroll
    for (int k = 0; k < NPOP; ++k) v[k] = *(const float *)((const char *)(in + sk * k) + off);
#pragma unroll
    for (int k = 0; k < NPOP; ++k) *(float *)((char *)(out + sk * k) + off) = v[k];
}

// C: one wave = 4 z-planes of one block with dwordx4 per lane (1 KiB per population per wave-instruction)
template <int NPOP>
__global__ __launch_bounds__(256) void k_quad_copy(const float *__restrict__ in, float *__restrict__ out, const int *__restrict__ items, size_t sk)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = items[blockIdx.x * 4 + wave];   // (block << 3) | z0, z0 in {0,4}
    const uint32_t off = ((uint32_t)(item >> 3) * 512u + (uint32_t)(item & 7) * 64u) * 4u + (threadIdx.x & 63) * 16u;
    float4 v[NPOP];
#pragma unroll
    for (int k = 0; k < NPOP; ++k) v[k] = *(const float4 *)((const char *)(in + sk * k) + off);
#pragma unroll
    for (int k = 0; k < NPOP; ++k) *(float4 *)((char *)(out + sk * k) + off) = v[k];
}

// A: pull with shifts (periodic NB^3 box, neighbours computed arithmetically), dword per lane, plane per wave
__device__ __forceinline__ int wrap(int v, int n) { return v < 0 ? v + n : (v >= n ? v - n : v); }
template <bool WRITE>
__global__ __launch_bounds__(256) void k_pull(const float *__restrict__ in, float *__restrict__ out, const int *__restrict__ items, size_t sk, int NB)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = items[blockIdx.x * 4 + wave];
    const int b = item >> 3, z = item & 7;
    const int bz = b % NB, by = (b / NB) % NB, bx = b / (NB * NB);
    const int lane = threadIdx.x & 63, x = lane & 7, y = lane >> 3;
    float v[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int cx = k % 3 - 1, cy = (k / 3) % 3 - 1, cz = k / 9 - 1;
        const int sx = x - cx, sy = y - cy, sz = z - cz;
        const int nbx = wrap(bx + (sx < 0 ? -1 : sx > 7 ? 1 : 0), NB), nby = wrap(by + (sy < 0 ? -1 : sy > 7 ? 1 : 0), NB);
        const int nbz = wrap(bz + (sz < 0 ? -1 : sz > 7 ? 1 : 0), NB);
        const uint32_t nb = (uint32_t)((nbx * NB + nby) * NB + nbz);
        const uint32_t off = (nb * 512u + (sx & 7) + 8 * (sy & 7) + 64 * (sz & 7)) * 4u;
        v[k] = *(const float *)((const char *)(in + sk * k) + off);
    }
    const uint32_t own = ((uint32_t)b * 512u + z * 64u + lane) * 4u;
    if (WRITE) {
#pragma unroll
        for (int k = 0; k < Q; ++k) *(float *)((char *)(out + sk * k) + own) = v[k];
    } else {
        float s = 0;
#pragma unroll
        for (int k = 0; k < Q; ++k) s += v[k];
        *(float *)((char *)out + own) = s;
    }
}


// L: LDS-staged patch pull. Workgroup = the same z-plane of PX x PY x/y-adjacent blocks (one wave per block plane).
// Every wave loads the 27 source planes of ITS block with aligned 256-B accesses into an LDS tile; the one-cell ring
// around the patch is loaded by a few lanes (partial lines, the only wasteful accesses); after one barrier every lane
// reads its 27 shifted values from LDS.
template <int PX, int PY, bool WRITE>
__global__ __launch_bounds__(64 * PX * PY) void k_patch_pull(const float *__restrict__ in, float *__restrict__ out, size_t sk, int NB)
{
    constexpr int TXS = 8 * PX + 8;          // row stride in floats: 8 (mod 32) -> conflict-free ds_read_b32 per half wave
    constexpr int TY = 8 * PY + 2;
    constexpr int TILE = TXS * TY;
    extern __shared__ float lds[];            // [27][TY][TXS]
    const int g = blockIdx.x;
    const int z = g & 7;                      // -> XCD z
    int pidx = g >> 3;
    const int npx = NB / PX, npy = NB / PY;
    const int ppx = pidx % npx; pidx /= npx;
    const int ppy = pidx % npy; const int bz = pidx / npy;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi = wave % PX, wj = wave / PX;
    const int bx = ppx * PX + wi, by = ppy * PY + wj;
    const int lane = threadIdx.x & 63, x = lane & 7, y = lane >> 3;
    const int tx = 8 * wi + x + 1, ty = 8 * wj + y + 1;     // position in the tile (ring at 0 and max)
    // aligned own loads: plane z - cz of block (bx,by,bz or its z neighbour)
    float v[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int cz = k / 9 - 1;
        const int sz = z - cz;
        const int nbz = wrap(bz + (sz < 0 ? -1 : sz > 7 ? 1 : 0), NB);
        const uint32_t nb = (uint32_t)((bx * NB + by) * NB + nbz);
        v[k] = *(const float *)((const char *)(in + sk * k) + (nb * 512u + 64 * (sz & 7) + lane) * 4u);
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) lds[k * TILE + ty * TXS + tx] = v[k];
    // ring: wave w takes populations k with k % (PX*PY) == w; lanes 0..(8*PY-1) column, next 8*PX row, +1 corner
    for (int k = wave; k < Q; k += PX * PY) {
        const int cx = k % 3 - 1, cy = (k / 3) % 3 - 1, cz = k / 9 - 1;
        const int sz = z - cz;
        const int nbz = wrap(bz + (sz < 0 ? -1 : sz > 7 ? 1 : 0), NB);
        int hx = -100, hy = -100;                     // tile coords (ring) this lane fills
        if (cx != 0 && lane < 8 * PY) { hx = cx == 1 ? 0 : 8 * PX + 1; hy = lane + 1; }
        else if (cy != 0 && lane >= 32 && lane < 32 + 8 * PX) { hx = lane - 32 + 1; hy = cy == 1 ? 0 : 8 * PY + 1; }
        else if (cx != 0 && cy != 0 && lane == 63) { hx = cx == 1 ? 0 : 8 * PX + 1; hy = cy == 1 ? 0 : 8 * PY + 1; }
        if (hx > -100) {
            // global cell of tile coord (hx,hy): patch origin cell = (8*ppx*PX, 8*ppy*PY)
            const int gx = 8 * ppx * PX + hx - 1, gy = 8 * ppy * PY + hy - 1;
            const int n = 8 * NB;
            const int wx = gx < 0 ? gx + n : (gx >= n ? gx - n : gx), wy = gy < 0 ? gy + n : (gy >= n ? gy - n : gy);
            const uint32_t nb = (uint32_t)(((wx >> 3) * NB + (wy >> 3)) * NB + nbz);
            lds[k * TILE + hy * TXS + hx] = *(const float *)((const char *)(in + sk * k) + (nb * 512u + 64 * (sz & 7) + 8 * (wy & 7) + (wx & 7)) * 4u);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int cx = k % 3 - 1, cy = (k / 3) % 3 - 1;
        v[k] = lds[k * TILE + (ty - cy) * TXS + (tx - cx)];
    }
    const uint32_t b = (uint32_t)((bx * NB + by) * NB + bz);
    const uint32_t own = (b * 512u + z * 64u + lane) * 4u;
    if (WRITE) {
#pragma unroll
        for (int k = 0; k < Q; ++k) *(float *)((char *)(out + sk * k) + own) = v[k];
    } else {
        float s = 0;
#pragma unroll
        for (int k = 0; k < Q; ++k) s += v[k];
        *(float *)((char *)out + own) = s;
    }
}


// R: read-only variants. each wave reads NPOP planes (aligned), sums, writes one dword per lane.
// STRIDE_MODE 0: reference layout f[k][b][512]; 1: block-major f[b][k][512]
template <int NPOP, int MODE>
__global__ __launch_bounds__(256) void k_plane_read(const float *__restrict__ in, float *__restrict__ out, const int *__restrict__ items, size_t sk)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = items[blockIdx.x * 4 + wave];
    const uint32_t b = item >> 3, z = item & 7, lane = threadIdx.x & 63;
    float s = 0;
    if (MODE == 0) {
        const uint32_t off = (b * 512u + z * 64u + lane) * 4u;
        float v[NPOP];
#pragma unroll
        for (int k = 0; k < NPOP; ++k) v[k] = *(const float *)((const char *)(in + sk * k) + off);
#pragma unroll
        for (int k = 0; k < NPOP; ++k) s += v[k];
    } else {
        const float *base = in + (size_t)b * 512 * NPOP;
        float v[NPOP];
#pragma unroll
        for (int k = 0; k < NPOP; ++k) v[k] = base[k * 512 + z * 64 + lane];
#pragma unroll
        for (int k = 0; k < NPOP; ++k) s += v[k];
    }
    out[(size_t)b * 512 + z * 64 + lane] = s;
}
__global__ void k_read4(const float4 *__restrict__ in, float *__restrict__ out, size_t n4)
{
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = in[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123.456f) out[threadIdx.x] = s;
}
__global__ void k_write4(float4 *__restrict__ out, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}


// L2: patch-persistent LDS pull. Workgroup = PX x PY x/y-adjacent blocks, loops over the 8 z-planes; the aligned loads
// of plane z+1 are issued before plane z is read back from LDS (software prefetch). Patches in natural (bz fastest) order.
template <int PX, int PY, bool WRITE>
__global__ __launch_bounds__(64 * PX * PY) void k_patch_loop(const float *__restrict__ in, float *__restrict__ out, size_t sk, int NB)
{
    constexpr int TXS = 8 * PX + 8;
    constexpr int TY = 8 * PY + 2;
    constexpr int TILE = TXS * TY;
    extern __shared__ float lds[];            // [27][TY][TXS]
    int pidx = blockIdx.x;
    const int bz = pidx % NB; pidx /= NB;     // bz fastest: consecutive workgroups = z-neighbour patches (adjacent memory)
    const int npy = NB / PY;
    const int ppy = pidx % npy; const int ppx = pidx / npy;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wi = wave % PX, wj = wave / PX;
    const int bx = ppx * PX + wi, by = ppy * PY + wj;
    const int lane = threadIdx.x & 63, x = lane & 7, y = lane >> 3;
    const int tx = 8 * wi + x + 1, ty = 8 * wj + y + 1;
    const uint32_t bcol = (uint32_t)((bx * NB + by) * NB);
    float v[Q];
    auto load_plane = [&](int z) {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            const int cz = k / 9 - 1;
            const int sz = z - cz;
            const int nbz = wrap(bz + (sz < 0 ? -1 : sz > 7 ? 1 : 0), NB);
            v[k] = *(const float *)((const char *)(in + sk * k) + ((bcol + nbz) * 512u + 64 * (sz & 7) + lane) * 4u);
        }
    };
    load_plane(0);
    for (int z = 0; z < 8; ++z) {
#pragma unroll
        for (int k = 0; k < Q; ++k) lds[k * TILE + ty * TXS + tx] = v[k];
        for (int k = wave; k < Q; k += PX * PY) {
            const int cx = k % 3 - 1, cy = (k / 3) % 3 - 1, cz = k / 9 - 1;
            const int sz = z - cz;
            const int nbz = wrap(bz + (sz < 0 ? -1 : sz > 7 ? 1 : 0), NB);
            int hx = -100, hy = -100;
            if (cx != 0 && lane < 8 * PY) { hx = cx == 1 ? 0 : 8 * PX + 1; hy = lane + 1; }
            else if (cy != 0 && lane >= 32 && lane < 32 + 8 * PX) { hx = lane - 32 + 1; hy = cy == 1 ? 0 : 8 * PY + 1; }
            else if (cx != 0 && cy != 0 && lane == 63) { hx = cx == 1 ? 0 : 8 * PX + 1; hy = cy == 1 ? 0 : 8 * PY + 1; }
            if (hx > -100) {
                const int gx = 8 * ppx * PX + hx - 1, gy = 8 * ppy * PY + hy - 1;
                const int n = 8 * NB;
                const int wx = gx < 0 ? gx + n : (gx >= n ? gx - n : gx), wy = gy < 0 ? gy + n : (gy >= n ? gy - n : gy);
                const uint32_t nb = (uint32_t)(((wx >> 3) * NB + (wy >> 3)) * NB + nbz);
                lds[k * TILE + hy * TXS + hx] = *(const float *)((const char *)(in + sk * k) + (nb * 512u + 64 * (sz & 7) + 8 * (wy & 7) + (wx & 7)) * 4u);
            }
        }
        __syncthreads();
        float r[Q];
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            const int cx = k % 3 - 1, cy = (k / 3) % 3 - 1;
            r[k] = lds[k * TILE + (ty - cy) * TXS + (tx - cx)];
        }
        if (z < 7) load_plane(z + 1);          // prefetch: in flight while plane z is consumed
        __syncthreads();
        const uint32_t own = ((bcol + bz) * 512u + z * 64u + lane) * 4u;
        if (WRITE) {
#pragma unroll
            for (int k = 0; k < Q; ++k) *(float *)((char *)(out + sk * k) + own) = r[k];
        } else {
            float s = 0;
#pragma unroll
            for (int k = 0; k < Q; ++k) s += r[k];
            *(float *)((char *)out + own) = s;
        }
    }
}


// S: pull-only with selectable shift directions (which of cx, cy, cz are honoured)
template <bool SX, bool SY, bool SZ>
__global__ __launch_bounds__(256) void k_pull_sel(const float *__restrict__ in, float *__restrict__ out, const int *__restrict__ items, size_t sk, int NB)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = items[blockIdx.x * 4 + wave];
    const int b = item >> 3, z = item & 7;
    const int bz = b % NB, by = (b / NB) % NB, bx = b / (NB * NB);
    const int lane = threadIdx.x & 63, x = lane & 7, y = lane >> 3;
    float v[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int cx = SX ? k % 3 - 1 : 0, cy = SY ? (k / 3) % 3 - 1 : 0, cz = SZ ? k / 9 - 1 : 0;
        const int sx = x - cx, sy = y - cy, sz = z - cz;
        const int nbx = wrap(bx + (sx < 0 ? -1 : sx > 7 ? 1 : 0), NB), nby = wrap(by + (sy < 0 ? -1 : sy > 7 ? 1 : 0), NB);
        const int nbz = wrap(bz + (sz < 0 ? -1 : sz > 7 ? 1 : 0), NB);
        const uint32_t nb = (uint32_t)((nbx * NB + nby) * NB + nbz);
        const uint32_t off = (nb * 512u + (sx & 7) + 8 * (sy & 7) + 64 * (sz & 7)) * 4u;
        v[k] = *(const float *)((const char *)(in + sk * k) + off);
    }
    const uint32_t own = ((uint32_t)b * 512u + z * 64u + lane) * 4u;
    float s = 0;
#pragma unroll
    for (int k = 0; k < Q; ++k) s += v[k];
    *(float *)((char *)out + own) = s;
}


// X: x-shift decomposed. MODE 0: shift wraps inside the own row (misalignment only, no neighbour block);
// MODE 1: only the face lane (x==0 / x==7) reads the neighbour block's column, all other lanes read their own cell
// (aligned) ; MODE 2: as the real pull
template <int MODE>
__global__ __launch_bounds__(256) void k_pull_x(const float *__restrict__ in, float *__restrict__ out, const int *__restrict__ items, size_t sk, int NB)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = items[blockIdx.x * 4 + wave];
    const int b = item >> 3, z = item & 7;
    const int bz = b % NB, by = (b / NB) % NB, bx = b / (NB * NB);
    const int lane = threadIdx.x & 63, x = lane & 7, y = lane >> 3;
    float v[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int cx = k % 3 - 1;
        int sx = x - cx;
        int nbx = bx;
        if (MODE == 0) { sx &= 7; }
        else if (MODE == 1) { if (sx < 0 || sx > 7) { nbx = wrap(bx + (sx < 0 ? -1 : 1), NB); sx &= 7; } else sx = x; }
        else { nbx = wrap(bx + (sx < 0 ? -1 : sx > 7 ? 1 : 0), NB); sx &= 7; }
        const uint32_t nb = (uint32_t)((nbx * NB + by) * NB + bz);
        const uint32_t off = (nb * 512u + sx + 8 * y + 64 * z) * 4u;
        v[k] = *(const float *)((const char *)(in + sk * k) + off);
    }
    const uint32_t own = ((uint32_t)b * 512u + z * 64u + lane) * 4u;
    float s = 0;
#pragma unroll
    for (int k = 0; k < Q; ++k) s += v[k];
    *(float *)((char *)out + own) = s;
}


// N: aligned plane copy with non-temporal loads and/or stores
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_plane_copy_nt(const float *__restrict__ in, float *__restrict__ out, const int *__restrict__ items, size_t sk)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = items[blockIdx.x * 4 + wave];
    const uint32_t off = ((uint32_t)(item >> 3) * 512u + (uint32_t)(item & 7) * 64u + (threadIdx.x & 63)) * 4u;
    float v[Q];
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const float *p = (const float *)((const char *)(in + sk * k) + off);
        v[k] = NTL ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        float *p = (float *)((char *)(out + sk * k) + off);
        if (NTS) __builtin_nontemporal_store(v[k], p); else *p = v[k];
    }
}

// write-only, plane per wave
__global__ __launch_bounds__(256) void k_plane_write(float *__restrict__ out, const int *__restrict__ items, size_t sk)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = items[blockIdx.x * 4 + wave];
    const uint32_t off = ((uint32_t)(item >> 3) * 512u + (uint32_t)(item & 7) * 64u + (threadIdx.x & 63)) * 4u;
#pragma unroll
    for (int k = 0; k < Q; ++k) *(float *)((char *)(out + sk * k) + off) = (float)k;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F> float run(F f, int reps)
    {
        f(); CK(hipDeviceSynchronize());
        std::vector<float> t;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        return t[t.size() / 2];
    }
};

int main(int argc, char **argv)
{
    const int NB = argc > 1 ? atoi(argv[1]) : 32;
    const size_t nblocks = (size_t)NB * NB * NB, sk = nblocks * 512, n = sk * Q;
    float *in, *out;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
    CK(hipMemset(in, 0, n * 4)); CK(hipMemset(out, 0, n * 4));
    // item lists
    auto upload = [&](const std::vector<int> &v) { int *d; CK(hipMalloc(&d, v.size() * 4)); CK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice)); return d; };
    std::vector<int> nat, quad, pxcd;
    for (size_t b = 0; b < nblocks; ++b) for (int z = 0; z < 8; ++z) nat.push_back((int)(b << 3) | z);
    for (size_t b = 0; b < nblocks; ++b) { quad.push_back((int)(b << 3)); quad.push_back((int)(b << 3) | 4); }
    // plane-per-XCD with 2x2 patches: slot g = 8*patch + z
    for (int bz = 0; bz < NB; ++bz) for (int py = 0; py < NB / 2; ++py) for (int px = 0; px < NB / 2; ++px)
        for (int z = 0; z < 8; ++z) for (int w = 0; w < 4; ++w) {
            const int bx = 2 * px + (w & 1), by = 2 * py + (w >> 1);
            pxcd.push_back((((bx * NB + by) * NB + bz) << 3) | z);
        }
    int *d_nat = upload(nat), *d_quad = upload(quad), *d_pxcd = upload(pxcd);
    Timer T;
    const double cells = (double)sk;
    auto report = [&](const char *name, float ms, double bytes_per_cell) {
        printf("%-34s %8.4f ms  %7.1f GB/s  (%5.1f B/cell)  -> %8.1f MLUPS-equivalent\n", name, ms, bytes_per_cell * cells / ms / 1e6, bytes_per_cell, cells / ms / 1e3);
    };
    const unsigned gw = (unsigned)(nblocks * 8 / 4), gq = (unsigned)(nblocks * 2 / 4);
    report("D float4 copy (27 pops)", T.run([&] { hipLaunchKernelGGL(k_copy4, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)in, (float4 *)out, n / 4); }, 7), 216);
    report("B plane copy dword, natural", T.run([&] { hipLaunchKernelGGL(k_plane_copy<27>, dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk); }, 7), 216);
    report("B plane copy dword, plane/XCD", T.run([&] { hipLaunchKernelGGL(k_plane_copy<27>, dim3(gw), dim3(256), 0, 0, in, out, d_pxcd, sk); }, 7), 216);
    report("C quad copy dwordx4, natural", T.run([&] { hipLaunchKernelGGL(k_quad_copy<27>, dim3(gq), dim3(256), 0, 0, in, out, d_quad, sk); }, 7), 216);
    report("A pull+write dword, natural", T.run([&] { hipLaunchKernelGGL(k_pull<true>, dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk, NB); }, 7), 216);
    report("A pull+write dword, plane/XCD", T.run([&] { hipLaunchKernelGGL(k_pull<true>, dim3(gw), dim3(256), 0, 0, in, out, d_pxcd, sk, NB); }, 7), 216);
    report("A pull only (read), natural", T.run([&] { hipLaunchKernelGGL(k_pull<false>, dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk, NB); }, 7), 112);
    report("A pull only (read), plane/XCD", T.run([&] { hipLaunchKernelGGL(k_pull<false>, dim3(gw), dim3(256), 0, 0, in, out, d_pxcd, sk, NB); }, 7), 112);
    report("write only 27 pops, natural", T.run([&] { hipLaunchKernelGGL(k_plane_write, dim3(gw), dim3(256), 0, 0, out, d_nat, sk); }, 7), 108);
    report("write only 27 pops, plane/XCD", T.run([&] { hipLaunchKernelGGL(k_plane_write, dim3(gw), dim3(256), 0, 0, out, d_pxcd, sk); }, 7), 108);

    {
        constexpr int PX = 4, PY = 2;
        const size_t ldsb = (size_t)27 * (8 * PX + 8) * (8 * PY + 2) * 4;
        const unsigned gp = (unsigned)(nblocks / (PX * PY) * 8);
        CK(hipFuncSetAttribute((const void *)k_patch_pull<PX, PY, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        CK(hipFuncSetAttribute((const void *)k_patch_pull<PX, PY, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        report("L patch 4x2 LDS pull+write", T.run([&] { hipLaunchKernelGGL((k_patch_pull<PX, PY, true>), dim3(gp), dim3(64 * PX * PY), ldsb, 0, in, out, sk, NB); }, 7), 216);
        report("L patch 4x2 LDS pull only", T.run([&] { hipLaunchKernelGGL((k_patch_pull<PX, PY, false>), dim3(gp), dim3(64 * PX * PY), ldsb, 0, in, out, sk, NB); }, 7), 112);
    }
    {
        constexpr int PX = 2, PY = 2;
        const size_t ldsb = (size_t)27 * (8 * PX + 8) * (8 * PY + 2) * 4;
        const unsigned gp = (unsigned)(nblocks / (PX * PY) * 8);
        CK(hipFuncSetAttribute((const void *)k_patch_pull<PX, PY, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        CK(hipFuncSetAttribute((const void *)k_patch_pull<PX, PY, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        report("L patch 2x2 LDS pull+write", T.run([&] { hipLaunchKernelGGL((k_patch_pull<PX, PY, true>), dim3(gp), dim3(64 * PX * PY), ldsb, 0, in, out, sk, NB); }, 7), 216);
        report("L patch 2x2 LDS pull only", T.run([&] { hipLaunchKernelGGL((k_patch_pull<PX, PY, false>), dim3(gp), dim3(64 * PX * PY), ldsb, 0, in, out, sk, NB); }, 7), 112);
    }
    {
        constexpr int PX = 4, PY = 4;
        const size_t ldsb = (size_t)27 * (8 * PX + 8) * (8 * PY + 2) * 4;
        const unsigned gp = (unsigned)(nblocks / (PX * PY) * 8);
        CK(hipFuncSetAttribute((const void *)k_patch_pull<PX, PY, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        CK(hipFuncSetAttribute((const void *)k_patch_pull<PX, PY, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
        report("L patch 4x4 LDS pull+write", T.run([&] { hipLaunchKernelGGL((k_patch_pull<PX, PY, true>), dim3(gp), dim3(64 * PX * PY), ldsb, 0, in, out, sk, NB); }, 7), 216);
        report("L patch 4x4 LDS pull only", T.run([&] { hipLaunchKernelGGL((k_patch_pull<PX, PY, false>), dim3(gp), dim3(64 * PX * PY), ldsb, 0, in, out, sk, NB); }, 7), 112);
    }

    report("R float4 read-only contiguous", T.run([&] { hipLaunchKernelGGL(k_read4, dim3(256 * 16), dim3(256), 0, 0, (const float4 *)in, out, n / 4); }, 7), 108);
    report("W float4 write-only contiguous", T.run([&] { hipLaunchKernelGGL(k_write4, dim3(256 * 16), dim3(256), 0, 0, (float4 *)out, n / 4); }, 7), 108);
    report("R plane read 27 streams natural", T.run([&] { hipLaunchKernelGGL((k_plane_read<27, 0>), dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk); }, 7), 112);
    report("R plane read 27 streams plane/XCD", T.run([&] { hipLaunchKernelGGL((k_plane_read<27, 0>), dim3(gw), dim3(256), 0, 0, in, out, d_pxcd, sk); }, 7), 112);
    report("R plane read block-major natural", T.run([&] { hipLaunchKernelGGL((k_plane_read<27, 1>), dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk); }, 7), 112);
    report("R plane read block-major plane/XCD", T.run([&] { hipLaunchKernelGGL((k_plane_read<27, 1>), dim3(gw), dim3(256), 0, 0, in, out, d_pxcd, sk); }, 7), 112);

    // column orders: WG = 4 consecutive planes of one block (1 KiB per population), bz fastest; (bx,by) columns
    // visited in different 2-D orders
    typedef std::vector<std::pair<int,int>> Cols;
    auto col_items = [&](const Cols &cols) {
        std::vector<int> v;
        for (auto &c : cols) for (int bz = 0; bz < NB; ++bz) { int b = (c.first * NB + c.second) * NB + bz; for (int z = 0; z < 8; ++z) v.push_back((b << 3) | z); }
        return upload(v);
    };
    auto tiled = [&](int tx, int ty) { Cols c;
        for (int Y = 0; Y < NB; Y += ty) for (int X = 0; X < NB; X += tx) for (int y = Y; y < Y + ty; ++y) for (int x = X; x < X + tx; ++x) c.push_back({x, y}); return c; };
    auto xouter = [&]() { Cols c; for (int x = 0; x < NB; ++x) for (int y = 0; y < NB; ++y) c.push_back({x, y}); return c; };
    auto morton = [&]() { Cols c; for (int m = 0; m < NB * NB; ++m) { int x = 0, y = 0; for (int i = 0; i < 8; ++i) { x |= ((m >> (2 * i)) & 1) << i; y |= ((m >> (2 * i + 1)) & 1) << i; } c.push_back({x, y}); } return c; };
    struct { const char *name; int *items; } orders[] = {
        {"cols x-outer (natural)", col_items(xouter())},
        {"cols x-inner", col_items(tiled(NB, 1))},
        {"cols tile 2x2", col_items(tiled(2, 2))},
        {"cols tile 4x4", col_items(tiled(4, 4))},
        {"cols tile 8x8", col_items(tiled(8, 8))},
        {"cols tile 8x2", col_items(tiled(8, 2))},
        {"cols morton", col_items(morton())},
    };
    for (auto &o : orders) {
        char nm[128];
        snprintf(nm, sizeof nm, "A pull only, %s", o.name);
        report(nm, T.run([&] { hipLaunchKernelGGL(k_pull<false>, dim3(gw), dim3(256), 0, 0, in, out, o.items, sk, NB); }, 7), 112);
        snprintf(nm, sizeof nm, "A pull+write, %s", o.name);
        report(nm, T.run([&] { hipLaunchKernelGGL(k_pull<true>, dim3(gw), dim3(256), 0, 0, in, out, o.items, sk, NB); }, 7), 216);
        snprintf(nm, sizeof nm, "R aligned read, %s", o.name);
        report(nm, T.run([&] { hipLaunchKernelGGL((k_plane_read<27, 0>), dim3(gw), dim3(256), 0, 0, in, out, o.items, sk); }, 7), 112);
    }

#define RUN_LOOP(PX_, PY_) { \
        const size_t ldsb = (size_t)27 * (8 * PX_ + 8) * (8 * PY_ + 2) * 4; \
        const unsigned gp = (unsigned)(nblocks / (PX_ * PY_)); \
        CK(hipFuncSetAttribute((const void *)k_patch_loop<PX_, PY_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)); \
        CK(hipFuncSetAttribute((const void *)k_patch_loop<PX_, PY_, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb)); \
        report("L2 patch-loop " #PX_ "x" #PY_ " pull+write", T.run([&] { hipLaunchKernelGGL((k_patch_loop<PX_, PY_, true>), dim3(gp), dim3(64 * PX_ * PY_), ldsb, 0, in, out, sk, NB); }, 7), 216); \
        report("L2 patch-loop " #PX_ "x" #PY_ " pull only", T.run([&] { hipLaunchKernelGGL((k_patch_loop<PX_, PY_, false>), dim3(gp), dim3(64 * PX_ * PY_), ldsb, 0, in, out, sk, NB); }, 7), 112); }
    RUN_LOOP(2, 2)
    RUN_LOOP(4, 2)
    RUN_LOOP(2, 4)
    RUN_LOOP(4, 4)
    RUN_LOOP(2, 1)

#define RUN_SEL(a, b, c) report("S pull-only shifts x" #a " y" #b " z" #c, T.run([&] { hipLaunchKernelGGL((k_pull_sel<a, b, c>), dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk, NB); }, 7), 112);
    RUN_SEL(0, 0, 0) RUN_SEL(0, 0, 1) RUN_SEL(0, 1, 0) RUN_SEL(1, 0, 0) RUN_SEL(1, 1, 0) RUN_SEL(0, 1, 1) RUN_SEL(1, 0, 1) RUN_SEL(1, 1, 1)

    report("X mode0 misaligned rows only", T.run([&] { hipLaunchKernelGGL((k_pull_x<0>), dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk, NB); }, 7), 112);
    report("X mode1 neighbour column only", T.run([&] { hipLaunchKernelGGL((k_pull_x<1>), dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk, NB); }, 7), 112);
    report("X mode2 real x pull", T.run([&] { hipLaunchKernelGGL((k_pull_x<2>), dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk, NB); }, 7), 112);

    report("N copy plain", T.run([&] { hipLaunchKernelGGL((k_plane_copy_nt<false, false>), dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk); }, 9), 216);
    report("N copy nt loads", T.run([&] { hipLaunchKernelGGL((k_plane_copy_nt<true, false>), dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk); }, 9), 216);
    report("N copy nt stores", T.run([&] { hipLaunchKernelGGL((k_plane_copy_nt<false, true>), dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk); }, 9), 216);
    report("N copy nt both", T.run([&] { hipLaunchKernelGGL((k_plane_copy_nt<true, true>), dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk); }, 9), 216);
    report("N copy plain (again)", T.run([&] { hipLaunchKernelGGL((k_plane_copy_nt<false, false>), dim3(gw), dim3(256), 0, 0, in, out, d_nat, sk); }, 9), 216);
    return 0;
}
