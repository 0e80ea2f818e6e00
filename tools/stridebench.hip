// Diagnostic micro-benchmark (not part of the product): step time of the stream-collide data movement against the DISTANCE
// between two populations in device memory, at block granularity, in one process with one allocation.
//   hipcc --offload-arch=gfx950 -O3 tools/stridebench.hip -o tools/stridebench
//   tools/stridebench NB reps mode from_blocks to_blocks step_blocks [in_out_gap_blocks]
// Data movement of the product's x-run kernel on the library's internal layout (x fastest in memory), nothing else: workgroup =
// the same 8x8 plane of 4 x-consecutive blocks, the 8 planes of a run on 8 consecutive workgroups with plane (slot - bz) mod 8,
// runs swept in memory order; per wave 27 aligned plane loads (plane z - cz of the block or of its z neighbour), the centre and
// z+-1 velocity planes, 27 + 3 non-temporal plane stores. The population stride S is a kernel argument: element (cell, block b,
// population k) lives at k * S + 512 b + cell, so a sweep over S costs one launch series per value, no re-allocation.
//   mode 0: f_in and f_out both with stride S, f_out placed `gap` blocks (default: 27 S rounded to 2 MiB) behind f_in.
//   mode 1: loads only; mode 2: stores only (which side the bad distances hurt).
//   mode 3 / 4: BLOCK-major storage f[block][k][512] with 32-bit / 64-bit per-lane addresses (no stride at all).
//   mode 5 / 6: the same with the planes outermost inside a block, f[block][z][k][64].
// Output: one line per S: blocks, MiB, ms per launch.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int Q = 27;

__device__ __forceinline__ float ldf(const float *base, uint32_t off) { return *(const float *)((const char *)base + off); }
__device__ __forceinline__ void stf(float *base, uint32_t off, float v) { __builtin_nontemporal_store(v, (float *)((char *)base + off)); }

template <int MODE>
__global__ __launch_bounds__(256) void k_move(const float *__restrict__ fin, float *__restrict__ fout, const float *__restrict__ vin,
                                              float *__restrict__ vout, size_t S, int NB)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x >> 3, slot = blockIdx.x & 7;          // run g (4 blocks, consecutive in memory), XCD slot
    const int b = 4 * g + wave;
    const int bz = b / (NB * NB);
    const int z = (slot - bz) & 7;
    const int bup = bz == NB - 1 ? b - (NB - 1) * NB * NB : b + NB * NB, bdn = bz == 0 ? b + (NB - 1) * NB * NB : b - NB * NB;
    const uint32_t own = ((uint32_t)b * 512u + z * 64u + lane) * 4u;
    float v[Q], u[3] = {0.f, 0.f, 0.f};
    if (MODE != 2) {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
            const int cz = k / 9 - 1, sz = z - cz;
            const uint32_t nb = sz < 0 ? bdn : (sz > 7 ? bup : b);
            v[k] = ldf(fin + S * k, ((uint32_t)nb * 512u + 64 * (sz & 7) + lane) * 4u);
        }
        const uint32_t oT = ((uint32_t)(z == 7 ? bup : b) * 512u + 64 * ((z + 1) & 7) + lane) * 4u;
        const uint32_t oB = ((uint32_t)(z == 0 ? bdn : b) * 512u + 64 * ((z - 1) & 7) + lane) * 4u;
#pragma unroll
        for (int c = 0; c < 3; ++c) u[c] = ldf(vin + S * c, own) + ldf(vin + S * c, oT) + ldf(vin + S * c, oB);
    } else {
#pragma unroll
        for (int k = 0; k < Q; ++k) v[k] = (float)(k + lane);
    }
    if (MODE != 1) {
#pragma unroll
        for (int k = 0; k < Q; ++k) stf(fout + S * k, own, v[k] + u[k % 3]);
#pragma unroll
        for (int c = 0; c < 3; ++c) stf(vout + S * c, own, u[c] + v[c]);
    } else {
        float s = u[0] + u[1] + u[2];
#pragma unroll
        for (int k = 0; k < Q; ++k) s += v[k];
        if (s == 123.456f) stf(vout, own, s);                      // keeps the loads alive
    }
}

// Block-major storage: f[block][k][512], vel[block][c][512] - the 27 populations of a block are contiguous (54 KiB), so there is no
// distance between populations to get wrong. WIDE: 64-bit per-lane addresses (what a level above 77 672 blocks - 4 GiB of f - needs).
template <bool WIDE, bool PLANE = false>
__global__ __launch_bounds__(256) void k_move_bm(const float *__restrict__ fin, float *__restrict__ fout, const float *__restrict__ vin,
                                                 float *__restrict__ vout, int NB)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x >> 3, slot = blockIdx.x & 7;
    const int b = 4 * g + wave;
    const int bz = b / (NB * NB);
    const int z = (slot - bz) & 7;
    const int bup = bz == NB - 1 ? b - (NB - 1) * NB * NB : b + NB * NB, bdn = bz == 0 ? b + (NB - 1) * NB * NB : b - NB * NB;
    // per-lane block select as in the product (there: y rows of the y neighbour); here lanes of row 0 / 7 take the same block, which
    // still forces the address through vector registers
    const int sel_own = (lane == 63 && NB < 0) ? bup : b;   // NB < 0 never holds: a per-lane value the compiler cannot fold
    float v[Q], u[3];
    // PLANE: plane-major inside the block, f[block][z][k][64]: the 27 planes a wave stores are one contiguous 6.75 KiB piece
    auto addr = [&](const float *base, int blk, uint32_t comps, uint32_t inner) -> const float * {
        if (PLANE) { const uint32_t k = inner >> 11, zz = (inner >> 8) & 7, l4 = inner & 255; inner = zz * (comps * 256u) + k * 256u + l4; }
        if (WIDE) return (const float *)((const char *)base + (uint64_t)(uint32_t)blk * (uint64_t)(comps * 2048u) + inner);
        return (const float *)((const char *)base + (uint32_t)((uint32_t)blk * (comps * 2048u) + inner));
    };
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const int cz = k / 9 - 1, sz = z - cz;
        const int nb = sz < 0 ? bdn : (sz > 7 ? bup : sel_own);
        v[k] = *addr(fin, nb, 27u, (uint32_t)(k * 2048 + 256 * (sz & 7) + lane * 4));
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
        u[c] = *addr(vin, sel_own, 3u, (uint32_t)(c * 2048 + 256 * z + lane * 4)) + *addr(vin, z == 7 ? bup : sel_own, 3u, (uint32_t)(c * 2048 + 256 * ((z + 1) & 7) + lane * 4)) +
               *addr(vin, z == 0 ? bdn : sel_own, 3u, (uint32_t)(c * 2048 + 256 * ((z - 1) & 7) + lane * 4));
#pragma unroll
    for (int k = 0; k < Q; ++k) __builtin_nontemporal_store(v[k] + u[k % 3], (float *)addr(fout, sel_own, 27u, (uint32_t)(k * 2048 + 256 * z + lane * 4)));
#pragma unroll
    for (int c = 0; c < 3; ++c) __builtin_nontemporal_store(u[c] + v[c], (float *)addr(vout, sel_own, 3u, (uint32_t)(c * 2048 + 256 * z + lane * 4)));
}

int main(int argc, char **argv)
{
    const int NB = argc > 1 ? atoi(argv[1]) : 32;
    const int reps = argc > 2 ? atoi(argv[2]) : 10;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const long nblk = (long)NB * NB * NB;
    const long from = argc > 4 ? atol(argv[4]) : nblk, to = argc > 5 ? atol(argv[5]) : nblk + nblk / 2, step = argc > 6 ? atol(argv[6]) : 64;
    const long gap_arg = argc > 7 ? atol(argv[7]) : -1;
    const size_t max_f = (size_t)to * 512 * Q * 4 + ((size_t)8 << 20);
    char *buf = nullptr;
    float *vin = nullptr, *vout = nullptr;
    CK(hipMalloc((void **)&buf, 2 * max_f + ((size_t)64 << 20)));
    CK(hipMalloc((void **)&vin, (size_t)to * 512 * 3 * 4)); CK(hipMalloc((void **)&vout, (size_t)to * 512 * 3 * 4));
    CK(hipMemset(buf, 0, 2 * max_f)); CK(hipMemset(vin, 0, (size_t)to * 512 * 3 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((unsigned)(nblk / 4 * 8)), block(256);
    printf("# NB %d (%ld blocks), mode %d, reps %d; stride from %ld to %ld step %ld blocks\n", NB, nblk, mode, reps, from, to, step);
    if (mode >= 3 && mode <= 6) {          // block-major storage, 32-bit / 64-bit per-lane addresses: one line, no stride to sweep
        float *fin = (float *)buf, *fout = (float *)(buf + (((size_t)nblk * 512 * Q * 4 + ((size_t)2 << 20) - 1) >> 21 << 21));
        if ((mode == 3 || mode == 5) && (size_t)nblk * 55296 >= ((size_t)1 << 32)) { printf("mode 3 needs < 77672 blocks\n"); return 1; }
        for (int rep = 0; rep < 5; ++rep) {
            auto launch = [&]() {
                if (mode == 3) hipLaunchKernelGGL((k_move_bm<false, false>), grid, block, 0, 0, fin, fout, vin, vout, NB);
                else if (mode == 4) hipLaunchKernelGGL((k_move_bm<true, false>), grid, block, 0, 0, fin, fout, vin, vout, NB);
                else if (mode == 5) hipLaunchKernelGGL((k_move_bm<false, true>), grid, block, 0, 0, fin, fout, vin, vout, NB);
                else hipLaunchKernelGGL((k_move_bm<true, true>), grid, block, 0, 0, fin, fout, vin, vout, NB);
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) { launch(); std::swap(fin, fout); }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s %s %.4f\n", mode <= 4 ? "block-major" : "plane-major", (mode & 1) ? "32-bit" : "64-bit", ms / reps);
        }
        return 0;
    }
    for (long sb = from; sb <= to; sb += step) {
        const size_t S = (size_t)sb * 512;
        size_t gap = gap_arg >= 0 ? (size_t)gap_arg * 2048 : (((size_t)sb * 512 * Q * 4 + ((size_t)2 << 20) - 1) >> 21 << 21);   // as hipMalloc would place the next array
        float *fin = (float *)buf, *fout = (float *)(buf + gap);
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k_move<0>, grid, block, 0, 0, fin, fout, vin, vout, S, NB);
            else if (mode == 1) hipLaunchKernelGGL(k_move<1>, grid, block, 0, 0, fin, fout, vin, vout, S, NB);
            else hipLaunchKernelGGL(k_move<2>, grid, block, 0, 0, fin, fout, vin, vout, S, NB);
        };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) { launch(); std::swap(fin, fout); }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%ld %.4f %.4f\n", sb, sb * 2048.0 / 1048576.0, ms / reps);
        fflush(stdout);
    }
    return 0;
}
