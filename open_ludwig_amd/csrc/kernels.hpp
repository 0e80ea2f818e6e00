// Device kernels of libludwig_hip.so (gfx950 / CDNA4 only).
//
// stream-collide  : reference src/physics_kernels.jl:9-358 (+ src/physics_utils.jl, src/physics_interpolation.jl)
// bouzidi         : reference src/bouzidi_kernel.jl:13-92
//
// Arithmetic contract: every floating-point expression keeps the reference's operand order and is
// compiled with -ffp-contract=off, so results are bit-identical to the scalar CPU restatement for
// finite inputs. Multiplications by the lattice constants 0/+1/-1 are resolved at compile time
// (x*1 = x, x*-1 = -x exactly; a dropped x*0 term only changes the sign of an exact zero).
//
// Mapping: one 64-lane wavefront = one 8x8 z-plane of one 8^3 block (lane = x + 8y); a 256-thread
// workgroup = 4 such planes chosen by the host's work list (by default the same plane of 4 x-consecutive
// blocks, which the library keeps consecutive in memory, see ludwig_hip.hip "Block order"). The block index and z
// are wave-uniform, so the 27 neighbour block ids come through the scalar cache and each population load is one
// coalesced, aligned 256-B access; the +-1 shift in x is a lane shift plus an LDS column between neighbouring waves.
//
// Storage (round 3): the device arrays are BLOCK-major - element (cell, block b, component k) of a K-component field lives
// at ((b * K + k) * 512 + cell): the 27 populations of a block are one contiguous 54-KiB piece. The reference's arrays are
// population-major, [8,8,8,n_blocks,K] (src/blocks.jl:118-150): 27 + 27 concurrent streams n_blocks x 2 KiB apart, and at
// some distances (which depend on nothing but n_blocks) they load MI355X's memory system unevenly - 5-30 % of the step
// (profiles/r03_stride_*). With block-major storage there is no such distance; the ABI translates (ludwig_hip.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "lattice.hpp"
#include "jl_math.h"

namespace lw {

struct SCParams {
    const float *f_in;
    float *f_out;
    float *f_post;            // nullptr unless store_post_collision
    const uint32_t *post_rows; // [n_blocks][2]: bit z*8+y of a block's 64 = the x-row (y, z) holds a cell f_post_collision is read at;
                              // nullptr = every row of a block flagged FLAG_STORE_POST
    const float *vel_in;
    float *vel_out;
    float *rho;
    const uint8_t *obstacle;
    const float *sponge;
    const float *wall_dist;
    const int32_t *meta;      // [n_blocks][NBR_STRIDE]
    const int32_t *items;     // work list, one entry per wave: (block << 3) | z, or -1
    // parent level (coarse -> fine interface), unused on level 1
    const float *pf_new, *pf_old, *prho_new, *prho_old, *pvel_new, *pvel_old;
    float tau, tau_parent, c_wale, nu_bg, u_inlet, inlet_turbulence, temporal_weight;
    int32_t is_level_1, is_symmetric, nx_g, ny_g, nz_g;
    int32_t wall_model, seed, use_temporal, sponge_blend;
    // coarse -> fine interface pass: values of interpolate_with_rescaling for every (cell, population) link of this
    // level that needs one, precomputed densely by k_interface_links: f_iface[(k * n_iface_blocks + gbi) * 512 + cell]
    float *f_iface;
    int32_t n_iface_blocks;
    // 0: this launch leaves `rho` unwritten - nobody reads it before the level's next step unless asked, and then
    // k_stream_collide_xrun<.., RHO_ONLY> recomputes it from the same inputs (ludwig_hip.hip "lazy rho")
    int32_t store_rho;
    // non-null: copy_to_old!'s rho part fused into this launch - every cell saves its old rho here before storing the new one
    // (whole-level launches of a level without ghost blocks only; ludwig_hip.hip "rho_old in the step")
    float *rho_old_save;
};

template <int K, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (K < N) {
        f(std::integral_constant<int, K>{});
        static_for<K + 1, N>(f);
    }
}

// Julia's max() propagates NaN (reference uses Base.max throughout)
__device__ __forceinline__ float jl_max(float a, float b)
{
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}
__device__ __forceinline__ float jl_clamp(float x, float lo, float hi)
{
    return x > hi ? hi : (x < lo ? lo : x);
}
// Base.^(::Float32, ::Float32) and Base.log(::Float32) evaluate in Float64 and round once. The double log2 / exp2 / log are
// jl_math.h's (IEEE +,-,*,/ only), the same source the CPU oracle compiles: identical bits on both sides.
__device__ __forceinline__ float jl_pow(float x, float y) { return lw_powf(x, y); }
__device__ __forceinline__ float jl_log(float x) { return lw_logf(x); }

// c * v for c in {-1,0,1} resolved at compile time; `first` says whether the running sum is empty
template <int C>
__device__ __forceinline__ void acc_signed(float &sum, bool &empty, float v)
{
    if constexpr (C != 0) {
        const float t = C > 0 ? v : -v;
        sum = empty ? t : sum + t;
        empty = false;
    }
}
// cx*a + cy*b + cz*c, left-associated like the reference, zero terms dropped
template <int K>
__device__ __forceinline__ float cdot(float a, float b, float c)
{
    float s = 0.0f;
    bool empty = true;
    acc_signed<CX(K)>(s, empty, a);
    acc_signed<CY(K)>(s, empty, b);
    acc_signed<CZ(K)>(s, empty, c);
    return s;
}

// reference src/physics_utils.jl:17-28 (Int32 products wrap)
__device__ __forceinline__ float gradient_noise(int32_t gx, int32_t gy, int32_t gz, int32_t seed)
{
    uint32_t h = (uint32_t)gx * 374761393u + (uint32_t)gy * 668265263u + (uint32_t)gz * 1274126177u + (uint32_t)seed;
    h = (h ^ (h >> 16)) * 0x85ebca6bu;
    h = (h ^ (h >> 13)) * 0xc2b2ae35u;
    h = h ^ (h >> 16);
    return ((float)(h & 0xFFFFu) / 32768.0f) - 1.0f;
}

// reference src/physics_utils.jl:34-39
__device__ __forceinline__ float calculate_equilibrium(float rho, float ux, float uy, float uz, float w_k,
                                                       float cx, float cy, float cz)
{
    const float cu = cx * ux + cy * uy + cz * uz;
    const float usq = ux * ux + uy * uy + uz * uz;
    return rho * w_k * (1.0f + 3.0f * cu + 4.5f * cu * cu - 1.5f * usq);
}

// ---- coarse -> fine interface value, reference src/physics_interpolation.jl:16-138: the trilinear kernel; the corner
// fetch, the blend in time and the rescaling live in k_interface_sources / k_interface_links below ----
__device__ __forceinline__ float trilin(float v000, float v100, float v010, float v110, float v001, float v101,
                                        float v011, float v111, float wx, float wy, float wz)
{
    const float c00 = v000 * (1.0f - wx) + v100 * wx;
    const float c01 = v001 * (1.0f - wx) + v101 * wx;
    const float c10 = v010 * (1.0f - wx) + v110 * wx;
    const float c11 = v011 * (1.0f - wx) + v111 * wx;
    const float c0 = c00 * (1.0f - wy) + c10 * wy;
    const float c1 = c01 * (1.0f - wy) + c11 * wy;
    return c0 * (1.0f - wz) + c1 * wz;
}

// ---- wall-model force, reference src/physics_kernels.jl:206-236 ----
#ifndef LW_WALL_INLINE
#define LW_WALL_INLINE __noinline__
#endif
// Returns the magnitude of the force, or -1 where the reference leaves F = 0; the caller turns it into F = -mag u / |u|.
// (One float in registers each way: with F returned through three references the call went through the stack - 16 B of
// scratch per lane - and the kernel lost a wave per SIMD.)
__device__ LW_WALL_INLINE float wall_model_force_mag(float dist_wall, float tau_molecular, float rho, float u_mag)
{
    float force_mag = -1.0f;
    if (dist_wall > 0.0f && dist_wall < 10.0f) {
        const float nu_visc = (tau_molecular - 0.5f) / 3.0f;
        if (u_mag > 1.0e-6f && nu_visc > 1.0e-10f) {
            float u_tau = u_mag * jl_pow(nu_visc / (dist_wall * u_mag + 1.0e-10f), 1.0f / 7.0f) *
                          jl_pow(2.0f * 8.3f, -1.0f / 7.0f);
            u_tau = jl_max(u_tau, 1.0e-6f);
            const float y_p = u_tau * dist_wall / nu_visc;
            if (y_p > 11.81f) {
                const float u_plus_law = (1.0f / KAPPA) * jl_log(y_p) + 5.2f;
                if (u_plus_law > 0.1f) {
                    u_tau = u_tau * ((u_mag / u_tau) / u_plus_law);
                    u_tau = jl_max(u_tau, 1.0e-6f);
                }
            }
            const float tau_wall = rho * u_tau * u_tau;
            const float tau_res = rho * nu_visc * (u_mag / dist_wall);
            if (tau_wall > tau_res) force_mag = (tau_wall - tau_res) / dist_wall;
        }
    }
    return force_mag;
}

// ---- addressing helpers -------------------------------------------------------------------------------------
// Block-major storage: byte offset of (cell, block b, component k) of a K-component float field = b * K * 2048 + k * 2048 + cell * 4.
// The wave's own block is wave-uniform, so its accesses are `SGPR base + 32-bit per-lane byte offset`
// (global_load/store_dword v, v_off, s[base:base+1]); only the populations pulled across a y face choose between two blocks per
// lane. NARROW (levels below 77 672 blocks = 4 GiB of f): that choice is a 32-bit byte offset from the array base too;
// WIDE: a 64-bit per-lane address (v_mad_u64_u32). Same loads, same values.
constexpr uint32_t F_BLOCK_BYTES = Q * CELLS * 4, V_BLOCK_BYTES = 3 * CELLS * 4, S_BLOCK_BYTES = CELLS * 4;
constexpr uint32_t COMP_BYTES = CELLS * 4;

template <bool WIDE>
__device__ __forceinline__ const float *cell_ptr(const float *base, int blk, uint32_t block_bytes, uint32_t inner)
{
    if constexpr (WIDE) return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (uint64_t)(uint32_t)blk * block_bytes + inner);
    else return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (uint32_t)((uint32_t)blk * block_bytes + inner));
}
template <bool WIDE>
__device__ __forceinline__ float ld_f32(const float *base, int blk, uint32_t block_bytes, uint32_t inner)
{
#ifdef LW_NT_LOADS
    return __builtin_nontemporal_load(cell_ptr<WIDE>(base, blk, block_bytes, inner));
#else
    return *cell_ptr<WIDE>(base, blk, block_bytes, inner);
#endif
}
// load of a line that has exactly ONE reader in the launch (non-temporal: do not keep it in L2 / Infinity Cache)
template <bool WIDE>
__device__ __forceinline__ float ld_f32_once(const float *base, int blk, uint32_t block_bytes, uint32_t inner)
{
#ifdef LW_NT_SINGLE_READER
    return __builtin_nontemporal_load(cell_ptr<WIDE>(base, blk, block_bytes, inner));
#else
    return ld_f32<WIDE>(base, blk, block_bytes, inner);
#endif
}
// the wave's own block (wave-uniform id): uniform 64-bit base, 32-bit per-lane offset, at any level size
__device__ __forceinline__ float ld_own(const float *base, int blk, uint32_t block_bytes, uint32_t inner)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (uint64_t)(uint32_t)blk * block_bytes + inner);
}
// stores go to the wave's own block only: uniform 64-bit base, 32-bit per-lane offset, at any level size
__device__ __forceinline__ void st_f32(float *base, int blk, uint32_t block_bytes, uint32_t inner, float v)
{
    float *q = reinterpret_cast<float *>(reinterpret_cast<char *>(base) + (uint64_t)(uint32_t)blk * block_bytes + inner);
#ifdef LW_NT_STORES
    __builtin_nontemporal_store(v, q);
#else
    *q = v;
#endif
}
// the wave's own cell: block (wave-uniform), plane and lane
struct Own {
    int b;            // block
    uint32_t cell4;   // (x + 8 y + 64 z) * 4
};

// The 27 neighbour block ids of the wave's block, held in SGPRs, regrouped by source z-layer:
// id[g][j] for g = 0 (cz=+1: source plane z-1), 1 (cz=0), 2 (cz=-1: source plane z+1), j = (ox+1) + 3(oy+1).
struct NeighbourIds {
    int id[3][9];
    int zs[3];      // source plane index inside the source block
};

__device__ __forceinline__ NeighbourIds load_neighbour_ids(const int32_t *__restrict__ meta, int z)
{
    int nb[27];
#ifdef LW_DIAG_ARITH_NBR   // timing-only: neighbour ids of the 32^3 periodic box computed, not loaded (no dependent scalar load)
    {
        const int b0 = (int)(meta - (const int32_t *)nullptr) / NBR_STRIDE;   // caller passes meta = nullptr + b * stride
        const int bz = b0 & 31, by = (b0 >> 5) & 31, bx = b0 >> 10;
#pragma unroll
        for (int d = 0; d < 27; ++d) {
            const int ox = d % 3 - 1, oy = (d / 3) % 3 - 1, oz = d / 9 - 1;
            nb[d] = ((((bx + ox) & 31) << 5) | ((by + oy) & 31)) << 5 | ((bz + oz) & 31);
        }
    }
#else
#pragma unroll
    for (int d = 0; d < 27; ++d) nb[d] = meta[d];            // wave-uniform -> scalar loads, one burst
#endif
    NeighbourIds n;
    const bool z_lo = z == 0, z_hi = z == 7;                 // wave-uniform
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int lo = nb[j], mid = nb[9 + j], hi = nb[18 + j];   // values, not lvalues (see source_block)
        n.id[0][j] = z_lo ? lo : mid;
        n.id[1][j] = mid;
        n.id[2][j] = z_hi ? hi : mid;
    }
    n.zs[0] = (z - 1) & 7; n.zs[1] = z; n.zs[2] = (z + 1) & 7;
    return n;
}

// per-lane position inside the 8x8 plane and the face flags the pull needs
struct LanePos {
    int x, y;
    bool x0, x7, y0, y7;
};

// source block id (per lane) of population k: own block or the x / y / xy neighbour in the layer cz selects
template <int K>
__device__ __forceinline__ int source_block(const NeighbourIds &n, const LanePos &l)
{
    constexpr int cx = CX(K), cy = CY(K), g = 1 - CZ(K);
    const bool xo = cx == 1 ? l.x0 : (cx == -1 ? l.x7 : false);
    const bool yo = cy == 1 ? l.y0 : (cy == -1 ? l.y7 : false);
    // copy to values first: a ?: between array ELEMENTS is an lvalue select, which keeps the table in scratch memory
    const int c00 = n.id[g][4], cX = n.id[g][4 - cx], cY = n.id[g][4 - 3 * cy], cXY = n.id[g][4 - cx - 3 * cy];
    int sel = c00;
    if constexpr (cx != 0 && cy != 0) sel = xo ? (yo ? cXY : cX) : (yo ? cY : c00);
    else if constexpr (cx != 0) sel = xo ? cX : c00;
    else if constexpr (cy != 0) sel = yo ? cY : c00;
    return sel;
}
// Everything after the loads: moments, obstacle bounce, sponge, wall model, WALE, regularized collision, stores.
// reference src/physics_kernels.jl:144-354. fs = the 27 pulled populations, u?_? = previous-step velocity of the six
// face neighbours. Shared by the per-wave kernel and the x-run kernel.
template <bool POST, bool WALL>
__device__ __forceinline__ void finish_cell(const SCParams &p, const int flags, const Own own, float (&fs)[Q],
                                            const float ux_E, const float uy_E, const float uz_E, const float ux_W, const float uy_W, const float uz_W,
                                            const float ux_N, const float uy_N, const float uz_N, const float ux_S, const float uy_S, const float uz_S,
                                            const float ux_T, const float uy_T, const float uz_T, const float ux_B, const float uy_B, const float uz_B)
{
#ifdef LW_DIAG_NO_MATH   // timing-only diagnostic build: same loads and stores, no collision arithmetic; results are wrong
    {
        float acc = ux_E + uy_E + uz_E + ux_W + uy_W + uz_W + ux_N + uy_N + uz_N + ux_S + uy_S + uz_S + ux_T + uy_T + uz_T + ux_B + uy_B + uz_B;
        st_f32(p.vel_out, own.b, V_BLOCK_BYTES, own.cell4, acc);
        st_f32(p.vel_out, own.b, V_BLOCK_BYTES, COMP_BYTES + own.cell4, fs[1]);
        st_f32(p.vel_out, own.b, V_BLOCK_BYTES, 2 * COMP_BYTES + own.cell4, fs[2]);
        st_f32(p.rho, own.b, S_BLOCK_BYTES, own.cell4, fs[0]);
        static_for<0, Q>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            st_f32(p.f_out, own.b, F_BLOCK_BYTES, k * COMP_BYTES + own.cell4, fs[k]);
        });
        return;
    }
#endif
    // moments in the reference's order: rho += f_k; j += f_k * c_k for k = 1..27
    float rho = 0.0f, jx = 0.0f, jy = 0.0f, jz = 0.0f;
    static_for<0, Q>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int cx = CX(k), cy = CY(k), cz = CZ(k);
        const float val = fs[k];
        rho += val;
        if constexpr (cx == 1) jx += val; else if constexpr (cx == -1) jx -= val;
        if constexpr (cy == 1) jy += val; else if constexpr (cy == -1) jy -= val;
        if constexpr (cz == 1) jz += val; else if constexpr (cz == -1) jz -= val;
    });

    // the reference stores f_post_collision for every cell of a Bouzidi level (src/physics_kernels.jl:350-352); its only
    // reader is the Bouzidi kernel, at boundary cells and their link neighbours, so blocks that neither hold nor touch a
    // boundary cell skip the dead store (wave-uniform flag; 108 of 357 B per cell update)
    // Round 3: inside such a block only the x-rows (8 cells = one 32-B sector per population) that hold a Bouzidi cell or the
    // cell one step behind a link are stored (`post_rows`, one bit per row; the word is wave-uniform: block and plane are).
    bool store_post = false;
    if constexpr (POST) {
        if (flags & FLAG_STORE_POST) {
            store_post = true;
            if (p.post_rows) {
                const int zu = __builtin_amdgcn_readfirstlane((int)(own.cell4 >> 8));
                const uint32_t w = p.post_rows[(size_t)own.b * 2 + (zu >> 2)];
                store_post = ((w >> ((zu & 3) * 8 + ((own.cell4 >> 5) & 7))) & 1u) != 0;
            }
        }
    }
    // ---- obstacle cell: full-way bounce-back of the pulled set, reference :154-166 ----
    bool is_obs = false;
    if (flags & FLAG_HAS_OBSTACLE) is_obs = p.obstacle[(size_t)own.b * CELLS + (own.cell4 >> 2)] != 0;
    if (is_obs) {
        st_f32(p.vel_out, own.b, V_BLOCK_BYTES, own.cell4, 0.0f);
        st_f32(p.vel_out, own.b, V_BLOCK_BYTES, COMP_BYTES + own.cell4, 0.0f);
        st_f32(p.vel_out, own.b, V_BLOCK_BYTES, 2 * COMP_BYTES + own.cell4, 0.0f);
        if (p.rho_old_save) st_f32(p.rho_old_save, own.b, S_BLOCK_BYTES, own.cell4, ld_own(p.rho, own.b, S_BLOCK_BYTES, own.cell4));
        if (p.store_rho) st_f32(p.rho, own.b, S_BLOCK_BYTES, own.cell4, 1.0f);
        static_for<0, Q>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const float f_coll = fs[OPP(k)];
            st_f32(p.f_out, own.b, F_BLOCK_BYTES, k * COMP_BYTES + own.cell4, f_coll);
            if constexpr (POST) { if (store_post) st_f32(p.f_post, own.b, F_BLOCK_BYTES, k * COMP_BYTES + own.cell4, f_coll); }
        });
        return;
    }

    // ---- macroscopic moments, reference :172-176 ----
    rho = jl_max(rho, 0.01f);
    const float inv_rho = 1.0f / rho;
    float ux = jx * inv_rho, uy = jy * inv_rho, uz = jz * inv_rho;

    // ---- sponge, reference :181-199 ----
    if (flags & FLAG_HAS_SPONGE) {
        const float sp = ld_own(p.sponge, own.b, S_BLOCK_BYTES, own.cell4);
        if (sp > 0.0f) {
            const float rho_target = 1.0f, ux_target = p.u_inlet;
            rho = rho * (1.0f - sp) + rho_target * sp;
            ux = ux * (1.0f - sp) + ux_target * sp;
            uy = uy * (1.0f - sp);
            uz = uz * (1.0f - sp);
            if (p.sponge_blend == 1) {
                static_for<0, Q>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    const float feq_target = calculate_equilibrium(rho_target, ux_target, 0.0f, 0.0f, WEIGHT(k),
                                                                   (float)CX(k), (float)CY(k), (float)CZ(k));
                    fs[k] = fs[k] * (1.0f - sp) + feq_target * sp;
                });
            }
        }
    }

    // ---- wall-model force, reference :202-241 ----
    float Fx = 0.0f, Fy = 0.0f, Fz = 0.0f;
    float ux_eq = ux, uy_eq = uy, uz_eq = uz;          // u + 0.5 F / rho with F = 0
    if constexpr (WALL) {
        if (flags & FLAG_HAS_NEAR_WALL) {
            const float u_mag = sqrtf(ux * ux + uy * uy + uz * uz);
            const float force_mag = wall_model_force_mag(ld_own(p.wall_dist, own.b, S_BLOCK_BYTES, own.cell4), p.tau, rho, u_mag);
            if (force_mag >= 0.0f) {
                Fx = -force_mag * ux / u_mag;
                Fy = -force_mag * uy / u_mag;
                Fz = -force_mag * uz / u_mag;
            }
        }
        ux_eq = ux + 0.5f * Fx * inv_rho;
        uy_eq = uy + 0.5f * Fy * inv_rho;
        uz_eq = uz + 0.5f * Fz * inv_rho;
    }
    const float usq_eq = ux_eq * ux_eq + uy_eq * uy_eq + uz_eq * uz_eq;

    st_f32(p.vel_out, own.b, V_BLOCK_BYTES, own.cell4, ux);
    st_f32(p.vel_out, own.b, V_BLOCK_BYTES, COMP_BYTES + own.cell4, uy);
    st_f32(p.vel_out, own.b, V_BLOCK_BYTES, 2 * COMP_BYTES + own.cell4, uz);
    if (p.rho_old_save) st_f32(p.rho_old_save, own.b, S_BLOCK_BYTES, own.cell4, ld_own(p.rho, own.b, S_BLOCK_BYTES, own.cell4));
    if (p.store_rho) st_f32(p.rho, own.b, S_BLOCK_BYTES, own.cell4, rho);

    // ---- WALE eddy viscosity from the previous step's velocity, reference :251-300 ----
    const float g11 = 0.5f * (ux_E - ux_W), g12 = 0.5f * (ux_N - ux_S), g13 = 0.5f * (ux_T - ux_B);
    const float g21 = 0.5f * (uy_E - uy_W), g22 = 0.5f * (uy_N - uy_S), g23 = 0.5f * (uy_T - uy_B);
    const float g31 = 0.5f * (uz_E - uz_W), g32 = 0.5f * (uz_N - uz_S), g33 = 0.5f * (uz_T - uz_B);

    const float gsq11 = g11 * g11 + g12 * g21 + g13 * g31;
    const float gsq12 = g11 * g12 + g12 * g22 + g13 * g32;
    const float gsq13 = g11 * g13 + g12 * g23 + g13 * g33;
    const float gsq21 = g21 * g11 + g22 * g21 + g23 * g31;
    const float gsq22 = g21 * g12 + g22 * g22 + g23 * g32;
    const float gsq23 = g21 * g13 + g22 * g23 + g23 * g33;
    const float gsq31 = g31 * g11 + g32 * g21 + g33 * g31;
    const float gsq32 = g31 * g12 + g32 * g22 + g33 * g32;
    const float gsq33 = g31 * g13 + g32 * g23 + g33 * g33;

    const float tr_gsq = gsq11 + gsq22 + gsq33;
    const float tr_term = tr_gsq / 3.0f;
    const float Sd11 = gsq11 - tr_term, Sd22 = gsq22 - tr_term, Sd33 = gsq33 - tr_term;
    const float Sd12 = 0.5f * (gsq12 + gsq21), Sd13 = 0.5f * (gsq13 + gsq31), Sd23 = 0.5f * (gsq23 + gsq32);
    const float S12 = 0.5f * (g12 + g21), S13 = 0.5f * (g13 + g31), S23 = 0.5f * (g23 + g32);
    const float OP1 = Sd11 * Sd11 + Sd22 * Sd22 + Sd33 * Sd33 + 2.0f * (Sd12 * Sd12 + Sd13 * Sd13 + Sd23 * Sd23);
    const float OP2 = g11 * g11 + g22 * g22 + g33 * g33 + 2.0f * (S12 * S12 + S13 * S13 + S23 * S23);

    float nu_eddy = 0.0f;
    if (OP1 > 1.0e-12f) {
        const float OP1_32 = OP1 * sqrtf(OP1);
        const float OP2_52 = OP2 * OP2 * sqrtf(jl_max(OP2, 1.0e-12f));
        const float denom = OP2_52 + OP1 * sqrtf(sqrtf(jl_max(OP1, 1.0e-12f)));
        if (denom > 1.0e-12f) nu_eddy = (p.c_wale * p.c_wale) * OP1_32 / denom;
    }
    nu_eddy = jl_max(nu_eddy, p.nu_bg);
    const float tau_turb = p.tau + nu_eddy * 3.0f;
    const float omega = 1.0f / jl_max(tau_turb, 0.500001f);

    // ---- non-equilibrium stress, reference :305-322 ----
    float Pi_xx = 0.0f, Pi_yy = 0.0f, Pi_zz = 0.0f, Pi_xy = 0.0f, Pi_yz = 0.0f, Pi_zx = 0.0f;
    static_for<0, Q>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int cx = CX(k), cy = CY(k), cz = CZ(k);
        const float cu = cdot<k>(ux_eq, uy_eq, uz_eq);
        const float feq = rho * WEIGHT(k) * (1.0f + 3.0f * cu + 4.5f * cu * cu - 1.5f * usq_eq);
        const float f_neq = fs[k] - feq;
        if constexpr (cx != 0) Pi_xx += f_neq;
        if constexpr (cy != 0) Pi_yy += f_neq;
        if constexpr (cz != 0) Pi_zz += f_neq;
        if constexpr (cx * cy == 1) Pi_xy += f_neq; else if constexpr (cx * cy == -1) Pi_xy -= f_neq;
        if constexpr (cy * cz == 1) Pi_yz += f_neq; else if constexpr (cy * cz == -1) Pi_yz -= f_neq;
        if constexpr (cz * cx == 1) Pi_zx += f_neq; else if constexpr (cz * cx == -1) Pi_zx -= f_neq;
    });

    // ---- regularized collision + write, reference :324-354 ----
    const float one_m_omega = 1.0f - omega;
    const float one_m_half_omega = 1.0f - 0.5f * omega;
    static_for<0, Q>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int cx = CX(k), cy = CY(k), cz = CZ(k);
        constexpr float w_k = WEIGHT(k);
        constexpr float cx_f = (float)cx, cy_f = (float)cy, cz_f = (float)cz;
        const float cu = cdot<k>(ux_eq, uy_eq, uz_eq);
        const float feq = rho * w_k * (1.0f + 3.0f * cu + 4.5f * cu * cu - 1.5f * usq_eq);
        constexpr float Q_xx = cx_f * cx_f - CS2_PHYSICS, Q_yy = cy_f * cy_f - CS2_PHYSICS, Q_zz = cz_f * cz_f - CS2_PHYSICS;
        // Pi_xy*cx*cy + Pi_yz*cy*cz + Pi_zx*cz*cx, left-associated, zero terms dropped
        float od = 0.0f;
        bool od_empty = true;
        acc_signed<cx * cy>(od, od_empty, Pi_xy);
        acc_signed<cy * cz>(od, od_empty, Pi_yz);
        acc_signed<cz * cx>(od, od_empty, Pi_zx);
        float inner = Pi_xx * Q_xx + Pi_yy * Q_yy + Pi_zz * Q_zz;
        if constexpr (cx * cy != 0 || cy * cz != 0 || cz * cx != 0) inner = inner + 2.0f * od;   // else + 2*0: exact no-op
        const float f_neq_reg = (w_k * 4.5f) * inner;
        float f_coll = feq + one_m_omega * f_neq_reg;
        if constexpr (WALL) {
            const float force_term = (w_k * 3.0f) * ((cx_f - ux + 3.0f * cu * cx_f) * Fx + (cy_f - uy + 3.0f * cu * cy_f) * Fy +
                                                     (cz_f - uz + 3.0f * cu * cz_f) * Fz);
            f_coll = f_coll + one_m_half_omega * force_term;
        }
        if constexpr (POST) { if (store_post) st_f32(p.f_post, own.b, F_BLOCK_BYTES, k * COMP_BYTES + own.cell4, f_coll); }
        st_f32(p.f_out, own.b, F_BLOCK_BYTES, k * COMP_BYTES + own.cell4, f_coll);
    });
}

// The density finish_cell would have stored for this cell, and nothing else: the same sums in the same order (reference
// src/physics_kernels.jl:144-148, :154-158 obstacle, :172 clamp, :181-186 sponge). Used to produce `rho` on demand after a
// step that skipped the store.
__device__ __forceinline__ void finish_rho_only(const SCParams &p, const int flags, const Own own, const float (&fs)[Q])
{
    float rho = 0.0f;
    static_for<0, Q>([&](auto kc) { rho += fs[decltype(kc)::value]; });
    bool is_obs = false;
    if (flags & FLAG_HAS_OBSTACLE) is_obs = p.obstacle[(size_t)own.b * CELLS + (own.cell4 >> 2)] != 0;
    if (is_obs) { st_f32(p.rho, own.b, S_BLOCK_BYTES, own.cell4, 1.0f); return; }
    rho = jl_max(rho, 0.01f);
    if (flags & FLAG_HAS_SPONGE) {
        const float sp = ld_own(p.sponge, own.b, S_BLOCK_BYTES, own.cell4);
        if (sp > 0.0f) rho = rho * (1.0f - sp) + 1.0f * sp;
    }
    st_f32(p.rho, own.b, S_BLOCK_BYTES, own.cell4, rho);
}

// Lanes whose source block is missing: the domain-edge chain of reference src/physics_kernels.jl:88-140 (1-based global
// coords) decides what replaces the pulled value - inlet / outlet equilibrium, mirror of the own cell, the coarse->fine
// interface value, or the weight. Runs after all loads were issued (the lane read a valid dummy address meanwhile).
__device__ __forceinline__ void patch_missing_sources(const SCParams &p, const int32_t *__restrict__ meta, const NeighbourIds &nbr,
                                                      const LanePos &l, const int z, const Own own, float (&fs)[Q])
{
    const int gx = (meta[NBR_BX] - 1) * BS + l.x + 1, gy = (meta[NBR_BY] - 1) * BS + l.y + 1, gz = (meta[NBR_BZ] - 1) * BS + z + 1;
    const float *iface_own = p.is_level_1 == 0 ? p.f_iface + (size_t)meta[NBR_GBI] * CELLS + (own.cell4 >> 2) : nullptr;
    static_for<0, Q>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int cx = CX(k), cy = CY(k), cz = CZ(k);
        if constexpr (k != 13) {
            if (source_block<k>(nbr, l) < 0) {
                const int src_gx = gx - cx, src_gy = gy - cy, src_gz = gz - cz;
                const bool is_inlet = src_gx < 1, is_outlet = src_gx > p.nx_g;
                const bool is_y_min = src_gy < 1, is_y_max = src_gy > p.ny_g;
                const bool is_z_min = src_gz < 1, is_z_max = src_gz > p.nz_g;
                float val;
                if (is_inlet) {
                    const float noise = p.inlet_turbulence > 0.0f
                                            ? gradient_noise(gy, gz, p.seed, 1234) * p.inlet_turbulence * p.u_inlet
                                            : 0.0f;
                    const float u_inst = p.u_inlet + noise;
                    const float cu_in = (float)cx * u_inst;
                    val = WEIGHT(k) * (1.0f + 3.0f * cu_in + 4.5f * cu_in * cu_in - 1.5f * u_inst * u_inst);
                } else if (is_outlet) {
                    const float cu_out = (float)cx * p.u_inlet;
                    val = WEIGHT(k) * (1.0f + 3.0f * cu_out + 4.5f * cu_out * cu_out - 1.5f * p.u_inlet * p.u_inlet);
                } else if (is_y_min && p.is_symmetric == 1) {
                    val = ld_own(p.f_in, own.b, F_BLOCK_BYTES, MIRROR_Y(k) * COMP_BYTES + own.cell4);
                } else if (is_y_min || is_y_max) {
                    val = ld_own(p.f_in, own.b, F_BLOCK_BYTES, MIRROR_Y(k) * COMP_BYTES + own.cell4);
                } else if (is_z_min || is_z_max) {
                    val = ld_own(p.f_in, own.b, F_BLOCK_BYTES, MIRROR_Z(k) * COMP_BYTES + own.cell4);
                } else if (p.is_level_1 == 0) {
                    // value computed by k_interface_links for exactly this (cell, k) link (same function, same inputs)
                    val = iface_own[(size_t)k * p.n_iface_blocks * CELLS];
                } else {
                    val = WEIGHT(k);
                }
                fs[k] = val;
            }
        }
    });
}

// ---- the stream-collide kernel --------------------------------------------------------------------------------
// GENERAL: blocks with a missing neighbour (domain edge / refinement interface); POST: also store f_post_collision
// (level has Bouzidi cells); WALL: wall model active; RHO_ONLY: reproduce an elided rho; WIDE: 64-bit per-lane addresses.
// Workgroup = the SAME z-plane of NW x-consecutive blocks (host guarantees: items 0..NW-1 of the group are valid,
// share z, item i+1 is the +x neighbour of item i, all blocks have their 26 neighbours). Every global access is an
// aligned 256-B plane row set (x unshifted; the y / z shift only changes the row / plane, i.e. stays 16-B aligned);
// the +-1 shift in x is a DPP lane shift, and the face column each block needs from its x neighbour is handed over
// between neighbouring waves through 24 x 8 floats of LDS. Only the two outer faces of the run are still read as
// strided columns from global memory. Measured motivation: DESIGN.md section 3.1 (the x-face column and the 4-byte
// misalignment are what separate the pull from an aligned copy on MI355X).
__device__ __forceinline__ float dpp_from_lower_lane(float v)   // lane i <- lane i-1 inside 16-lane rows (row_shr:1)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x111, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_from_upper_lane(float v)   // lane i <- lane i+1 inside 16-lane rows (row_shl:1)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x101, 0xF, 0xF, false));
}
// slot of population k in the exchange buffer: cx=+1 populations publish their x=7 column in slots 0..8,
// cx=-1 populations their x=0 column in slots 9..17 (k = (cx+1) + 3 j, j = 0..8)
__host__ __device__ constexpr int XSLOT(int k) { return CX(k) == 1 ? k / 3 : 9 + k / 3; }

// work item of the x-run kernel: (block << 3) | z in the low bits, plus which of its lateral faces are served by the
// neighbouring wave of the workgroup (LDS) rather than by a strided column read from global memory
constexpr int ITEM_LINK_W = 1 << 30;   // wave - 1 of this workgroup holds the -x neighbour block, same plane
constexpr int ITEM_LINK_E = 1 << 29;   // wave + 1 holds the +x neighbour block, same plane
constexpr int ITEM_ID_MASK = (1 << 29) - 1;

#ifndef LW_MIN_WAVES
#define LW_MIN_WAVES 5      // waves per SIMD the register allocator must leave room for (96 VGPRs)
#endif
template <int NW, bool GENERAL, bool POST, bool WALL, bool RHO_ONLY = false, bool WIDE = false>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu((WIDE && GENERAL) ? LW_MIN_WAVES - 1 : LW_MIN_WAVES))) void k_stream_collide_xrun(const SCParams p)
{
    __shared__ float xch[NW][24][8];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int raw = p.items[blockIdx.x * NW + wave];
    const bool active = raw >= 0;                             // -1 = idle wave (padding); it still meets the barrier
    const int item = active ? (raw & ITEM_ID_MASK) : 0;
    const int b = item >> 3;
    const int z = item & 7;
    const int lane = threadIdx.x & 63;
    LanePos l;
    l.x = lane & 7; l.y = lane >> 3;
    l.x0 = l.x == 0; l.x7 = l.x == 7; l.y0 = l.y == 0; l.y7 = l.y == 7;
#ifdef LW_DIAG_ARITH_NBR
    const int32_t *__restrict__ meta = (const int32_t *)nullptr + (size_t)b * NBR_STRIDE;
    const int flags = FLAG_ALL_NEIGHBOURS;
#else
    const int32_t *__restrict__ meta = p.meta + (size_t)b * NBR_STRIDE;
    const int flags = meta[NBR_FLAGS];
#endif
    const NeighbourIds nbr = load_neighbour_ids(meta, z);
    const Own own{b, (uint32_t)((l.x + 8 * l.y + 64 * z) * 4)};
    // wave-uniform: a run may be shorter than the workgroup (several short runs, or single blocks, share one)
    const bool first = !active || (raw & ITEM_LINK_W) == 0, last = !active || (raw & ITEM_LINK_E) == 0;
    float fs[Q];
    float halo[Q];                                            // outer-face column (run ends only)
    float uc[3], uT[3], uB[3], uy_edge[3], ux_edge_lo[3], ux_edge_hi[3];
    if (active) {
    // ---- aligned loads: value of population k at (x, y - cy, z - cz) ----
    static_for<0, Q>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
#ifdef LW_DIAG_NO_YSHIFT   // timing-only
        constexpr int cx = CX(k), cy = 0, g = 1 - CZ(k);
#else
        constexpr int cx = CX(k), cy = CY(k), g = 1 - CZ(k);
#endif
        const bool yo = cy == 1 ? l.y0 : (cy == -1 ? l.y7 : false);
        const int c00 = nbr.id[g][4], cY = nbr.id[g][4 - 3 * cy];
        int sel = cy != 0 ? (yo ? cY : c00) : c00;
        if constexpr (GENERAL) sel = sel >= 0 ? sel : b;      // missing block: in-bounds dummy, patched after the exchange
        const uint32_t rowz = (uint32_t)(k * COMP_BYTES) + (uint32_t)((8 * ((l.y - cy) & 7)) * 4) + (uint32_t)(nbr.zs[g] * 256);
        // populations with cy = 0: the block is wave-uniform (SGPR base) and the two lines of this plane are read by this wave
        // only (the x neighbours get their column through LDS, nobody shifts rows) -> single reader
        if constexpr (cy == 0) fs[k] = ld_f32_once<true>(p.f_in, sel, F_BLOCK_BYTES, rowz + (uint32_t)(l.x * 4));
        else fs[k] = ld_f32<WIDE>(p.f_in, sel, F_BLOCK_BYTES, rowz + (uint32_t)(l.x * 4));
        halo[k] = 0.0f;
        if constexpr (cx != 0) {
#ifdef LW_DIAG_NO_OUTER   // timing-only
            if (false) {
#else
            if (cx == 1 ? first : last) {                     // the run's outer face: strided column of the x / xy neighbour
#endif
                const int cX = nbr.id[g][4 - cx], cXY = nbr.id[g][4 - cx - 3 * cy];
                int selx = cy != 0 ? (yo ? cXY : cX) : cX;
                if constexpr (GENERAL) selx = selx >= 0 ? selx : b;
                if (cx == 1 ? l.x0 : l.x7)
                    halo[k] = ld_f32<(cy == 0) || WIDE>(p.f_in, selx, F_BLOCK_BYTES, rowz + (uint32_t)((cx == 1 ? 7 : 0) * 4));
            }
        }
    });
    // previous-step velocity: centre plane, planes z+-1 (aligned), y-face rows and outer x-face columns (masked)
    if constexpr (!RHO_ONLY) {
        int bT = z == 7 ? nbr.id[2][4] : b, bB = z == 0 ? nbr.id[0][4] : b;      // wave-uniform
        const uint32_t xy = (uint32_t)((l.x + 8 * l.y) * 4);
        uint32_t inT = xy + (uint32_t)(((z + 1) & 7) * 256), inB = xy + (uint32_t)(((z - 1) & 7) * 256);
        // lanes y==0 fetch row 7 of the -y neighbour, lanes y==7 row 0 of the +y neighbour (one masked load per component)
        int by_ = l.y0 ? nbr.id[1][4 - 3] : nbr.id[1][4 + 3];
        uint32_t inY = (uint32_t)((l.x + 8 * (l.y0 ? 7 : 0) + 64 * z) * 4);
        int bXlo = nbr.id[1][4 - 1], bXhi = nbr.id[1][4 + 1];                    // wave-uniform
        uint32_t inXlo = (uint32_t)((7 + 8 * l.y + 64 * z) * 4), inXhi = (uint32_t)((0 + 8 * l.y + 64 * z) * 4);
        if constexpr (GENERAL) {      // missing neighbour block -> the cell's own velocity (reference src/physics_utils.jl:45-70)
            if (bT < 0) { bT = b; inT = own.cell4; }
            if (bB < 0) { bB = b; inB = own.cell4; }
            if (by_ < 0) { by_ = b; inY = own.cell4; }
            if (bXlo < 0) { bXlo = b; inXlo = own.cell4; }
            if (bXhi < 0) { bXhi = b; inXhi = own.cell4; }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t cb = (uint32_t)c * COMP_BYTES;
#if defined(LW_DIAG_NO_VEL)   // timing-only
            uc[c] = __int_as_float(own.cell4 + c); uT[c] = uc[c]; uB[c] = uc[c]; uy_edge[c] = 0.0f; ux_edge_lo[c] = 0.0f; ux_edge_hi[c] = 0.0f;
            (void)cb; (void)inT; (void)inB; (void)inY; (void)inXlo; (void)inXhi;
#elif defined(LW_DIAG_NO_VEL_TB)
            uc[c] = ld_own(p.vel_in, b, V_BLOCK_BYTES, cb + own.cell4); uT[c] = uc[c]; uB[c] = uc[c];
            uy_edge[c] = 0.0f; ux_edge_lo[c] = 0.0f; ux_edge_hi[c] = 0.0f;
            if (l.y0 || l.y7) uy_edge[c] = ld_f32<WIDE>(p.vel_in, by_, V_BLOCK_BYTES, cb + inY);
            if (first) { if (l.x0) ux_edge_lo[c] = ld_own(p.vel_in, bXlo, V_BLOCK_BYTES, cb + inXlo); }
            if (last) { if (l.x7) ux_edge_hi[c] = ld_own(p.vel_in, bXhi, V_BLOCK_BYTES, cb + inXhi); }
#else
            uc[c] = ld_own(p.vel_in, b, V_BLOCK_BYTES, cb + own.cell4);
            uT[c] = ld_own(p.vel_in, bT, V_BLOCK_BYTES, cb + inT);
            uB[c] = ld_own(p.vel_in, bB, V_BLOCK_BYTES, cb + inB);
            uy_edge[c] = 0.0f; ux_edge_lo[c] = 0.0f; ux_edge_hi[c] = 0.0f;
            if (l.y0 || l.y7) uy_edge[c] = ld_f32<WIDE>(p.vel_in, by_, V_BLOCK_BYTES, cb + inY);
            if (first) { if (l.x0) ux_edge_lo[c] = ld_own(p.vel_in, bXlo, V_BLOCK_BYTES, cb + inXlo); }
            if (last) { if (l.x7) ux_edge_hi[c] = ld_own(p.vel_in, bXhi, V_BLOCK_BYTES, cb + inXhi); }
#endif
        }
    }

    // ---- publish the face columns the neighbouring waves need ----
#ifndef LW_DIAG_NO_XCH
    static_for<0, Q>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int cx = CX(k);
        if constexpr (cx == 1) { if (l.x7) xch[wave][XSLOT(k)][l.y] = fs[k]; }
        if constexpr (cx == -1) { if (l.x0) xch[wave][XSLOT(k)][l.y] = fs[k]; }
    });
    if constexpr (!RHO_ONLY) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (l.x7) xch[wave][18 + c][l.y] = uc[c];
            if (l.x0) xch[wave][21 + c][l.y] = uc[c];
        }
    }
#endif
    }   // if (active)
#ifndef LW_DIAG_NO_XCH
    __syncthreads();
#endif
    if (!active) return;
#ifdef LW_DIAG_NO_XCH   // timing-only: no LDS traffic at all
#define LW_XCH(w, s, y) 0.0f
#else
#define LW_XCH(w, s, y) xch[w][s][y]
#endif
    const int wlo = first ? 0 : wave - 1, whi = last ? NW - 1 : wave + 1;
    static_for<0, Q>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int cx = CX(k);
        if constexpr (cx == 1) {
            const float inner = dpp_from_lower_lane(fs[k]);
            const float edge = first ? halo[k] : LW_XCH(wlo, XSLOT(k), l.y);
            fs[k] = l.x0 ? edge : inner;
        }
        if constexpr (cx == -1) {
            const float inner = dpp_from_upper_lane(fs[k]);
            const float edge = last ? halo[k] : LW_XCH(whi, XSLOT(k), l.y);
            fs[k] = l.x7 ? edge : inner;
        }
    });
    // merged launches run all-neighbour blocks through the GENERAL instantiation too: nothing to patch there (wave-uniform flag)
    const bool needs_patch = GENERAL && (flags & FLAG_ALL_NEIGHBOURS) == 0;
    if constexpr (RHO_ONLY) {
        if constexpr (GENERAL) { if (needs_patch) patch_missing_sources(p, meta, nbr, l, z, own, fs); }
        finish_rho_only(p, flags, own, fs);
        return;
    }
    float uE[3], uW[3], uN[3], uS[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float e_edge = last ? ux_edge_hi[c] : LW_XCH(whi, 21 + c, l.y);
        const float w_edge = first ? ux_edge_lo[c] : LW_XCH(wlo, 18 + c, l.y);
        // cross-lane reads must execute with ALL lanes active: never inside an arm of ?: (that arm runs under a
        // reduced EXEC mask and a DPP read of an inactive lane silently keeps the old value)
        const float e_in = dpp_from_upper_lane(uc[c]), w_in = dpp_from_lower_lane(uc[c]);
        uE[c] = l.x7 ? e_edge : e_in;
        uW[c] = l.x0 ? w_edge : w_in;
        const float n_in = __shfl_down(uc[c], 8, 64), s_in = __shfl_up(uc[c], 8, 64);
        uN[c] = l.y7 ? uy_edge[c] : n_in;
        uS[c] = l.y0 ? uy_edge[c] : s_in;
    }
    if constexpr (GENERAL) { if (needs_patch) patch_missing_sources(p, meta, nbr, l, z, own, fs); }
    finish_cell<POST, WALL>(p, flags, own, fs, uE[0], uE[1], uE[2], uW[0], uW[1], uW[2], uN[0], uN[1], uN[2], uS[0], uS[1], uS[2],
                            uT[0], uT[1], uT[2], uB[0], uB[1], uB[2]);
}

// ---- coarse -> fine interface pass (reference src/physics_kernels.jl:122-137 + src/physics_interpolation.jl) ----
// The reference evaluates interpolate_with_rescaling inline, for the few lanes of a refinement-edge block whose source
// block is missing. Done that way on a 64-wide wavefront it is 27 divergent call sites at ~12 % lane utilisation and
// took 86 % of the GPU time of a 3-level case (profiles/r01_ball1m_kernel_stats_before_interface_pass.csv). The links
// that need it are a static property of the level (topology + global box), so the host lists them once and two small
// kernels evaluate them (per source cell, then per link); the stream-collide kernel loads the value. Same expressions on
// the same inputs as the reference's interpolate_with_rescaling: bit-identical.
__device__ __forceinline__ float weight_rt(int k)
{
    const int cx = k % 3 - 1, cy = (k / 3) % 3 - 1, cz = k / 9 - 1;
    const int d2 = cx * cx + cy * cy + cz * cz;
    return d2 == 0 ? WEIGHT(13) : d2 == 1 ? WEIGHT(12) : d2 == 2 ? WEIGHT(9) : WEIGHT(0);
}

// The two sub-steps a child level takes per parent step see the SAME parent buffers and differ only in the temporal
// weight (0.0 then 0.5, reference src/solver_control.jl:63-83). With TWO = true both values are produced from one set of
// loads (second outputs mac2 / f_iface2 for weight tw2); the second sub-step then skips the pass.
struct InterfaceArgs {
    const int4 *corners;      // 2 per source: parent-cell offsets of the 8 stencil corners, -1 = absent (static, host)
    const float4 *weights;    // per source: wx, wy, wz (static)
    float4 *mac, *mac2;       // per source: interpolated rho, ux, uy, uz for tw / tw2
    const int4 *links;        // per link: (block << 9) | cell, (gbi << 5) | k, source index
    float *f_iface2;
    float tw2;
    int n_sources, n_links;
};

__device__ __forceinline__ float blend_in_time(bool blend, float v_old, float v_new, float tw)
{
    return blend ? v_old * (1.0f - tw) + v_new * tw : v_new;      // reference src/physics_interpolation.jl:83-93
}

// Pass 1, one thread per SOURCE cell (a fine-grid cell just outside this level's blocks): trilinear rho / u of
// reference src/physics_interpolation.jl:64-124 - the same for every population pulled from that cell, so evaluated once.
template <bool TWO>
__global__ __launch_bounds__(256) void k_interface_sources(const SCParams p, const InterfaceArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_sources) return;
    const int4 c0 = a.corners[2 * i], c1 = a.corners[2 * i + 1];
    const int cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float4 w = a.weights[i];
    const float tw = p.temporal_weight, tw2 = a.tw2;
    const bool blend = p.use_temporal == 1 && tw < 0.99f, blend2 = TWO && p.use_temporal == 1 && tw2 < 0.99f;
    float v1[4][8], v2[4][8];                                  // rho, ux, uy, uz at the 8 corners
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        v1[0][n] = 1.0f; v1[1][n] = 0.0f; v1[2][n] = 0.0f; v1[3][n] = 0.0f;      // (w_k, 1, 0, 0, 0, false) default
        v2[0][n] = 1.0f; v2[1][n] = 0.0f; v2[2][n] = 0.0f; v2[3][n] = 0.0f;
        if (cc[n] >= 0) {
            const int c = cc[n];                                 // parent cell: block * 512 + cell
            const size_t cv = (size_t)(c >> 9) * (3 * CELLS) + (c & 511);      // component 0 of the block-major velocity array
            const float vn[4] = {p.prho_new[c], p.pvel_new[cv], p.pvel_new[cv + CELLS], p.pvel_new[cv + 2 * CELLS]};
            float vo[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (blend || blend2) { vo[0] = p.prho_old[c]; vo[1] = p.pvel_old[cv]; vo[2] = p.pvel_old[cv + CELLS]; vo[3] = p.pvel_old[cv + 2 * CELLS]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v1[q][n] = blend_in_time(blend, vo[q], vn[q], tw);
                if (TWO) v2[q][n] = blend_in_time(blend2, vo[q], vn[q], tw2);
            }
        }
    }
    // invalid corners take corner 000's tuple (which may itself be the default), reference :100-107
#pragma unroll
    for (int n = 1; n < 8; ++n)
        if (cc[n] < 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { v1[q][n] = v1[q][0]; v2[q][n] = v2[q][0]; }
        }
#define LW_TL(v, q) trilin(v[q][0], v[q][1], v[q][2], v[q][3], v[q][4], v[q][5], v[q][6], v[q][7], w.x, w.y, w.z)
    a.mac[i] = make_float4(LW_TL(v1, 0), LW_TL(v1, 1), LW_TL(v1, 2), LW_TL(v1, 3));
    if (TWO) a.mac2[i] = make_float4(LW_TL(v2, 0), LW_TL(v2, 1), LW_TL(v2, 2), LW_TL(v2, 3));
#undef LW_TL
}

// Pass 2, one thread per LINK (cell, population k): f_k interpolated over the same 8 corners, equilibrium from pass 1's
// moments, non-equilibrium rescaled (reference src/physics_interpolation.jl:110-135). Expression by expression the
// reference's interpolate_with_rescaling, so the value is bit-identical to the inline call.
template <bool TWO>
__global__ __launch_bounds__(256) void k_interface_links(const SCParams p, const InterfaceArgs a)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.n_links) return;
    const int4 l = a.links[j];
    const int cell = l.x & 511, k = l.y & 31, gbi = l.y >> 5, src = l.z;
    const int4 c0 = a.corners[2 * src], c1 = a.corners[2 * src + 1];
    const int cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float4 w = a.weights[src];
    const float tw = p.temporal_weight, tw2 = a.tw2;
    const bool blend = p.use_temporal == 1 && tw < 0.99f, blend2 = TWO && p.use_temporal == 1 && tw2 < 0.99f;
    const float w_k = weight_rt(k);
    float fc[8], fc2[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        fc[n] = w_k; fc2[n] = w_k;
        if (cc[n] >= 0) {
            const size_t cf = ((size_t)(cc[n] >> 9) * Q + k) * CELLS + (cc[n] & 511);      // population k of the parent cell
            const float fn = p.pf_new[cf];
            float fo = 0.0f;
            if (blend || blend2) fo = p.pf_old[cf];
            fc[n] = blend_in_time(blend, fo, fn, tw);
            if (TWO) fc2[n] = blend_in_time(blend2, fo, fn, tw2);
        }
    }
#pragma unroll
    for (int n = 1; n < 8; ++n)
        if (cc[n] < 0) { fc[n] = fc[0]; fc2[n] = fc2[0]; }
    const float cxf = (float)(k % 3 - 1), cyf = (float)((k / 3) % 3 - 1), czf = (float)(k / 9 - 1);
    const float tau_c = p.tau_parent - 0.5f, tau_f = p.tau - 0.5f;
    const float scale = tau_c > 1.0e-6f ? jl_clamp(tau_f / tau_c, 0.01f, 100.0f) : 1.0f;
    const size_t out = ((size_t)k * p.n_iface_blocks + gbi) * CELLS + cell;
    {
        const float4 m = a.mac[src];
        const float f_int = trilin(fc[0], fc[1], fc[2], fc[3], fc[4], fc[5], fc[6], fc[7], w.x, w.y, w.z);
        const float feq_int = calculate_equilibrium(m.x, m.y, m.z, m.w, w_k, cxf, cyf, czf);
        const float f_neq = f_int - feq_int;
        p.f_iface[out] = feq_int + f_neq * scale;
    }
    if (TWO) {
        const float4 m = a.mac2[src];
        const float f_int = trilin(fc2[0], fc2[1], fc2[2], fc2[3], fc2[4], fc2[5], fc2[6], fc2[7], w.x, w.y, w.z);
        const float feq_int = calculate_equilibrium(m.x, m.y, m.z, m.w, w_k, cxf, cyf, czf);
        const float f_neq = f_int - feq_int;
        a.f_iface2[out] = feq_int + f_neq * scale;
    }
}

// (Round 3 tried both passes in ONE kernel, one thread per source cell walking the populations pulled from it: nine dependent rounds
// of loads per thread at 100 VGPRs - 97 us against 42 + 22 on the wing. profiles/r03_interface_pass_fused_experiment.txt; not kept.)

// ---- Bouzidi correction, reference src/bouzidi_kernel.jl:13-92 ----
struct BouzidiParams {
    float *f_out;
    const float *f_post;
    const _Float16 *q_map;
    const int32_t *cell_block;   // 0-based
    const int8_t *cell_x, *cell_y, *cell_z;   // 0-based
    const int32_t *meta;
    int32_t n_cells;
    float q_min;
    const int4 *links;           // compact list of the links with q > 0, everything static about one link in one record:
                                 // (block * 512 + cell, k, bits of q as Float32, block' * 512 + cell' of the cell one step behind or -1)
    int32_t n_links;
};

// one thread per (listed cell, link k): the 27 links of a cell are independent (each writes its own f_out[cell][opp k]
// and reads only f_post_collision), so spreading them over lanes changes nothing but the latency that is exposed
// With a link list (the q map is static, so the host lists the links with q > 0 once) the threads that would only find
// q = 0 - four out of five - do not exist; q_min is still applied here, per call.
__global__ __launch_bounds__(256) void k_bouzidi(const BouzidiParams p)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    auto at = [](int blk, int kk, int c) { return ((size_t)blk * Q + kk) * CELLS + c; };      // block-major f / q map
    if (p.links) {
        // the link list: one dependent load less than reading q from the map and the neighbour block from the table -
        // record -> two independent f_post loads -> store (this kernel is a few microseconds of pure latency, four times per coarse step)
        if (i >= p.n_links) return;
        const int4 l = p.links[i];
        const float q = __int_as_float(l.z);                  // (float)q_map[...]: Float16 -> Float32 is exact, done on the host
        if (!(q > p.q_min && q <= 1.0f)) return;
        const int b = l.x >> 9, cell = l.x & 511, k = l.y, opp_k = 26 - k;
        const float f_k = p.f_post[at(b, k, cell)];
        if (q < 0.5f) {
            const float f_ff = l.w >= 0 ? p.f_post[at(l.w >> 9, k, l.w & 511)] : f_k;
            const float coeff1 = 2.0f * q;
            p.f_out[at(b, opp_k, cell)] = coeff1 * f_k + (1.0f - coeff1) * f_ff;
        } else {
            const float f_opp_post = p.f_post[at(b, opp_k, cell)];
            const float inv_2q = 1.0f / (2.0f * q);
            const float coeff2 = (2.0f * q - 1.0f) * inv_2q;
            p.f_out[at(b, opp_k, cell)] = inv_2q * f_k + coeff2 * f_opp_post;
        }
        return;
    }
    int b, x, y, z, k;
    {
        if (i >= p.n_cells * Q) return;
        const int c = i / Q;
        k = i - c * Q;
        b = p.cell_block[c];
        x = p.cell_x[c]; y = p.cell_y[c]; z = p.cell_z[c];
    }
    const int opp_k = 26 - k;
    const int cell = x + 8 * y + 64 * z;
    const float q = (float)p.q_map[at(b, k, cell)];
    if (q > p.q_min && q <= 1.0f) {
        const float f_k = p.f_post[at(b, k, cell)];
        if (q < 0.5f) {
            const int nx = x + (opp_k % 3 - 1), ny = y + ((opp_k / 3) % 3 - 1), nz = z + (opp_k / 9 - 1);
            float f_ff = f_k;
            if (nx >= 0 && nx < BS && ny >= 0 && ny < BS && nz >= 0 && nz < BS) {
                f_ff = p.f_post[at(b, k, nx + 8 * ny + 64 * nz)];
            } else {
                const int ox = nx < 0 ? -1 : (nx >= BS ? 1 : 0);
                const int oy = ny < 0 ? -1 : (ny >= BS ? 1 : 0);
                const int oz = nz < 0 ? -1 : (nz >= BS ? 1 : 0);
                const int nbb = p.meta[(int64_t)b * NBR_STRIDE + DIR(ox, oy, oz)];
                if (nbb >= 0) f_ff = p.f_post[at(nbb, k, (nx & 7) + 8 * (ny & 7) + 64 * (nz & 7))];
            }
            const float coeff1 = 2.0f * q;
            p.f_out[at(b, opp_k, cell)] = coeff1 * f_k + (1.0f - coeff1) * f_ff;
        } else {
            const float f_opp_post = p.f_post[at(b, opp_k, cell)];
            const float inv_2q = 1.0f / (2.0f * q);
            const float coeff2 = (2.0f * q - 1.0f) * inv_2q;
            p.f_out[at(b, opp_k, cell)] = inv_2q * f_k + coeff2 * f_opp_post;
        }
    }
}

// ---- init_eq!, reference src/main.jl:109-124 ----
__global__ void k_fill_weights(float *f, int64_t n_cells)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // cell index: block * 512 + cell
    if (i >= n_cells) return;
    float *q = f + (i >> 9) * (int64_t)(Q * CELLS) + (i & 511);
    static_for<0, Q>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        q[k * CELLS] = WEIGHT(k);
    });
}
__global__ void k_fill(float *a, int64_t n, float v)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = v;
}

// ---- surface stresses, reference src/forces/surface.jl:32-76 (compute_stress_from_cell) and :138-266 (map_stresses_kernel!) ----
struct SurfaceParams {
    const float *rho, *vel;
    const uint8_t *obstacle;
    const int32_t *block_pointer;    // [gdx,gdy,gdz] 1-based, 0 = absent
    int32_t gdx, gdy, gdz, n_tri, radius;
    float dx, tau, off_x, off_y, off_z, pressure_scale, stress_scale;
    const float *centers, *normals;  // [n_tri * 3]
    float *p, *tx, *ty, *tz;         // [n_tri]
};

__global__ __launch_bounds__(128) void k_map_stresses(const SurfaceParams s)
{
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= s.n_tri) return;
    const float tx = s.centers[3 * i] + s.off_x, ty = s.centers[3 * i + 1] + s.off_y, tz = s.centers[3 * i + 2] + s.off_z;
    const float nx = s.normals[3 * i], ny = s.normals[3 * i + 1], nz = s.normals[3 * i + 2];
    const int g_x = (int)floorf(tx / s.dx) + 1, g_y = (int)floorf(ty / s.dx) + 1, g_z = (int)floorf(tz / s.dx) + 1;
    float best_d = 1.0e10f, best_rho = 1.0f, ux = 0.0f, uy = 0.0f, uz = 0.0f, best_wd = 0.5f;
    bool found = false;
    for (int radius = 0; radius <= s.radius; ++radius) {
        if (found && radius > 1) break;
        for (int dz = -radius; dz <= radius; ++dz)
            for (int dy = -radius; dy <= radius; ++dy)
                for (int dx = -radius; dx <= radius; ++dx) {
                    if (radius > 0 && !(abs(dx) == radius || abs(dy) == radius || abs(dz) == radius)) continue;   // the shell only
                    const int cgx = g_x + dx, cgy = g_y + dy, cgz = g_z + dz;
                    if (cgx < 1 || cgy < 1 || cgz < 1) continue;
                    const int bx = (cgx - 1) / BS + 1, by = (cgy - 1) / BS + 1, bz = (cgz - 1) / BS + 1;
                    if (bx > s.gdx || by > s.gdy || bz > s.gdz) continue;
                    const int32_t b = s.block_pointer[(size_t)(bx - 1) + (size_t)s.gdx * ((size_t)(by - 1) + (size_t)s.gdy * (bz - 1))];
                    if (b <= 0) continue;
                    const int64_t c = (int64_t)((cgx - 1) % BS) + 8 * ((cgy - 1) % BS) + 64 * ((cgz - 1) % BS) + 512 * (int64_t)(b - 1);
                    if (s.obstacle[c]) continue;
                    const float ccx = ((float)cgx - 0.5f) * s.dx, ccy = ((float)cgy - 0.5f) * s.dx, ccz = ((float)cgz - 0.5f) * s.dx;
                    const float ex = tx - ccx, ey = ty - ccy, ez = tz - ccz;
                    const float d2 = ex * ex + ey * ey + ez * ez;
                    if (d2 < best_d) {
                        best_d = d2;
                        best_rho = s.rho[c];
                        const int64_t cv = (c >> 9) * (int64_t)(3 * CELLS) + (c & 511);      // block-major velocity
                        ux = s.vel[cv]; uy = s.vel[cv + CELLS]; uz = s.vel[cv + 2 * CELLS];
                        best_wd = sqrtf(d2) / s.dx;
                        found = true;
                    }
                }
    }
    float p = 0.0f, sx = 0.0f, sy = 0.0f, sz = 0.0f;
    if (found) {
        const float wall_dist = fmaxf(best_wd, 0.5f);
        p = ((best_rho - 1.0f) / 3.0f) * s.pressure_scale;
        const float udn = ux * nx + uy * ny + uz * nz;
        const float utx = ux - udn * nx, uty = uy - udn * ny, utz = uz - udn * nz;
        const float umag = sqrtf(utx * utx + uty * uty + utz * utz);
        const float nu_lat = (s.tau - 0.5f) / 3.0f;
        if (umag > 1.0e-10f && wall_dist > 0.01f) {
            const float tmag = (best_rho * nu_lat * umag / wall_dist) * s.stress_scale;
            sx = (utx / umag) * tmag; sy = (uty / umag) * tmag; sz = (utz / umag) * tmag;
        }
    }
    s.p[i] = p; s.tx[i] = sx; s.ty[i] = sy; s.tz[i] = sz;
}

// ---- compute_flow_stats, reference src/diagnostics.jl:56-94 (CUDA branch): minimum of rho over non-obstacle cells ----
// Finite rho is > 0 (clamped at 0.01, obstacle cells hold 1), so the IEEE bit pattern orders like the value and an integer
// atomicMin does the job; a minimum does not depend on the order of its operands: identical to any host reduction.
// A diverged run holds NaN, and the reference's minimum() PROPAGATES it (Julia's min, like jl_max above) where fminf and the
// integer compare would silently drop it: a NaN in any counted cell raises out[1], and the caller reports NaN.
__global__ __launch_bounds__(256) void k_rho_min(const float *__restrict__ rho, const uint8_t *__restrict__ obstacle, int64_t n, int *__restrict__ out)
{
    float m = __int_as_float(0x7f800000);   // +inf
    bool nan = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        if (!obstacle[i]) {
            const float v = rho[i];
            nan = nan || v != v;
            m = fminf(m, v);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fminf(m, __shfl_xor(m, o, 64));
    const bool any_nan = __any(nan);
    if ((threadIdx.x & 63) == 0) {
        atomicMin(out, __float_as_int(m));
        if (any_nan) atomicOr(out + 1, 1);
    }
}

// ---- internal storage (ludwig_hip.hip "block order", "block-major"): the caller's arrays keep the reference's layout,
// [8,8,8,n_blocks,K] with the reference's block order; the device arrays hold the blocks in the library's own order, block-major.
// ref2int[b_reference] = b_internal (nullptr = same order) ----
// element offset in the reference layout (cell + 512 b + 512 n_blocks k) -> the same element in the device array
__device__ __forceinline__ int64_t to_internal_offset(int64_t off, const int32_t *__restrict__ ref2int, int64_t n_cells, int K)
{
    const int64_t k = off / n_cells, r = off - k * n_cells;
    const int64_t blk = ref2int ? (int64_t)ref2int[r >> 9] : (r >> 9);
    return (blk * K + k) * CELLS + (r & 511);
}
// one component k of a K-component field: src / dst = that component in the reference layout, n = 512 n_blocks elements
template <class T>
__global__ __launch_bounds__(256) void k_component_to_internal(T *__restrict__ dev, const T *__restrict__ src, const int32_t *__restrict__ ref2int, int64_t n, int K, int k)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dev[((ref2int ? (int64_t)ref2int[i >> 9] : (i >> 9)) * K + k) * CELLS + (i & 511)] = src[i];
}
template <class T>
__global__ __launch_bounds__(256) void k_component_to_reference(T *__restrict__ dst, const T *__restrict__ dev, const int32_t *__restrict__ ref2int, int64_t n, int K, int k)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = dev[((ref2int ? (int64_t)ref2int[i >> 9] : (i >> 9)) * K + k) * CELLS + (i & 511)];
}

// ---- halo pack / unpack: index holds element offsets in the REFERENCE layout ----
__global__ void k_gather(const float *__restrict__ field, const int64_t *__restrict__ index, int64_t n, float *__restrict__ dst,
                         const int32_t *__restrict__ ref2int, int64_t n_cells, int K)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = field[to_internal_offset(index[i], ref2int, n_cells, K)];
}
__global__ void k_scatter(float *__restrict__ field, const int64_t *__restrict__ index, int64_t n, const float *__restrict__ src,
                          const int32_t *__restrict__ ref2int, int64_t n_cells, int K)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) field[to_internal_offset(index[i], ref2int, n_cells, K)] = src[i];
}

// ---- structured halo pack / unpack (ludwig_halo_plan_*): the elements of a message grouped into OCTETS - aligned groups of 8
// consecutive floats of the device array = one 32-B sector - with a mask of the members that travel. desc = (octet index in the
// field, position of the octet's first member in the message, mask, unused). Eight threads share a descriptor: the sector is
// read / written once, whole, and the message side is contiguous. A z face (whole 8x8 planes) and a y face (rows of 8) are full
// octets: 16 B of descriptor per 32 B of payload, against 8 B of index per 4 B with k_gather / k_scatter. Octets with one or two
// members (an x face: single cells 32 B apart) would waste six or seven of their eight threads: those members go to a second
// list, one thread each, single = (octet index, message position << 3 | member) - threads [8 n_oct, 8 n_oct + n_single). ----
__global__ __launch_bounds__(256) void k_pack_octets(const float *__restrict__ field, const uint4 *__restrict__ desc, int64_t n_oct,
                                                     const uint2 *__restrict__ single, int64_t n_single, float *__restrict__ msg)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, o = t >> 3;
    if (o < n_oct) {
        const int j = (int)(t & 7);
        const uint4 d = desc[o];
        if ((d.z >> j) & 1u) msg[(size_t)d.y + __popc(d.z & ((1u << j) - 1u))] = field[(size_t)d.x * 8 + j];
    } else {
        const int64_t i = t - n_oct * 8;
        if (i >= n_single) return;
        const uint2 d = single[i];
        msg[d.y >> 3] = field[(size_t)d.x * 8 + (d.y & 7u)];
    }
}
__global__ __launch_bounds__(256) void k_unpack_octets(float *__restrict__ field, const uint4 *__restrict__ desc, int64_t n_oct,
                                                       const uint2 *__restrict__ single, int64_t n_single, const float *__restrict__ msg)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, o = t >> 3;
    if (o < n_oct) {
        const int j = (int)(t & 7);
        const uint4 d = desc[o];
        if ((d.z >> j) & 1u) field[(size_t)d.x * 8 + j] = msg[(size_t)d.y + __popc(d.z & ((1u << j) - 1u))];
    } else {
        const int64_t i = t - n_oct * 8;
        if (i >= n_single) return;
        const uint2 d = single[i];
        field[(size_t)d.x * 8 + (d.y & 7u)] = msg[d.y >> 3];
    }
}

}  // namespace lw
