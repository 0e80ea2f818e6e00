// D3Q27 lattice as compile-time constants.
// Restates build_lattice_arrays_gpu (reference src/physics_v2.jl:99-117) with k 0-based:
//   k = (cx+1) + 3(cy+1) + 9(cz+1);  opp[k] = 26-k;  weights 8/27, 2/27, 1/54, 1/216 (Float32 division).
#pragma once

namespace lw {

constexpr int BS = 8;        // BLOCK_SIZE, reference src/blocks.jl:14
constexpr int CELLS = 512;   // cells per block
constexpr int Q = 27;

__host__ __device__ constexpr int CX(int k) { return k % 3 - 1; }
__host__ __device__ constexpr int CY(int k) { return (k / 3) % 3 - 1; }
__host__ __device__ constexpr int CZ(int k) { return k / 9 - 1; }
__host__ __device__ constexpr int OPP(int k) { return 26 - k; }
__host__ __device__ constexpr int MIRROR_Y(int k) { return k - 6 * CY(k); }
__host__ __device__ constexpr int MIRROR_Z(int k) { return k - 18 * CZ(k); }
__host__ __device__ constexpr float WEIGHT(int k)
{
    const int d2 = CX(k) * CX(k) + CY(k) * CY(k) + CZ(k) * CZ(k);
    return d2 == 0 ? 8.0f / 27.0f : d2 == 1 ? 2.0f / 27.0f : d2 == 2 ? 1.0f / 54.0f : 1.0f / 216.0f;
}
// 27-neighbourhood direction index used by neighbor_table (reference src/domain_topology.jl:150), 0-based
__host__ __device__ constexpr int DIR(int ox, int oy, int oz) { return (ox + 1) + 3 * (oy + 1) + 9 * (oz + 1); }

constexpr float KAPPA = 0.41f;              // reference src/physics_v2.jl:15
constexpr float CS2_PHYSICS = 1.0f / 3.0f;  // reference src/physics_v2.jl:16

// per-block metadata row: [0..26] neighbour block (0-based, -1 absent), [27] flags, [28..30] bx,by,bz (1-based), [31] gbi
constexpr int NBR_STRIDE = 32;
constexpr int NBR_FLAGS = 27;
constexpr int NBR_BX = 28, NBR_BY = 29, NBR_BZ = 30;
constexpr int NBR_GBI = 31;              // compact index of the block among the level's interface (general) blocks, or -1

constexpr int FLAG_ALL_NEIGHBOURS = 1;   // all 26 neighbour blocks present -> no domain-edge / interface code
constexpr int FLAG_HAS_OBSTACLE = 2;
constexpr int FLAG_HAS_SPONGE = 4;
constexpr int FLAG_HAS_NEAR_WALL = 8;    // some cell with 0 < wall_dist < 10
constexpr int FLAG_STORE_POST = 16;      // f_post_collision has a reader here: the block holds a Bouzidi cell or touches one

}  // namespace lw
