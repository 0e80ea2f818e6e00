/* libludwig_setup.so - host-side case set-up for large levels. Plain C ABI, no GPU code.
 *
 * Every entry restates one host function of the reference's set-up for a level given as its sorted list of active blocks
 * (`BlockLevel.active_block_coords`: 1-based block coordinates, int32 [n_blocks][3]) and the surface mesh with the mesh offset
 * already added (Float64 [n_tri][3 corners][3]). Cell arrays are the reference's [8,8,8,n_blocks] column-major arrays, i.e. linear
 * index x + 8 y + 64 z + 512 b (0-based). Arithmetic is Float64 in the reference's operation order (library built with
 * -ffp-contract=off). Calls return a count >= 0 or 0 on success and -1 on a bad argument (`lws_last_error`).
 * n_threads <= 0 = all hardware threads (the reference: Base.Threads over blocks).
 */
#ifndef LUDWIG_SETUP_H
#define LUDWIG_SETUP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* lws_last_error(void);

/* voxelize_blocks!(obstacle_arr, sorted_blocks, mesh, dx, mesh_offset), src/domain_generation.jl:74-112 with
 * build_block_triangle_map :34-72 (margin 2 dx) and triangle_intersects_aabb :10-32 (half box 0.75 dx, tolerance 1.001).
 * Sets obstacle[cell] = 1 for surface cells; the array must come in zeroed (or holding cells to keep). */
int lws_voxelize(const double* tri, int64_t n_tri, double dx, const int32_t* coords, int64_t n_blocks, uint8_t* obstacle, int n_threads);

/* perform_flood_fill!(obstacle_arr, sorted_blocks, block_ptr, ...), src/domain_generation.jl:114-203. Fluid cells not reachable
 * (6-neighbourhood, through existing blocks only) from the fluid cells of the blocks with the smallest bx turn solid.
 * Returns the number of cells filled - the count the reference prints. */
int64_t lws_flood_fill(const int32_t* coords, int64_t n_blocks, uint8_t* obstacle);

/* compute_wall_distances!(wall_dist_arr, sorted_blocks, obstacle_arr, mesh, dx, mesh_offset), src/domain_generation.jl:371-434.
 * `wall` must come in filled with 100.0f (the BlockLevel constructor's default, src/blocks.jl:150). Returns the number of near-wall
 * cells - the count the reference prints. */
int64_t lws_wall_distance(const int32_t* coords, int64_t n_blocks, const uint8_t* obstacle, double dx, float* wall, int n_threads);

/* compute_bouzidi_qmap_sparse(level_active_coords, mesh, dx, mesh_offset, block_size), src/bouzidi_setup.jl:64-167 with
 * compute_q_for_cell / ray_triangle_intersection, src/bouzidi_math.jl:9-102 (triangles binned with margin 2.5 dx).
 * q_map: Float16 bits, [27][n_blocks][512] = the reference's [8,8,8,n_blocks,27] column-major, zeroed by the caller.
 * boundary[n_blocks * 512]: zeroed by the caller, set to 1 for every cell that got a q (the reference's boundary-cell list, whose
 * order is thread-dependent there, is these cells). Returns their number. The triangle map (`tri_map`) is not produced: nothing on the
 * stepping path reads it. */
int64_t lws_bouzidi_qmap(const double* tri, int64_t n_tri, double dx, const int32_t* coords, int64_t n_blocks, uint16_t* q_map,
                         uint8_t* boundary, int n_threads);

/* Float16(x::Float64) as the q-map stores it (one rounding to nearest even); exported for the tests. */
uint16_t lws_f64_to_f16(double x);

#ifdef __cplusplus
}
#endif
#endif
