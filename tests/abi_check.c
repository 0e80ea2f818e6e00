/* C99 caller of include/ludwig_hip.h (tests/test_library_abi.py builds it with `gcc -std=c99 -Wall -Werror -Iinclude`):
 *   1. the header is valid C99 on its own;
 *   2. prints sizeof / offsetof of the ABI structs, which the test compares with the ctypes mirrors (open_ludwig_amd/_lib.py)
 *      and with the struct definitions of the Julia binding (julia/LudwigHIP.jl) - the only check that file can get here;
 *   3. dlopens the library given as argv[1] and calls it through the header's prototypes, without a GPU. */
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "ludwig_hip.h"

#define FIELD(S, f) printf("field %s %s %zu %zu\n", #S, #f, offsetof(S, f), sizeof(((S *)0)->f))
#define SIZE(S) printf("struct %s %zu\n", #S, sizeof(S))

int main(int argc, char **argv)
{
    SIZE(LudwigLevelHost);
    FIELD(LudwigLevelHost, level_id); FIELD(LudwigLevelHost, n_blocks); FIELD(LudwigLevelHost, n_owned); FIELD(LudwigLevelHost, tau);
    FIELD(LudwigLevelHost, grid_dim_x); FIELD(LudwigLevelHost, grid_dim_y); FIELD(LudwigLevelHost, grid_dim_z);
    FIELD(LudwigLevelHost, block_pointer); FIELD(LudwigLevelHost, neighbor_table);
    FIELD(LudwigLevelHost, map_x); FIELD(LudwigLevelHost, map_y); FIELD(LudwigLevelHost, map_z);
    FIELD(LudwigLevelHost, obstacle); FIELD(LudwigLevelHost, sponge); FIELD(LudwigLevelHost, wall_dist);
    FIELD(LudwigLevelHost, enable_temporal_interpolation); FIELD(LudwigLevelHost, n_boundary_cells);
    FIELD(LudwigLevelHost, bouzidi_q_map); FIELD(LudwigLevelHost, bouzidi_cell_block);
    FIELD(LudwigLevelHost, bouzidi_cell_x); FIELD(LudwigLevelHost, bouzidi_cell_y); FIELD(LudwigLevelHost, bouzidi_cell_z);
    FIELD(LudwigLevelHost, comm_boundary); FIELD(LudwigLevelHost, store_post_collision_everywhere);
    SIZE(LudwigStepFlags);
    FIELD(LudwigStepFlags, domain_nx); FIELD(LudwigStepFlags, domain_ny); FIELD(LudwigStepFlags, domain_nz);
    FIELD(LudwigStepFlags, is_symmetric); FIELD(LudwigStepFlags, wall_model_active); FIELD(LudwigStepFlags, use_temporal_interp);
    FIELD(LudwigStepFlags, sponge_blend_distributions); FIELD(LudwigStepFlags, c_wale); FIELD(LudwigStepFlags, nu_sgs_background);
    FIELD(LudwigStepFlags, inlet_turbulence); FIELD(LudwigStepFlags, q_min_threshold);
    SIZE(LudwigSurfaceParams);
    FIELD(LudwigSurfaceParams, dx); FIELD(LudwigSurfaceParams, tau); FIELD(LudwigSurfaceParams, offset_x); FIELD(LudwigSurfaceParams, offset_y);
    FIELD(LudwigSurfaceParams, offset_z); FIELD(LudwigSurfaceParams, pressure_scale); FIELD(LudwigSurfaceParams, stress_scale);
    FIELD(LudwigSurfaceParams, search_radius);
    SIZE(LudwigLevelInfo);
    FIELD(LudwigLevelInfo, n_blocks); FIELD(LudwigLevelInfo, n_owned); FIELD(LudwigLevelInfo, n_fast_blocks); FIELD(LudwigLevelInfo, n_general_blocks);
    FIELD(LudwigLevelInfo, n_boundary_cells); FIELD(LudwigLevelInfo, has_temporal_storage); FIELD(LudwigLevelInfo, has_post_collision);
    FIELD(LudwigLevelInfo, n_xrun_blocks); FIELD(LudwigLevelInfo, device_bytes);
    SIZE(LudwigHaloPlanDesc);
    FIELD(LudwigHaloPlanDesc, n_peers); FIELD(LudwigHaloPlanDesc, peer_ranks); FIELD(LudwigHaloPlanDesc, send_count); FIELD(LudwigHaloPlanDesc, recv_count);
    FIELD(LudwigHaloPlanDesc, send_index); FIELD(LudwigHaloPlanDesc, recv_index);
    printf("enum LUDWIG_HALO_GROUPS %d\n", (int)LUDWIG_HALO_GROUPS);
    printf("enum LUDWIG_UNIQUE_ID_BYTES %d\n", (int)LUDWIG_UNIQUE_ID_BYTES);
    printf("enum LUDWIG_FIELD_COUNT %d\n", (int)LUDWIG_FIELD_COUNT);
    printf("enum LUDWIG_WALL_DIST %d\n", (int)LUDWIG_WALL_DIST);
    printf("enum LUDWIG_PART_INTERIOR %d\n", (int)LUDWIG_PART_INTERIOR);
    if (argc < 2) return 0;

    void *so = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!so) { printf("dlopen failed: %s\n", dlerror()); return 2; }
    /* typed through the header's own prototypes: a signature drift between header and caller is a compile error */
    int (*abi_version)(void) = (int (*)(void))dlsym(so, "ludwig_abi_version");
    const char *(*last_error)(void) = (const char *(*)(void))dlsym(so, "ludwig_last_error");
    int (*level_create)(const LudwigLevelHost *, int, LudwigLevel **) =
        (int (*)(const LudwigLevelHost *, int, LudwigLevel **))dlsym(so, "ludwig_level_create");
    int (*level_info)(const LudwigLevel *, LudwigLevelInfo *) = (int (*)(const LudwigLevel *, LudwigLevelInfo *))dlsym(so, "ludwig_level_info");
    void (*level_destroy)(LudwigLevel *) = (void (*)(LudwigLevel *))dlsym(so, "ludwig_level_destroy");
    if (!abi_version || !last_error || !level_create || !level_info || !level_destroy) { printf("missing symbol\n"); return 3; }
    if (0) {   /* never run: only checks that the pointers above have the header's types */
        abi_version = ludwig_abi_version; last_error = ludwig_last_error; level_create = ludwig_level_create;
        level_info = ludwig_level_info; level_destroy = ludwig_level_destroy;
    }
    printf("call abi_version %d %d\n", abi_version(), LUDWIG_ABI_VERSION);
    LudwigLevel *lvl = (LudwigLevel *)0x1;
    int rc = level_create(NULL, 0, &lvl);
    printf("call level_create_null %d %s\n", rc, lvl == NULL ? "out_cleared" : "out_kept");
    printf("call last_error %s\n", last_error());
    LudwigLevelHost h;
    memset(&h, 0, sizeof h);
    h.level_id = 1; h.n_blocks = -5;
    rc = level_create(&h, 0, &lvl);
    printf("call level_create_bad %d\n", rc);
    LudwigLevelInfo info;
    printf("call level_info_null %d\n", level_info(NULL, &info));
    level_destroy(NULL);
    dlclose(so);
    return 0;
}
