/*
 * ludwig_oracle.c - scalar FP32 CPU restatement of the OPEN_Ludwig hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ludwig_oracle.h). Pinned by the reference's run-log series
 * (tests/test_case_ball1m.py); there are no unit-level reference vectors (the reference
 * has none and cannot be executed in this image).
 *
 * Every function cites the reference lines it restates (paths under /root/reference).
 * Operation ORDER follows the reference expression by expression; build with
 * -ffp-contract=off so that a*b+c is two roundings as in Julia's CPU code generation.
 * Julia semantics that differ from C and are restated explicitly:
 *   - Int32 arithmetic wraps                     -> uint32_t arithmetic
 *   - max(a,b) propagates NaN                    -> jl_maxf
 *   - Float32 ^, log, cos evaluate in Float64 and round once -> (float)pow((double)..)
 *   - n-ary + and * associate to the left        -> same as C
 */
#include "ludwig_oracle.h"
#include "../open_ludwig_amd/csrc/jl_math.h"

#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define BS 8            /* BLOCK_SIZE, src/blocks.jl:14 */
#define CELLS 512

/* src/physics_v2.jl:15-17 */
static const float KAPPA = 0.41f;
#define CS2_PHYSICS (1.0f / 3.0f)

/* ---- D3Q27 tables, src/physics_v2.jl:99-117 (k is 0-based here, reference is 1-based) ---- */
static int32_t  L_cx[27], L_cy[27], L_cz[27], L_opp[27], L_my[27], L_mz[27];
static float    L_w[27];
static int      L_ready = 0;

static void lattice_init(void)
{
    if (L_ready) return;
    int k = 0;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                L_cx[k] = dx; L_cy[k] = dy; L_cz[k] = dz;
                int d2 = dx*dx + dy*dy + dz*dz;
                L_w[k] = d2 == 0 ? 8.0f/27.0f : d2 == 1 ? 2.0f/27.0f : d2 == 2 ? 1.0f/54.0f : 1.0f/216.0f;
                ++k;
            }
    for (int i = 0; i < 27; ++i)
        for (int j = 0; j < 27; ++j) {
            if (L_cx[j] == -L_cx[i] && L_cy[j] == -L_cy[i] && L_cz[j] == -L_cz[i]) L_opp[i] = j;
            if (L_cx[j] ==  L_cx[i] && L_cy[j] == -L_cy[i] && L_cz[j] ==  L_cz[i]) L_my[i]  = j;
            if (L_cx[j] ==  L_cx[i] && L_cy[j] ==  L_cy[i] && L_cz[j] == -L_cz[i]) L_mz[i]  = j;
        }
    L_ready = 1;
}

void oracle_lattice(int32_t *cx, int32_t *cy, int32_t *cz, float *w,
                    int32_t *opp, int32_t *mirror_y, int32_t *mirror_z)
{
    lattice_init();
    for (int k = 0; k < 27; ++k) {
        cx[k] = L_cx[k]; cy[k] = L_cy[k]; cz[k] = L_cz[k]; w[k] = L_w[k];
        opp[k] = L_opp[k] + 1; mirror_y[k] = L_my[k] + 1; mirror_z[k] = L_mz[k] + 1;  /* 1-based like the reference */
    }
}

int  oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---- Julia Base semantics ---- */
static inline float jl_maxf(float a, float b)
{
    if (a != a) return a;
    if (b != b) return b;
    return a > b ? a : b;
}
static inline float jl_clampf(float x, float lo, float hi)
{
    return x > hi ? hi : (x < lo ? lo : x);
}
/* Base.^(::Float32,::Float32) widens, evaluates exp2(log2(x)*y) in Float64 and rounds once; Base.log(::Float32) is a Float64
 * kernel rounded once. The double log2 / exp2 / log come from the header the HIP kernels compile too (jl_math.h): the one
 * piece of arithmetic the oracle shares with the product, so that wall-model cells are bit-comparable. It is itself
 * checked against glibc in tests/test_jl_math.py through the two hooks below. */
#ifdef ORACLE_LIBM_WALL_MODEL
/* Independence check (tests/test_oracle_libm_flavour.py): the wall model's two transcendental calls through glibc's double pow / log
 * instead of the shared header, so that an error in jl_math.h cannot cancel between the oracle and the product. Never the parity build. */
static inline float jl_powf(float x, float y) { return (float)pow((double)x, (double)y); }
static inline float jl_logf(float x)          { return (float)log((double)x); }
#else
static inline float jl_powf(float x, float y) { return lw_powf(x, y); }
static inline float jl_logf(float x)          { return lw_logf(x); }
#endif
void oracle_jl_math(int which, const double *x, double *out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = which == 0 ? lw_log2(x[i]) : which == 1 ? lw_exp2(x[i]) : lw_log(x[i]);
}
void oracle_jl_powf(const float *x, const float *y, float *out, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) out[i] = lw_powf(x[i], y[i]);
}
static inline float jl_cosf(float x)          { return (float)cos((double)x); }

/* Float16 -> Float32, used at src/bouzidi_kernel.jl:36 */
float oracle_half_to_float(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp  = (h >> 10) & 0x1Fu;
    uint32_t man  = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            man &= 0x3FFu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f; memcpy(&f, &bits, 4); return f;
}

/* ---- addressing (src/blocks.jl:118-150); x,y,z 1-based local, b 1-based, k 0-based ---- */
static inline int64_t idx4(int x, int y, int z, int b)
{
    return (int64_t)(x - 1) + 8 * (int64_t)(y - 1) + 64 * (int64_t)(z - 1) + 512 * (int64_t)(b - 1);
}
static inline int64_t idx5(int x, int y, int z, int b, int k, int64_t nb)
{
    return idx4(x, y, z, b) + 512 * nb * (int64_t)k;
}

/* src/physics_utils.jl:17-22 */
static inline uint32_t gpu_hash(uint32_t h)
{
    h = (h ^ (h >> 16)) * 0x85ebca6bu;
    h = (h ^ (h >> 13)) * 0xc2b2ae35u;
    return h ^ (h >> 16);
}
/* src/physics_utils.jl:24-28 (Int32 products wrap) */
float oracle_gradient_noise(int32_t gx, int32_t gy, int32_t gz, int32_t seed)
{
    uint32_t combined = (uint32_t)gx * 374761393u + (uint32_t)gy * 668265263u
                      + (uint32_t)gz * 1274126177u + (uint32_t)seed;
    uint32_t h = gpu_hash(combined);
    return ((float)(h & 0xFFFFu) / 32768.0f) - 1.0f;
}

/* src/physics_utils.jl:34-39 */
static inline float calculate_equilibrium(float rho, float ux, float uy, float uz,
                                          float w_k, float cx, float cy, float cz)
{
    float cu  = cx*ux + cy*uy + cz*uz;
    float usq = ux*ux + uy*uy + uz*uz;
    return rho * w_k * (1.0f + 3.0f*cu + 4.5f*cu*cu - 1.5f*usq);
}

/* src/physics_utils.jl:45-70 */
static inline void get_velocity_neighbor(const float *vel_in, int x, int y, int z, int b,
                                         int dx, int dy, int dz, const int32_t *nbt, int64_t nb,
                                         float *o1, float *o2, float *o3)
{
    int nx = x + dx, ny = y + dy, nz = z + dz;
    if (nx >= 1 && nx <= BS && ny >= 1 && ny <= BS && nz >= 1 && nz <= BS) {
        *o1 = vel_in[idx5(nx, ny, nz, b, 0, nb)];
        *o2 = vel_in[idx5(nx, ny, nz, b, 1, nb)];
        *o3 = vel_in[idx5(nx, ny, nz, b, 2, nb)];
        return;
    }
    int off_x = nx < 1 ? -1 : (nx > BS ? 1 : 0);
    int off_y = ny < 1 ? -1 : (ny > BS ? 1 : 0);
    int off_z = nz < 1 ? -1 : (nz > BS ? 1 : 0);
    int dir = (off_x + 1) + (off_y + 1) * 3 + (off_z + 1) * 9;       /* 0-based */
    int32_t nbi = nbt[(int64_t)(b - 1) + nb * dir];
    if (nbi > 0) {
        int nnx = nx < 1 ? nx + BS : (nx > BS ? nx - BS : nx);
        int nny = ny < 1 ? ny + BS : (ny > BS ? ny - BS : ny);
        int nnz = nz < 1 ? nz + BS : (nz > BS ? nz - BS : nz);
        *o1 = vel_in[idx5(nnx, nny, nnz, nbi, 0, nb)];
        *o2 = vel_in[idx5(nnx, nny, nnz, nbi, 1, nb)];
        *o3 = vel_in[idx5(nnx, nny, nnz, nbi, 2, nb)];
        return;
    }
    *o1 = vel_in[idx5(x, y, z, b, 0, nb)];
    *o2 = vel_in[idx5(x, y, z, b, 1, nb)];
    *o3 = vel_in[idx5(x, y, z, b, 2, nb)];
}

/* ---- src/physics_interpolation.jl:16-138 ---- */
typedef struct { float f, rho, ux, uy, uz; int valid; } Blend;

static inline Blend get_blended(const OracleLevel *par, const float *pf_new, const float *pvel_new,
                                int pgx, int pgy, int pgz, int k, float w_k,
                                float temporal_weight, int use_temporal_interp)
{
    /* :49-62 - Julia div/rem truncate toward zero, like C */
    int pbx = (pgx - 1) / BS + 1;
    int pby = (pgy - 1) / BS + 1;
    int pbz = (pgz - 1) / BS + 1;
    int64_t pnb = par->n_blocks;
    if (pbx >= 1 && pbx <= par->grid_dim_x && pby >= 1 && pby <= par->grid_dim_y &&
        pbz >= 1 && pbz <= par->grid_dim_z) {
        int32_t pb = par->block_pointer[(int64_t)(pbx - 1) + (int64_t)par->grid_dim_x *
                                        ((int64_t)(pby - 1) + (int64_t)par->grid_dim_y * (pbz - 1))];
        if (pb > 0) {
            int plx = (pgx - 1) % BS + 1;
            int ply = (pgy - 1) % BS + 1;
            int plz = (pgz - 1) % BS + 1;
            Blend r;
            float f_new   = pf_new[idx5(plx, ply, plz, pb, k, pnb)];
            float rho_new = par->rho[idx4(plx, ply, plz, pb)];
            float ux_new  = pvel_new[idx5(plx, ply, plz, pb, 0, pnb)];
            float uy_new  = pvel_new[idx5(plx, ply, plz, pb, 1, pnb)];
            float uz_new  = pvel_new[idx5(plx, ply, plz, pb, 2, pnb)];
            if (use_temporal_interp == 1 && temporal_weight < 0.99f) {      /* :69-82 */
                float f_old   = par->f_old[idx5(plx, ply, plz, pb, k, pnb)];
                float rho_old = par->rho_old[idx4(plx, ply, plz, pb)];
                float ux_old  = par->vel_old[idx5(plx, ply, plz, pb, 0, pnb)];
                float uy_old  = par->vel_old[idx5(plx, ply, plz, pb, 1, pnb)];
                float uz_old  = par->vel_old[idx5(plx, ply, plz, pb, 2, pnb)];
                float tw = temporal_weight;
                r.f   = f_old  *(1.0f - tw) + f_new  *tw;
                r.rho = rho_old*(1.0f - tw) + rho_new*tw;
                r.ux  = ux_old *(1.0f - tw) + ux_new *tw;
                r.uy  = uy_old *(1.0f - tw) + uy_new *tw;
                r.uz  = uz_old *(1.0f - tw) + uz_new *tw;
                r.valid = 1;
                return r;
            }
            r.f = f_new; r.rho = rho_new; r.ux = ux_new; r.uy = uy_new; r.uz = uz_new; r.valid = 1;
            return r;
        }
    }
    Blend d = { w_k, 1.0f, 0.0f, 0.0f, 0.0f, 0 };     /* :86 */
    return d;
}

static inline float trilin(float v000, float v100, float v010, float v110,
                           float v001, float v101, float v011, float v111,
                           float wx, float wy, float wz)
{   /* :110-118 */
    float c00 = v000*(1.0f - wx) + v100*wx;
    float c01 = v001*(1.0f - wx) + v101*wx;
    float c10 = v010*(1.0f - wx) + v110*wx;
    float c11 = v011*(1.0f - wx) + v111*wx;
    float c0 = c00*(1.0f - wy) + c10*wy;
    float c1 = c01*(1.0f - wy) + c11*wy;
    return c0*(1.0f - wz) + c1*wz;
}

static float interpolate_with_rescaling(const OracleLevel *par, const float *pf_new, const float *pvel_new,
                                        int fine_gx, int fine_gy, int fine_gz, int k,
                                        float w_k, float cx, float cy, float cz,
                                        float tau_coarse, float tau_fine,
                                        float temporal_weight, int use_temporal_interp)
{
    float px_cont = ((float)fine_gx - 0.5f) * 0.5f;     /* :29-31 */
    float py_cont = ((float)fine_gy - 0.5f) * 0.5f;
    float pz_cont = ((float)fine_gz - 0.5f) * 0.5f;
    int px0 = (int)floorf(px_cont), py0 = (int)floorf(py_cont), pz0 = (int)floorf(pz_cont);
    int px1 = px0 + 1, py1 = py0 + 1, pz1 = pz0 + 1;     /* :36-38, before the clamp (Appendix A.14) */
    float wx = px_cont - (float)px0, wy = py_cont - (float)py0, wz = pz_cont - (float)pz0;
    px0 = px0 < 1 ? 1 : px0; py0 = py0 < 1 ? 1 : py0; pz0 = pz0 < 1 ? 1 : pz0;   /* :44-46 */

#define GB(a, b, c) get_blended(par, pf_new, pvel_new, a, b, c, k, w_k, temporal_weight, use_temporal_interp)
    Blend d000 = GB(px0, py0, pz0), d100 = GB(px1, py0, pz0), d010 = GB(px0, py1, pz0), d110 = GB(px1, py1, pz0);
    Blend d001 = GB(px0, py0, pz1), d101 = GB(px1, py0, pz1), d011 = GB(px0, py1, pz1), d111 = GB(px1, py1, pz1);
#undef GB
    Blend v000 = d000;                                   /* :100-107 */
    Blend v100 = d100.valid ? d100 : v000, v010 = d010.valid ? d010 : v000, v110 = d110.valid ? d110 : v000;
    Blend v001 = d001.valid ? d001 : v000, v101 = d101.valid ? d101 : v000, v011 = d011.valid ? d011 : v000;
    Blend v111 = d111.valid ? d111 : v000;

#define TL(m) trilin(v000.m, v100.m, v010.m, v110.m, v001.m, v101.m, v011.m, v111.m, wx, wy, wz)
    float f_int = TL(f), rho_int = TL(rho), ux_int = TL(ux), uy_int = TL(uy), uz_int = TL(uz);
#undef TL
    float feq_int = calculate_equilibrium(rho_int, ux_int, uy_int, uz_int, w_k, cx, cy, cz);   /* :127 */
    float f_neq = f_int - feq_int;
    float tau_c = tau_coarse - 0.5f, tau_f = tau_fine - 0.5f;
    float scale = tau_c > 1.0e-6f ? jl_clampf(tau_f / tau_c, 0.01f, 100.0f) : 1.0f;             /* :135 */
    return feq_int + f_neq * scale;
}

/* ---- one cell of stream_collide_kernel_v2!, src/physics_kernels.jl:39-357 ---- */
static void stream_collide_cell(const OracleLevel *L, const OracleLevel *par,
                                const float *parent_f, const float *parent_vel, float tau_parent,
                                float *f_out, const float *f_in, float *vel_out, const float *vel_in,
                                int x, int y, int z, int b,
                                float u_inlet, int32_t time_step_seed, float temporal_weight,
                                int nx_global, int ny_global, int nz_global, int store_post,
                                const OracleParams *p)
{
    const int64_t nb = L->n_blocks;
    const int32_t *nbt = L->neighbor_table;
    const float tau_molecular = L->tau;
    const int is_level_1 = par == NULL;

    int bx = L->map_x[b - 1], by = L->map_y[b - 1], bz = L->map_z[b - 1];     /* :43-48 */
    int gx = (bx - 1) * BS + x, gy = (by - 1) * BS + y, gz = (bz - 1) * BS + z;
    int is_obs = L->obstacle[idx4(x, y, z, b)] != 0;

    float rho = 0.0f, jx = 0.0f, jy = 0.0f, jz = 0.0f;
    float f_stored[27];

    for (int k = 0; k < 27; ++k) {                                             /* :62-149 */
        int cx = L_cx[k], cy = L_cy[k], cz = L_cz[k];
        int sx = x - cx, sy = y - cy, sz = z - cz;
        float val = 0.0f;
        if (sx >= 1 && sx <= BS && sy >= 1 && sy <= BS && sz >= 1 && sz <= BS) {
            val = f_in[idx5(sx, sy, sz, b, k, nb)];
        } else {
            int ox = sx < 1 ? -1 : (sx > BS ? 1 : 0);
            int oy = sy < 1 ? -1 : (sy > BS ? 1 : 0);
            int oz = sz < 1 ? -1 : (sz > BS ? 1 : 0);
            int dir = (ox + 1) + (oy + 1) * 3 + (oz + 1) * 9;
            int32_t nbg = nbt[(int64_t)(b - 1) + nb * dir];
            if (nbg > 0) {
                int nsx = sx < 1 ? sx + BS : (sx > BS ? sx - BS : sx);
                int nsy = sy < 1 ? sy + BS : (sy > BS ? sy - BS : sy);
                int nsz = sz < 1 ? sz + BS : (sz > BS ? sz - BS : sz);
                val = f_in[idx5(nsx, nsy, nsz, nbg, k, nb)];
            } else {
                int src_gx = gx - cx, src_gy = gy - cy, src_gz = gz - cz;       /* :88-97 */
                int is_inlet = src_gx < 1, is_outlet = src_gx > nx_global;
                int is_y_min = src_gy < 1, is_y_max = src_gy > ny_global;
                int is_z_min = src_gz < 1, is_z_max = src_gz > nz_global;
                if (is_inlet) {                                                 /* :99-104 */
                    float noise = p->inlet_turbulence > 0.0f
                        ? oracle_gradient_noise(gy, gz, time_step_seed, 1234) * p->inlet_turbulence * u_inlet
                        : 0.0f;
                    float u_inst = u_inlet + noise;
                    float cu_in = (float)cx * u_inst;
                    val = L_w[k] * (1.0f + 3.0f*cu_in + 4.5f*cu_in*cu_in - 1.5f*u_inst*u_inst);
                } else if (is_outlet) {                                         /* :106-113 */
                    float cu_out = (float)cx * u_inlet;
                    val = L_w[k] * (1.0f + 3.0f*cu_out + 4.5f*cu_out*cu_out - 1.5f*u_inlet*u_inlet);
                } else if (is_y_min && p->is_symmetric == 1) {                  /* :115-116 */
                    val = f_in[idx5(x, y, z, b, L_my[k], nb)];
                } else if (is_y_min || is_y_max) {                              /* :117-118 */
                    val = f_in[idx5(x, y, z, b, L_my[k], nb)];
                } else if (is_z_min || is_z_max) {                              /* :119-120 */
                    val = f_in[idx5(x, y, z, b, L_mz[k], nb)];
                } else if (!is_level_1) {                                       /* :122-137 */
                    val = interpolate_with_rescaling(par, parent_f, parent_vel,
                                                     src_gx, src_gy, src_gz, k, L_w[k],
                                                     (float)cx, (float)cy, (float)cz,
                                                     tau_parent, tau_molecular,
                                                     temporal_weight, p->use_temporal_interp);
                } else {
                    val = L_w[k];                                               /* :139 */
                }
            }
        }
        f_stored[k] = val;                                                      /* :144-148 */
        rho += val;
        jx += val * (float)L_cx[k];
        jy += val * (float)L_cy[k];
        jz += val * (float)L_cz[k];
    }

    if (is_obs) {                                                               /* :154-166 */
        vel_out[idx5(x, y, z, b, 0, nb)] = 0.0f;
        vel_out[idx5(x, y, z, b, 1, nb)] = 0.0f;
        vel_out[idx5(x, y, z, b, 2, nb)] = 0.0f;
        L->rho[idx4(x, y, z, b)] = 1.0f;
        for (int k = 0; k < 27; ++k) {
            float f_coll = f_stored[L_opp[k]];
            f_out[idx5(x, y, z, b, k, nb)] = f_coll;
            if (store_post) L->f_post_collision[idx5(x, y, z, b, k, nb)] = f_coll;
        }
        return;
    }

    rho = jl_maxf(rho, 0.01f);                                                  /* :172-176 */
    float inv_rho = 1.0f / rho;
    float ux = jx * inv_rho, uy = jy * inv_rho, uz = jz * inv_rho;

    float sp = L->sponge[idx4(x, y, z, b)];                                     /* :181-199 */
    if (sp > 0.0f) {
        float rho_target = 1.0f, ux_target = u_inlet;
        rho = rho * (1.0f - sp) + rho_target * sp;
        ux  = ux  * (1.0f - sp) + ux_target * sp;
        uy  = uy  * (1.0f - sp);
        uz  = uz  * (1.0f - sp);
        if (p->sponge_blend_distributions == 1) {
            for (int k = 0; k < 27; ++k) {
                float feq_target = calculate_equilibrium(rho_target, ux_target, 0.0f, 0.0f, L_w[k],
                                                         (float)L_cx[k], (float)L_cy[k], (float)L_cz[k]);
                f_stored[k] = f_stored[k] * (1.0f - sp) + feq_target * sp;
            }
        }
    }

    float Fx_wall = 0.0f, Fy_wall = 0.0f, Fz_wall = 0.0f;                       /* :202-236 */
    if (p->wall_model_active == 1) {
        float dist_wall = L->wall_dist[idx4(x, y, z, b)];
        if (dist_wall > 0.0f && dist_wall < 10.0f) {
            float u_mag = sqrtf(ux*ux + uy*uy + uz*uz);
            float nu_visc = (tau_molecular - 0.5f) / 3.0f;
            if (u_mag > 1.0e-6f && nu_visc > 1.0e-10f) {
                float u_tau = u_mag * jl_powf(nu_visc / (dist_wall * u_mag + 1.0e-10f), 1.0f/7.0f)
                                    * jl_powf(2.0f * 8.3f, -1.0f/7.0f);
                u_tau = jl_maxf(u_tau, 1.0e-6f);
                float y_p = u_tau * dist_wall / nu_visc;
                if (y_p > 11.81f) {
                    float u_plus_law = (1.0f / KAPPA) * jl_logf(y_p) + 5.2f;
                    if (u_plus_law > 0.1f) {
                        u_tau = u_tau * ((u_mag / u_tau) / u_plus_law);
                        u_tau = jl_maxf(u_tau, 1.0e-6f);
                    }
                }
                float tau_wall = rho * u_tau * u_tau;
                float tau_res  = rho * nu_visc * (u_mag / dist_wall);
                if (tau_wall > tau_res) {
                    float force_mag = (tau_wall - tau_res) / dist_wall;
                    Fx_wall = -force_mag * ux / u_mag;
                    Fy_wall = -force_mag * uy / u_mag;
                    Fz_wall = -force_mag * uz / u_mag;
                }
            }
        }
    }

    float ux_eq = ux + 0.5f * Fx_wall * inv_rho;                                /* :238-241 */
    float uy_eq = uy + 0.5f * Fy_wall * inv_rho;
    float uz_eq = uz + 0.5f * Fz_wall * inv_rho;
    float usq_eq = ux_eq*ux_eq + uy_eq*uy_eq + uz_eq*uz_eq;

    vel_out[idx5(x, y, z, b, 0, nb)] = ux;                                      /* :243-246 */
    vel_out[idx5(x, y, z, b, 1, nb)] = uy;
    vel_out[idx5(x, y, z, b, 2, nb)] = uz;
    L->rho[idx4(x, y, z, b)] = rho;

    /* velocity gradients from vel_in, src/physics_utils.jl:72-83 */
    float ux_E, uy_E, uz_E, ux_W, uy_W, uz_W, ux_N, uy_N, uz_N, ux_S, uy_S, uz_S, ux_T, uy_T, uz_T, ux_B, uy_B, uz_B;
    get_velocity_neighbor(vel_in, x, y, z, b,  1, 0, 0, nbt, nb, &ux_E, &uy_E, &uz_E);
    get_velocity_neighbor(vel_in, x, y, z, b, -1, 0, 0, nbt, nb, &ux_W, &uy_W, &uz_W);
    get_velocity_neighbor(vel_in, x, y, z, b, 0,  1, 0, nbt, nb, &ux_N, &uy_N, &uz_N);
    get_velocity_neighbor(vel_in, x, y, z, b, 0, -1, 0, nbt, nb, &ux_S, &uy_S, &uz_S);
    get_velocity_neighbor(vel_in, x, y, z, b, 0, 0,  1, nbt, nb, &ux_T, &uy_T, &uz_T);
    get_velocity_neighbor(vel_in, x, y, z, b, 0, 0, -1, nbt, nb, &ux_B, &uy_B, &uz_B);
    float g11 = 0.5f*(ux_E-ux_W), g12 = 0.5f*(ux_N-ux_S), g13 = 0.5f*(ux_T-ux_B);
    float g21 = 0.5f*(uy_E-uy_W), g22 = 0.5f*(uy_N-uy_S), g23 = 0.5f*(uy_T-uy_B);
    float g31 = 0.5f*(uz_E-uz_W), g32 = 0.5f*(uz_N-uz_S), g33 = 0.5f*(uz_T-uz_B);

    float gsq11 = g11*g11 + g12*g21 + g13*g31;                                  /* :256-264 */
    float gsq12 = g11*g12 + g12*g22 + g13*g32;
    float gsq13 = g11*g13 + g12*g23 + g13*g33;
    float gsq21 = g21*g11 + g22*g21 + g23*g31;
    float gsq22 = g21*g12 + g22*g22 + g23*g32;
    float gsq23 = g21*g13 + g22*g23 + g23*g33;
    float gsq31 = g31*g11 + g32*g21 + g33*g31;
    float gsq32 = g31*g12 + g32*g22 + g33*g32;
    float gsq33 = g31*g13 + g32*g23 + g33*g33;

    float tr_gsq = gsq11 + gsq22 + gsq33;                                       /* :266-281 */
    float tr_term = tr_gsq / 3.0f;
    float Sd11 = gsq11 - tr_term, Sd22 = gsq22 - tr_term, Sd33 = gsq33 - tr_term;
    float Sd12 = 0.5f * (gsq12 + gsq21), Sd13 = 0.5f * (gsq13 + gsq31), Sd23 = 0.5f * (gsq23 + gsq32);
    float S12 = 0.5f * (g12 + g21), S13 = 0.5f * (g13 + g31), S23 = 0.5f * (g23 + g32);
    float OP1 = Sd11*Sd11 + Sd22*Sd22 + Sd33*Sd33 + 2.0f*(Sd12*Sd12 + Sd13*Sd13 + Sd23*Sd23);
    float OP2 = g11*g11 + g22*g22 + g33*g33 + 2.0f*(S12*S12 + S13*S13 + S23*S23);

    float nu_eddy = 0.0f;                                                       /* :283-291 */
    if (OP1 > 1.0e-12f) {
        float OP1_32 = OP1 * sqrtf(OP1);
        float OP2_52 = OP2 * OP2 * sqrtf(jl_maxf(OP2, 1.0e-12f));
        float denom = OP2_52 + OP1 * sqrtf(sqrtf(jl_maxf(OP1, 1.0e-12f)));
        if (denom > 1.0e-12f) nu_eddy = (p->c_wale * p->c_wale) * OP1_32 / denom;
    }
    nu_eddy = jl_maxf(nu_eddy, p->nu_sgs_background);                           /* :297 */
    float tau_turb = tau_molecular + nu_eddy * 3.0f;                            /* :299-300 */
    float omega = 1.0f / jl_maxf(tau_turb, 0.500001f);

    float Pi_xx = 0.0f, Pi_yy = 0.0f, Pi_zz = 0.0f, Pi_xy = 0.0f, Pi_yz = 0.0f, Pi_zx = 0.0f;
    for (int k = 0; k < 27; ++k) {                                              /* :308-322 */
        float cx_f = (float)L_cx[k], cy_f = (float)L_cy[k], cz_f = (float)L_cz[k];
        float cu = cx_f*ux_eq + cy_f*uy_eq + cz_f*uz_eq;
        float feq = rho * L_w[k] * (1.0f + 3.0f*cu + 4.5f*cu*cu - 1.5f*usq_eq);
        float f_neq = f_stored[k] - feq;
        Pi_xx += f_neq * cx_f * cx_f;
        Pi_yy += f_neq * cy_f * cy_f;
        Pi_zz += f_neq * cz_f * cz_f;
        Pi_xy += f_neq * cx_f * cy_f;
        Pi_yz += f_neq * cy_f * cz_f;
        Pi_zx += f_neq * cz_f * cx_f;
    }
    for (int k = 0; k < 27; ++k) {                                              /* :324-354 */
        float cx_f = (float)L_cx[k], cy_f = (float)L_cy[k], cz_f = (float)L_cz[k];
        float w_k = L_w[k];
        float cu = cx_f*ux_eq + cy_f*uy_eq + cz_f*uz_eq;
        float feq = rho * w_k * (1.0f + 3.0f*cu + 4.5f*cu*cu - 1.5f*usq_eq);
        float force_term = w_k * 3.0f * (
            (cx_f - ux + 3.0f*cu*cx_f) * Fx_wall +
            (cy_f - uy + 3.0f*cu*cy_f) * Fy_wall +
            (cz_f - uz + 3.0f*cu*cz_f) * Fz_wall);
        float Q_xx = cx_f*cx_f - CS2_PHYSICS;
        float Q_yy = cy_f*cy_f - CS2_PHYSICS;
        float Q_zz = cz_f*cz_f - CS2_PHYSICS;
        float f_neq_reg = w_k * 4.5f * (
            Pi_xx * Q_xx + Pi_yy * Q_yy + Pi_zz * Q_zz +
            2.0f * (Pi_xy * cx_f*cy_f + Pi_yz * cy_f*cz_f + Pi_zx * cz_f*cx_f));
        float f_coll = feq + (1.0f - omega) * f_neq_reg + (1.0f - 0.5f*omega) * force_term;
        if (store_post) L->f_post_collision[idx5(x, y, z, b, k, nb)] = f_coll;
        f_out[idx5(x, y, z, b, k, nb)] = f_coll;
    }
}

/* launch of stream_collide_kernel_v2!, src/physics_v2.jl:43-83 */
void oracle_stream_collide(const OracleLevel *level, const OracleLevel *parent,
                           const float *parent_f, const float *parent_vel, float tau_parent,
                           float *f_out, const float *f_in, float *vel_out, const float *vel_in,
                           float u_inlet, int64_t timestep, float temporal_weight,
                           const OracleParams *p)
{
    lattice_init();
    if (level->n_blocks == 0) return;
    int scale = 1 << (level->level_id - 1);                                     /* physics_v2.jl:55-56 */
    int nx_g = p->domain_nx * scale, ny_g = p->domain_ny * scale, nz_g = p->domain_nz * scale;
    int32_t seed = (int32_t)(timestep % 1000000);                               /* :76 */
    int store_post = (level->bouzidi_enabled && level->n_boundary_cells > 0) ? 1 : 0;   /* :77 */
    const int nb = level->n_blocks;
#pragma omp parallel for schedule(static)
    for (int b = 1; b <= nb; ++b)
        for (int z = 1; z <= BS; ++z)
            for (int y = 1; y <= BS; ++y)
                for (int x = 1; x <= BS; ++x)
                    stream_collide_cell(level, parent, parent_f, parent_vel, tau_parent,
                                        f_out, f_in, vel_out, vel_in, x, y, z, b,
                                        u_inlet, seed, temporal_weight, nx_g, ny_g, nz_g, store_post, p);
}

/* bouzidi_correction_kernel_fixed!, src/bouzidi_kernel.jl:13-92 */
void oracle_bouzidi_correction(const OracleLevel *L, float *f_out, float q_min_threshold)
{
    lattice_init();
    const int64_t nb = L->n_blocks;
    const float *f_post = L->f_post_collision;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < L->n_boundary_cells; ++c) {
        int b = L->bouzidi_cell_block[c];
        int x = L->bouzidi_cell_x[c], y = L->bouzidi_cell_y[c], z = L->bouzidi_cell_z[c];
        for (int k = 0; k < 27; ++k) {
            float q = oracle_half_to_float(L->bouzidi_q_map[idx5(x, y, z, b, k, nb)]);
            if (q > q_min_threshold && q <= 1.0f) {
                int opp_k = L_opp[k];
                float f_k = f_post[idx5(x, y, z, b, k, nb)];
                if (q < 0.5f) {
                    int nx = x + L_cx[opp_k], ny = y + L_cy[opp_k], nz = z + L_cz[opp_k];
                    float f_ff = f_k;
                    if (nx >= 1 && nx <= BS && ny >= 1 && ny <= BS && nz >= 1 && nz <= BS) {
                        f_ff = f_post[idx5(nx, ny, nz, b, k, nb)];
                    } else {
                        int ox = nx < 1 ? -1 : (nx > BS ? 1 : 0);
                        int oy = ny < 1 ? -1 : (ny > BS ? 1 : 0);
                        int oz = nz < 1 ? -1 : (nz > BS ? 1 : 0);
                        int dir = (ox + 1) + (oy + 1) * 3 + (oz + 1) * 9;
                        int32_t nbb = L->neighbor_table[(int64_t)(b - 1) + nb * dir];
                        if (nbb > 0) {
                            int nnx = nx < 1 ? nx + BS : (nx > BS ? nx - BS : nx);
                            int nny = ny < 1 ? ny + BS : (ny > BS ? ny - BS : ny);
                            int nnz = nz < 1 ? nz + BS : (nz > BS ? nz - BS : nz);
                            f_ff = f_post[idx5(nnx, nny, nnz, nbb, k, nb)];
                        }
                    }
                    float coeff1 = 2.0f * q;
                    f_out[idx5(x, y, z, b, opp_k, nb)] = coeff1 * f_k + (1.0f - coeff1) * f_ff;
                } else {
                    float f_opp_post = f_post[idx5(x, y, z, b, opp_k, nb)];
                    float inv_2q = 1.0f / (2.0f * q);
                    float coeff2 = (2.0f * q - 1.0f) * inv_2q;
                    f_out[idx5(x, y, z, b, opp_k, nb)] = inv_2q * f_k + coeff2 * f_opp_post;
                }
            }
        }
    }
}

/* perform_timestep_v2!, src/physics_v2.jl:26-97 */
void oracle_perform_timestep(const OracleLevel *level, const OracleLevel *parent,
                             const float *parent_f, const float *parent_vel, float tau_parent,
                             float *f_out, const float *f_in, float *vel_out, const float *vel_in,
                             float u_inlet, int64_t timestep, float temporal_weight,
                             const OracleParams *p)
{
    if (level->n_blocks == 0) return;
    oracle_stream_collide(level, parent, parent_f, parent_vel, tau_parent, f_out, f_in, vel_out, vel_in,
                          u_inlet, timestep, temporal_weight, p);
    if (level->bouzidi_enabled && level->n_boundary_cells > 0)
        oracle_bouzidi_correction(level, f_out, p->q_min_threshold);
}

/* copy_to_old!, src/blocks.jl:199-205 */
static void copy_to_old(OracleLevel *L, const float *f_cur, const float *vel_cur)
{
    if (!L->has_temporal_storage) return;
    size_t n = (size_t)L->n_blocks * CELLS;
    memcpy(L->f_old, f_cur, n * 27 * sizeof(float));
    memcpy(L->rho_old, L->rho, n * sizeof(float));
    memcpy(L->vel_old, vel_cur, n * 3 * sizeof(float));
}

/* recursive_step! / recursive_step_temporal!, src/solver_control.jl:21-143 (identical bodies
 * apart from the temporal weight, which recursive_step! fixes at 0.0f0). */
static void recursive_step(OracleLevel *levels, int n_levels, int lvl /*1-based*/, int64_t t_sub,
                           const OracleLevel *parent, const float *parent_f, const float *parent_vel,
                           float parent_tau, float temporal_weight, float u_vel, const OracleParams *p)
{
    if (lvl > n_levels) return;
    OracleLevel *L = &levels[lvl - 1];
    float *f_in, *f_out, *vel_in, *vel_out;
    if ((t_sub % 2) == 0) { f_in = L->f;      f_out = L->f_temp; vel_in = L->vel;      vel_out = L->vel_temp; }
    else                  { f_in = L->f_temp; f_out = L->f;      vel_in = L->vel_temp; vel_out = L->vel; }
    int has_children = lvl < n_levels;
    if (has_children && p->use_temporal_interp && L->has_temporal_storage) copy_to_old(L, f_in, vel_in);
    oracle_perform_timestep(L, parent, parent_f, parent_vel, parent_tau, f_out, f_in, vel_out, vel_in,
                            u_vel, t_sub, temporal_weight, p);
    if (has_children) {
        recursive_step(levels, n_levels, lvl + 1, 2 * t_sub,     L, f_out, vel_out, L->tau, 0.0f, u_vel, p);
        recursive_step(levels, n_levels, lvl + 1, 2 * t_sub + 1, L, f_out, vel_out, L->tau, 0.5f, u_vel, p);
    }
}

void oracle_execute_timestep_batch(OracleLevel *levels, int32_t n_levels, int64_t t_start,
                                   int32_t batch_size, float u_curr, const OracleParams *p)
{
    lattice_init();
    for (int t_off = 0; t_off < batch_size; ++t_off)
        recursive_step(levels, n_levels, 1, t_start + t_off, NULL, NULL, NULL, 0.5f, 0.0f, u_curr, p);
}

/* init_eq!, src/main.jl:109-134 */
void oracle_init_equilibrium(OracleLevel *L)
{
    lattice_init();
    size_t n = (size_t)L->n_blocks * CELLS;
    for (int k = 0; k < 27; ++k)
        for (size_t i = 0; i < n; ++i) {
            L->f[n * k + i] = L_w[k];
            L->f_temp[n * k + i] = L_w[k];
            if (L->has_temporal_storage) L->f_old[n * k + i] = L_w[k];
        }
    if (L->has_temporal_storage) {
        for (size_t i = 0; i < n; ++i) L->rho_old[i] = 1.0f;
        memset(L->vel_old, 0, n * 3 * sizeof(float));
    }
}

/* src/main.jl:173 */
float oracle_ramp_progress(int64_t batch_end, int64_t ramp_steps)
{
    if (batch_end <= ramp_steps) {
        float arg = 3.14159265358979323846f * (float)batch_end / (float)ramp_steps;
        return 0.5f * (1.0f - jl_cosf(arg));
    }
    return 1.0f;
}
