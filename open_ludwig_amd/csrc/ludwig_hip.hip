// libludwig_hip.so - C ABI (include/ludwig_hip.h) over the gfx950 kernels in kernels.hpp.
// Host side only: memory ownership, block classification, work lists, launches.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>
#include <string>
#include <vector>

#include "../../include/ludwig_hip.h"
#include "kernels.hpp"

using namespace lw;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define LW_HIP(call)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (call);                                                                           \
        if (e_ != hipSuccess) return fail(LUDWIG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr int N_PARTS = 3, N_CLASSES = 3;   // class 1 = blocks with a missing neighbour (GENERAL instantiation), 2 = all-neighbour blocks; 0 unused
constexpr int XRUN = 4, XRUN_MAX = 4;       // waves per workgroup = blocks of an x-run (8 and 16 were tried in rounds 1-2: slower)

}  // namespace

struct LudwigLevel {
    int device = 0;
    hipStream_t stream = nullptr;
    int level_id = 1, n_blocks = 0, n_owned = 0;
    float tau = 1.0f;
    int gdx = 0, gdy = 0, gdz = 0;
    // Storage. The caller's arrays are [8,8,8,n_blocks,K], population-major (reference src/blocks.jl:118-150); the device arrays are
    // BLOCK-major: element (cell, block b, component k) at ((b * K + k) * 512 + cell) - the 27 populations of a block are one
    // contiguous 54-KiB piece. With the reference's layout the step is 27 + 27 concurrent streams n_blocks x 2 KiB apart, and at
    // some distances - 64.5, 65.75, 68, 76-77 ... MiB for boxes around 256^3, far denser for the small levels of a nested case -
    // they load MI355X's memory system unevenly: 5-30 % of the step, decided by nothing but n_blocks (profiles/r03_stride_*).
    // Block-major storage has no such distance, and the step's data movement is 5-12 % faster than at the best stride.
    // Translated at the ABI like the block order; only raw pointers see it (ludwig_level_field_layout).
    int64_t sk = 0;        // cells of the level = 512 * n_blocks (the reference's population stride; here only a count)
    bool wide = false;     // 64-bit per-lane addresses: the level's f array is 4 GiB or more (>= 77 672 blocks), or LUDWIG_WIDE_ADDR
    float *f[2] = {nullptr, nullptr};      // f, f_temp
    float *vel[2] = {nullptr, nullptr};    // vel, vel_temp
    float *rho = nullptr, *f_post = nullptr, *f_old = nullptr, *rho_old = nullptr, *vel_old = nullptr;
    uint8_t *obstacle = nullptr;
    float *sponge = nullptr, *wall_dist = nullptr;
    int32_t *meta = nullptr;
    bool has_temporal = false, has_post = false, bouzidi_enabled = false;
    // copy_to_old! without the copies: a pull step never writes its input buffers, so after ludwig_save_old(t_sub) the saved
    // f / vel ARE f[in] / vel[in] until something else writes that buffer. old_alias = that buffer index, or -1 when the
    // saved state lives in f_old / vel_old (materialize_old() copies it there before any such write). rho is copied.
    int old_alias = -1;
    // ludwig_level_field_ptr handed out a WRITABLE device pointer (f, f_temp, vel, vel_temp, rho, ...): from then on the caller
    // may write into the level at any time without the library seeing it, so the three shortcuts that rely on seeing every
    // write are off for this level - saved state is copied (no aliasing), its children's interface values are not computed
    // ahead, rho is stored by every step.
    bool external_writer = false;
    int n_bc = 0;
    int4 *bouzidi_links = nullptr;      // every listed link with q > 0 (the q map is static): kernels.hpp BouzidiParams::links
    int n_bouzidi_links = 0;
    uint32_t *post_rows = nullptr;      // [n_blocks][2] bits: the x-rows f_post_collision is read at by the links with q > 0 (SCParams::post_rows)
    std::vector<uint32_t> h_post_rows;  // host copy (ludwig_level_add_post_collision_readers ORs into it)
    int post_mode = 0;                  // 0: rows / blocks with a reader; 1: every block because asked to (flag, environment);
                                        // 2: every block because no cell list is owned and nobody has named the readers yet
    bool post_rows_used = false;        // the last stream-collide stored f_post_collision by rows (valid for q_min >= 0 only)
    _Float16 *q_map = nullptr;
    int32_t *cell_block = nullptr;
    int8_t *cell_x = nullptr, *cell_y = nullptr, *cell_z = nullptr;
    // Block order. The caller's arrays keep the reference's block order (sort of (bx,by,bz) tuples, src/domain.jl:171: bz fastest
    // in memory, x neighbours 2 MB apart at 256^3). The device arrays hold the blocks in the library's own order - owned blocks
    // first, each group sorted with bx FASTEST - so that the four x-consecutive blocks an x-run workgroup steps are consecutive
    // in memory and a sweep "x, then y, then z" reads and writes every population stream sequentially while x / y neighbours stay
    // close in time in one L2 (the block order's DRAM floor AND the x-run order's halo hits). Everything inside the
    // library uses internal ids; the ABI translates: create (tables, geometry, Bouzidi lists), upload / download (a block-permuting
    // copy through `scratch`, one population at a time), halo pack / unpack (offsets translated in the kernel), set_order.
    std::vector<int32_t> ref2int, int2ref;      // empty = the reference order is kept (LUDWIG_REFERENCE_BLOCK_ORDER, or <= 1 block)
    bool x_fastest_memory = false;              // known while the default launch order is built (before ref2int is attached)
    int32_t *d_ref2int = nullptr;
    void *scratch = nullptr;                    // one population in the reference order (sk floats), allocated on first use
    std::vector<int32_t> h_meta;
    std::vector<uint8_t> h_comm_boundary;
    int32_t *items[N_PARTS][N_CLASSES] = {};
    int64_t n_items[N_PARTS][N_CLASSES] = {};
    // small levels step both kinds of blocks in ONE launch (merge_classes): the all-neighbour workgroups followed by the general ones,
    // through the GENERAL instantiation. The separate lists stay for the rho replay and for levels that are not merged.
    int32_t *items_merged[N_PARTS] = {};
    int64_t n_items_merged[N_PARTS] = {};
    int n_fast_blocks = 0;
    bool general_in_runs[N_PARTS] = {};     // class-1 items are in x-run format (link bits, XRUN waves per workgroup)
    int64_t n_linked_items[N_PARTS] = {};   // class-2 waves that exchange a face column with a neighbouring wave
    int64_t device_bytes = 0;
    // coarse -> fine interface pass (levels >= 2): links per part, built lazily for the global box of the first step
    int n_iface_blocks = 0;
    float *f_iface = nullptr;
    int4 *links[N_PARTS] = {};           // per link: (block << 9) | cell, (gbi << 5) | k, source index
    int4 *sources[N_PARTS] = {};        // per source cell: 8 parent-cell offsets of its trilinear stencil (-1 = absent), 2 x int4
    float4 *source_w[N_PARTS] = {};     // per source cell: interpolation weights wx, wy, wz
    float4 *source_mac[N_PARTS] = {};   // per source cell, rewritten every pass: interpolated rho, ux, uy, uz
    float4 *source_mac2[N_PARTS] = {};  // the same for the speculated weight of the next sub-step
    float *f_iface2 = nullptr;
    // Interface values computed ahead for the second sub-step of a pair (reference src/solver_control.jl:63-83: the child
    // steps 2t with weight 0.0 and 2t+1 with 0.5 against the same parent buffers): valid while the parent was not written.
    struct IfaceAhead {
        bool valid = false;
        const LudwigLevel *parent = nullptr;
        uint64_t parent_version = 0;
        int64_t t_sub = 0;            // the sub-step the speculated values are for
        float tw = 0.0f, tau_parent = 0.0f;
        int use_temporal = 0;
    } ahead[N_PARTS];
    // Interface values produced BEFORE the step that uses them (interface_prepass: level streams run a parent level's interface
    // pass ahead of its wait for the children, see recursive_step): same keys as IfaceAhead, values in f_iface.
    IfaceAhead prepared[N_PARTS];
    // Lazy rho. The step writes `rho` (4 of 244 B per cell update) for readers that mostly are not there: the next step
    // overwrites it unread unless a child level interpolates from it (every step), or a diagnostic / download / save asks
    // for it (now and then). A level nobody has read `rho` of in between therefore skips the store and remembers the
    // launch (rho_replay); ensure_rho() reproduces the array on demand with the RHO_ONLY instantiation, from the same
    // input buffers (a pull step never writes its inputs) - bit for bit what the step would have stored.
    // Only whole-level launches (LUDWIG_PART_ALL) elide the store. Boundary / interior part launches - the multi-GPU schedule, where
    // the next step's interior blocks start before this step's halo has landed - always store: a part's elided rho could no longer
    // be reproduced once another part's launch has reused its input buffer.
    bool rho_eager = false;             // a child reads rho after every step (or LUDWIG_EAGER_RHO): always store
    struct RhoReplay {
        bool stale = false;             // the launch left rho unwritten
        int64_t t_sub = -1;
        SCParams p;
    } rho_replay[N_PARTS];              // only [LUDWIG_PART_ALL] is ever stale
    int64_t step_count = 0, last_step_t = -1, last_replay_step = -10;   // a level asked for rho after two steps in a row turns eager
    // ludwig_execute_timestep_batch runs every level on a stream of its own (level_streams below): events that order them
    hipStream_t own_stream = nullptr;
    hipEvent_t parent_wait = nullptr;   // set by recursive_step: the parent's step this sub-step's interface pass has to wait for
    // rho_old in the step: copy_to_old!'s copy of rho (the one field a step overwrites in place) costs a launch and two transitions on
    // a parent level's stream per step. The batch driver leaves it to the step that follows: every cell saves its old rho before it
    // stores the new one (kernels.hpp rho_old_save). Only whole-level launches of a level without ghost blocks; anything that
    // would read rho_old or write rho in between performs the copy first (resolve_rho_old).
    bool rho_old_pending = false;
    // Parent-side interface pass (level streams, recursive_step): the PARENT's stream computes this level's interface values for
    // the pair of sub-steps 2t, 2t + 1 right after the parent's step t, into set t & 1 of a second pair of side buffers - while this
    // level is still stepping pair t - 1 out of the other set. This level's own stream then carries no interface kernels at all.
    float *f_iface_b = nullptr, *f_iface2_b = nullptr;             // set 1 (set 0 = f_iface, f_iface2)
    float4 *mac_b = nullptr, *mac2_b = nullptr;                    // scratch of set 1's pass (LUDWIG_PART_ALL)
    struct PairReady {
        bool valid = false;
        const LudwigLevel *parent = nullptr;
        uint64_t parent_version = 0;
        int64_t pair = -1;                                         // t_sub >> 1 of the sub-steps the values are for
        float tau_parent = 0.0f;
        int use_temporal = 0, set = 0;
    } pair_ready;
    hipEvent_t ev_pair_done[2] = {nullptr, nullptr};               // recorded on THIS level's stream after the second sub-step of a pair
    bool pair_done_set[2] = {false, false};
    hipEvent_t ev_stepped = nullptr;    // recorded on this level's stream after each of its steps (collision + Bouzidi)
    hipEvent_t ev_consumed = nullptr;   // recorded on the CHILD's stream once its interface pass has read this level's buffers
    bool ev_consumed_set = false;
    uint64_t stepped_gen = 0;           // how often ev_stepped has been recorded
    const LudwigLevel *waited_parent = nullptr;   // the parent step this level's stream has already been made to wait for
    uint64_t waited_gen = 0;
    uint64_t version = 0;               // bumped by everything that writes this level's fields
    const LudwigLevel *iface_parent = nullptr;
    std::vector<int32_t> h_block_pointer;   // [gdx,gdy,gdz] 1-based, 0 = absent (src/blocks.jl:111-114)
    int n_links[N_PARTS] = {};
    int n_sources[N_PARTS] = {};
    int iface_dims[3] = {-1, -1, -1};
};

namespace {

template <class T>
int dev_alloc(LudwigLevel *L, T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) return LUDWIG_OK;
    hipError_t e = hipMalloc((void **)p, count * sizeof(T));
    if (e != hipSuccess) return fail(LUDWIG_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    L->device_bytes += (int64_t)(count * sizeof(T));
    return LUDWIG_OK;
}

int fill(LudwigLevel *L, float *a, int64_t n, float v)
{
    if (n == 0) return LUDWIG_OK;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, L->stream, a, n, v);
    LW_HIP(hipGetLastError());
    return LUDWIG_OK;
}

struct FieldDesc {
    void *ptr;
    size_t bytes;
    int comps = 1;     // K: components per cell
    int es = 4;        // element size in bytes
};

int ensure_rho(LudwigLevel *L);

int materialize_old(LudwigLevel *L)
{
    if (L->old_alias < 0) return LUDWIG_OK;
    ++L->version;
    const size_t c = (size_t)L->sk;
    const int a = L->old_alias;
    L->old_alias = -1;
    LW_HIP(hipSetDevice(L->device));
    LW_HIP(hipMemcpyAsync(L->f_old, L->f[a], c * Q * 4, hipMemcpyDeviceToDevice, L->stream));
    LW_HIP(hipMemcpyAsync(L->vel_old, L->vel[a], c * 3 * 4, hipMemcpyDeviceToDevice, L->stream));
    return LUDWIG_OK;
}

// call before anything but a stream-collide step writes `field` (upload, halo unpack, a raw pointer handed out)
int before_external_write(LudwigLevel *L, int field)
{
    // an elided rho is recomputed from the input populations / sponge / obstacle of the last step: produce it before any
    // of them changes (a halo unpack into the OUTPUT buffer of that step, the usual case, does not touch them)
    bool feeds_rho = field == LUDWIG_SPONGE || field == LUDWIG_OBSTACLE || field == LUDWIG_RHO;
    for (int a = 0; a < N_PARTS; ++a)
        if (L->rho_replay[a].stale && (field == LUDWIG_F || field == LUDWIG_F_TEMP) && L->f[field == LUDWIG_F ? 0 : 1] == L->rho_replay[a].p.f_in) feeds_rho = true;
    if (feeds_rho) {
        const int r = ensure_rho(L);
        if (r) return r;
    }
    if (L->old_alias < 0) return LUDWIG_OK;
    const int a = L->old_alias;
    const bool hits = field == LUDWIG_F_OLD || field == LUDWIG_VEL_OLD || (field == LUDWIG_F && a == 0) || (field == LUDWIG_F_TEMP && a == 1) ||
                      (field == LUDWIG_VEL && a == 0) || (field == LUDWIG_VEL_TEMP && a == 1);
    return hits ? materialize_old(L) : LUDWIG_OK;
}

FieldDesc field_desc(const LudwigLevel *L, int field)
{
    const size_t c = (size_t)L->sk;
    if (L->old_alias >= 0) {          // readers of the saved state follow the alias
        if (field == LUDWIG_F_OLD) return {L->f[L->old_alias], L->has_temporal ? c * Q * 4 : 0, Q, 4};
        if (field == LUDWIG_VEL_OLD) return {L->vel[L->old_alias], L->has_temporal ? c * 3 * 4 : 0, 3, 4};
    }
    switch (field) {
    case LUDWIG_F: return {L->f[0], c * Q * 4, Q, 4};
    case LUDWIG_F_TEMP: return {L->f[1], c * Q * 4, Q, 4};
    case LUDWIG_F_POST: return {L->f_post, L->has_post ? c * Q * 4 : 0, Q, 4};
    case LUDWIG_F_OLD: return {L->f_old, L->has_temporal ? c * Q * 4 : 0, Q, 4};
    case LUDWIG_RHO: return {L->rho, c * 4, 1, 4};
    case LUDWIG_RHO_OLD: return {L->rho_old, L->has_temporal ? c * 4 : 0, 1, 4};
    case LUDWIG_VEL: return {L->vel[0], c * 3 * 4, 3, 4};
    case LUDWIG_VEL_TEMP: return {L->vel[1], c * 3 * 4, 3, 4};
    case LUDWIG_VEL_OLD: return {L->vel_old, L->has_temporal ? c * 3 * 4 : 0, 3, 4};
    case LUDWIG_OBSTACLE: return {L->obstacle, c, 1, 1};
    case LUDWIG_SPONGE: return {L->sponge, c * 4, 1, 4};
    case LUDWIG_WALL_DIST: return {L->wall_dist, c * 4, 1, 4};
    default: return {nullptr, 0, 1, 4};
    }
}

// per-block flags that let the kernel skip loads: recomputed whenever the host hands us the field
void scan_flags(LudwigLevel *L, int field, const void *host)
{
    const int flag = field == LUDWIG_OBSTACLE ? FLAG_HAS_OBSTACLE : field == LUDWIG_SPONGE ? FLAG_HAS_SPONGE : FLAG_HAS_NEAR_WALL;
    for (int b = 0; b < L->n_blocks; ++b) {
        bool any = false;
        if (host) {
            if (field == LUDWIG_OBSTACLE) {
                const uint8_t *o = (const uint8_t *)host + (size_t)b * CELLS;
                for (int i = 0; i < CELLS && !any; ++i) any = o[i] != 0;
            } else if (field == LUDWIG_SPONGE) {
                const float *s = (const float *)host + (size_t)b * CELLS;
                for (int i = 0; i < CELLS && !any; ++i) any = s[i] > 0.0f;
            } else {
                const float *d = (const float *)host + (size_t)b * CELLS;
                for (int i = 0; i < CELLS && !any; ++i) any = d[i] > 0.0f && d[i] < 10.0f;
            }
        }
        int32_t &fl = L->h_meta[(size_t)b * NBR_STRIDE + NBR_FLAGS];
        fl = any ? (fl | flag) : (fl & ~flag);
    }
}

int upload_meta(LudwigLevel *L)
{
    if (L->n_blocks == 0) return LUDWIG_OK;
    LW_HIP(hipMemcpyAsync(L->meta, L->h_meta.data(), L->h_meta.size() * 4, hipMemcpyHostToDevice, L->stream));
    LW_HIP(hipStreamSynchronize(L->stream));
    return LUDWIG_OK;
}

bool block_in_part(const LudwigLevel *L, int b, int part)
{
    if (b >= L->n_owned) return false;
    if (part == LUDWIG_PART_ALL) return true;
    const bool bnd = !L->h_comm_boundary.empty() && L->h_comm_boundary[b] != 0;
    return part == LUDWIG_PART_BOUNDARY ? bnd : !bnd;
}

// LUDWIG_MERGE_CLASSES: 1 = always one launch per pass, 0 = never, unset = levels below MERGE_BELOW_BLOCKS owned blocks
constexpr int MERGE_BELOW_BLOCKS = 8192;      // 4.2 M cells: above that a pass is > 0.2 ms and the extra launch is noise
bool merge_classes(const LudwigLevel *L, const std::vector<int32_t> &general, const std::vector<int32_t> &fast)
{
    if (general.empty() || fast.empty()) return false;
    const char *e = getenv("LUDWIG_MERGE_CLASSES");
    if (e) return atoi(e) != 0;
    return L->n_owned < MERGE_BELOW_BLOCKS;
}

int set_items(LudwigLevel *L, int part, const int32_t *items, int64_t n)
{
    // one item per wave: (block << 3) | z, or -1 = idle wave. XRUN consecutive items form one workgroup.
    // class 2: all-neighbour blocks, class 1: blocks with a missing neighbour - both through the x-run kernel;
    // class 0 (wave-by-wave kernel) only with LUDWIG_NO_XRUN (diagnostics).
    std::vector<int32_t> cls[N_CLASSES];
    auto is_fast = [&](int b) { return (L->h_meta[(size_t)b * NBR_STRIDE + NBR_FLAGS] & FLAG_ALL_NEIGHBOURS) != 0; };
    for (int64_t i = 0; i < n; ++i) {
        if (items[i] >= 0) {
            const int b = items[i] >> 3;
            if (b < 0 || b >= L->n_owned) return fail(LUDWIG_ERR_INVALID, "work item %lld: block %d is not an owned block", (long long)i, b);
            if (!block_in_part(L, b, part)) return fail(LUDWIG_ERR_INVALID, "work item %lld: block %d is not in part %d", (long long)i, b, part);
        }
    }
    const bool use_xrun = true;
    L->general_in_runs[part] = true;
    {
        // All-neighbour blocks -> x-run kernel (class 2), XRUN waves per workgroup; a wave is LINKED to the next one when
        // that one holds its +x neighbour block at the same plane (then the face column travels through LDS). Workgroups
        // of the caller's order that hold only such blocks keep their composition (and with it their XCD slot); loose
        // all-neighbour items of mixed workgroups are re-packed at the end. Blocks with a missing neighbour -> class 1.
        std::vector<int32_t> loose[2];                     // [0] all-neighbour, [1] general items of mixed workgroups
        for (int64_t g = 0; g < n; g += XRUN) {
            const int64_t m = std::min<int64_t>(XRUN, n - g);
            int n_fast = 0, n_gen = 0;
            for (int64_t w = 0; w < m; ++w)
                if (items[g + w] >= 0) (is_fast(items[g + w] >> 3) ? n_fast : n_gen)++;
            const bool pure = m == XRUN && (n_fast == 0 || n_gen == 0);
            for (int64_t w = 0; w < m; ++w) {
                const int32_t it = items[g + w];
                if (pure) { if (n_fast + n_gen > 0) cls[n_gen > 0 ? 1 : 2].push_back(it); }
                else if (it >= 0) loose[is_fast(it >> 3) ? 0 : 1].push_back(it);
            }
        }
        for (int c = 1; c <= 2; ++c) {
            const std::vector<int32_t> &lo = loose[c == 2 ? 0 : 1];
            cls[c].insert(cls[c].end(), lo.begin(), lo.end());
            while (cls[c].size() % XRUN) cls[c].push_back(-1);
            for (size_t g = 0; g < cls[c].size(); g += XRUN)
                for (int w = 0; w + 1 < XRUN; ++w) {
                    const int32_t a = cls[c][g + w], d = cls[c][g + w + 1];
                    if (a < 0 || d < 0 || (a & 7) != (d & 7)) continue;
                    const int ba = (a & ITEM_ID_MASK) >> 3, bd = d >> 3;     // `a` may already carry its west link
                    if (ba == bd || L->h_meta[(size_t)ba * NBR_STRIDE + DIR(1, 0, 0)] != bd) continue;
                    cls[c][g + w] |= ITEM_LINK_E;
                    cls[c][g + w + 1] |= ITEM_LINK_W;
                }
        }
        L->n_linked_items[part] = 0;
        for (int32_t it : cls[2])
            if (it >= 0 && (it & (ITEM_LINK_E | ITEM_LINK_W))) ++L->n_linked_items[part];
    }
    std::vector<int32_t> merged;
    if (use_xrun && merge_classes(L, cls[1], cls[2])) {
        // one launch for the whole pass: the all-neighbour workgroups ride along in the GENERAL instantiation (its patch
        // phase finds nothing to do for them) - on small levels a launch boundary costs more than that
        merged = cls[2];
        merged.insert(merged.end(), cls[1].begin(), cls[1].end());
    }
    if (L->items_merged[part]) { (void)hipFree(L->items_merged[part]); L->items_merged[part] = nullptr; }
    L->n_items_merged[part] = (int64_t)merged.size();
    if (!merged.empty()) {
        LW_HIP(hipMalloc((void **)&L->items_merged[part], merged.size() * 4));
        LW_HIP(hipMemcpy(L->items_merged[part], merged.data(), merged.size() * 4, hipMemcpyHostToDevice));
    }
    for (int c = 0; c < N_CLASSES; ++c) {
        bool any = false;
        for (int32_t it : cls[c]) any = any || it >= 0;
        if (!any) cls[c].clear();
        while (cls[c].size() % XRUN) cls[c].push_back(-1);
        if (L->items[part][c]) { (void)hipFree(L->items[part][c]); L->items[part][c] = nullptr; }
        L->n_items[part][c] = (int64_t)cls[c].size();
        if (!cls[c].empty()) {
            LW_HIP(hipMalloc((void **)&L->items[part][c], cls[c].size() * 4));
            LW_HIP(hipMemcpy(L->items[part][c], cls[c].data(), cls[c].size() * 4, hipMemcpyHostToDevice));
        }
    }
    return LUDWIG_OK;
}

// Default launch order: x-runs, plane-per-XCD.
// A cache line of population k, z-plane p of a block is read only by waves working on plane p + cz(k): in the
// owning block and in its x/y neighbours (faces, edges); nothing is shared across different plane indices.
//  * workgroup = the same plane of 4 x-consecutive all-neighbour blocks -> the x-run kernel: aligned loads only,
//    x-face columns handed over in LDS (kernels.hpp). Runs are cut greedily along every (by,bz) row of blocks.
//  * MI355X deals workgroup g to XCD g % 8 (private L2 each): the 8 planes of a group occupy 8 consecutive slots,
//    plane z on XCD (z + bz) % 8, so a line is fetched by one XCD only; groups are swept y-fastest, then x, then z.
//  * chains shorter than a workgroup (leftovers, lone blocks) are packed several to a workgroup, same slot rule; a work
//    item's link bits say which lateral faces arrive through LDS. Blocks with a missing neighbour (domain edges,
//    refinement interfaces) are chained the same way among themselves and take the GENERAL instantiation.
// Measured at 256^3 against the alternatives with tools/order_sweep.py (DESIGN.md "Launch order").
int default_items(LudwigLevel *L, int part)
{
    struct Blk { int32_t bx, by, bz, b; };
    std::vector<Blk> blks;
    for (int b = 0; b < L->n_owned; ++b) {
        if (!block_in_part(L, b, part)) continue;
        const int32_t *row = &L->h_meta[(size_t)b * NBR_STRIDE];
        blks.push_back({row[NBR_BX], row[NBR_BY], row[NBR_BZ], b});
    }
    std::sort(blks.begin(), blks.end(), [](const Blk &a, const Blk &c) {
        if (a.by != c.by) return a.by < c.by;
        if (a.bz != c.bz) return a.bz < c.bz;
        return a.bx < c.bx;
    });
    auto fast = [&](int b) { return (L->h_meta[(size_t)b * NBR_STRIDE + NBR_FLAGS] & FLAG_ALL_NEIGHBOURS) != 0; };
    struct Group { int32_t by, bx0, bz; int32_t n; int32_t b[XRUN_MAX]; };
    // kind 0: all-neighbour blocks, kind 1: blocks with a missing neighbour. Chains never mix kinds (different kernels).
    std::vector<Group> runs[2], shorts[2];      // full runs of XRUN; leftover chains of 1 .. XRUN-1 blocks
    size_t i = 0;
    while (i < blks.size()) {
        // maximal chain of x-consecutive blocks of one kind starting at i (same by,bz row)
        size_t j = i;
        const int kind = fast(blks[i].b) ? 0 : 1;
        while (j + 1 < blks.size() && blks[j + 1].by == blks[i].by && blks[j + 1].bz == blks[i].bz && (fast(blks[j + 1].b) ? 0 : 1) == kind &&
               L->h_meta[(size_t)blks[j].b * NBR_STRIDE + DIR(1, 0, 0)] == blks[j + 1].b && blks[j + 1].b != blks[i].b)
            ++j;
        size_t k = i;
        for (; k + XRUN <= j + 1; k += XRUN) {
            Group g{blks[k].by, blks[k].bx, blks[k].bz, XRUN, {}};
            for (int w = 0; w < XRUN; ++w) g.b[w] = blks[k + w].b;
            runs[kind].push_back(g);
        }
        if (k <= j) {
            Group g{blks[k].by, blks[k].bx, blks[k].bz, (int32_t)(j + 1 - k), {}};
            for (size_t w = k; w <= j; ++w) g.b[w - k] = blks[w].b;
            shorts[kind].push_back(g);
        }
        i = j + 1;
    }
    // sweep. Internal block order (x fastest in memory): x fastest, then y, then z = memory order, every population stream is
    // read and written sequentially, and the x / y neighbour groups follow within a few workgroups on the same XCD.
    // Reference block order kept (LUDWIG_REFERENCE_BLOCK_ORDER): y fastest, then x, then z (round 1's choice for that layout).
    const bool x_fastest = L->x_fastest_memory;
    auto by_zxy = [x_fastest](const Group &a, const Group &c) {
        if (a.bz != c.bz) return a.bz < c.bz;
        if (x_fastest) {
            if (a.by != c.by) return a.by < c.by;
            return a.bx0 < c.bx0;
        }
        if (a.bx0 != c.bx0) return a.bx0 < c.bx0;
        return a.by < c.by;
    };
    // slot 8 * group + x runs on XCD x and steps plane z = (x - bz) mod 8: x/y neighbours (same bz, same plane) share an
    // XCD, and over bz every XCD sees every plane index, i.e. every value of address bits 8..10 (no L2-channel aliasing)
    std::vector<int32_t> seq;
    for (int kind = 0; kind < 2; ++kind) {
        std::sort(runs[kind].begin(), runs[kind].end(), by_zxy);
        std::sort(shorts[kind].begin(), shorts[kind].end(), by_zxy);
        for (const Group &g : runs[kind])
            for (int x = 0; x < 8; ++x) {
                const int z = ((x - g.bz) % 8 + 8) % 8;
                for (int w = 0; w < XRUN; ++w) seq.push_back((g.b[w] << 3) | z);
            }
        // leftover chains: whole chains packed into workgroups of XRUN waves (never split), same slot rule
        for (size_t s0 = 0; s0 < shorts[kind].size();) {
            int32_t wg[XRUN_MAX];
            int used = 0;
            const int bz0 = shorts[kind][s0].bz;
            while (s0 < shorts[kind].size() && used + shorts[kind][s0].n <= XRUN) {
                for (int w = 0; w < shorts[kind][s0].n; ++w) wg[used++] = shorts[kind][s0].b[w];
                ++s0;
            }
            for (int x = 0; x < 8; ++x) {
                const int z = ((x - bz0) % 8 + 8) % 8;
                for (int w = 0; w < XRUN; ++w) seq.push_back(w < used ? (wg[w] << 3) | z : -1);
            }
        }
    }
    return set_items(L, part, seq.data(), (int64_t)seq.size());
}

// Links (cell, population) of the interface blocks whose source block is missing and for which the reference's
// domain-edge chain (src/physics_kernels.jl:88-140) selects the parent interpolation: source inside the global box.
int build_interface_links(LudwigLevel *L, const LudwigLevel *parent, int nx_g, int ny_g, int nz_g)
{
    // Everything about a link that does not depend on the flow is fixed here, once: which (cell, population) pairs pull
    // from outside the level (reference src/physics_kernels.jl:122-137), and for each such source cell the 8 parent cells
    // of its trilinear stencil and the weights (reference src/physics_interpolation.jl:29-62: px_cont, floor, +1 before
    // the clamp to 1, block lookup through the parent's block_pointer).
    if (L->iface_parent == parent && L->iface_dims[0] == nx_g && L->iface_dims[1] == ny_g && L->iface_dims[2] == nz_g) return LUDWIG_OK;
    if (parent->h_block_pointer.size() != (size_t)parent->gdx * parent->gdy * parent->gdz)
        return fail(LUDWIG_ERR_STATE, "parent level %d was created without block_pointer: its children cannot interpolate", parent->level_id);
    LW_HIP(hipStreamSynchronize(L->stream));
    struct Link { int64_t key; int sx, sy, sz; int2 e; };
    std::vector<Link> raw[N_PARTS];
    for (int b = 0; b < L->n_owned; ++b) {
        const int32_t *row = &L->h_meta[(size_t)b * NBR_STRIDE];
        const int gbi = row[NBR_GBI];
        if (gbi < 0) continue;
        const bool bnd = !L->h_comm_boundary.empty() && L->h_comm_boundary[b] != 0;
        for (int cell = 0; cell < CELLS; ++cell) {
            const int x = cell & 7, y = (cell >> 3) & 7, z = cell >> 6;
            if (x > 0 && x < 7 && y > 0 && y < 7 && z > 0 && z < 7) continue;
            const int gx = (row[NBR_BX] - 1) * BS + x + 1, gy = (row[NBR_BY] - 1) * BS + y + 1, gz = (row[NBR_BZ] - 1) * BS + z + 1;
            for (int k = 0; k < Q; ++k) {
                const int sx = x - CX(k), sy = y - CY(k), sz = z - CZ(k);
                const int ox = sx < 0 ? -1 : (sx > 7 ? 1 : 0), oy = sy < 0 ? -1 : (sy > 7 ? 1 : 0), oz = sz < 0 ? -1 : (sz > 7 ? 1 : 0);
                if ((ox | oy | oz) == 0 || row[DIR(ox, oy, oz)] >= 0) continue;
                const int src_gx = gx - CX(k), src_gy = gy - CY(k), src_gz = gz - CZ(k);
                if (src_gx < 1 || src_gx > nx_g || src_gy < 1 || src_gy > ny_g || src_gz < 1 || src_gz > nz_g) continue;   // inlet / outlet / mirror win
                const Link l{((int64_t)src_gz * (ny_g + 2) + src_gy) * (nx_g + 2) + src_gx, src_gx, src_gy, src_gz, make_int2((b << 9) | cell, (gbi << 5) | k)};
                raw[LUDWIG_PART_ALL].push_back(l);
                raw[bnd ? LUDWIG_PART_BOUNDARY : LUDWIG_PART_INTERIOR].push_back(l);
            }
        }
    }
    for (int a2 = 0; a2 < N_PARTS; ++a2) {
        std::stable_sort(raw[a2].begin(), raw[a2].end(), [](const Link &u, const Link &v) { return u.key < v.key; });
        std::vector<int4> links;       // (block << 9) | cell, (gbi << 5) | k, source index, unused
        std::vector<int4> corners;     // 2 per source
        std::vector<float4> weights;
        for (size_t i = 0; i < raw[a2].size(); ++i) {
            const Link &l = raw[a2][i];
            if (i == 0 || l.key != raw[a2][i - 1].key) {
                const float pc[3] = {((float)l.sx - 0.5f) * 0.5f, ((float)l.sy - 0.5f) * 0.5f, ((float)l.sz - 0.5f) * 0.5f};
                int p0[3], p1[3];
                float w[3];
                for (int d = 0; d < 3; ++d) {
                    p0[d] = (int)floorf(pc[d]);
                    p1[d] = p0[d] + 1;
                    w[d] = pc[d] - (float)p0[d];
                    p0[d] = std::max(1, p0[d]);
                }
                int cc[8];
                for (int n = 0; n < 8; ++n) {        // corner order 000,100,010,110,001,101,011,111
                    const int pgx = (n & 1) ? p1[0] : p0[0], pgy = (n & 2) ? p1[1] : p0[1], pgz = (n & 4) ? p1[2] : p0[2];
                    const int pbx = (pgx - 1) / BS + 1, pby = (pgy - 1) / BS + 1, pbz = (pgz - 1) / BS + 1;
                    cc[n] = -1;
                    if (pbx >= 1 && pbx <= parent->gdx && pby >= 1 && pby <= parent->gdy && pbz >= 1 && pbz <= parent->gdz) {
                        const int32_t pb = parent->h_block_pointer[(size_t)(pbx - 1) + (size_t)parent->gdx * ((size_t)(pby - 1) + (size_t)parent->gdy * (pbz - 1))];
                        if (pb > 0) cc[n] = ((pgx - 1) % BS) + 8 * ((pgy - 1) % BS) + 64 * ((pgz - 1) % BS) + 512 * (pb - 1);
                    }
                }
                corners.push_back(make_int4(cc[0], cc[1], cc[2], cc[3]));
                corners.push_back(make_int4(cc[4], cc[5], cc[6], cc[7]));
                weights.push_back(make_float4(w[0], w[1], w[2], 0.0f));
            }
            links.push_back(make_int4(l.e.x, l.e.y, (int)weights.size() - 1, 0));
        }
        // launch order of the links: population by population, sources in (z, y, x) order inside. Neighbouring lanes then
        // read neighbouring parent cells of ONE population array (a few 128-B lines per wave-load instead of ~30) and
        // write neighbouring cells of f_iface.
        std::stable_sort(links.begin(), links.end(), [](const int4 &u, const int4 &v) { return (u.y & 31) < (v.y & 31); });
        L->ahead[a2].valid = false;
        void **owned[] = {(void **)&L->links[a2], (void **)&L->sources[a2], (void **)&L->source_w[a2], (void **)&L->source_mac[a2],
                          (void **)&L->source_mac2[a2]};
        for (void **q : owned)
            if (*q) { (void)hipFree(*q); *q = nullptr; }
        L->n_links[a2] = (int)links.size();
        L->n_sources[a2] = (int)weights.size();
        if (!links.empty()) {
            LW_HIP(hipMalloc((void **)&L->links[a2], links.size() * sizeof(int4)));
            LW_HIP(hipMemcpy(L->links[a2], links.data(), links.size() * sizeof(int4), hipMemcpyHostToDevice));
            LW_HIP(hipMalloc((void **)&L->sources[a2], corners.size() * sizeof(int4)));
            LW_HIP(hipMemcpy(L->sources[a2], corners.data(), corners.size() * sizeof(int4), hipMemcpyHostToDevice));
            LW_HIP(hipMalloc((void **)&L->source_w[a2], weights.size() * sizeof(float4)));
            LW_HIP(hipMemcpy(L->source_w[a2], weights.data(), weights.size() * sizeof(float4), hipMemcpyHostToDevice));
            LW_HIP(hipMalloc((void **)&L->source_mac[a2], weights.size() * sizeof(float4)));
            LW_HIP(hipMalloc((void **)&L->source_mac2[a2], weights.size() * sizeof(float4)));

        }
    }
    if (!L->f_iface && L->n_iface_blocks > 0) {
        LW_HIP(hipMalloc((void **)&L->f_iface, (size_t)L->n_iface_blocks * CELLS * Q * sizeof(float)));
        LW_HIP(hipMalloc((void **)&L->f_iface2, (size_t)L->n_iface_blocks * CELLS * Q * sizeof(float)));
        L->device_bytes += 2 * (int64_t)L->n_iface_blocks * CELLS * Q * 4;
    }
    L->iface_parent = parent;
    L->iface_dims[0] = nx_g; L->iface_dims[1] = ny_g; L->iface_dims[2] = nz_g;
    return LUDWIG_OK;
}

// the parent-side pointers of a step / an interface pass for sub-step t_sub
static void fill_parent_params(SCParams &p, const LudwigLevel *parent, int64_t t_sub)
{
    if (parent) {
        const int pout = ((t_sub >> 1) % 2 == 0) ? 1 : 0;   // output buffer of the parent's step t_sub >> 1
        p.pf_new = parent->f[pout];
        p.pvel_new = parent->vel[pout];
        p.prho_new = parent->rho;
        // without temporal storage the reference passes 1-element dummies that are never read
        // (use_temporal_interp is then false at every call site that matters); alias "new" to stay in bounds
        p.pf_old = parent->has_temporal ? (parent->old_alias >= 0 ? parent->f[parent->old_alias] : parent->f_old) : parent->f[pout];
        p.prho_old = parent->has_temporal ? parent->rho_old : parent->rho;
        p.pvel_old = parent->has_temporal ? (parent->old_alias >= 0 ? parent->vel[parent->old_alias] : parent->vel_old) : parent->vel[pout];
        p.is_level_1 = 0;
    } else {
        p.is_level_1 = 1;
    }
}

// Coarse -> fine interface pass for the general blocks of `part` (reference src/physics_kernels.jl:122-137), in two halves.
// interface_decide: leaves p.f_iface pointing at the values sub-step t_sub loads and says whether kernels have to run for them
// (IFACE_LAUNCH) or they exist already - produced ahead by interface_prepass (READY) or together with the previous sub-step's (HIT).
// interface_launch: the two kernels on `st`. ahead_of_step: called by interface_prepass, before the step's own launch.
enum IfaceState { IFACE_NONE, IFACE_READY, IFACE_HIT, IFACE_LAUNCH };

static int interface_decide(LudwigLevel *L, const LudwigLevel *parent, int part, SCParams &p, int64_t t_sub, float parent_tau,
                            float temporal_weight, bool ahead_of_step, IfaceState *state)
{
    *state = IFACE_NONE;
    if (!parent || L->n_items[part][1] == 0) return LUDWIG_OK;
    const int r = build_interface_links(L, parent, p.nx_g, p.ny_g, p.nz_g);
    if (r) return r;
    p.f_iface = L->f_iface;
    p.n_iface_blocks = L->n_iface_blocks;
    if (L->n_links[part] == 0) return LUDWIG_OK;
    {   // values the parent's stream has produced for this pair of sub-steps (recursive_step: parent-side interface pass)
        const LudwigLevel::PairReady &pr = L->pair_ready;
        if (!ahead_of_step && part == LUDWIG_PART_ALL && pr.valid && pr.parent == parent && pr.parent_version == parent->version &&
            pr.pair == (t_sub >> 1) && pr.tau_parent == parent_tau && pr.use_temporal == p.use_temporal &&
            temporal_weight == ((t_sub & 1) ? 0.5f : 0.0f)) {
            float *const first = pr.set ? L->f_iface_b : L->f_iface, *const second = pr.set ? L->f_iface2_b : L->f_iface2;
            p.f_iface = (t_sub & 1) ? second : first;
            *state = IFACE_READY;
            return LUDWIG_OK;
        }
    }
    LudwigLevel::IfaceAhead &pre = L->prepared[part];
    const bool ready = pre.valid && pre.parent == parent && pre.parent_version == parent->version && pre.t_sub == t_sub &&
                       pre.tw == temporal_weight && pre.tau_parent == parent_tau && pre.use_temporal == p.use_temporal;
    pre.valid = ready && ahead_of_step;
    *state = IFACE_READY;
    if (ready) return LUDWIG_OK;                 // produced by interface_prepass; the look-ahead record for t_sub + 1 stays as it is
    LudwigLevel::IfaceAhead &ah = L->ahead[part];
    const bool hit = ah.valid && ah.parent == parent && ah.parent_version == parent->version && ah.t_sub == t_sub &&
                     ah.tw == temporal_weight && ah.tau_parent == parent_tau && ah.use_temporal == p.use_temporal;
    if (hit && ahead_of_step) return LUDWIG_OK;  // nothing to do ahead: the step will find the values in f_iface2
    ah.valid = false;
    *state = IFACE_HIT;
    if (hit) {
        p.f_iface = L->f_iface2;                 // computed together with the previous sub-step's values
        return LUDWIG_OK;
    }
    *state = IFACE_LAUNCH;
    return LUDWIG_OK;
}

static int interface_launch(LudwigLevel *L, const LudwigLevel *parent, int part, const SCParams &p, int64_t t_sub, float parent_tau,
                            float temporal_weight, bool ahead_of_step, hipStream_t st)
{
    LudwigLevel::IfaceAhead &pre = L->prepared[part];
    LudwigLevel::IfaceAhead &ah = L->ahead[part];
    // first sub-step of a pair (even t_sub): also produce the values for t_sub + 1 at weight 0.5
    const bool two = (t_sub % 2 == 0) && !parent->external_writer && getenv("LUDWIG_NO_IFACE_AHEAD") == nullptr;
    InterfaceArgs a{};
    a.corners = L->sources[part]; a.weights = L->source_w[part];
    a.mac = L->source_mac[part]; a.mac2 = L->source_mac2[part];
    a.links = L->links[part];
    a.f_iface2 = L->f_iface2;
    a.tw2 = 0.5f;
    a.n_sources = L->n_sources[part]; a.n_links = L->n_links[part];
    const dim3 gs((unsigned)((a.n_sources + 255) / 256)), gl((unsigned)((a.n_links + 255) / 256));
    if (two) {
        hipLaunchKernelGGL(k_interface_sources<true>, gs, dim3(256), 0, st, p, a);
        hipLaunchKernelGGL(k_interface_links<true>, gl, dim3(256), 0, st, p, a);
        ah.valid = true; ah.parent = parent; ah.parent_version = parent->version; ah.t_sub = t_sub + 1;
        ah.tw = a.tw2; ah.tau_parent = parent_tau; ah.use_temporal = p.use_temporal;
    } else {
        hipLaunchKernelGGL(k_interface_sources<false>, gs, dim3(256), 0, st, p, a);
        hipLaunchKernelGGL(k_interface_links<false>, gl, dim3(256), 0, st, p, a);
    }
    LW_HIP(hipGetLastError());
    if (ahead_of_step) {
        pre.valid = true; pre.parent = parent; pre.parent_version = parent->version; pre.t_sub = t_sub;
        pre.tw = temporal_weight; pre.tau_parent = parent_tau; pre.use_temporal = p.use_temporal;
    }
    // level streams (ludwig_execute_timestep_batch): the parent's buffers have been read - the last time for this
    // pair of sub-steps when the values for the second one were produced alongside
    if (parent->ev_consumed && L->own_stream && L->stream == L->own_stream) {
        LW_HIP(hipEventRecord(parent->ev_consumed, st));
        const_cast<LudwigLevel *>(parent)->ev_consumed_set = true;
    }
    return LUDWIG_OK;
}

static int interface_pass(LudwigLevel *L, const LudwigLevel *parent, int part, SCParams &p, int64_t t_sub, float parent_tau,
                          float temporal_weight, bool ahead_of_step)
{
    IfaceState state;
    const int r = interface_decide(L, parent, part, p, t_sub, parent_tau, temporal_weight, ahead_of_step, &state);
    if (r || state != IFACE_LAUNCH) return r;
    return interface_launch(L, parent, part, p, t_sub, parent_tau, temporal_weight, ahead_of_step, L->stream);
}

// Parent-side interface pass: `parent` has just stepped t_pair on ITS stream; the interface values its child C needs for sub-steps
// 2 t_pair (temporal weight 0.0) and 2 t_pair + 1 (0.5) are computed right there, behind that step, into set t_pair & 1 of C's side
// buffers. C's stream, the critical chain of a nested case, then runs nothing but stream-collide and Bouzidi launches; the pass
// shares the parent's (low-priority) stream with the parent's own kernels and fills the gaps the finest level leaves.
// Dependencies: the pass reads the parent's new and saved state - same stream, right after the step that made them, before the
// next one that overwrites them (the ev_consumed hand-shake of the child-side pass is not needed); it overwrites the set C read
// two pairs ago - waited for through C's ev_pair_done. Same kernels, same arguments as the child-side pass: same bits.
static int interface_pass_for_child(LudwigLevel *parent, LudwigLevel *C, int64_t t_pair, const LudwigStepFlags *fl)
{
    if (C->n_blocks == 0 || C->n_items[LUDWIG_PART_ALL][1] == 0 || parent->external_writer) return LUDWIG_OK;
    SCParams p{};
    const int64_t t_sub = 2 * t_pair;
    fill_parent_params(p, parent, t_sub);
    p.tau = C->tau;
    p.tau_parent = parent->tau;
    p.temporal_weight = 0.0f;
    const int scale = 1 << (C->level_id - 1);
    p.nx_g = fl->domain_nx * scale; p.ny_g = fl->domain_ny * scale; p.nz_g = fl->domain_nz * scale;
    p.use_temporal = (fl->use_temporal_interp && parent->has_temporal) ? 1 : 0;
    int rc = build_interface_links(C, parent, p.nx_g, p.ny_g, p.nz_g);
    if (rc) return rc;
    const int part = LUDWIG_PART_ALL;
    if (C->n_links[part] == 0) return LUDWIG_OK;
    const int set = (int)(t_pair & 1);
    const size_t n_side = (size_t)C->n_iface_blocks * CELLS * Q, n_src = (size_t)C->n_sources[part];
    if (!C->f_iface_b) {
        LW_HIP(hipMalloc((void **)&C->f_iface_b, n_side * sizeof(float)));
        LW_HIP(hipMalloc((void **)&C->f_iface2_b, n_side * sizeof(float)));
        LW_HIP(hipMalloc((void **)&C->mac_b, n_src * sizeof(float4)));
        LW_HIP(hipMalloc((void **)&C->mac2_b, n_src * sizeof(float4)));
        C->device_bytes += (int64_t)(2 * n_side * sizeof(float) + 2 * n_src * sizeof(float4));
        for (hipEvent_t &ev : C->ev_pair_done) LW_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    if (C->pair_done_set[set]) LW_HIP(hipStreamWaitEvent(parent->stream, C->ev_pair_done[set], 0));      // C has finished with this set
    p.f_iface = set ? C->f_iface_b : C->f_iface;
    p.n_iface_blocks = C->n_iface_blocks;
    InterfaceArgs a{};
    a.corners = C->sources[part]; a.weights = C->source_w[part];
    a.mac = set ? C->mac_b : C->source_mac[part]; a.mac2 = set ? C->mac2_b : C->source_mac2[part];
    a.links = C->links[part];
    a.f_iface2 = set ? C->f_iface2_b : C->f_iface2;
    a.tw2 = 0.5f;
    a.n_sources = C->n_sources[part]; a.n_links = C->n_links[part];
    const dim3 gs((unsigned)((a.n_sources + 255) / 256)), gl((unsigned)((a.n_links + 255) / 256));
    hipLaunchKernelGGL(k_interface_sources<true>, gs, dim3(256), 0, parent->stream, p, a);
    hipLaunchKernelGGL(k_interface_links<true>, gl, dim3(256), 0, parent->stream, p, a);
    LW_HIP(hipGetLastError());
    LudwigLevel::PairReady &pr = C->pair_ready;
    pr.valid = true; pr.parent = parent; pr.parent_version = parent->version; pr.pair = t_pair;
    pr.tau_parent = parent->tau; pr.use_temporal = p.use_temporal; pr.set = set;
    C->ahead[part].valid = false;          // the child-side look-ahead buffers (set 0's second half) are about to be / have been reused
    C->prepared[part].valid = false;
    return LUDWIG_OK;
}

// The interface pass of sub-step t_sub, launched before the step itself. It reads the PARENT's buffers and writes this level's
// interface side buffers only, so a level with children can run it before it waits for them (recursive_step).
static int interface_prepass(LudwigLevel *L, const LudwigLevel *parent, int64_t t_sub, float parent_tau, float temporal_weight,
                             const LudwigStepFlags *fl)
{
    if (!parent || L->n_blocks == 0 || L->n_items[LUDWIG_PART_ALL][1] == 0) return LUDWIG_OK;
    LW_HIP(hipSetDevice(L->device));
    SCParams p{};
    fill_parent_params(p, parent, t_sub);
    p.tau = L->tau;
    p.tau_parent = parent_tau;
    p.temporal_weight = temporal_weight;
    const int scale = 1 << (L->level_id - 1);
    p.nx_g = fl->domain_nx * scale; p.ny_g = fl->domain_ny * scale; p.nz_g = fl->domain_nz * scale;
    p.use_temporal = (fl->use_temporal_interp && parent->has_temporal) ? 1 : 0;
    return interface_pass(L, parent, LUDWIG_PART_ALL, p, t_sub, parent_tau, temporal_weight, true);
}

int launch_stream_collide(LudwigLevel *L, const LudwigLevel *parent, int64_t t_sub, float u_curr, float parent_tau,
                          float temporal_weight, const LudwigStepFlags *fl, int part)
{
    if (!L || !fl) return fail(LUDWIG_ERR_INVALID, "null level or flags");
    if (part < 0 || part >= N_PARTS) return fail(LUDWIG_ERR_INVALID, "bad part %d", part);
    if (t_sub < 0) return fail(LUDWIG_ERR_INVALID, "t_sub must be >= 0");
    if (L->n_blocks == 0) return LUDWIG_OK;   // reference src/physics_v2.jl:41
    if (parent && parent->device != L->device) return fail(LUDWIG_ERR_INVALID, "parent level lives on another device");
    LW_HIP(hipSetDevice(L->device));
    const int in = (t_sub % 2 == 0) ? 0 : 1, out = 1 - in;   // reference src/solver_control.jl:35-41
    ++L->version;                                            // this level's fields are about to change
    if (L->old_alias == out) {                               // about to overwrite the buffer that holds the saved state
        const int r = materialize_old(L);
        if (r) return r;
    }
    SCParams p{};
    p.f_in = L->f[in];
    p.f_out = L->f[out];
    p.vel_in = L->vel[in];
    p.vel_out = L->vel[out];
    p.rho = L->rho;
    // reference src/physics_v2.jl:77: store_post_collision = bouzidi_enabled && n_boundary_cells > 0; a rank that owns no
    // Bouzidi cell of a Bouzidi level still stores it when asked to (n_boundary_cells < 0): a peer's cells read its face layer
    p.f_post = L->has_post ? L->f_post : nullptr;
    // rows: the links with q > 0 are the only readers as long as q_min >= 0 (a negative threshold makes every direction of a listed
    // cell a link, src/bouzidi_kernel.jl:44: whole blocks then)
    p.post_rows = (p.f_post && fl->q_min_threshold >= 0.0f) ? L->post_rows : nullptr;
    if (p.f_post) L->post_rows_used = p.post_rows != nullptr;
    p.obstacle = L->obstacle;
    p.sponge = L->sponge;
    p.wall_dist = L->wall_dist;
    p.meta = L->meta;
    fill_parent_params(p, parent, t_sub);
    p.tau = L->tau;
    p.tau_parent = parent_tau;
    p.c_wale = fl->c_wale;
    p.nu_bg = fl->nu_sgs_background;
    p.u_inlet = u_curr;
    p.inlet_turbulence = fl->inlet_turbulence;
    p.temporal_weight = temporal_weight;
    p.is_symmetric = fl->is_symmetric ? 1 : 0;
    const int scale = 1 << (L->level_id - 1);   // reference src/physics_v2.jl:55-56
    p.nx_g = fl->domain_nx * scale; p.ny_g = fl->domain_ny * scale; p.nz_g = fl->domain_nz * scale;
    p.wall_model = fl->wall_model_active ? 1 : 0;
    p.seed = (int32_t)(t_sub % 1000000);        // reference src/physics_v2.jl:76
    // a parent without temporal storage cannot be blended with (reference would index a dummy array)
    p.use_temporal = (fl->use_temporal_interp && (!parent || parent->has_temporal)) ? 1 : 0;
    p.sponge_blend = fl->sponge_blend_distributions ? 1 : 0;

    // lazy rho: who reads rho before this level's next step?
    if (parent && !parent->rho_eager) {            // a child interpolates from its parent's rho after every parent step
        LudwigLevel *pm = const_cast<LudwigLevel *>(parent);
        pm->rho_eager = true;
        const int r = ensure_rho(pm);
        if (r) return r;
    }
    {
        static const bool eager_env = getenv("LUDWIG_EAGER_RHO") != nullptr;
        if (t_sub != L->last_step_t) { ++L->step_count; L->last_step_t = t_sub; }
        const bool store = L->rho_eager || eager_env || part != LUDWIG_PART_ALL;
        // This launch reuses the previous step's INPUT buffer as its output. A whole-level launch supersedes the elided rho of
        // the previous one (the reference would be overwriting it right now, unread); a part launch covers only some of the
        // cells, so the rest is produced first, while its inputs still exist.
        if (part != LUDWIG_PART_ALL && L->rho_replay[LUDWIG_PART_ALL].stale) {
            const int r = ensure_rho(L);
            if (r) return r;
        }
        L->rho_replay[LUDWIG_PART_ALL].stale = false;
        p.store_rho = store ? 1 : 0;
        // rho_old in the step (save_old_impl): this launch covers every cell and stores rho
        if (L->rho_old_pending) {
            if (part == LUDWIG_PART_ALL && store) p.rho_old_save = L->rho_old;
            else LW_HIP(hipMemcpyAsync(L->rho_old, L->rho, (size_t)L->sk * 4, hipMemcpyDeviceToDevice, L->stream));
            L->rho_old_pending = false;
        }
    }

    const bool post = p.f_post != nullptr, wall = p.wall_model != 0;
    const hipStream_t cs = L->stream;
    // one launch of a work list through the instantiation its blocks need
    auto launch_list = [&](const int32_t *items, int64_t n_items, bool general) -> int {
        if (n_items == 0) return LUDWIG_OK;
        p.items = items;
        const dim3 grid((unsigned)(n_items / XRUN)), block(64 * XRUN);
#define LW_LAUNCH_X(G, P, W) do { if (L->wide) hipLaunchKernelGGL((k_stream_collide_xrun<XRUN, G, P, W, false, true>), grid, block, 0, cs, p); \
            else hipLaunchKernelGGL((k_stream_collide_xrun<XRUN, G, P, W, false, false>), grid, block, 0, cs, p); } while (0)
        // WALL without POST goes through the <POST, WALL> instantiation (no block is flagged for the f_post store, so the null
        // pointer is never used): the register allocator gets <POST = false, WALL = true> down to 96 VGPRs only by spilling
        // 208 B per lane, while the POST variant sits at 88-90 without - same arithmetic, same bits
        if (general) {       // GENERAL without POST needs a 104-byte spill to stay at 96 VGPRs: the POST variant serves it too
            if (wall) LW_LAUNCH_X(true, true, true);
            else LW_LAUNCH_X(true, true, false);
        } else {
            if (wall) LW_LAUNCH_X(false, true, true);
            else if (post) LW_LAUNCH_X(false, true, false);
            else LW_LAUNCH_X(false, false, false);
        }
#undef LW_LAUNCH_X
        LW_HIP(hipGetLastError());
        return LUDWIG_OK;
    };
    const hipEvent_t parent_wait = L->parent_wait;          // level streams: the parent step this sub-step's interface values come from
    L->parent_wait = nullptr;
    IfaceState istate;
    {
        const int r = interface_decide(L, parent, part, p, t_sub, parent_tau, temporal_weight, false, &istate);
        if (r) return r;
    }
    // (Round 3 tried to take the interface pass - the one piece of a sub-step that reads the PARENT, and whose result only the general
    // blocks read - off the chain: all-neighbour blocks first, before the wait for the parent's step; and the pass on a side stream
    // under them. Both measured slower than this merged launch, at any stream priority: profiles/r03_split_substep_ab.txt. Removed.)
    {
        if (parent_wait) LW_HIP(hipStreamWaitEvent(cs, parent_wait, 0));
        int r;
        if (istate == IFACE_LAUNCH && (r = interface_launch(L, parent, part, p, t_sub, parent_tau, temporal_weight, false, cs))) return r;
        if (L->n_items_merged[part] > 0) {
            if ((r = launch_list(L->items_merged[part], L->n_items_merged[part], true))) return r;
        } else {
            if ((r = launch_list(L->items[part][1], L->n_items[part][1], true))) return r;
            if ((r = launch_list(L->items[part][2], L->n_items[part][2], false))) return r;
        }
    }
    if (!p.store_rho) {
        LudwigLevel::RhoReplay &rr = L->rho_replay[part];
        rr.stale = true;
        rr.t_sub = t_sub;
        rr.p = p;                                    // pointers into this level's (and the parent's) buffers, scalars of the step
    }
    return LUDWIG_OK;
}

// Produce `rho` now if the last step left it unwritten (see LudwigLevel::rho_replay).
int ensure_rho(LudwigLevel *L)
{
    bool any = false;
    for (int a = 0; a < N_PARTS; ++a) any = any || L->rho_replay[a].stale;
    if (!any) return LUDWIG_OK;
    LW_HIP(hipSetDevice(L->device));
    // somebody reads rho after every step (e.g. a peer rank's finer blocks interpolate from this rank's cells: the halo pack
    // of rho): replaying costs a second pass over f each time, storing costs 4 B per cell - switch
    if (L->last_replay_step == L->step_count - 1) L->rho_eager = true;
    L->last_replay_step = L->step_count;
    for (int part = 0; part < N_PARTS; ++part) {
        LudwigLevel::RhoReplay &rr = L->rho_replay[part];
        if (!rr.stale) continue;
        SCParams p = rr.p;
        p.store_rho = 1;
        for (int c = 1; c < N_CLASSES; ++c) {
            if (L->n_items[part][c] == 0) continue;
            p.items = L->items[part][c];
            const bool general = c == 1;
            const dim3 grid((unsigned)(L->n_items[part][c] / XRUN)), block(64 * XRUN);
            if (L->wide) {
                if (general) hipLaunchKernelGGL((k_stream_collide_xrun<XRUN, true, false, false, true, true>), grid, block, 0, L->stream, p);
                else hipLaunchKernelGGL((k_stream_collide_xrun<XRUN, false, false, false, true, true>), grid, block, 0, L->stream, p);
            } else {
                if (general) hipLaunchKernelGGL((k_stream_collide_xrun<XRUN, true, false, false, true, false>), grid, block, 0, L->stream, p);
                else hipLaunchKernelGGL((k_stream_collide_xrun<XRUN, false, false, false, true, false>), grid, block, 0, L->stream, p);
            }
            LW_HIP(hipGetLastError());
        }
        rr.stale = false;
    }
    return LUDWIG_OK;
}

int launch_bouzidi(LudwigLevel *L, int64_t t_sub, float q_min)
{
    if (!L) return fail(LUDWIG_ERR_INVALID, "null level");
    if (!(L->bouzidi_enabled && L->n_bc > 0)) return LUDWIG_OK;   // reference src/bouzidi_kernel.jl:107-109
    if (q_min < 0.0f && L->post_rows_used)
        return fail(LUDWIG_ERR_STATE, "q_min_threshold < 0 makes every direction a link: the stream-collide call must be given the same "
                                      "threshold (it stored f_post_collision for the links with q > 0 only)");
    ++L->version;
    LW_HIP(hipSetDevice(L->device));
    const int out = (t_sub % 2 == 0) ? 1 : 0;
    BouzidiParams p{};
    p.f_out = L->f[out];
    p.f_post = L->f_post;
    p.q_map = L->q_map;
    p.cell_block = L->cell_block;
    p.cell_x = L->cell_x; p.cell_y = L->cell_y; p.cell_z = L->cell_z;
    p.meta = L->meta;
    p.n_cells = L->n_bc;
    // q > q_min with q_min >= 0 can only hold where q > 0: the compact list; a negative threshold takes every (cell, k)
    const bool compact = q_min >= 0.0f && L->bouzidi_links;
    p.links = compact ? L->bouzidi_links : nullptr;
    p.n_links = L->n_bouzidi_links;
    p.q_min = q_min;
    const int64_t n_threads = compact ? (int64_t)L->n_bouzidi_links : (int64_t)L->n_bc * Q;
    if (n_threads == 0) return LUDWIG_OK;
    hipLaunchKernelGGL(k_bouzidi, dim3((unsigned)((n_threads + 255) / 256)), dim3(256), 0, L->stream, p);
    LW_HIP(hipGetLastError());
    return LUDWIG_OK;
}

// Host <-> device copy of a whole field between the caller's array (reference layout: population-major, reference block order)
// and the device array (block-major, internal block order). One component and the same block order: one copy; otherwise one
// component at a time through `scratch` and a kernel that places every block. K = components, es = element size (1: obstacle,
// 2: q map, else 4). Synchronous on the level's stream.
int copy_field(LudwigLevel *L, void *dev, void *host, int K, size_t es, bool to_device)
{
    const size_t plane = (size_t)L->sk * es;                   // one component in bytes
    if (plane == 0) return LUDWIG_OK;
    if (K == 1 && !L->d_ref2int) {
        if (to_device) LW_HIP(hipMemcpyAsync(dev, host, plane, hipMemcpyHostToDevice, L->stream));
        else LW_HIP(hipMemcpyAsync(host, dev, plane, hipMemcpyDeviceToHost, L->stream));
        LW_HIP(hipStreamSynchronize(L->stream));
        return LUDWIG_OK;
    }
    if (!L->scratch) LW_HIP(hipMalloc(&L->scratch, (size_t)L->sk * 4));
    const int64_t n = L->sk;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    for (int k = 0; k < K; ++k) {
        char *h = (char *)host + (size_t)k * plane;
        if (to_device) {
            LW_HIP(hipMemcpyAsync(L->scratch, h, plane, hipMemcpyHostToDevice, L->stream));
            if (es == 1) hipLaunchKernelGGL(k_component_to_internal<uint8_t>, grid, block, 0, L->stream, (uint8_t *)dev, (const uint8_t *)L->scratch, L->d_ref2int, n, K, k);
            else if (es == 2) hipLaunchKernelGGL(k_component_to_internal<uint16_t>, grid, block, 0, L->stream, (uint16_t *)dev, (const uint16_t *)L->scratch, L->d_ref2int, n, K, k);
            else hipLaunchKernelGGL(k_component_to_internal<float>, grid, block, 0, L->stream, (float *)dev, (const float *)L->scratch, L->d_ref2int, n, K, k);
        } else {
            if (es == 1) hipLaunchKernelGGL(k_component_to_reference<uint8_t>, grid, block, 0, L->stream, (uint8_t *)L->scratch, (const uint8_t *)dev, L->d_ref2int, n, K, k);
            else if (es == 2) hipLaunchKernelGGL(k_component_to_reference<uint16_t>, grid, block, 0, L->stream, (uint16_t *)L->scratch, (const uint16_t *)dev, L->d_ref2int, n, K, k);
            else hipLaunchKernelGGL(k_component_to_reference<float>, grid, block, 0, L->stream, (float *)L->scratch, (const float *)dev, L->d_ref2int, n, K, k);
            LW_HIP(hipMemcpyAsync(h, L->scratch, plane, hipMemcpyDeviceToHost, L->stream));
            LW_HIP(hipStreamSynchronize(L->stream));          // pageable destination: the copy must be over before scratch is reused
        }
        LW_HIP(hipGetLastError());
    }
    LW_HIP(hipStreamSynchronize(L->stream));
    return LUDWIG_OK;
}

}  // namespace

extern "C" {

int ludwig_abi_version(void) { return LUDWIG_ABI_VERSION; }

const char *ludwig_last_error(void) { return g_err.c_str(); }

int ludwig_device_count(int *count)
{
    if (!count) return fail(LUDWIG_ERR_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(LUDWIG_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return LUDWIG_OK;
}

void ludwig_level_destroy(LudwigLevel *L)
{
    if (!L) return;
    (void)hipSetDevice(L->device);
    void *ptrs[] = {L->f[0], L->f[1], L->vel[0], L->vel[1], L->rho, L->f_post, L->f_old, L->rho_old, L->vel_old, L->obstacle,
                    L->sponge, L->wall_dist, L->meta, L->q_map, L->cell_block, L->cell_x, L->cell_y, L->cell_z};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (int a = 0; a < N_PARTS; ++a) {
        for (int c = 0; c < N_CLASSES; ++c)
            if (L->items[a][c]) (void)hipFree(L->items[a][c]);
        if (L->items_merged[a]) (void)hipFree(L->items_merged[a]);
        if (L->links[a]) (void)hipFree(L->links[a]);
        if (L->sources[a]) (void)hipFree(L->sources[a]);
        if (L->source_w[a]) (void)hipFree(L->source_w[a]);
        if (L->source_mac[a]) (void)hipFree(L->source_mac[a]);
        if (L->source_mac2[a]) (void)hipFree(L->source_mac2[a]);
    }
    if (L->f_iface) (void)hipFree(L->f_iface);
    if (L->d_ref2int) (void)hipFree(L->d_ref2int);
    if (L->scratch) (void)hipFree(L->scratch);
    if (L->own_stream) (void)hipStreamDestroy(L->own_stream);
    if (L->ev_stepped) (void)hipEventDestroy(L->ev_stepped);
    if (L->ev_consumed) (void)hipEventDestroy(L->ev_consumed);
    if (L->bouzidi_links) (void)hipFree(L->bouzidi_links);
    if (L->post_rows) (void)hipFree(L->post_rows);
    if (L->f_iface2) (void)hipFree(L->f_iface2);
    {
        void *more[] = {L->f_iface_b, L->f_iface2_b, L->mac_b, L->mac2_b};
        for (void *q : more)
            if (q) (void)hipFree(q);
        for (hipEvent_t ev : L->ev_pair_done)
            if (ev) (void)hipEventDestroy(ev);
    }
    delete L;
}

static int level_create_impl(const LudwigLevelHost *h, int device, LudwigLevel **out, bool x_fastest_memory);

int ludwig_level_create(const LudwigLevelHost *h, int device, LudwigLevel **out)
{
    if (out) *out = nullptr;
    if (!h || !out) return fail(LUDWIG_ERR_INVALID, "null argument");
    const int32_t nb = h->n_blocks;
    if (nb <= 1 || getenv("LUDWIG_REFERENCE_BLOCK_ORDER") || !h->neighbor_table || !h->map_x || !h->map_y || !h->map_z)
        return level_create_impl(h, device, out, false);
    const int n_owned = h->n_owned > 0 ? h->n_owned : (h->n_owned < 0 ? 0 : nb);
    if (n_owned > nb) return fail(LUDWIG_ERR_INVALID, "n_owned > n_blocks");
    // internal order: owned blocks first (as given), inside each group sorted by (bz, by, bx) - bx fastest
    std::vector<int32_t> int2ref((size_t)nb), ref2int((size_t)nb);
    for (int32_t b = 0; b < nb; ++b) int2ref[b] = b;
    std::stable_sort(int2ref.begin(), int2ref.end(), [&](int32_t a, int32_t c) {
        const bool oa = a < n_owned, oc = c < n_owned;
        if (oa != oc) return oa;
        if (h->map_z[a] != h->map_z[c]) return h->map_z[a] < h->map_z[c];
        if (h->map_y[a] != h->map_y[c]) return h->map_y[a] < h->map_y[c];
        return h->map_x[a] < h->map_x[c];
    });
    for (int32_t i = 0; i < nb; ++i) ref2int[int2ref[i]] = i;
    // the same description with every per-block array permuted and every block id translated (ids are 1-based, 0 = absent)
    LudwigLevelHost hp = *h;
    const size_t n = (size_t)nb;
    std::vector<int32_t> nt(n * 27), mx(n), my(n), mz(n), bp, cellb;
    std::vector<uint8_t> obs, cbnd;
    std::vector<float> spo, wal;
    std::vector<uint16_t> qm;
    for (size_t i = 0; i < n; ++i) {
        const size_t r = (size_t)int2ref[i];
        mx[i] = h->map_x[r]; my[i] = h->map_y[r]; mz[i] = h->map_z[r];
        for (int d = 0; d < 27; ++d) {
            const int32_t v = h->neighbor_table[r + n * d];
            if (v < 0 || v > nb) return fail(LUDWIG_ERR_INVALID, "neighbor_table[%zu,%d] = %d out of range", r + 1, d + 1, v);
            nt[i + n * d] = v > 0 ? ref2int[v - 1] + 1 : 0;
        }
    }
    hp.neighbor_table = nt.data(); hp.map_x = mx.data(); hp.map_y = my.data(); hp.map_z = mz.data();
    auto permute_cells = [&](auto &dst, const auto *src, size_t comps) {
        dst.resize(n * CELLS * comps);
        for (size_t k = 0; k < comps; ++k)
            for (size_t i = 0; i < n; ++i)
                memcpy(&dst[(k * n + i) * CELLS], &src[(k * n + (size_t)int2ref[i]) * CELLS], CELLS * sizeof(dst[0]));
    };
    if (h->obstacle) { permute_cells(obs, h->obstacle, 1); hp.obstacle = obs.data(); }
    if (h->sponge) { permute_cells(spo, h->sponge, 1); hp.sponge = spo.data(); }
    if (h->wall_dist) { permute_cells(wal, h->wall_dist, 1); hp.wall_dist = wal.data(); }
    if (h->bouzidi_q_map && h->n_boundary_cells > 0) { permute_cells(qm, h->bouzidi_q_map, Q); hp.bouzidi_q_map = qm.data(); }
    if (h->bouzidi_cell_block && h->n_boundary_cells > 0) {
        cellb.resize((size_t)h->n_boundary_cells);
        for (int32_t i = 0; i < h->n_boundary_cells; ++i) {
            const int32_t v = h->bouzidi_cell_block[i];
            if (v < 1 || v > nb) return fail(LUDWIG_ERR_INVALID, "bouzidi cell %d out of range", i + 1);
            cellb[i] = ref2int[v - 1] + 1;
        }
        hp.bouzidi_cell_block = cellb.data();
    }
    if (h->comm_boundary) {
        cbnd.resize(n);
        for (size_t i = 0; i < n; ++i) cbnd[i] = h->comm_boundary[int2ref[i]];
        hp.comm_boundary = cbnd.data();
    }
    const size_t nptr = (size_t)h->grid_dim_x * h->grid_dim_y * h->grid_dim_z;
    if (h->block_pointer && nptr > 0) {
        bp.resize(nptr);
        for (size_t i = 0; i < nptr; ++i) {
            const int32_t v = h->block_pointer[i];
            if (v < 0 || v > nb) return fail(LUDWIG_ERR_INVALID, "block_pointer[%zu] = %d out of range", i, v);
            bp[i] = v > 0 ? ref2int[v - 1] + 1 : 0;
        }
        hp.block_pointer = bp.data();
    }
    const int rc = level_create_impl(&hp, device, out, true);   // the description is in the internal order now
    if (rc) return rc;
    LudwigLevel *L = *out;
    L->ref2int.swap(ref2int);
    L->int2ref.swap(int2ref);
    hipError_t e = hipMalloc((void **)&L->d_ref2int, n * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(L->d_ref2int, L->ref2int.data(), n * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) { ludwig_level_destroy(L); *out = nullptr; return fail(LUDWIG_ERR_HIP, "block order table: %s", hipGetErrorString(e)); }
    return LUDWIG_OK;
}

static int level_create_impl(const LudwigLevelHost *h, int device, LudwigLevel **out, bool x_fastest_memory)
{
    if (out) *out = nullptr;
    if (!h || !out) return fail(LUDWIG_ERR_INVALID, "null argument");
    if (h->n_blocks < 0 || h->level_id < 1 || h->level_id > 30) return fail(LUDWIG_ERR_INVALID, "bad n_blocks/level_id");
    if (h->n_blocks > 0 && (!h->neighbor_table || !h->map_x || !h->map_y || !h->map_z)) return fail(LUDWIG_ERR_INVALID, "neighbor_table and map_x/y/z are required");
    if ((int64_t)h->n_blocks * CELLS * 4 >= (int64_t)1 << 32) return fail(LUDWIG_ERR_INVALID, "n_blocks too large for 32-bit byte offsets (max 2^21 - 1 blocks per level)");
    const int n_owned = h->n_owned > 0 ? h->n_owned : (h->n_owned < 0 ? 0 : h->n_blocks);
    if (n_owned > h->n_blocks) return fail(LUDWIG_ERR_INVALID, "n_owned > n_blocks");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(LUDWIG_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(LUDWIG_ERR_INVALID, "device %d out of range (have %d)", device, ndev);
    LW_HIP(hipSetDevice(device));

    LudwigLevel *L = new (std::nothrow) LudwigLevel();
    if (!L) return fail(LUDWIG_ERR_ALLOC, "host allocation failed");
    L->device = device;
    L->x_fastest_memory = x_fastest_memory;
    L->level_id = h->level_id;
    L->n_blocks = h->n_blocks;
    L->n_owned = n_owned;
    L->tau = h->tau;
    L->gdx = h->grid_dim_x; L->gdy = h->grid_dim_y; L->gdz = h->grid_dim_z;
    L->sk = (int64_t)h->n_blocks * CELLS;
    L->wide = (int64_t)h->n_blocks * (int64_t)F_BLOCK_BYTES >= ((int64_t)1 << 32) || getenv("LUDWIG_WIDE_ADDR") != nullptr;
    const size_t c = (size_t)L->sk, nb = (size_t)h->n_blocks;
    L->has_temporal = h->enable_temporal_interpolation && nb > 0;            // reference src/blocks.jl:123
    L->bouzidi_enabled = h->n_boundary_cells > 0 && h->bouzidi_q_map;        // reference src/blocks.jl:152
    L->n_bc = L->bouzidi_enabled ? h->n_boundary_cells : 0;
    L->has_post = h->n_boundary_cells != 0 && nb > 0;                         // reference src/blocks.jl:135 (< 0: forced, multi-GPU)
    if (L->bouzidi_enabled && (!h->bouzidi_cell_block || !h->bouzidi_cell_x || !h->bouzidi_cell_y || !h->bouzidi_cell_z)) {
        delete L;
        return fail(LUDWIG_ERR_INVALID, "bouzidi cell lists missing");
    }

    int rc = LUDWIG_OK;
#define LW_TRY(expr) do { if (rc == LUDWIG_OK) rc = (expr); } while (0)
    LW_TRY(dev_alloc(L, &L->f[0], c * Q));
    LW_TRY(dev_alloc(L, &L->f[1], c * Q));
    LW_TRY(dev_alloc(L, &L->vel[0], c * 3));
    LW_TRY(dev_alloc(L, &L->vel[1], c * 3));
    LW_TRY(dev_alloc(L, &L->rho, c));
    if (L->has_post) LW_TRY(dev_alloc(L, &L->f_post, c * Q));
    if (L->has_temporal) {
        LW_TRY(dev_alloc(L, &L->f_old, c * Q));
        LW_TRY(dev_alloc(L, &L->rho_old, c));
        LW_TRY(dev_alloc(L, &L->vel_old, c * 3));
    }
    LW_TRY(dev_alloc(L, &L->obstacle, c));
    LW_TRY(dev_alloc(L, &L->sponge, c));
    LW_TRY(dev_alloc(L, &L->wall_dist, c));
    LW_TRY(dev_alloc(L, &L->meta, nb * NBR_STRIDE));
    const size_t nptr = (size_t)h->grid_dim_x * h->grid_dim_y * h->grid_dim_z;
    if (L->bouzidi_enabled) {
        LW_TRY(dev_alloc(L, &L->q_map, c * Q));
        LW_TRY(dev_alloc(L, &L->cell_block, (size_t)L->n_bc));
        LW_TRY(dev_alloc(L, &L->cell_x, (size_t)L->n_bc));
        LW_TRY(dev_alloc(L, &L->cell_y, (size_t)L->n_bc));
        LW_TRY(dev_alloc(L, &L->cell_z, (size_t)L->n_bc));
    }
    if (rc != LUDWIG_OK) { ludwig_level_destroy(L); return rc; }

    // metadata rows: neighbours 0-based (-1 absent), flags, block coords
    L->h_meta.assign(nb * NBR_STRIDE, 0);
    for (size_t b = 0; b < nb; ++b) {
        int32_t *row = &L->h_meta[b * NBR_STRIDE];
        bool all = true;
        for (int d = 0; d < 27; ++d) {
            const int32_t v = h->neighbor_table[b + nb * d];
            if (v < 0 || v > h->n_blocks) { ludwig_level_destroy(L); return fail(LUDWIG_ERR_INVALID, "neighbor_table[%zu,%d] = %d out of range", b + 1, d + 1, v); }
            row[d] = v - 1;
            if (d != 13 && v == 0) all = false;
        }
        row[13] = (int32_t)b;   // the block itself; the reference never reads this entry
        row[NBR_FLAGS] = all ? FLAG_ALL_NEIGHBOURS : 0;
        row[NBR_BX] = h->map_x[b]; row[NBR_BY] = h->map_y[b]; row[NBR_BZ] = h->map_z[b];
    }
    if (h->comm_boundary) L->h_comm_boundary.assign(h->comm_boundary, h->comm_boundary + nb);
    scan_flags(L, LUDWIG_OBSTACLE, h->obstacle);
    scan_flags(L, LUDWIG_SPONGE, h->sponge);
    scan_flags(L, LUDWIG_WALL_DIST, h->wall_dist);
    for (int b = 0; b < n_owned; ++b)
        if (L->h_meta[(size_t)b * NBR_STRIDE + NBR_FLAGS] & FLAG_ALL_NEIGHBOURS) ++L->n_fast_blocks;
    for (size_t b = 0; b < nb; ++b) L->h_meta[b * NBR_STRIDE + NBR_GBI] = -1;
    for (int b = 0; b < n_owned; ++b)
        if (!(L->h_meta[(size_t)b * NBR_STRIDE + NBR_FLAGS] & FLAG_ALL_NEIGHBOURS)) L->h_meta[(size_t)b * NBR_STRIDE + NBR_GBI] = L->n_iface_blocks++;

    auto init = [&]() -> int {
        // constructor defaults, reference src/blocks.jl:118-150
        LW_HIP(hipMemsetAsync(L->f[0], 0, c * Q * 4, L->stream));
        LW_HIP(hipMemsetAsync(L->f[1], 0, c * Q * 4, L->stream));
        LW_HIP(hipMemsetAsync(L->vel[0], 0, c * 3 * 4, L->stream));
        LW_HIP(hipMemsetAsync(L->vel[1], 0, c * 3 * 4, L->stream));
        int r = fill(L, L->rho, L->sk, 1.0f);
        if (r) return r;
        if (L->has_post) LW_HIP(hipMemsetAsync(L->f_post, 0, c * Q * 4, L->stream));
        if (L->has_temporal) {
            LW_HIP(hipMemsetAsync(L->f_old, 0, c * Q * 4, L->stream));
            LW_HIP(hipMemsetAsync(L->vel_old, 0, c * 3 * 4, L->stream));
            if ((r = fill(L, L->rho_old, L->sk, 1.0f))) return r;
        }
        if (h->obstacle) LW_HIP(hipMemcpyAsync(L->obstacle, h->obstacle, c, hipMemcpyHostToDevice, L->stream));
        else LW_HIP(hipMemsetAsync(L->obstacle, 0, c, L->stream));
        if (h->sponge) LW_HIP(hipMemcpyAsync(L->sponge, h->sponge, c * 4, hipMemcpyHostToDevice, L->stream));
        else LW_HIP(hipMemsetAsync(L->sponge, 0, c * 4, L->stream));
        if (h->wall_dist) LW_HIP(hipMemcpyAsync(L->wall_dist, h->wall_dist, c * 4, hipMemcpyHostToDevice, L->stream));
        else if ((r = fill(L, L->wall_dist, L->sk, 100.0f))) return r;
        // block_pointer stays on the host: its only use is the static corner lookup of a child's interface links
        if (h->block_pointer && nptr > 0) L->h_block_pointer.assign(h->block_pointer, h->block_pointer + nptr);
        std::vector<int4> bl;               // the links with q > 0: (own cell, k, q bits, cell behind or -1)
        if (L->bouzidi_enabled) {
            {   // host: population-major (already in the internal block order here), device: block-major
                const int r = copy_field(L, L->q_map, const_cast<uint16_t *>(h->bouzidi_q_map), Q, 2, true);
                if (r) return r;
            }
            std::vector<int32_t> cb((size_t)L->n_bc);
            std::vector<int8_t> cx((size_t)L->n_bc), cy((size_t)L->n_bc), cz((size_t)L->n_bc);
            for (int i = 0; i < L->n_bc; ++i) {
                cb[i] = h->bouzidi_cell_block[i] - 1;
                cx[i] = (int8_t)(h->bouzidi_cell_x[i] - 1); cy[i] = (int8_t)(h->bouzidi_cell_y[i] - 1); cz[i] = (int8_t)(h->bouzidi_cell_z[i] - 1);
                if (cb[i] < 0 || cb[i] >= L->n_blocks || cx[i] < 0 || cx[i] > 7 || cy[i] < 0 || cy[i] > 7 || cz[i] < 0 || cz[i] > 7)
                    return fail(LUDWIG_ERR_INVALID, "bouzidi cell %d out of range", i + 1);
            }
            LW_HIP(hipMemcpy(L->cell_block, cb.data(), cb.size() * 4, hipMemcpyHostToDevice));
            LW_HIP(hipMemcpy(L->cell_x, cx.data(), cx.size(), hipMemcpyHostToDevice));
            LW_HIP(hipMemcpy(L->cell_y, cy.data(), cy.size(), hipMemcpyHostToDevice));
            LW_HIP(hipMemcpy(L->cell_z, cz.data(), cz.size(), hipMemcpyHostToDevice));
            const _Float16 *qh = reinterpret_cast<const _Float16 *>(h->bouzidi_q_map);
            for (int i = 0; i < L->n_bc; ++i) {
                const int own = cb[i] * CELLS + cx[i] + 8 * cy[i] + 64 * cz[i];
                for (int k = 0; k < Q; ++k) {
                    const float q = (float)qh[(size_t)own + c * k];
                    if (!(q > 0.0f)) continue;
                    // the cell one step behind along the link (reference src/bouzidi_kernel.jl:47-58): same block, the neighbour block
                    // of the table, or none (-1: the reference falls back to the cell's own value)
                    const int ok = 26 - k;
                    const int nx = cx[i] + CX(ok), ny = cy[i] + CY(ok), nz = cz[i] + CZ(ok);
                    const int ox = nx < 0 ? -1 : (nx >= BS ? 1 : 0), oy = ny < 0 ? -1 : (ny >= BS ? 1 : 0), oz = nz < 0 ? -1 : (nz >= BS ? 1 : 0);
                    const int nbb = (ox | oy | oz) == 0 ? cb[i] : L->h_meta[(size_t)cb[i] * NBR_STRIDE + DIR(ox, oy, oz)];
                    const int behind = nbb >= 0 ? nbb * CELLS + (nx & 7) + 8 * (ny & 7) + 64 * (nz & 7) : -1;
                    int qbits;
                    memcpy(&qbits, &q, 4);
                    bl.push_back(make_int4(own, k, qbits, behind));
                }
            }
            // population-major, cells in memory order inside a population: the lanes of a wave then work on ONE population array at
            // cells that are mostly neighbours along x - their f_post loads and f_out stores share 32-B sectors instead of touching 64
            // different 2-KiB population slices of a few cells (every link writes its own (cell, opp k): the order is free)
            std::sort(bl.begin(), bl.end(), [](const int4 &a, const int4 &c) { return a.y != c.y ? a.y < c.y : a.x < c.x; });
            L->n_bouzidi_links = (int)bl.size();
            if (!bl.empty()) {
                LW_HIP(hipMalloc((void **)&L->bouzidi_links, bl.size() * sizeof(int4)));
                LW_HIP(hipMemcpy(L->bouzidi_links, bl.data(), bl.size() * sizeof(int4), hipMemcpyHostToDevice));
            }
        }
        if (L->has_post) {
            // where f_post_collision has a reader: blocks that hold a Bouzidi cell or a cell next to one (a link q < 1/2 reads
            // the cell one step behind, possibly across a block face). No cell list on this rank (forced store, multi-GPU:
            // the readers are a peer's cells), store_post_collision_everywhere or LUDWIG_FULL_POST_COLLISION set: every block, as the reference.
            const bool fixed = h->store_post_collision_everywhere != 0 || getenv("LUDWIG_FULL_POST_COLLISION") != nullptr;
            const bool everywhere = L->n_bc == 0 || fixed;
            L->post_mode = fixed ? 1 : (L->n_bc == 0 ? 2 : 0);
            for (int b = 0; b < L->n_blocks && everywhere; ++b) L->h_meta[(size_t)b * NBR_STRIDE + NBR_FLAGS] |= FLAG_STORE_POST;
            for (int i = 0; i < L->n_bc && !everywhere; ++i) {
                // the cell itself and the 26 cells around it (a link reads f_post one cell behind the boundary cell)
                const int32_t *row = &L->h_meta[(size_t)(h->bouzidi_cell_block[i] - 1) * NBR_STRIDE];
                const int x = h->bouzidi_cell_x[i] - 1, y = h->bouzidi_cell_y[i] - 1, z = h->bouzidi_cell_z[i] - 1;
                for (int oz = (z == 0 ? -1 : 0); oz <= (z == 7 ? 1 : 0); ++oz)
                    for (int oy = (y == 0 ? -1 : 0); oy <= (y == 7 ? 1 : 0); ++oy)
                        for (int ox = (x == 0 ? -1 : 0); ox <= (x == 7 ? 1 : 0); ++ox) {
                            const int nb2 = row[DIR(ox, oy, oz)];
                            if (nb2 >= 0) L->h_meta[(size_t)nb2 * NBR_STRIDE + NBR_FLAGS] |= FLAG_STORE_POST;
                        }
            }
            // inside those blocks: the x-rows a link with q > 0 reads - its own cell and the cell one step behind
            // (LUDWIG_POST_ROWS=0: whole blocks, round 2's granularity)
            const char *pr = getenv("LUDWIG_POST_ROWS");
            if (!fixed && !(pr && pr[0] == '0')) {
                L->h_post_rows.assign((size_t)L->n_blocks * 2, 0u);
                auto mark = [&](int cell) { const int row = (cell & (CELLS - 1)) >> 3; L->h_post_rows[(size_t)(cell / CELLS) * 2 + (row >> 5)] |= 1u << (row & 31); };
                for (const int4 &l : bl) { mark(l.x); if (l.w >= 0) mark(l.w); }
                if (!everywhere) {
                    LW_HIP(hipMalloc((void **)&L->post_rows, L->h_post_rows.size() * 4));
                    LW_HIP(hipMemcpy(L->post_rows, L->h_post_rows.data(), L->h_post_rows.size() * 4, hipMemcpyHostToDevice));
                }
            }
        }
        LW_HIP(hipStreamSynchronize(L->stream));
        int r2 = upload_meta(L);
        if (r2) return r2;
        for (int part = 0; part < N_PARTS; ++part)
            if ((r2 = default_items(L, part))) return r2;
        return LUDWIG_OK;
    };
    rc = nb > 0 ? init() : LUDWIG_OK;
    if (rc != LUDWIG_OK) { ludwig_level_destroy(L); return rc; }
#undef LW_TRY
    *out = L;
    return LUDWIG_OK;
}

int ludwig_level_add_post_collision_readers(LudwigLevel *L, const int64_t *offsets, int64_t n)
{
    if (!L || (!offsets && n > 0) || n < 0) return fail(LUDWIG_ERR_INVALID, "null argument");
    if (!L->has_post) return fail(LUDWIG_ERR_STATE, "level %d has no f_post_collision", L->level_id);
    if (L->post_mode == 1 || L->n_blocks == 0) return LUDWIG_OK;      // every block is stored already, and stays so
    LW_HIP(hipSetDevice(L->device));
    LW_HIP(hipStreamSynchronize(L->stream));
    const int64_t sk = (int64_t)L->n_blocks * CELLS;
    for (int64_t i = 0; i < n; ++i)
        if (offsets[i] < 0 || offsets[i] >= sk * Q) return fail(LUDWIG_ERR_INVALID, "reader offset %lld outside f_post_collision", (long long)i);
    if (L->post_mode == 2) {             // "everything, because nobody said who reads": now somebody has
        for (int b = 0; b < L->n_blocks; ++b) L->h_meta[(size_t)b * NBR_STRIDE + NBR_FLAGS] &= ~FLAG_STORE_POST;
        L->post_mode = 0;
    }
    const bool by_rows = !L->h_post_rows.empty();
    for (int64_t i = 0; i < n; ++i) {
        const int64_t e = offsets[i] % sk;
        int b = (int)(e / CELLS);
        if (!L->ref2int.empty()) b = L->ref2int[b];
        const int row = (int)(e % CELLS) >> 3;
        L->h_meta[(size_t)b * NBR_STRIDE + NBR_FLAGS] |= FLAG_STORE_POST;
        if (by_rows) L->h_post_rows[(size_t)b * 2 + (row >> 5)] |= 1u << (row & 31);
    }
    if (by_rows) {
        if (!L->post_rows) LW_HIP(hipMalloc((void **)&L->post_rows, L->h_post_rows.size() * 4));
        LW_HIP(hipMemcpy(L->post_rows, L->h_post_rows.data(), L->h_post_rows.size() * 4, hipMemcpyHostToDevice));
    }
    ++L->version;
    return upload_meta(L);
}

int ludwig_level_set_stream(LudwigLevel *L, void *hip_stream)
{
    if (!L) return fail(LUDWIG_ERR_INVALID, "null level");
    L->stream = (hipStream_t)hip_stream;
    return LUDWIG_OK;
}

int ludwig_stream_create(int device, int reserved_cus, void **stream_out)
{
    if (!stream_out) return fail(LUDWIG_ERR_INVALID, "null argument");
    *stream_out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return fail(LUDWIG_ERR_NO_DEVICE, "no HIP device %d", device);
    LW_HIP(hipSetDevice(device));
    hipStream_t st = nullptr;
    if (reserved_cus <= 0) {
        LW_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        *stream_out = st;
        return LUDWIG_OK;
    }
    int n_cu = 0;
    LW_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device));
    if (reserved_cus > n_cu / 2) return fail(LUDWIG_ERR_INVALID, "%d of %d compute units reserved: at most half", reserved_cus, n_cu);
    std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
    for (int i = 0; i < n_cu; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
    // Bit i of the mask is a CU of XCD i mod 8, and inside an XCD four consecutive CU indices sit in its four shader engines
    // (measured, tools/cu_mask_patterns.py, profiles/r03_cu_mask_patterns.txt: CUs 0, 8, 16, 24 of every XCD off = +51 %, CUs 0-3 of
    // every XCD off = +4.5 %; one CU per XCD off = +6 %, a single CU = +8 %). What a mask costs the stepping kernel is the IMBALANCE between
    // shader engines, not the number of CUs: the cheapest mask that reserves anything takes one CU out of every shader engine of every
    // XCD - 32 mask bits - and costs less than round 2's one-per-XCD pattern. So on 256 CUs the request is rounded up to a multiple of 32
    // (bits 224-255, then 192-223, ...; 64 CUs: +12.6 %).
    int done = 0;
    if (n_cu == 256) {
        const int groups = (reserved_cus + 31) / 32;
        for (int bit = 256 - 32 * groups; bit < 256; ++bit, ++done) mask[(size_t)bit / 32] &= ~(1u << (bit % 32));
    }
    for (int k = 0; done < reserved_cus; ++k) {      // generic / remainder: evenly spaced, skipping bits already cleared
        const int bit = (int)(((int64_t)k * n_cu) / reserved_cus + 5) % n_cu;
        if (mask[(size_t)bit / 32] & (1u << (bit % 32))) { mask[(size_t)bit / 32] &= ~(1u << (bit % 32)); ++done; }
        if (k > 4 * n_cu) break;
    }
    LW_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    *stream_out = st;
    return LUDWIG_OK;
}

int ludwig_stream_destroy(int device, void *hip_stream)
{
    if (!hip_stream) return LUDWIG_OK;
    LW_HIP(hipSetDevice(device));
    LW_HIP(hipStreamDestroy((hipStream_t)hip_stream));
    return LUDWIG_OK;
}

int ludwig_level_set_order(LudwigLevel *L, int part, const int32_t *items, int64_t n_items)
{
    if (!L || (!items && n_items > 0)) return fail(LUDWIG_ERR_INVALID, "null argument");
    if (part < 0 || part >= N_PARTS) return fail(LUDWIG_ERR_INVALID, "bad part %d", part);
    LW_HIP(hipSetDevice(L->device));
    LW_HIP(hipStreamSynchronize(L->stream));
    // every (block, z-plane) of the part exactly once
    std::vector<uint8_t> seen((size_t)L->n_owned * 8, 0);
    int64_t expect = 0, real = 0;
    for (int b = 0; b < L->n_owned; ++b)
        if (block_in_part(L, b, part)) expect += 8;
    for (int64_t i = 0; i < n_items; ++i) {
        if (items[i] < 0) continue;
        ++real;
        const int b = items[i] >> 3;
        if (b >= L->n_owned) return fail(LUDWIG_ERR_INVALID, "bad work item %lld", (long long)i);
        if (seen[(size_t)items[i]]++) return fail(LUDWIG_ERR_INVALID, "work item %lld listed twice", (long long)i);
    }
    if (real != expect) return fail(LUDWIG_ERR_INVALID, "order has %lld work items, part has %lld", (long long)real, (long long)expect);
    if (L->ref2int.empty()) return set_items(L, part, items, n_items);
    std::vector<int32_t> tr(items, items + n_items);           // the caller counts blocks in the reference order
    for (int32_t &it : tr)
        if (it >= 0) it = (L->ref2int[it >> 3] << 3) | (it & 7);
    return set_items(L, part, tr.data(), n_items);
}

int ludwig_level_upload(LudwigLevel *L, int field, const void *host, size_t bytes)
{
    if (!L || !host) return fail(LUDWIG_ERR_INVALID, "null argument");
    const FieldDesc d = field_desc(L, field);
    if (!d.ptr || d.bytes == 0) return fail(LUDWIG_ERR_STATE, "field %d is not allocated on this level", field);
    if (bytes != d.bytes) return fail(LUDWIG_ERR_INVALID, "field %d: got %zu bytes, expected %zu", field, bytes, d.bytes);
    LW_HIP(hipSetDevice(L->device));
    ++L->version;
    {
        const int r = before_external_write(L, field);
        if (r) return r;
    }
    const size_t es = (size_t)d.es;
    {
        const FieldDesc dn = field_desc(L, field);             // after before_external_write: the saved state may have moved
        const int r = copy_field(L, dn.ptr, const_cast<void *>(host), dn.comps, es, true);
        if (r) return r;
    }
    if (field == LUDWIG_OBSTACLE || field == LUDWIG_SPONGE || field == LUDWIG_WALL_DIST) {
        if (!L->ref2int.empty()) {                             // the per-block flags are indexed by internal block id
            std::vector<char> tmp(bytes);
            for (size_t i = 0; i < (size_t)L->n_blocks; ++i)
                memcpy(&tmp[i * CELLS * es], (const char *)host + (size_t)L->int2ref[i] * CELLS * es, CELLS * es);
            scan_flags(L, field, tmp.data());
        } else {
            scan_flags(L, field, host);
        }
        return upload_meta(L);
    }
    return LUDWIG_OK;
}

int ludwig_level_download(const LudwigLevel *L, int field, void *host, size_t bytes)
{
    if (!L || !host) return fail(LUDWIG_ERR_INVALID, "null argument");
    const FieldDesc d = field_desc(L, field);
    if (!d.ptr || d.bytes == 0) return fail(LUDWIG_ERR_STATE, "field %d is not allocated on this level", field);
    if (bytes != d.bytes) return fail(LUDWIG_ERR_INVALID, "field %d: got %zu bytes, expected %zu", field, bytes, d.bytes);
    LW_HIP(hipSetDevice(L->device));
    if (field == LUDWIG_RHO) {
        const int r = ensure_rho(const_cast<LudwigLevel *>(L));
        if (r) return r;
    }
    return copy_field(const_cast<LudwigLevel *>(L), d.ptr, host, d.comps, (size_t)d.es, false);
}

int ludwig_level_block_order(const LudwigLevel *L, int32_t *ref_to_internal)
{
    if (!L || (!ref_to_internal && L->n_blocks > 0)) return fail(LUDWIG_ERR_INVALID, "null argument");
    for (int32_t b = 0; b < L->n_blocks; ++b) ref_to_internal[b] = L->ref2int.empty() ? b : L->ref2int[b];
    return LUDWIG_OK;
}

int ludwig_level_field_ptr(const LudwigLevel *L, int field, void **device_ptr, size_t *bytes)
{
    if (!L || !device_ptr) return fail(LUDWIG_ERR_INVALID, "null argument");
    ++const_cast<LudwigLevel *>(L)->version;
    {   // the caller may write through the pointer: give the saved state its own storage first
        const int r = before_external_write(const_cast<LudwigLevel *>(L), field);
        if (r) return r;
    }
    const FieldDesc d = field_desc(L, field);
    if (!d.ptr || d.bytes == 0) return fail(LUDWIG_ERR_STATE, "field %d is not allocated on this level", field);
    if (field != LUDWIG_OBSTACLE && field != LUDWIG_SPONGE && field != LUDWIG_WALL_DIST) {   // those three feed per-block flags: upload them
        LudwigLevel *m = const_cast<LudwigLevel *>(L);
        m->external_writer = true;
        m->rho_eager = true;
    }
    *device_ptr = d.ptr;
    if (bytes) *bytes = d.bytes;
    return LUDWIG_OK;
}

int ludwig_level_set_rho_store(LudwigLevel *L, int every_step)
{
    if (!L) return fail(LUDWIG_ERR_INVALID, "null level");
    if (every_step) {
        const int r = ensure_rho(L);                // what the last step elided is produced now; from here on every step stores
        if (r) return r;
        L->rho_eager = true;
    } else {
        L->rho_eager = false;                       // steps may elide again; a reader that turns up twice in a row makes it eager again
        L->last_replay_step = -10;
    }
    return LUDWIG_OK;
}

int ludwig_level_field_layout(const LudwigLevel *L, int field, int32_t *components, int64_t *block_stride, int64_t *component_stride)
{
    if (!L) return fail(LUDWIG_ERR_INVALID, "null level");
    const FieldDesc d = field_desc(L, field);
    if (field < 0 || field >= LUDWIG_FIELD_COUNT) return fail(LUDWIG_ERR_INVALID, "bad field %d", field);
    if (components) *components = d.comps;
    if (block_stride) *block_stride = (int64_t)d.comps * CELLS;      // block-major: the components of a block are contiguous
    if (component_stride) *component_stride = CELLS;
    return LUDWIG_OK;
}

int ludwig_init_equilibrium(LudwigLevel *L)
{
    if (!L) return fail(LUDWIG_ERR_INVALID, "null level");
    if (L->n_blocks == 0) return LUDWIG_OK;
    LW_HIP(hipSetDevice(L->device));
    const unsigned grid = (unsigned)((L->sk + 255) / 256);
    ++L->version;
    {
        const int r = ensure_rho(L);                 // init_eq! leaves rho alone: it must hold what the last step made of it
        if (r) return r;
    }
    hipLaunchKernelGGL(k_fill_weights, dim3(grid), dim3(256), 0, L->stream, L->f[0], L->sk);
    hipLaunchKernelGGL(k_fill_weights, dim3(grid), dim3(256), 0, L->stream, L->f[1], L->sk);
    if (L->has_temporal) {
        L->old_alias = -1;
        hipLaunchKernelGGL(k_fill_weights, dim3(grid), dim3(256), 0, L->stream, L->f_old, L->sk);
        int r = fill(L, L->rho_old, L->sk, 1.0f);
        if (r) return r;
        LW_HIP(hipMemsetAsync(L->vel_old, 0, (size_t)L->sk * 3 * 4, L->stream));
    }
    LW_HIP(hipGetLastError());
    return LUDWIG_OK;
}

int ludwig_stream_collide(LudwigLevel *L, const LudwigLevel *parent, int64_t t_sub, float u_curr, float parent_tau,
                          float temporal_weight, const LudwigStepFlags *flags, int part)
{
    return launch_stream_collide(L, parent, t_sub, u_curr, parent_tau, temporal_weight, flags, part);
}

int ludwig_bouzidi_correction(LudwigLevel *L, int64_t t_sub, float q_min_threshold) { return launch_bouzidi(L, t_sub, q_min_threshold); }

int ludwig_step(LudwigLevel *L, const LudwigLevel *parent, int64_t t_sub, float u_curr, float parent_tau, float temporal_weight,
                const LudwigStepFlags *flags)
{
    int r = launch_stream_collide(L, parent, t_sub, u_curr, parent_tau, temporal_weight, flags, LUDWIG_PART_ALL);
    if (r) return r;
    return launch_bouzidi(L, t_sub, flags->q_min_threshold);
}

static int save_old_impl(LudwigLevel *L, int64_t t_sub, bool defer_rho);

int ludwig_save_old(LudwigLevel *L, int64_t t_sub) { return save_old_impl(L, t_sub, false); }

static int save_old_impl(LudwigLevel *L, int64_t t_sub, bool defer_rho)
{
    if (!L) return fail(LUDWIG_ERR_INVALID, "null level");
    if (!L->has_temporal) return LUDWIG_OK;   // reference src/blocks.jl:200
    LW_HIP(hipSetDevice(L->device));
    const int in = (t_sub % 2 == 0) ? 0 : 1;
    const size_t c = (size_t)L->sk;
    // f and vel: no copy - the step that follows reads f[in] / vel[in] and never writes them (see old_alias)
    ++L->version;
    if (L->external_writer) {            // somebody holds raw pointers into f / vel: a real copy, as the reference does
        L->old_alias = -1;
        LW_HIP(hipMemcpyAsync(L->f_old, L->f[in], c * Q * 4, hipMemcpyDeviceToDevice, L->stream));
        LW_HIP(hipMemcpyAsync(L->vel_old, L->vel[in], c * 3 * 4, hipMemcpyDeviceToDevice, L->stream));
    } else {
        L->old_alias = in;
    }
    L->rho_eager = true;                             // a level that saves its old state has children reading rho every step
    {
        const int r = ensure_rho(L);
        if (r) return r;
    }
    if (defer_rho && L->n_owned == L->n_blocks && !L->external_writer && getenv("LUDWIG_RHO_OLD_COPY") == nullptr) {
        L->rho_old_pending = true;                   // the step that follows saves it cell by cell (launch_stream_collide)
        return LUDWIG_OK;
    }
    L->rho_old_pending = false;
    LW_HIP(hipMemcpyAsync(L->rho_old, L->rho, c * 4, hipMemcpyDeviceToDevice, L->stream));
    return LUDWIG_OK;
}

// Level streams. The reference steps its levels strictly one after the other (src/solver_control.jl:21-143), and every launch
// of a small level leaves most of the 256 CUs idle. The data dependencies are weaker than the call order: coupling is one-way,
// coarse -> fine, and a child reads its parent's buffers only in its interface pass (k_interface_sources / _links, once per
// pair of sub-steps). So each level gets a HIP stream of its own and two events:
//   * a child's sub-step waits for its parent's step (parent->ev_stepped) before it interpolates from it;
//   * a parent's NEXT step - which overwrites the buffer holding its old state, rho and rho_old - waits until the child's
//     interface pass has read them (ev_consumed, recorded on the child's stream right after that pass).
// Launches are still issued in the reference's order; the GPU then runs level 1's step t + 1 under the finer levels' sub-steps
// of step t, and a middle level's second sub-step under its children's first pair. Same kernels, same inputs: same bits.
// The finest level is the critical chain (2^(n-1) sub-steps per coarse step): its stream gets the highest priority, the others the
// lowest, so the coarser levels only fill what it leaves free. Measured on one box, alternating (profiles/
// r02_level_streams_with_priorities_ab_one_box.txt): 3-level sphere 0.419 -> 0.393 ms per coarse step, real wing 1.011 -> 0.949,
// 4-level sphere 1.72 -> 1.52; without the priorities 0.402 / 0.982 / 1.58 (and on another box the 4-level case got slower).
// LUDWIG_BATCH_SERIAL=1 keeps everything on one stream; LUDWIG_LEVEL_STREAM_PRIORITY=0 gives every level the same priority.
static bool level_streams() { static const bool v = getenv("LUDWIG_BATCH_SERIAL") == nullptr; return v; }

static int recursive_step(LudwigLevel *const *levels, int n_levels, int lvl /*1-based*/, int64_t t_sub, const LudwigLevel *parent,
                          float parent_tau, float temporal_weight, float u_vel, const LudwigStepFlags *fl, bool concurrent)
{
    // recursive_step! / recursive_step_temporal!, reference src/solver_control.jl:21-143
    if (lvl > n_levels) return LUDWIG_OK;
    LudwigLevel *L = levels[lvl - 1];
    const bool has_children = lvl < n_levels;
    int rc;
    if (concurrent) {
        // every event operation costs the stream ~7 us between two kernels (kernel trace of the 3-level sphere): wait for a
        // parent step once, not once per sub-step
        // the only reader of the parent is this level's interface pass: the wait for the parent's step goes where that pass goes
        // (launch_stream_collide) - unless the pass is hoisted below
        const bool need_parent = parent && (L->waited_parent != parent || L->waited_gen != parent->stepped_gen);
        bool hoisting = false;
        if (has_children && L->ev_consumed_set) {
            static const char *he0 = getenv("LUDWIG_IFACE_HOIST");
            hoisting = he0 ? atoi(he0) != 0 : 20 * (int64_t)L->n_blocks >= 9 * (int64_t)levels[lvl]->n_blocks;
        }
        if (need_parent) {
            // handed to launch_stream_collide, which waits right before the interface pass
            if (hoisting) LW_HIP(hipStreamWaitEvent(L->stream, parent->ev_stepped, 0));
            else L->parent_wait = parent->ev_stepped;
            L->waited_parent = parent; L->waited_gen = parent->stepped_gen;
        }
        if (has_children && L->ev_consumed_set) {
            // This level's own interface pass reads the parent and writes side buffers: it need not wait for the children to have
            // read THIS level's buffers, only the step behind it must. Running it ahead shortens what is left to do after the wait.
            // That pays when this level is not much smaller than its child - its step then comes in late for the child's next
            // pair of sub-steps (3-level sphere, 1000 blocks under 1728: 25 us late per coarse step, 0.370 -> 0.337 ms) - and costs
            // when the child dwarfs it and nothing was late (wing, 1728 under 5256: 0.848 -> 0.883 ms of added contention).
            // LUDWIG_IFACE_HOIST=0 / 1 overrides the size rule.
            static const char *he = getenv("LUDWIG_IFACE_HOIST");
            const bool hoist = he ? atoi(he) != 0 : 20 * (int64_t)L->n_blocks >= 9 * (int64_t)levels[lvl]->n_blocks;
            if (hoist && (rc = interface_prepass(L, parent, t_sub, parent_tau, temporal_weight, fl))) return rc;
            LW_HIP(hipStreamWaitEvent(L->stream, L->ev_consumed, 0));
        }
    }
    if (has_children && fl->use_temporal_interp && L->has_temporal)
        if ((rc = save_old_impl(L, t_sub, true))) return rc;      // rho's part rides in the step's launch where it can
    if ((rc = ludwig_step(L, parent, t_sub, u_vel, parent_tau, temporal_weight, fl))) return rc;
    static const bool parent_side = getenv("LUDWIG_CHILD_SIDE_IFACE") == nullptr;      // LUDWIG_CHILD_SIDE_IFACE=1: round 2's placement
    if (concurrent && parent && parent_side && (t_sub & 1) && L->ev_pair_done[0]) {
        // this level has read the last of set (t_sub >> 1) & 1 of its interface values: the parent's stream may overwrite it
        const int set = (int)((t_sub >> 1) & 1);
        LW_HIP(hipEventRecord(L->ev_pair_done[set], L->stream));
        L->pair_done_set[set] = true;
    }
    if (concurrent && has_children) {
        if (parent_side && (rc = interface_pass_for_child(L, levels[lvl], t_sub, fl))) return rc;
        LW_HIP(hipEventRecord(L->ev_stepped, L->stream));
        ++L->stepped_gen;
    }
    if (has_children) {
        if ((rc = recursive_step(levels, n_levels, lvl + 1, 2 * t_sub, L, L->tau, 0.0f, u_vel, fl, concurrent))) return rc;
        if ((rc = recursive_step(levels, n_levels, lvl + 1, 2 * t_sub + 1, L, L->tau, 0.5f, u_vel, fl, concurrent))) return rc;
    }
    return LUDWIG_OK;
}

int ludwig_execute_timestep_batch(LudwigLevel *const *levels, int32_t n_levels, int64_t t_start, int32_t batch_size, float u_curr,
                                  const LudwigStepFlags *flags)
{
    if (!levels || !flags || n_levels < 1 || batch_size < 0) return fail(LUDWIG_ERR_INVALID, "bad argument");
    for (int i = 0; i < n_levels; ++i) {
        if (!levels[i]) return fail(LUDWIG_ERR_INVALID, "null level %d", i + 1);
        if (levels[i]->device != levels[0]->device || levels[i]->stream != levels[0]->stream)
            return fail(LUDWIG_ERR_INVALID, "all levels must share one device and one stream");
    }
    const bool concurrent = n_levels > 1 && level_streams();
    hipStream_t user_stream = levels[0]->stream;
    if (concurrent) {
        LW_HIP(hipSetDevice(levels[0]->device));
        LW_HIP(hipStreamSynchronize(user_stream));             // everything queued before the batch is done
        // 1. every level gets its stream and its two events - or none does: a half-made set is destroyed again, so that a later
        //    batch never finds a level with a stream but no events. The levels keep the caller's stream until all of it exists.
        int pr_least = 0, pr_greatest = 0;
        LW_HIP(hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest));
        const char *pe = getenv("LUDWIG_LEVEL_STREAM_PRIORITY");
        const bool use_pr = !pe || atoi(pe) != 0;
        hipError_t e = hipSuccess;
        for (int i = 0; i < n_levels && e == hipSuccess; ++i) {
            LudwigLevel *L = levels[i];
            if (L->own_stream && L->ev_stepped && L->ev_consumed) continue;
            // the finest level is the critical chain (2^(n-1) sub-steps per coarse step): its stream gets the highest priority, the
            // coarser levels fill what it leaves free (graded priorities: no better)
            // LUDWIG_LEVEL_STREAM_PRIORITY=2: graded - the finest level highest, its parent (whose stream carries the finest level's interface
            // pass since round 3) one below, the rest lowest
            int pr = !use_pr ? pr_least : (i == n_levels - 1 ? pr_greatest : pr_least);
            if (pe && atoi(pe) == 2) pr = std::min(pr_least, pr_greatest + (n_levels - 1 - i));
            if (!L->own_stream) e = hipStreamCreateWithPriority(&L->own_stream, hipStreamNonBlocking, pr);
            if (e == hipSuccess && !L->ev_stepped) e = hipEventCreateWithFlags(&L->ev_stepped, hipEventDisableTiming);
            if (e == hipSuccess && !L->ev_consumed) e = hipEventCreateWithFlags(&L->ev_consumed, hipEventDisableTiming);
        }
        if (e != hipSuccess) {
            for (int i = 0; i < n_levels; ++i) {
                LudwigLevel *L = levels[i];
                if (L->own_stream && L->ev_stepped && L->ev_consumed) continue;      // complete from an earlier batch: keep
                hipEvent_t *evs[] = {&L->ev_stepped, &L->ev_consumed};
                for (hipEvent_t *ev : evs)
                    if (*ev) { (void)hipEventDestroy(*ev); *ev = nullptr; }
                if (L->own_stream) { (void)hipStreamDestroy(L->own_stream); L->own_stream = nullptr; }
            }
            return fail(LUDWIG_ERR_HIP, "batch: level streams: %s", hipGetErrorString(e));
        }
        // 2. switch over; one way back for every early return
        auto restore = [&]() { for (int j = 0; j < n_levels; ++j) levels[j]->stream = user_stream; };
        for (int i = 0; i < n_levels; ++i) {
            LudwigLevel *L = levels[i];
            L->ev_consumed_set = false;
            L->waited_parent = nullptr;
            L->parent_wait = nullptr;
            L->pair_ready.valid = false;
            L->pair_done_set[0] = L->pair_done_set[1] = false;      // the previous batch ended with every stream idle
            L->stream = L->own_stream;
        }
        for (int i = 0; i + 1 < n_levels; ++i) {
            LudwigLevel *L = levels[i];
            if (!L->rho_eager) {                               // children interpolate from rho after every step
                L->rho_eager = true;
                const int r = ensure_rho(L);
                if (r) { restore(); return r; }
            }
        }
    }
    for (int i = 0; i < n_levels; ++i) levels[i]->rho_old_pending = false;      // (only an aborted batch could have left one)
    int rc = LUDWIG_OK;
    for (int32_t o = 0; o < batch_size && rc == LUDWIG_OK; ++o)
        rc = recursive_step(levels, n_levels, 1, t_start + o, nullptr, 0.5f, 0.0f, u_curr, flags, concurrent);
    if (concurrent) {
        hipError_t e = hipSuccess;
        for (int i = 0; i < n_levels; ++i) {
            const hipError_t ei = hipStreamSynchronize(levels[i]->own_stream);
            if (e == hipSuccess) e = ei;
            levels[i]->stream = user_stream;
        }
        if (rc == LUDWIG_OK && e != hipSuccess) return fail(LUDWIG_ERR_HIP, "batch: %s", hipGetErrorString(e));
    }
    if (rc) return rc;
    return ludwig_sync(levels[0]);
}

int ludwig_sync(const LudwigLevel *L)
{
    if (!L) return fail(LUDWIG_ERR_INVALID, "null level");
    LW_HIP(hipSetDevice(L->device));
    LW_HIP(hipStreamSynchronize(L->stream));
    return LUDWIG_OK;
}

int ludwig_map_surface_stresses(const LudwigLevel *L, int vel_field, int32_t n_tri, const float *centers, const float *normals,
                                const LudwigSurfaceParams *sp, float *pressure, float *shear_x, float *shear_y, float *shear_z)
{
    if (!L || !sp || n_tri < 0 || (n_tri > 0 && (!centers || !normals || !pressure || !shear_x || !shear_y || !shear_z)))
        return fail(LUDWIG_ERR_INVALID, "null argument");
    if (vel_field != LUDWIG_VEL && vel_field != LUDWIG_VEL_TEMP) return fail(LUDWIG_ERR_INVALID, "vel_field must be LUDWIG_VEL or LUDWIG_VEL_TEMP");
    if (n_tri == 0) return LUDWIG_OK;
    if (L->n_blocks == 0 || L->h_block_pointer.size() != (size_t)L->gdx * L->gdy * L->gdz)
        return fail(LUDWIG_ERR_STATE, "level has no blocks or was created without block_pointer");
    if (!(sp->dx > 0.0f) || sp->search_radius < 0 || sp->search_radius > 16) return fail(LUDWIG_ERR_INVALID, "bad dx or search radius");
    LW_HIP(hipSetDevice(L->device));
    {
        const int r = ensure_rho(const_cast<LudwigLevel *>(L));
        if (r) return r;
    }
    // scratch: [block_pointer | centers | normals | 4 outputs]; a diagnostics call every few hundred steps, so allocated per call
    const size_t nptr = L->h_block_pointer.size(), n = (size_t)n_tri;
    char *buf = nullptr;
    const size_t bytes = nptr * 4 + n * 10 * 4;
    LW_HIP(hipMalloc((void **)&buf, bytes));
    int32_t *d_ptr = (int32_t *)buf;
    float *d_c = (float *)(buf + nptr * 4), *d_n = d_c + 3 * n, *d_out = d_n + 3 * n;
    hipError_t e = hipMemcpyAsync(d_ptr, L->h_block_pointer.data(), nptr * 4, hipMemcpyHostToDevice, L->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_c, centers, n * 12, hipMemcpyHostToDevice, L->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_n, normals, n * 12, hipMemcpyHostToDevice, L->stream);
    if (e == hipSuccess) {
        SurfaceParams s{};
        s.rho = L->rho;
        s.vel = L->vel[vel_field == LUDWIG_VEL ? 0 : 1];
        s.obstacle = L->obstacle;
        s.block_pointer = d_ptr;
        s.gdx = L->gdx; s.gdy = L->gdy; s.gdz = L->gdz; s.n_tri = n_tri; s.radius = sp->search_radius;
        s.dx = sp->dx; s.tau = sp->tau; s.off_x = sp->offset_x; s.off_y = sp->offset_y; s.off_z = sp->offset_z;
        s.pressure_scale = sp->pressure_scale; s.stress_scale = sp->stress_scale;
        s.centers = d_c; s.normals = d_n;
        s.p = d_out; s.tx = d_out + n; s.ty = d_out + 2 * n; s.tz = d_out + 3 * n;
        hipLaunchKernelGGL(k_map_stresses, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, L->stream, s);
        e = hipGetLastError();
    }
    float *outs[4] = {pressure, shear_x, shear_y, shear_z};
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipMemcpyAsync(outs[i], d_out + (size_t)i * n, n * 4, hipMemcpyDeviceToHost, L->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(L->stream);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail(LUDWIG_ERR_HIP, "surface stresses: %s", hipGetErrorString(e));
    return LUDWIG_OK;
}

int ludwig_level_rho_min(const LudwigLevel *L, float *rho_min)
{
    if (!L || !rho_min) return fail(LUDWIG_ERR_INVALID, "null argument");
    *rho_min = __builtin_inff();
    if (L->n_owned == 0) return LUDWIG_OK;
    LW_HIP(hipSetDevice(L->device));
    {
        const int r = ensure_rho(const_cast<LudwigLevel *>(L));
        if (r) return r;
    }
    int *d = nullptr;
    LW_HIP(hipMalloc((void **)&d, 2 * sizeof(int)));
    const int init[2] = {0x7f800000, 0};                  // +inf, "no NaN seen"
    hipError_t e = hipMemcpyAsync(d, init, sizeof init, hipMemcpyHostToDevice, L->stream);
    const int64_t n = (int64_t)L->n_owned * CELLS;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_rho_min, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 2048)), dim3(256), 0, L->stream, L->rho, L->obstacle, n, d);
        e = hipGetLastError();
    }
    int res[2] = {init[0], 0};
    if (e == hipSuccess) e = hipMemcpyAsync(res, d, sizeof res, hipMemcpyDeviceToHost, L->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(L->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(LUDWIG_ERR_HIP, "rho_min: %s", hipGetErrorString(e));
    memcpy(rho_min, &res[0], sizeof(float));
    if (res[1]) *rho_min = __builtin_nanf("");            // minimum() of the reference propagates NaN (src/diagnostics.jl:71)
    return LUDWIG_OK;
}

int ludwig_halo_pack(const LudwigLevel *L, int field, const int64_t *index_dev, int64_t n, float *dst_dev, void *hip_stream)
{
    if (!L || (n > 0 && (!index_dev || !dst_dev))) return fail(LUDWIG_ERR_INVALID, "null argument");
    if (n == 0) return LUDWIG_OK;
    const FieldDesc d = field_desc(L, field);
    if (!d.ptr || field == LUDWIG_OBSTACLE) return fail(LUDWIG_ERR_STATE, "field %d cannot be packed", field);
    LW_HIP(hipSetDevice(L->device));
    if (field == LUDWIG_RHO) {
        bool stale = false;
        for (int a = 0; a < N_PARTS; ++a) stale = stale || L->rho_replay[a].stale;
        const int r = ensure_rho(const_cast<LudwigLevel *>(L));
        if (r) return r;
        // the replay ran on the level's stream: a pack queued on another stream must not overtake it
        if (stale && hip_stream && (hipStream_t)hip_stream != L->stream) LW_HIP(hipStreamSynchronize(L->stream));
    }
    hipLaunchKernelGGL(k_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hip_stream ? (hipStream_t)hip_stream : L->stream, (const float *)d.ptr, index_dev, n, dst_dev, (const int32_t *)L->d_ref2int, L->sk, d.comps);
    LW_HIP(hipGetLastError());
    return LUDWIG_OK;
}

int ludwig_halo_unpack(LudwigLevel *L, int field, const int64_t *index_dev, int64_t n, const float *src_dev, void *hip_stream)
{
    if (!L || (n > 0 && (!index_dev || !src_dev))) return fail(LUDWIG_ERR_INVALID, "null argument");
    if (n == 0) return LUDWIG_OK;
    ++L->version;
    {
        const bool aliased = L->old_alias >= 0;
        const int r = before_external_write(L, field);
        if (r) return r;
        if (aliased && L->old_alias < 0 && hip_stream && (hipStream_t)hip_stream != L->stream) LW_HIP(hipStreamSynchronize(L->stream));
    }
    const FieldDesc d = field_desc(L, field);
    if (!d.ptr || field == LUDWIG_OBSTACLE) return fail(LUDWIG_ERR_STATE, "field %d cannot be unpacked", field);
    LW_HIP(hipSetDevice(L->device));
    hipLaunchKernelGGL(k_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, hip_stream ? (hipStream_t)hip_stream : L->stream, (float *)d.ptr, index_dev, n, src_dev, (const int32_t *)L->d_ref2int, L->sk, d.comps);
    LW_HIP(hipGetLastError());
    return LUDWIG_OK;
}

int ludwig_level_info(const LudwigLevel *L, LudwigLevelInfo *info)
{
    if (!L || !info) return fail(LUDWIG_ERR_INVALID, "null argument");
    info->n_blocks = L->n_blocks;
    info->n_owned = L->n_owned;
    info->n_fast_blocks = L->n_fast_blocks;
    info->n_general_blocks = L->n_owned - L->n_fast_blocks;
    info->n_boundary_cells = L->n_bc;
    info->has_temporal_storage = L->has_temporal;
    info->has_post_collision = L->has_post;
    info->n_xrun_blocks = (int32_t)(L->n_linked_items[LUDWIG_PART_ALL] / 8);   // 8 planes per block
    info->device_bytes = L->device_bytes;
    return LUDWIG_OK;
}

// ============================================================================================================================
// Multi-GPU: communicator, halo plan, exchange, distributed step (include/ludwig_hip.h "multi-GPU")
// ============================================================================================================================
}  // extern "C"  (helpers below are internal)

#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>      // types and prototypes only: the functions are resolved at run time, the library is not linked

namespace {

struct RcclApi {
    void *handle = nullptr;
    std::string path;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

int find_loaded_rccl(struct dl_phdr_info *info, size_t, void *data)
{
    if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) {
        *static_cast<std::string *>(data) = info->dlpi_name;
        return 1;
    }
    return 0;
}

// The RCCL already mapped into the process wins (a host that also uses PyTorch has torch's copy loaded, and two RCCLs in one process
// each bring their own kernels, proxy threads and IPC state); else LUDWIG_RCCL_LIB, else the loader's librccl.so.1 / librccl.so.
RcclApi *rccl()
{
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);
    const char *env = getenv("LUDWIG_RCCL_LIB");
    std::vector<std::string> names;
    if (!loaded.empty()) names.push_back(loaded);
    if (env) names.push_back(env);
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    for (const std::string &n : names) {
        void *h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) continue;
        api.handle = h;
        api.path = n;
        break;
    }
    if (!api.handle) { g_err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return nullptr; }
#define LW_SYM(f) api.f = reinterpret_cast<decltype(api.f)>(dlsym(api.handle, "nccl" #f)); if (!api.f) { g_err = "librccl lacks nccl" #f; api.handle = nullptr; return nullptr; }
    LW_SYM(GetUniqueId) LW_SYM(CommInitRank) LW_SYM(CommDestroy) LW_SYM(GroupStart) LW_SYM(GroupEnd) LW_SYM(Send) LW_SYM(Recv)
    LW_SYM(AllReduce) LW_SYM(GetErrorString)
#undef LW_SYM
    return &api;
}

#define LW_NCCL(api, call)                                                                                          \
    do {                                                                                                            \
        ncclResult_t r_ = (call);                                                                                   \
        if (r_ != ncclSuccess) return fail(LUDWIG_ERR_HIP, "%s failed: %s (%s:%d)", #call, (api)->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

constexpr int HALO_GROUP_COMPS[LUDWIG_HALO_GROUPS] = {Q, 3, Q, 1};      // populations, velocity, f_post_collision, rho
constexpr int TIMING_RING = 256;

}  // namespace

struct LudwigComm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;       // small collectives (diagnostics)
    float *scratch = nullptr;           // device, 2 x 256 floats
};

struct LudwigHaloPlan {
    LudwigLevel *L = nullptr;
    LudwigComm *comm = nullptr;
    std::vector<int32_t> peers;
    struct Group {
        int64_t n_send = 0, n_recv = 0, n_send_oct = 0, n_recv_oct = 0, n_send_single = 0, n_recv_single = 0;
        float *send_buf = nullptr, *recv_buf = nullptr;
        uint4 *send_desc = nullptr, *recv_desc = nullptr;
        uint2 *send_single = nullptr, *recv_single = nullptr;
        std::vector<int64_t> send_off, recv_off;       // [n_peers + 1] prefix sums
    } g[LUDWIG_HALO_GROUPS];
    hipStream_t s_comm = nullptr;                      // high priority: a queue of its own beside the stepping stream
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    bool pending = false;                              // an exchange has been queued since the last wait
    bool self_via_rccl = false;                        // LUDWIG_HALO_SELF_VIA_RCCL: messages to this rank itself go through ncclSend / ncclRecv
    bool timing = false;
    bool in_stream = false;                            // ludwig_halo_plan_in_stream: the exchange runs ON the level's stream (no event hand-over)
    hipEvent_t t0[TIMING_RING] = {}, t1[TIMING_RING] = {};
    int64_t n_timed = 0, n_read = 0;
};

namespace {

// message elements (reference-layout offsets, in message order) -> octet descriptors of the device array (kernels.hpp: k_pack_octets)
int build_octets(const LudwigLevel *L, int K, const int64_t *index, int64_t n, std::vector<uint4> &out, std::vector<uint2> &singles)
{
    out.clear();
    singles.clear();
    const int64_t sk = L->sk, total = sk * K;
    int64_t cur_oct = -1;
    int last_j = -1;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t off = index[i];
        if (off < 0 || off >= total) return fail(LUDWIG_ERR_INVALID, "halo plan: element offset %lld outside the field (%lld elements)", (long long)off, (long long)total);
        const int64_t k = off / sk, r = off - k * sk;
        const int64_t blk = L->ref2int.empty() ? (r >> 9) : (int64_t)L->ref2int[(size_t)(r >> 9)];
        const int64_t in = (blk * K + k) * CELLS + (r & 511);
        const int64_t oct = in >> 3;
        const int j = (int)(in & 7);
        if (oct != cur_oct || j <= last_j) {             // a new sector, or not ascending inside it: a new descriptor
            if (oct >= ((int64_t)1 << 32) || i >= ((int64_t)1 << 32)) return fail(LUDWIG_ERR_INVALID, "halo plan: level too large for 32-bit octet descriptors");
            out.push_back(make_uint4((uint32_t)oct, (uint32_t)i, 0u, 0u));
            cur_oct = oct;
        }
        out.back().z |= 1u << j;
        last_j = j;
    }
    // octets with one or two members: one thread per member instead of eight per octet (kernels.hpp)
    if (n >= ((int64_t)1 << 29)) return fail(LUDWIG_ERR_INVALID, "halo plan: message too long for 29-bit positions");
    std::vector<uint4> full;
    full.reserve(out.size());
    for (const uint4 &d : out) {
        if (__builtin_popcount(d.z) >= 3) { full.push_back(d); continue; }
        uint32_t pos = d.y;
        for (uint32_t j = 0; j < 8; ++j)
            if ((d.z >> j) & 1u) singles.push_back(make_uint2(d.x, (pos++ << 3) | j));
    }
    out.swap(full);
    return LUDWIG_OK;
}

int halo_pack_group(LudwigHaloPlan *P, int group, int field, hipStream_t st)
{
    LudwigLevel *L = P->L;
    LudwigHaloPlan::Group &G = P->g[group];
    if (G.n_send == 0) return LUDWIG_OK;
    const FieldDesc d = field_desc(L, field);
    if (!d.ptr || d.es != 4 || d.comps != HALO_GROUP_COMPS[group]) return fail(LUDWIG_ERR_INVALID, "halo group %d cannot move field %d", group, field);
    hipLaunchKernelGGL(k_pack_octets, dim3((unsigned)((G.n_send_oct * 8 + G.n_send_single + 255) / 256)), dim3(256), 0, st, (const float *)d.ptr, G.send_desc,
                       G.n_send_oct, G.send_single, G.n_send_single, G.send_buf);
    LW_HIP(hipGetLastError());
    return LUDWIG_OK;
}

int halo_unpack_group(LudwigHaloPlan *P, int group, int field, hipStream_t st)
{
    LudwigLevel *L = P->L;
    LudwigHaloPlan::Group &G = P->g[group];
    if (G.n_recv == 0) return LUDWIG_OK;
    const FieldDesc d = field_desc(L, field);
    if (!d.ptr || d.es != 4 || d.comps != HALO_GROUP_COMPS[group]) return fail(LUDWIG_ERR_INVALID, "halo group %d cannot move field %d", group, field);
    hipLaunchKernelGGL(k_unpack_octets, dim3((unsigned)((G.n_recv_oct * 8 + G.n_recv_single + 255) / 256)), dim3(256), 0, st, (float *)d.ptr, G.recv_desc,
                       G.n_recv_oct, G.recv_single, G.n_recv_single, G.recv_buf);
    LW_HIP(hipGetLastError());
    return LUDWIG_OK;
}

// what has to be true on the LEVEL's stream before an exchange may read `field` (pack side) / write its ghosts (unpack side)
int halo_prepare_field(LudwigLevel *L, int group, int field)
{
    if (field == LUDWIG_RHO) {
        const int r = ensure_rho(L);                    // an elided rho is produced now, on the level's stream
        if (r) return r;
    }
    (void)group;
    return before_external_write(L, field);             // the ghosts of `field` are about to be written: saved-state alias, lazy rho inputs
}

}  // namespace

extern "C" {

int ludwig_comm_unique_id(void *id_out)
{
    if (!id_out) return fail(LUDWIG_ERR_INVALID, "null argument");
    RcclApi *api = rccl();
    if (!api) return fail(LUDWIG_ERR_STATE, "RCCL unavailable: %s", g_err.c_str());
    static_assert(sizeof(ncclUniqueId) == LUDWIG_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    LW_NCCL(api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return LUDWIG_OK;
}

int ludwig_comm_create(const void *unique_id, int rank, int world, int device, LudwigComm **out)
{
    if (out) *out = nullptr;
    if (!unique_id || !out || world < 1 || rank < 0 || rank >= world) return fail(LUDWIG_ERR_INVALID, "bad argument");
    RcclApi *api = rccl();
    if (!api) return fail(LUDWIG_ERR_STATE, "RCCL unavailable: %s", g_err.c_str());
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(LUDWIG_ERR_NO_DEVICE, "no HIP device %d", device);
    LW_HIP(hipSetDevice(device));
    LudwigComm *c = new (std::nothrow) LudwigComm();
    if (!c) return fail(LUDWIG_ERR_ALLOC, "host allocation failed");
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclResult_t r = api->CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) { delete c; return fail(LUDWIG_ERR_HIP, "ncclCommInitRank: %s", api->GetErrorString(r)); }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&c->scratch, 512 * sizeof(float));
    if (e != hipSuccess) { ludwig_comm_destroy(c); return fail(LUDWIG_ERR_HIP, "communicator set-up: %s", hipGetErrorString(e)); }
    *out = c;
    return LUDWIG_OK;
}

void ludwig_comm_destroy(LudwigComm *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    RcclApi *api = rccl();
    if (c->comm && api) (void)api->CommDestroy(c->comm);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int ludwig_comm_allreduce_f32(LudwigComm *c, float *buf, int32_t n, int32_t op)
{
    if (!c || !buf || n < 0 || n > 256) return fail(LUDWIG_ERR_INVALID, "bad argument (at most 256 values)");
    if (op != 0 && op != 2 && op != 3) return fail(LUDWIG_ERR_INVALID, "op must be 0 (sum), 2 (max) or 3 (min)");
    if (n == 0) return LUDWIG_OK;
    RcclApi *api = rccl();
    if (!api) return fail(LUDWIG_ERR_STATE, "RCCL unavailable");
    LW_HIP(hipSetDevice(c->device));
    // min / max: what the backend makes of a NaN is its own business, so NaN travels as a flag (second reduction, max)
    float host[512];
    for (int i = 0; i < n; ++i) {
        const bool nan = buf[i] != buf[i];
        host[i] = (nan && op != 0) ? (op == 3 ? __builtin_inff() : -__builtin_inff()) : buf[i];
        host[256 + i] = nan ? 1.0f : 0.0f;
    }
    LW_HIP(hipMemcpyAsync(c->scratch, host, 512 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    LW_NCCL(api, api->AllReduce(c->scratch, c->scratch, (size_t)n, ncclFloat, (ncclRedOp_t)op, c->comm, c->stream));
    if (op != 0) LW_NCCL(api, api->AllReduce(c->scratch + 256, c->scratch + 256, (size_t)n, ncclFloat, ncclMax, c->comm, c->stream));
    LW_HIP(hipMemcpyAsync(host, c->scratch, 512 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    LW_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i) buf[i] = (op != 0 && host[256 + i] > 0.0f) ? __builtin_nanf("") : host[i];
    return LUDWIG_OK;
}

void ludwig_halo_plan_destroy(LudwigHaloPlan *P)
{
    if (!P) return;
    if (P->L) (void)hipSetDevice(P->L->device);
    if (P->s_comm) (void)hipStreamSynchronize(P->s_comm);
    for (auto &G : P->g) {
        void *ptrs[] = {G.send_buf, G.recv_buf, G.send_desc, G.recv_desc, G.send_single, G.recv_single};
        for (void *q : ptrs)
            if (q) (void)hipFree(q);
    }
    for (int i = 0; i < TIMING_RING; ++i) {
        if (P->t0[i]) (void)hipEventDestroy(P->t0[i]);
        if (P->t1[i]) (void)hipEventDestroy(P->t1[i]);
    }
    if (P->ev_ready) (void)hipEventDestroy(P->ev_ready);
    if (P->ev_done) (void)hipEventDestroy(P->ev_done);
    if (P->s_comm) (void)hipStreamDestroy(P->s_comm);
    delete P;
}

int ludwig_halo_plan_create(LudwigLevel *L, LudwigComm *comm, const LudwigHaloPlanDesc *desc, LudwigHaloPlan **out)
{
    if (out) *out = nullptr;
    if (!L || !desc || !out || desc->n_peers < 0 || (desc->n_peers > 0 && !desc->peer_ranks)) return fail(LUDWIG_ERR_INVALID, "bad argument");
    if (comm && comm->device != L->device) return fail(LUDWIG_ERR_INVALID, "communicator and level live on different devices");
    for (int p = 0; p < desc->n_peers; ++p) {
        const int r = desc->peer_ranks[p];
        if (comm ? (r < 0 || r >= comm->world) : r != 0) return fail(LUDWIG_ERR_INVALID, "peer %d: rank %d not in the communicator", p, r);
    }
    LW_HIP(hipSetDevice(L->device));
    LudwigHaloPlan *P = new (std::nothrow) LudwigHaloPlan();
    if (!P) return fail(LUDWIG_ERR_ALLOC, "host allocation failed");
    P->L = L; P->comm = comm;
    P->peers.assign(desc->peer_ranks, desc->peer_ranks + desc->n_peers);
    P->self_via_rccl = comm && getenv("LUDWIG_HALO_SELF_VIA_RCCL") != nullptr;
    int rc = LUDWIG_OK;
    auto upload = [&](const auto &v, auto **dst) -> int {
        *dst = nullptr;
        if (v.empty()) return LUDWIG_OK;
        LW_HIP(hipMalloc((void **)dst, v.size() * sizeof(v[0])));
        LW_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice));
        return LUDWIG_OK;
    };
    for (int g = 0; g < LUDWIG_HALO_GROUPS && rc == LUDWIG_OK; ++g) {
        LudwigHaloPlan::Group &G = P->g[g];
        G.send_off.assign((size_t)desc->n_peers + 1, 0);
        G.recv_off.assign((size_t)desc->n_peers + 1, 0);
        for (int p = 0; p < desc->n_peers; ++p) {
            const int64_t ns = desc->send_count[g] ? desc->send_count[g][p] : 0, nr = desc->recv_count[g] ? desc->recv_count[g][p] : 0;
            if (ns < 0 || nr < 0) rc = fail(LUDWIG_ERR_INVALID, "negative count");
            G.send_off[(size_t)p + 1] = G.send_off[(size_t)p] + ns;
            G.recv_off[(size_t)p + 1] = G.recv_off[(size_t)p] + nr;
        }
        if (rc) break;
        G.n_send = G.send_off.back(); G.n_recv = G.recv_off.back();
        if ((G.n_send > 0 && !desc->send_index[g]) || (G.n_recv > 0 && !desc->recv_index[g])) { rc = fail(LUDWIG_ERR_INVALID, "group %d: index list missing", g); break; }
        if (g == 2 && (G.n_send || G.n_recv) && !L->has_post) { rc = fail(LUDWIG_ERR_STATE, "group 2 (f_post_collision) on a level without that array"); break; }
        std::vector<uint4> sd, rd;
        std::vector<uint2> ss, rs;
        if ((rc = build_octets(L, HALO_GROUP_COMPS[g], desc->send_index[g], G.n_send, sd, ss))) break;
        if ((rc = build_octets(L, HALO_GROUP_COMPS[g], desc->recv_index[g], G.n_recv, rd, rs))) break;
        G.n_send_oct = (int64_t)sd.size(); G.n_recv_oct = (int64_t)rd.size();
        G.n_send_single = (int64_t)ss.size(); G.n_recv_single = (int64_t)rs.size();
        if ((rc = upload(sd, &G.send_desc)) || (rc = upload(rd, &G.recv_desc)) || (rc = upload(ss, &G.send_single)) || (rc = upload(rs, &G.recv_single))) break;
        hipError_t e = hipSuccess;
        if (G.n_send) e = hipMalloc((void **)&G.send_buf, (size_t)G.n_send * 4);
        if (e == hipSuccess && G.n_recv) e = hipMalloc((void **)&G.recv_buf, (size_t)G.n_recv * 4);
        if (e != hipSuccess) rc = fail(LUDWIG_ERR_ALLOC, "halo buffers: %s", hipGetErrorString(e));
    }
    if (rc == LUDWIG_OK) {
        int pr_least = 0, pr_greatest = 0;
        hipError_t e = hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
        if (e == hipSuccess) e = hipStreamCreateWithPriority(&P->s_comm, hipStreamNonBlocking, pr_greatest);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&P->ev_ready, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&P->ev_done, hipEventDisableTiming);
        if (e != hipSuccess) rc = fail(LUDWIG_ERR_HIP, "halo plan streams: %s", hipGetErrorString(e));
    }
    if (rc) { ludwig_halo_plan_destroy(P); return rc; }
    *out = P;
    return LUDWIG_OK;
}

int ludwig_halo_plan_timing(LudwigHaloPlan *P, int enable)
{
    if (!P) return fail(LUDWIG_ERR_INVALID, "null plan");
    LW_HIP(hipSetDevice(P->L->device));
    if (enable && !P->t0[0])
        for (int i = 0; i < TIMING_RING; ++i) { LW_HIP(hipEventCreate(&P->t0[i])); LW_HIP(hipEventCreate(&P->t1[i])); }
    P->timing = enable != 0;
    P->n_read = P->n_timed;
    return LUDWIG_OK;
}

int ludwig_halo_plan_in_stream(LudwigHaloPlan *P, int enable)
{
    if (!P) return fail(LUDWIG_ERR_INVALID, "null plan");
    if (P->pending) {                    // an exchange of the other kind is in flight: the level's stream joins it first
        const int r = ludwig_halo_wait(P);
        if (r) return r;
    }
    P->in_stream = enable != 0;
    return LUDWIG_OK;
}

int ludwig_halo_plan_exchange_ms(LudwigHaloPlan *P, float *ms_out, int32_t max, int32_t *n_out)
{
    if (!P || !n_out || (max > 0 && !ms_out)) return fail(LUDWIG_ERR_INVALID, "null argument");
    *n_out = 0;
    if (P->n_timed - P->n_read > TIMING_RING) P->n_read = P->n_timed - TIMING_RING;
    while (P->n_read < P->n_timed && *n_out < max) {
        const int i = (int)(P->n_read % TIMING_RING);
        float ms = 0.0f;
        LW_HIP(hipEventElapsedTime(&ms, P->t0[i], P->t1[i]));
        ms_out[(*n_out)++] = ms;
        ++P->n_read;
    }
    return LUDWIG_OK;
}

int ludwig_halo_plan_pack(LudwigHaloPlan *P, int32_t group, int32_t field, void *hip_stream)
{
    if (!P || group < 0 || group >= LUDWIG_HALO_GROUPS) return fail(LUDWIG_ERR_INVALID, "bad argument");
    LW_HIP(hipSetDevice(P->L->device));
    if (field == LUDWIG_RHO) { const int r = ensure_rho(P->L); if (r) return r; }
    return halo_pack_group(P, group, field, hip_stream ? (hipStream_t)hip_stream : P->L->stream);
}

int ludwig_halo_plan_unpack(LudwigHaloPlan *P, int32_t group, int32_t field, void *hip_stream)
{
    if (!P || group < 0 || group >= LUDWIG_HALO_GROUPS) return fail(LUDWIG_ERR_INVALID, "bad argument");
    LW_HIP(hipSetDevice(P->L->device));
    ++P->L->version;
    { const int r = before_external_write(P->L, field); if (r) return r; }
    return halo_unpack_group(P, group, field, hip_stream ? (hipStream_t)hip_stream : P->L->stream);
}

int ludwig_halo_plan_buffers(const LudwigHaloPlan *P, int32_t group, void **send_dev, int64_t *n_send, void **recv_dev, int64_t *n_recv)
{
    if (!P || group < 0 || group >= LUDWIG_HALO_GROUPS) return fail(LUDWIG_ERR_INVALID, "bad argument");
    const LudwigHaloPlan::Group &G = P->g[group];
    if (send_dev) *send_dev = G.send_buf;
    if (n_send) *n_send = G.n_send;
    if (recv_dev) *recv_dev = G.recv_buf;
    if (n_recv) *n_recv = G.n_recv;
    return LUDWIG_OK;
}

static double host_now_us()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

int ludwig_halo_exchange(LudwigHaloPlan *P, int32_t n, const int32_t *groups, const int32_t *fields)
{
    if (!P || n < 0 || n > LUDWIG_HALO_GROUPS || (n > 0 && (!groups || !fields))) return fail(LUDWIG_ERR_INVALID, "bad argument");
    // LUDWIG_HALO_TRACE=1: host microseconds of the three parts of this call on stderr (what an exchange costs the host)
    static const bool trace = getenv("LUDWIG_HALO_TRACE") != nullptr;
    const double h0 = trace ? host_now_us() : 0.0;
    double h1 = 0.0, h2 = 0.0;
    LudwigLevel *L = P->L;
    LW_HIP(hipSetDevice(L->device));
    for (int i = 0; i < n; ++i) {
        if (groups[i] < 0 || groups[i] >= LUDWIG_HALO_GROUPS) return fail(LUDWIG_ERR_INVALID, "bad group %d", groups[i]);
        const int r = halo_prepare_field(L, groups[i], fields[i]);     // may queue work on the level's stream: before the event below
        if (r) return r;
    }
    // nothing to send or to receive in these groups (a level this rank shares with nobody): no event, no kernel
    {
        int64_t moved = 0;
        for (int i = 0; i < n; ++i) moved += P->g[groups[i]].n_send + P->g[groups[i]].n_recv;
        if (moved == 0) return LUDWIG_OK;
    }
    ++L->version;
    // in-stream mode: pack, transfer and unpack are queued on the level's own stream - what follows on that stream follows the exchange,
    // at the price of no overlap and to the gain of two cross-stream hand-overs (tens of microseconds of idle device each, more than a
    // small level's whole exchange)
    const hipStream_t xs = P->in_stream ? L->stream : P->s_comm;
    if (!P->in_stream) {
        LW_HIP(hipEventRecord(P->ev_ready, L->stream));
        LW_HIP(hipStreamWaitEvent(P->s_comm, P->ev_ready, 0));
    }
    const int slot = (int)(P->n_timed % TIMING_RING);
    if (P->timing) LW_HIP(hipEventRecord(P->t0[slot], xs));
    for (int i = 0; i < n; ++i) {
        const int r = halo_pack_group(P, groups[i], fields[i], xs);
        if (r) return r;
    }
    if (trace) h1 = host_now_us();
    RcclApi *api = P->comm ? rccl() : nullptr;
    bool any_remote = false;
    for (size_t p = 0; p < P->peers.size(); ++p)
        if (P->comm && (P->peers[p] != P->comm->rank || P->self_via_rccl)) any_remote = true;
    if (any_remote && !api) return fail(LUDWIG_ERR_STATE, "RCCL unavailable");
    if (any_remote) LW_NCCL(api, api->GroupStart());
    for (size_t p = 0; p < P->peers.size(); ++p) {
        const bool self = !P->comm || (P->peers[p] == P->comm->rank && !P->self_via_rccl);
        for (int i = 0; i < n; ++i) {
            LudwigHaloPlan::Group &G = P->g[groups[i]];
            const int64_t s0 = G.send_off[p], ns = G.send_off[p + 1] - s0, r0 = G.recv_off[p], nr = G.recv_off[p + 1] - r0;
            if (self) {
                if (ns != nr) return fail(LUDWIG_ERR_INVALID, "peer %zu is this rank itself but sends %lld and receives %lld elements", p, (long long)ns, (long long)nr);
                if (ns) LW_HIP(hipMemcpyAsync(G.recv_buf + r0, G.send_buf + s0, (size_t)ns * 4, hipMemcpyDeviceToDevice, xs));
            } else {
                if (ns) LW_NCCL(api, api->Send(G.send_buf + s0, (size_t)ns, ncclFloat, P->peers[p], P->comm->comm, xs));
                if (nr) LW_NCCL(api, api->Recv(G.recv_buf + r0, (size_t)nr, ncclFloat, P->peers[p], P->comm->comm, xs));
            }
        }
    }
    if (any_remote) LW_NCCL(api, api->GroupEnd());
    if (trace) h2 = host_now_us();
    for (int i = 0; i < n; ++i) {
        const int r = halo_unpack_group(P, groups[i], fields[i], xs);
        if (r) return r;
    }
    if (P->timing) { LW_HIP(hipEventRecord(P->t1[slot], xs)); ++P->n_timed; }
    if (!P->in_stream) {
        LW_HIP(hipEventRecord(P->ev_done, P->s_comm));
        P->pending = true;
    }
    if (trace) fprintf(stderr, "[ludwig_halo_exchange] host us: events + pack %.1f, send/recv group (%zu peers) %.1f, unpack + event %.1f\n", h1 - h0, P->peers.size(), h2 - h1, host_now_us() - h2);
    return LUDWIG_OK;
}

int ludwig_halo_wait(LudwigHaloPlan *P)
{
    if (!P) return fail(LUDWIG_ERR_INVALID, "null plan");
    if (!P->pending) return LUDWIG_OK;
    LW_HIP(hipSetDevice(P->L->device));
    LW_HIP(hipStreamWaitEvent(P->L->stream, P->ev_done, 0));
    P->pending = false;
    return LUDWIG_OK;
}

int ludwig_step_distributed(LudwigLevel *L, LudwigHaloPlan *P, const LudwigLevel *parent, int64_t t_sub, float u_curr, float parent_tau,
                            float temporal_weight, const LudwigStepFlags *fl)
{
    if (!L || !P || !fl || P->L != L) return fail(LUDWIG_ERR_INVALID, "bad argument (the plan must belong to the level)");
    int rc;
    // interior blocks read and write owned cells only: they run while the ghosts of their input are still arriving
    if ((rc = launch_stream_collide(L, parent, t_sub, u_curr, parent_tau, temporal_weight, fl, LUDWIG_PART_INTERIOR))) return rc;
    if ((rc = ludwig_halo_wait(P))) return rc;
    if ((rc = launch_stream_collide(L, parent, t_sub, u_curr, parent_tau, temporal_weight, fl, LUDWIG_PART_BOUNDARY))) return rc;
    if (L->has_post) {
        // the correction rewrites f_out from post-collision values of neighbour cells (src/bouzidi_kernel.jl:44-77): the few that
        // live across a cut are fetched in between, and waited for
        if (P->g[2].n_send || P->g[2].n_recv) {
            // a few thousand elements, needed at once: on the level's own stream (two cross-stream hand-overs cost more than the messages)
            const int32_t grp = 2, fld = LUDWIG_F_POST;
            const bool mode = P->in_stream;
            P->in_stream = true;
            rc = ludwig_halo_exchange(P, 1, &grp, &fld);
            P->in_stream = mode;
            if (rc) return rc;
        }
        if ((rc = launch_bouzidi(L, t_sub, fl->q_min_threshold))) return rc;
    }
    const bool out_temp = t_sub % 2 == 0;                    // reference src/solver_control.jl:35-41
    const int32_t grps[2] = {0, 1}, flds[2] = {out_temp ? LUDWIG_F_TEMP : LUDWIG_F, out_temp ? LUDWIG_VEL_TEMP : LUDWIG_VEL};
    return ludwig_halo_exchange(P, 2, grps, flds);           // left in flight: the next call's interior blocks run under it
}

}  // extern "C"
