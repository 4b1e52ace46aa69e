// mcrt_api.cpp -- the C-ABI of include/mcrt.h: context, uploads, frame orchestration.
// Host C++ only; kernels live in mcrt_kernels.hip.  No CPU fallback exists: every compute entry point
// needs the GPU context.
#include "../../include/mcrt.h"
#include "mcrt_internal.h"
#include "mcrt_kernels.h"
#include "mcrt_lbvh.h"

#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

using mcrt::set_error;

// (an allocation the device cannot satisfy is MCRT_ERR_NOMEM, every other HIP failure MCRT_ERR_HIP)
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); return set_error(e_ == hipErrorOutOfMemory ? MCRT_ERR_NOMEM : MCRT_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)
#define CTX_TRY(ctx) do { if (!(ctx)) return set_error(MCRT_ERR_INVALID, "null context"); HIP_TRY(hipSetDevice((ctx)->device)); } while (0)

struct Consts {   // main.cpp:23-37, rfimage.h:48-51,178-180 evaluated at run time
    float axial_res_f; double axial_res_mm, time_step_us, row_dt_us, max_travel_us; uint32_t axial_res_um, max_rows;
};
static Consts derive_consts(const mcrt_params &p)
{
    Consts c;
    c.axial_res_f = 1.45f / p.frequency;                               // main.cpp:25
    c.axial_res_mm = (double)c.axial_res_f;
    c.axial_res_um = (uint32_t)(c.axial_res_f * 1000.0f);              // main.cpp:36
    c.time_step_us = (c.axial_res_mm * 1000.0) / (double)p.speed_of_sound;   // main.cpp:118 (mm -> um is *1000, units.h:1365)
    c.row_dt_us = (double)c.axial_res_um / (double)p.speed_of_sound;   // rfimage.h:35
    c.max_travel_us = (p.depth_cm / (double)p.speed_of_sound) * 10000.0;     // main.cpp:31 (cm s/m -> us)
    c.max_rows = (uint32_t)((p.speed_of_sound * (uint32_t)c.max_travel_us) / c.axial_res_um);   // rfimage.h:180
    return c;
}

// Work set of ONE wavefront pipeline: path state, queues, rays, closest-hit words, march records (segments on request), and the
// streams it runs on (k_march of bounce b runs on a low-priority side stream beside k_trace of bounce b+1).  A context can own
// several, to trace the scan-lines of a pass as independent groups on separate streams (MCRT_GROUPS, a tuning knob: one group
// measured best).
struct Work {
    hipStream_t stream = nullptr, side[MCRT_SIDE_STREAMS] = {};   // k_march of bounce b runs on side[b % n] (n = 1, or 2 in large passes: side_streams)
    hipEvent_t ev_bounce[MCRT_MAX_BOUNCES] = {}, ev_join[MCRT_SIDE_STREAMS] = {}, ev_done = nullptr;
    float4 *d_st0 = nullptr, *d_st1 = nullptr, *d_st2 = nullptr;
    unsigned long long *d_key0 = nullptr, *d_key1 = nullptr;
    int *d_stack_ovf = nullptr; size_t ovf_cap = 0;            // traversal-stack overflow of THIS work set's walk (its launches run beside the other groups')
    uint32_t *d_q = nullptr, *d_counts = nullptr, *d_seg_count = nullptr, *d_cursors = nullptr;
    mcrt_segment *d_segs = nullptr; size_t segs_cap = 0;       // [paths][depth], only for the callers that ask for segments
    int32_t *d_hits = nullptr; size_t hits_cap = 0;            // [paths][depth], only for the callers that ask for hit indices
    float4 *d_mrec = nullptr; size_t paths = 0; uint32_t depth = 0;
};

// tuning knobs from the environment, read ONCE at mcrt_create (never on the frame path) -- and only in a process started with
// MCRT_TUNING=1 (mcrt::tuning_env): linked into someone else's program the library has its defaults and nothing else.
struct Knobs {
    uint32_t ksplit_limit = MCRT_KSPLIT_DEFAULT, trace_blocks = 0, trace_blocks_wide = 0, wide_from = 0 /* 0: the kernels' own default */, wide_max_tree_mb = 128, groups = MCRT_GROUPS_DEFAULT, march_streams = MCRT_SIDE_STREAMS_DEFAULT, march_blocks = 0;   // march_blocks 0: launch_march picks
    bool no_overlap = false, no_priority = false, no_fast_div = false, no_lean = false;
    uint32_t path_groups = MCRT_PATH_GROUPS_DEFAULT;   // ... as this many scan-line groups on their own streams: a group's accumulation runs beside the other groups' last walks
    uint32_t path_max = MCRT_PATH_MAX_DEFAULT;    // passes of at most this many paths run as ONE launch that carries every path through all of its bounces (k_path: the latency form)
    uint32_t packet_mask = MCRT_PACKET_MASK_DEFAULT, packet_from = MCRT_PACKET_FROM;   // bit b: bounce b is walked by k_trace_packet (one wavefront per packet of 64 queue neighbours), in passes of at least packet_from paths
    uint32_t march_cus = 0;                    // CUs the accumulation's side stream is confined to (0 = no mask); the mask's bit order is the driver's
    bool main_mask = false;                    // with march_cus: the walk / shade chain runs on its own stream confined to the OTHER CUs
    bool test_hooks = false;                   // MCRT_TEST_HOOKS: mcrt_debug_set_error may poison the context (tests only)
};
static Knobs read_knobs()
{
    using mcrt::tuning_env;
    Knobs k;
    if (const char *e = tuning_env("MCRT_KSPLIT_LIMIT")) { long v = atol(e); if (v >= 0 && v <= MCRT_KSPLIT_MAX) k.ksplit_limit = (uint32_t)v; }   // 0 = off
    if (const char *e = tuning_env("MCRT_TRACE_BLOCKS")) { int v = atoi(e); if (v >= 1) k.trace_blocks = (uint32_t)v; }
    if (const char *e = tuning_env("MCRT_TRACE_BLOCKS_WIDE")) { int v = atoi(e); if (v >= 1) k.trace_blocks_wide = (uint32_t)v; }
    if (const char *e = tuning_env("MCRT_WIDE_MAX_TREE_MB")) { long long v = atoll(e); if (v >= 0 && v <= 0xffffffffll) k.wide_max_tree_mb = (uint32_t)v; }
    if (const char *e = tuning_env("MCRT_WIDE_FROM")) { long long v = atoll(e); if (v >= 1 && v <= 0xffffffffll) k.wide_from = (uint32_t)v; }   // rays in a launch from which the walk takes its five-wavefront form (1: always; 4294967295: never)
    if (const char *e = tuning_env("MCRT_GROUPS")) { int v = atoi(e); if (v >= 1 && v <= 16) k.groups = (uint32_t)v; }
    if (const char *e = tuning_env("MCRT_PACKET_BOUNCES")) { long v = strtol(e, nullptr, 0); if (v >= 0) k.packet_mask = (uint32_t)v; }
    if (const char *e = tuning_env("MCRT_PATH_GROUPS")) { int v = atoi(e); if (v >= 1 && v <= 16) k.path_groups = (uint32_t)v; }
    if (const char *e = tuning_env("MCRT_PATH_MAX")) { long long v = atoll(e); if (v >= 0 && v <= 0xffffffffll) k.path_max = (uint32_t)v; }
    if (const char *e = tuning_env("MCRT_PACKET_FROM")) { long long v = atoll(e); if (v >= 0 && v <= 0xffffffffll) k.packet_from = (uint32_t)v; }
    if (const char *e = tuning_env("MCRT_MARCH_STREAMS")) { int v = atoi(e); if (v >= 1 && v <= MCRT_SIDE_STREAMS) k.march_streams = (uint32_t)v; }
    if (const char *e = tuning_env("MCRT_MARCH_BLOCKS")) { int v = atoi(e); if (v >= 1) k.march_blocks = (uint32_t)v; }
    k.no_overlap = tuning_env("MCRT_NO_OVERLAP") != nullptr; k.no_priority = tuning_env("MCRT_NO_PRIORITY") != nullptr;
    k.no_fast_div = tuning_env("MCRT_NO_FAST_DIV") != nullptr; k.no_lean = tuning_env("MCRT_NO_LEAN") != nullptr;
    if (const char *e = tuning_env("MCRT_MARCH_CUS")) { int v = atoi(e); if (v >= 0 && v <= 248) k.march_cus = (uint32_t)v; }
    k.main_mask = tuning_env("MCRT_MAIN_MASK") != nullptr;
    k.test_hooks = tuning_env("MCRT_TEST_HOOKS") != nullptr;
    return k;
}

struct mcrt_ctx {
    int device = 0;
    Knobs knobs;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::vector<Work> work;                               // one per concurrent scan-line group (see mcrt_trace_frame)
    hipEvent_t ev_start = nullptr;
    mcrt_params p{};
    Consts c{};
    // scene
    mcrt_bvh bvh{};
    mcrt_bvh4 bvh4{};
    uint32_t *d_error = nullptr;
    float4 *d_nodes = nullptr, *d_tris = nullptr, *d_mats = nullptr;
    mcrt_bvh4_node *walked_nodes = nullptr; bool walked_stale = true;   // host copy of the tree as the lane walk sees it (mcrt_get_bvh4)
    uint4 *d_nodes_walk = nullptr; uint32_t nodes_walk_cap = 0;   // the walk's child-transposed half-float nodes
    uint4 *d_meshes = nullptr;
    uint32_t *d_tri_slot = nullptr;
    float4 *d_tris_id = nullptr; uint32_t tris_id_cap = 0;      // the triangle records in id order (refresh_soa)
    uint32_t n_mesh = 0, n_mat = 0, start_mat = 0, n_cu = 256;
    int builder = MCRT_BVH_HOST_SAH; bool host_bvh_stale = false;   // device-built tree: host copies are downloaded on demand
    std::vector<uint32_t> tri_mesh;   // per-triangle mesh index of the uploaded scene (for mcrt_update_triangles)
    float scene_lo[3] = { 0, 0, 0 }, scene_hi[3] = { 0, 0, 0 };
    float spacing[3] = { 1, 1, 1 };
    bool have_scene = false;
    // texture
    float2 *d_tex = nullptr; uint32_t tex_n = 0; bool tex_finite = false;
    // transducer
    float *d_pos = nullptr, *d_dir = nullptr; uint32_t n_el = 0;
    const float *pose_pos = nullptr, *pose_dir = nullptr;      // set for the duration of mcrt_trace_frames_poses: device [F][E][3] per-frame probe poses
    float *d_pose[2] = { nullptr, nullptr }; size_t pose_cap[2] = { 0, 0 };   // staging for pose tables handed over as host memory:
    float *h_pose[2] = { nullptr, nullptr }; hipEvent_t ev_pose = nullptr; bool pose_copy_pending = false;   // the caller's table is copied into pinned memory the context owns before the call returns
    hipEvent_t ev_scene = nullptr; hipStream_t scene_stream = nullptr; bool scene_pending = false;   // the last scene update's device work (refresh_soa), for traces issued on ANOTHER stream
    // accumulators
    long long *d_acc = nullptr; uint32_t *d_flags = nullptr; size_t acc_cap = 0, flag_cap = 0;
    uint32_t acc_clean_ne = 0, acc_clean_rows = 0;   // bins known to be all-zero for this shape (k_finalize leaves them so)
    float *d_tmp = nullptr; size_t tmp_cap = 0;
    // row thresholds (exact replacement of the per-echo double division) and the verified fast division by tex_res
    double *d_row_thr = nullptr; uint32_t thr_rows = 0; double thr_dt = 0.0;
    float verified_res = 0.0f; bool fast_div = false, fast_div_all = false;
    float last_lean_bound = 0.0f; uint32_t last_march_rows = 0;   // what the last frame's kernels were given (mcrt_debug_fast_paths)
    // per-material table of k_march (depends on the materials, the axial step and the frequency)
    float4 *d_mtab = nullptr; uint32_t mtab_n = 0; float mtab_key[2] = { 0.0f, 0.0f }; bool mtab_valid = false;
    // scan-conversion maps
    float *d_map_col = nullptr, *d_map_row = nullptr; uint32_t map_key[6] = { 0, 0, 0, 0, 0, 0 }; double map_keyd[2] = { 0, 0 };
    // instrumentation
    unsigned long long *d_stats = nullptr; bool stats_on = false;
    bool timing_on = false; int timing_level = 0;       // 1: the walk's launches are bracketed by HIP events; 2: k_shade's and k_march's too
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; std::vector<unsigned char> ev_kind; size_t ev_used = 0;
};


static int prepare_tables(mcrt_ctx *c)
{
    if (c->thr_rows != c->p.n_rows || c->thr_dt != c->c.row_dt_us || !c->d_row_thr) {
        std::vector<double> thr((size_t)c->p.n_rows + 1);
        { int rc = mcrt_row_thresholds(c->c.row_dt_us, c->p.n_rows, thr.data()); if (rc) return rc; }
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->d_row_thr); c->d_row_thr = nullptr;
        HIP_TRY(hipMalloc(&c->d_row_thr, thr.size() * 8));
        HIP_TRY(hipMemcpy(c->d_row_thr, thr.data(), thr.size() * 8, hipMemcpyHostToDevice));
        c->thr_rows = c->p.n_rows; c->thr_dt = c->c.row_dt_us;
    }
    if (c->verified_res != c->p.tex_res) {
        // the GPU checks, exhaustively, that its fma-corrected reciprocal multiply IS IEEE division by tex_res
        unsigned long long *d_bad = nullptr, bad = 1;
        HIP_TRY(hipMalloc(&d_bad, 8));
        HIP_TRY(hipMemsetAsync(d_bad, 0, 8, c->stream));
        HIP_TRY(mcrt::launch_verify_div(c->p.tex_res, 1.0f / c->p.tex_res, d_bad, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
        hipFree(d_bad);
        c->fast_div = (bad == 0) && !c->knobs.no_fast_div;
        c->fast_div_all = c->fast_div && c->p.tex_res > 1e-16f && !c->knobs.no_lean;
        c->verified_res = c->p.tex_res;
    }
    if (c->have_scene && (!c->mtab_valid || c->mtab_key[0] != c->c.axial_res_f || c->mtab_key[1] != c->p.frequency)) {
        if (c->mtab_n < c->n_mat) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            hipFree(c->d_mtab); c->d_mtab = nullptr; c->mtab_n = 0;
            HIP_TRY(hipMalloc(&c->d_mtab, 16 * (size_t)c->n_mat));
            c->mtab_n = c->n_mat;
        }
        HIP_TRY(mcrt::launch_material_table(c->d_mats, c->n_mat, c->c.axial_res_f, c->p.frequency, c->d_mtab, c->stream));
        c->mtab_key[0] = c->c.axial_res_f; c->mtab_key[1] = c->p.frequency; c->mtab_valid = true;
    }
    return MCRT_OK;
}

extern "C" int mcrt_version(void) { return MCRT_VERSION; }
extern "C" int mcrt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int mcrt_default_params(mcrt_params *p)
{
    if (!p) return set_error(MCRT_ERR_INVALID, "null params");
    memset(p, 0, sizeof *p);
    p->n_elements = 512; p->n_samples = 5; p->max_depth = 10; p->n_rows = 465;
    p->frequency = 4.5f; p->intensity_epsilon = 1e-10f; p->initial_intensity = 1.0f; p->ray_start_offset = 0.1f;
    p->speed_of_sound = 1500; p->depth_cm = 15.0; p->seed = 0x5EED; p->sanitize_tir = 0; p->tex_n = 256; p->tex_res = 0.145f;
    return MCRT_OK;
}

extern "C" int mcrt_create(int device, mcrt_ctx **out)
{
    if (!out) return set_error(MCRT_ERR_INVALID, "null out pointer");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return set_error(MCRT_ERR_NO_DEVICE, "no HIP device visible: libmcrt_hip has no CPU fallback");
    if (device < 0 || device >= n) return set_error(MCRT_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (!strstr(prop.gcnArchName, "gfx950"))
        return set_error(MCRT_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only", device, prop.gcnArchName);
    mcrt_ctx *c = new (std::nothrow) mcrt_ctx();
    if (!c) return set_error(MCRT_ERR_NOMEM, "out of host memory");
    c->device = device;
    c->knobs = read_knobs();
    if (prop.multiProcessorCount > 0) c->n_cu = (uint32_t)prop.multiProcessorCount;
    if (c->knobs.march_cus + 8u > c->n_cu) c->knobs.march_cus = c->n_cu > 8u ? c->n_cu - 8u : 0u;   // (the CU-mask knobs always leave both sides at least 8 CUs)
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return set_error(MCRT_ERR_HIP, "hipStreamCreate failed"); }
    c->stream = c->own_stream;
    hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming);
    c->work.reserve(16);   // pointers into this vector are held across get_work() calls; never more than 16 groups
    mcrt_default_params(&c->p);
    c->c = derive_consts(c->p);
    c->stream = c->own_stream;
    if (hipMalloc(&c->d_stats, MCRT_STATS_WORDS * sizeof(unsigned long long)) != hipSuccess || hipMemsetAsync(c->d_stats, 0, MCRT_STATS_WORDS * sizeof(unsigned long long), c->stream) != hipSuccess ||
        hipMalloc(&c->d_error, 4) != hipSuccess || hipMemsetAsync(c->d_error, 0, 4, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        hipStreamDestroy(c->own_stream); delete c; return set_error(MCRT_ERR_HIP, "hipMalloc failed");
    }
    { int rc = prepare_tables(c); if (rc) { mcrt_destroy(c); return rc; } }
    *out = c;
    return MCRT_OK;
}

static void free_work_buffers(Work &w)
{
    hipFree(w.d_st0); hipFree(w.d_st1); hipFree(w.d_st2); hipFree(w.d_key0); hipFree(w.d_key1);
    hipFree(w.d_q); hipFree(w.d_counts); hipFree(w.d_seg_count); hipFree(w.d_cursors); w.d_cursors = nullptr; hipFree(w.d_segs); hipFree(w.d_hits); hipFree(w.d_mrec);
    w.d_st0 = w.d_st1 = w.d_st2 = nullptr; w.d_key0 = w.d_key1 = nullptr; w.d_q = w.d_counts = w.d_seg_count = nullptr;
    w.d_segs = nullptr; w.segs_cap = 0; w.d_hits = nullptr; w.hits_cap = 0; w.d_mrec = nullptr; w.paths = 0; w.depth = 0;
}

static void free_work(mcrt_ctx *c)
{
    for (Work &w : c->work) {
        free_work_buffers(w);
        hipFree(w.d_stack_ovf); w.d_stack_ovf = nullptr; w.ovf_cap = 0;
        for (int i = 0; i < MCRT_MAX_BOUNCES; i++) { if (w.ev_bounce[i]) hipEventDestroy(w.ev_bounce[i]); }
        for (int i = 0; i < MCRT_SIDE_STREAMS; i++) { if (w.ev_join[i]) hipEventDestroy(w.ev_join[i]); if (w.side[i]) hipStreamDestroy(w.side[i]); }
        if (w.ev_done) hipEventDestroy(w.ev_done);
        if (w.stream) hipStreamDestroy(w.stream);
    }
    c->work.clear();
}

// work set g (created on first use).  Streams are created only when a pipeline asks for them (work_stream / side_stream):
// HIP multiplexes streams onto a few hardware queues, where one stream's event wait holds up whatever shares its queue, so a
// context keeps no stream it does not use.
static int get_work(mcrt_ctx *c, size_t g, Work **out)
{
    while (c->work.size() <= g) {
        Work w;
        for (int i = 0; i < MCRT_SIDE_STREAMS; i++) HIP_TRY(hipEventCreateWithFlags(&w.ev_join[i], hipEventDisableTiming));
        for (int i = 0; i < MCRT_MAX_BOUNCES; i++) HIP_TRY(hipEventCreateWithFlags(&w.ev_bounce[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&w.ev_done, hipEventDisableTiming));
        c->work.push_back(w);
    }
    *out = &c->work[g];
    return MCRT_OK;
}
// a stream confined to CUs [lo, hi) of the device (hipExtStreamCreateWithCUMask; bit i of the mask = CU i in the driver's numbering)
static int masked_stream(mcrt_ctx *c, uint32_t lo, uint32_t hi, hipStream_t *out)
{
    uint32_t mask[16] = {};
    const uint32_t words = (c->n_cu + 31u) / 32u;
    for (uint32_t i = lo; i < hi && i < c->n_cu; i++) mask[i >> 5] |= 1u << (i & 31u);
    HIP_TRY(hipExtStreamCreateWithCUMask(out, words, mask));
    return MCRT_OK;
}

// the stream of scan-line group g of the wavefront pipeline: group 0 runs on the context's stream, the others on their own
static int work_stream(mcrt_ctx *c, Work &w, bool first, hipStream_t *out)
{
    const bool masked = c->knobs.march_cus && c->knobs.main_mask;
    if (first && !masked) { *out = c->stream; return MCRT_OK; }
    if (!w.stream && masked) { int rc = masked_stream(c, 0, c->n_cu - c->knobs.march_cus, &w.stream); if (rc) return rc; }
    if (!w.stream) HIP_TRY(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    *out = w.stream;
    return MCRT_OK;
}
// k_march runs beside the walk on a LOW-priority stream: k_trace / k_shade are the critical chain, and their workgroups must
// not queue behind k_march's (measured: k_shade took 0.4-0.7 ms instead of 0.1 ms when they did)
static int side_stream(mcrt_ctx *c, Work &w, uint32_t i, hipStream_t *out)
{
    if (!w.side[i] && c->knobs.march_cus) { int rc = masked_stream(c, c->n_cu - c->knobs.march_cus, c->n_cu, &w.side[i]); if (rc) return rc; }
    if (!w.side[i]) {
        int prio_low = 0, prio_high = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        if (c->knobs.no_priority) prio_low = 0;   // tuning knob
        HIP_TRY(hipStreamCreateWithPriority(&w.side[i], hipStreamNonBlocking, prio_low));
    }
    *out = w.side[i];
    return MCRT_OK;
}

// the walk's view of the tree: child-transposed half-float nodes, rebuilt whenever d_nodes changes.  The buffer is kept while the
// node count stays (a refit -- the per-frame path of a deforming scene -- then costs one kernel on the context's stream and no
// allocation HERE; mcrt_refit_triangles itself still frees its staging copy of the vertices, which synchronises the device).
// Nothing here waits: the rebuild is ordered on the stream it was issued on, and an event recorded behind it orders a trace that
// is issued on ANOTHER stream after mcrt_set_stream (enqueue_frame waits for it).
static int refresh_soa(mcrt_ctx *c)
{
    c->walked_stale = true;
    if (c->bvh4.n_nodes == 0) { hipFree(c->d_nodes_walk); c->d_nodes_walk = nullptr; c->nodes_walk_cap = 0; return MCRT_OK; }
    if (c->nodes_walk_cap != c->bvh4.n_nodes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->d_nodes_walk); c->d_nodes_walk = nullptr; c->nodes_walk_cap = 0;
        HIP_TRY(hipMalloc(&c->d_nodes_walk, 64 * (size_t)c->bvh4.n_nodes));
        c->nodes_walk_cap = c->bvh4.n_nodes;
    }
    HIP_TRY(mcrt::launch_nodes_walk(c->d_nodes, c->bvh4.n_nodes, c->d_nodes_walk, c->stream));
    if (c->tris_id_cap != c->bvh.n_tri) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->d_tris_id); c->d_tris_id = nullptr; c->tris_id_cap = 0;
        if (c->bvh.n_tri) { HIP_TRY(hipMalloc(&c->d_tris_id, 16 * (size_t)MCRT_TRI_PIECES * c->bvh.n_tri)); c->tris_id_cap = c->bvh.n_tri; }
    }
    HIP_TRY(mcrt::launch_tris_by_id((const float4 *)c->d_tris, c->bvh.n_tri, c->d_tris_id, c->stream));
    if (!c->ev_scene) HIP_TRY(hipEventCreateWithFlags(&c->ev_scene, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->ev_scene, c->stream));
    c->scene_stream = c->stream; c->scene_pending = true;
    return MCRT_OK;
}

static void free_scene(mcrt_ctx *c)
{
    free(c->walked_nodes); c->walked_nodes = nullptr; c->walked_stale = true;
    hipFree(c->d_tris_id); c->d_tris_id = nullptr; c->tris_id_cap = 0;
    hipFree(c->d_nodes_walk); c->d_nodes_walk = nullptr; c->nodes_walk_cap = 0;
    hipFree(c->d_nodes); hipFree(c->d_tris); hipFree(c->d_mats); hipFree(c->d_meshes); hipFree(c->d_tri_slot); c->d_tri_slot = nullptr;
    c->d_nodes = c->d_tris = c->d_mats = nullptr; c->d_meshes = nullptr;
    mcrt_free_bvh(&c->bvh);
    mcrt_free_bvh4(&c->bvh4);
    c->have_scene = false;
}

extern "C" int mcrt_destroy(mcrt_ctx *c)
{
    if (!c) return MCRT_OK;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    free_scene(c);
    free_work(c);
    free(c->walked_nodes); c->walked_nodes = nullptr;
    hipFree(c->d_pose[0]); hipFree(c->d_pose[1]);
    if (c->h_pose[0]) hipHostFree(c->h_pose[0]);
    if (c->h_pose[1]) hipHostFree(c->h_pose[1]);
    if (c->ev_pose) hipEventDestroy(c->ev_pose);
    if (c->ev_scene) hipEventDestroy(c->ev_scene);
    hipFree(c->d_tex); hipFree(c->d_pos); hipFree(c->d_dir); hipFree(c->d_acc); hipFree(c->d_flags); hipFree(c->d_tmp);
    hipFree(c->d_map_col); hipFree(c->d_map_row); hipFree(c->d_stats); hipFree(c->d_row_thr); hipFree(c->d_error); hipFree(c->d_mtab);
    for (auto &e : c->ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
    if (c->ev_start) hipEventDestroy(c->ev_start);
    hipStreamDestroy(c->own_stream);
    delete c;
    return MCRT_OK;
}

namespace mcrt { hipStream_t ctx_stream(mcrt_ctx *c) { return c->stream; } }   // (mcrt_group.cpp: the root context's stream)
extern "C" int mcrt_set_stream(mcrt_ctx *c, void *s) { CTX_TRY(c); c->stream = s ? (hipStream_t)s : c->own_stream; return MCRT_OK; }
static int check_device_error(mcrt_ctx *c)
{
    uint32_t e = 0;
    HIP_TRY(hipMemcpy(&e, c->d_error, 4, hipMemcpyDeviceToHost));
    if (e) { HIP_TRY(hipMemsetAsync(c->d_error, 0, 4, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); return set_error(MCRT_ERR_LIMIT, "device error flag 0x%x:%s%s", e, (e & 1u) ? " BVH traversal stack overflow" : "", (e & 2u) ? " kernel watchdog expired (a persistent kernel ran for more than its time limit and was abandoned)" : ""); }
    return MCRT_OK;
}
extern "C" int mcrt_synchronize(mcrt_ctx *c) { CTX_TRY(c); HIP_TRY(hipStreamSynchronize(c->stream)); return check_device_error(c); }

extern "C" int mcrt_set_params(mcrt_ctx *c, const mcrt_params *p)
{
    CTX_TRY(c);
    if (!p) return set_error(MCRT_ERR_INVALID, "null params");
    if (p->n_elements == 0 || p->n_samples == 0) return set_error(MCRT_ERR_INVALID, "n_elements and n_samples must be positive");
    if (p->max_depth == 0 || p->max_depth > MCRT_MAX_BOUNCES) return set_error(MCRT_ERR_LIMIT, "max_depth must be 1..%d", MCRT_MAX_BOUNCES);
    if (p->n_rows == 0 || p->n_rows > MCRT_MAX_ROWS) return set_error(MCRT_ERR_LIMIT, "n_rows must be 1..%d", MCRT_MAX_ROWS);
    if (!(p->frequency > 0.f) || p->speed_of_sound == 0 || !(p->depth_cm > 0.0)) return set_error(MCRT_ERR_INVALID, "frequency, speed_of_sound and depth must be positive");
    if (p->tex_n == 0 || !(p->tex_res > 0.f)) return set_error(MCRT_ERR_INVALID, "texture size/resolution must be positive");
    Consts k = derive_consts(*p);
    if (k.axial_res_um == 0) return set_error(MCRT_ERR_INVALID, "axial resolution rounds to 0 um at %g MHz", (double)p->frequency);
    c->p = *p; c->c = k;
    return prepare_tables(c);
}

extern "C" int mcrt_get_params(mcrt_ctx *c, mcrt_params *out)
{
    if (!c || !out) return set_error(MCRT_ERR_INVALID, "null argument");
    *out = c->p;
    return MCRT_OK;
}

// a deep copy of a host-built tree (the context frees its trees with mcrt_free_bvh / mcrt_free_bvh4: malloc'ed arrays)
static int copy_tree(const mcrt::HostTree &src, mcrt_bvh *bvh, mcrt_bvh4 *bvh4)
{
    *bvh = *src.bvh; *bvh4 = *src.bvh4;
    bvh->nodes = nullptr; bvh->tri = nullptr; bvh4->nodes = nullptr;
    const size_t nb = sizeof(mcrt_bvh_node) * (size_t)src.bvh->n_nodes, tb = 48 * (size_t)src.bvh->n_tri, n4 = sizeof(mcrt_bvh4_node) * (size_t)src.bvh4->n_nodes;
    bvh->nodes = (mcrt_bvh_node *)malloc(nb ? nb : 1); bvh->tri = (float *)malloc(tb ? tb : 1); bvh4->nodes = (mcrt_bvh4_node *)malloc(n4 ? n4 : 1);
    if (!bvh->nodes || !bvh->tri || !bvh4->nodes) { mcrt_free_bvh(bvh); mcrt_free_bvh4(bvh4); return set_error(MCRT_ERR_NOMEM, "out of host memory"); }
    memcpy(bvh->nodes, src.bvh->nodes, nb); memcpy(bvh->tri, src.bvh->tri, tb); memcpy(bvh4->nodes, src.bvh4->nodes, n4);
    return MCRT_OK;
}

// builds the BVH over tri[n_tri][9] (host or device pointer) with the context's builder and installs it on the device.
// pre: a tree the HOST builder has already made of exactly these triangles (mcrt_group builds once for all its ranks); ignored by the
// device builder, which needs no host work
static int index_triangles(mcrt_ctx *c, const float *tri, uint32_t n_tri, const mcrt::HostTree *pre = nullptr)
{
    // k_trace addresses nodes (64 B as walked) and triangle records (64 B) with 32-bit byte offsets
    if (n_tri >= (1u << 25)) return set_error(MCRT_ERR_LIMIT, "%u triangles: the walk addresses at most 2^25 (32-bit byte offsets into 64-byte nodes and records)", n_tri);
    hipFree(c->d_nodes); hipFree(c->d_tris); hipFree(c->d_tri_slot); c->d_nodes = c->d_tris = nullptr; c->d_tri_slot = nullptr;
    mcrt_free_bvh(&c->bvh); mcrt_free_bvh4(&c->bvh4);
    c->host_bvh_stale = false;
    if (c->builder == MCRT_BVH_DEVICE_LBVH) {
        float *d_tri = nullptr; uint32_t *d_mesh = nullptr;
        HIP_TRY(hipMalloc(&d_tri, 36 * (size_t)n_tri));
        if (hipMalloc(&d_mesh, 4 * (size_t)n_tri) != hipSuccess) { hipFree(d_tri); return set_error(MCRT_ERR_NOMEM, "out of device memory"); }
        mcrt::LbvhResult r;
        int rc = MCRT_OK;
        if (hipMemcpyAsync(d_tri, tri, 36 * (size_t)n_tri, hipMemcpyDefault, c->stream) != hipSuccess ||
            hipMemcpyAsync(d_mesh, c->tri_mesh.data(), 4 * (size_t)n_tri, hipMemcpyHostToDevice, c->stream) != hipSuccess)
            rc = set_error(MCRT_ERR_HIP, "triangle upload failed");
        if (!rc) rc = mcrt::lbvh_build(d_tri, d_mesh, n_tri, c->stream, &r);
        hipFree(d_tri); hipFree(d_mesh);
        if (rc) return rc;
        c->d_nodes = r.d_nodes; c->d_tri_slot = r.d_tri_slot;
        {   // the walk's 64-byte records from the builder's 48-byte leaf-order array
            hipError_t e = hipMalloc(&c->d_tris, 16 * MCRT_TRI_PIECES * (size_t)n_tri);
            if (e == hipSuccess) e = mcrt::launch_expand_tris(r.d_tris, n_tri, c->d_tris, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            hipFree(r.d_tris);
            if (e != hipSuccess) return set_error(MCRT_ERR_HIP, "triangle records: %s", hipGetErrorString(e));
        }
        c->bvh.n_nodes = 0; c->bvh.n_tri = n_tri; c->bvh.max_depth = r.max_depth; c->bvh.pad_abs = r.pad_abs; c->bvh.nodes = nullptr; c->bvh.tri = nullptr;
        c->bvh4.n_nodes = r.n_nodes4; c->bvh4.max_stack = r.max_stack; c->bvh4.nodes = nullptr;
        c->host_bvh_stale = true;
        for (int i = 0; i < 3; i++) { c->scene_lo[i] = r.lo[i]; c->scene_hi[i] = r.hi[i]; }
        if (c->bvh4.max_stack > MCRT_STACK)
            return set_error(MCRT_ERR_LIMIT, "device-built BVH4 needs a %u-entry traversal stack, the kernel has %d", c->bvh4.max_stack, MCRT_STACK);
        if (c->bvh4.n_nodes >= (1u << 25)) return set_error(MCRT_ERR_LIMIT, "%u BVH4 nodes: the walk addresses at most 2^25", c->bvh4.n_nodes);
        return MCRT_OK;
    }
    std::vector<float> host_copy;
    if (!pre) {   // the host builder reads host memory
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, tri) == hipSuccess && at.type == hipMemoryTypeDevice) {
            host_copy.resize((size_t)n_tri * 9);
            HIP_TRY(hipMemcpy(host_copy.data(), tri, 36 * (size_t)n_tri, hipMemcpyDeviceToHost));
            tri = host_copy.data();
        } else (void)hipGetLastError();
    }
    int rc;
    if (pre) {
        if (pre->bvh->n_tri != n_tri) return set_error(MCRT_ERR_INVALID, "prebuilt tree has %u triangles, the scene %u", pre->bvh->n_tri, n_tri);
        rc = copy_tree(*pre, &c->bvh, &c->bvh4);
        if (rc) return rc;
    } else {
        rc = mcrt_build_bvh(tri, c->tri_mesh.data(), n_tri, &c->bvh);
        if (rc) return rc;
        rc = mcrt_build_bvh4(&c->bvh, &c->bvh4);
        if (rc) return rc;
    }
    for (int i = 0; i < 3; i++) { c->scene_lo[i] = INFINITY; c->scene_hi[i] = -INFINITY; }
    for (int k = 0; k < 4; k++) {
        const mcrt_bvh4_child &ch = c->bvh4.nodes[0].c[k];
        if (ch.ref == MCRT_BVH4_EMPTY) continue;
        const float hi[3] = { ch.hi_x, ch.hi_y, ch.hi_z };
        for (int i = 0; i < 3; i++) { c->scene_lo[i] = std::min(c->scene_lo[i], ch.lo[i]); c->scene_hi[i] = std::max(c->scene_hi[i], hi[i]); }
    }
    if (c->bvh4.max_stack > MCRT_STACK)
        return set_error(MCRT_ERR_LIMIT, "BVH4 needs a %u-entry traversal stack, the kernel has %d", c->bvh4.max_stack, MCRT_STACK);
    if (c->bvh4.n_nodes >= (1u << 25)) return set_error(MCRT_ERR_LIMIT, "%u BVH4 nodes: the walk addresses at most 2^25", c->bvh4.n_nodes);
    HIP_TRY(hipMalloc(&c->d_nodes, sizeof(mcrt_bvh4_node) * (size_t)c->bvh4.n_nodes));
    HIP_TRY(hipMemcpy(c->d_nodes, c->bvh4.nodes, sizeof(mcrt_bvh4_node) * (size_t)c->bvh4.n_nodes, hipMemcpyHostToDevice));
    {   // the walk's 64-byte records from the builder's 48-byte leaf-order array
        float4 *d_in = nullptr;
        HIP_TRY(hipMalloc(&d_in, 48 * (size_t)n_tri));
        hipError_t e = hipMalloc(&c->d_tris, 16 * MCRT_TRI_PIECES * (size_t)n_tri);
        if (e == hipSuccess) e = hipMemcpy(d_in, c->bvh.tri, 48 * (size_t)n_tri, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = mcrt::launch_expand_tris(d_in, n_tri, c->d_tris, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        hipFree(d_in);
        if (e != hipSuccess) return set_error(MCRT_ERR_HIP, "triangle records: %s", hipGetErrorString(e));
    }
    {   // triangle id -> leaf-order slot (k_shade re-derives the winning triangle's normal from its vertices)
        std::vector<uint32_t> slot(n_tri);
        for (uint32_t k = 0; k < n_tri; k++) { uint32_t id; memcpy(&id, &c->bvh.tri[(size_t)k * 12 + 3], 4); slot[id] = k; }
        HIP_TRY(hipMalloc(&c->d_tri_slot, 4 * (size_t)n_tri));
        HIP_TRY(hipMemcpy(c->d_tri_slot, slot.data(), 4 * (size_t)n_tri, hipMemcpyHostToDevice));
    }
    return MCRT_OK;
}

// host copies of a device-built tree, for mcrt_get_bvh / mcrt_get_bvh4
static int download_bvh(mcrt_ctx *c)
{
    if (!c->host_bvh_stale) return MCRT_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->bvh.tri = (float *)malloc(48 * (size_t)c->bvh.n_tri);
    c->bvh4.nodes = (mcrt_bvh4_node *)malloc(sizeof(mcrt_bvh4_node) * (size_t)c->bvh4.n_nodes);
    if (!c->bvh.tri || !c->bvh4.nodes) return set_error(MCRT_ERR_NOMEM, "out of memory");
    {   // back from the walk's records to the ABI's 48-byte layout (v0|id, v1|mesh, v2|0)
        const size_t W = 4 * MCRT_TRI_PIECES;      // floats per record
        std::vector<float> rec((size_t)c->bvh.n_tri * W);
        HIP_TRY(hipMemcpy(rec.data(), c->d_tris, 4 * W * (size_t)c->bvh.n_tri, hipMemcpyDeviceToHost));
        for (size_t t = 0; t < c->bvh.n_tri; t++) {
            const float *r = &rec[t * W]; float *o = &c->bvh.tri[t * 12];
            memcpy(o, r, 32);                                          // v0 | id, v1 | mesh
            o[8] = r[8]; o[9] = r[9]; o[10] = r[10]; o[11] = 0.0f;      // v2 | 0 (the record keeps the edge tolerance there)
        }
    }
    HIP_TRY(hipMemcpy(c->bvh4.nodes, c->d_nodes, sizeof(mcrt_bvh4_node) * (size_t)c->bvh4.n_nodes, hipMemcpyDeviceToHost));
    c->host_bvh_stale = false;
    return MCRT_OK;
}

extern "C" int mcrt_set_bvh_builder(mcrt_ctx *c, int builder)
{
    CTX_TRY(c);
    if (builder != MCRT_BVH_HOST_SAH && builder != MCRT_BVH_DEVICE_LBVH) return set_error(MCRT_ERR_INVALID, "unknown BVH builder %d", builder);
    c->builder = builder;
    return MCRT_OK;
}

static int update_triangles(mcrt_ctx *c, const float *tri, uint32_t n_tri, const mcrt::HostTree *pre)
{
    CTX_TRY(c);
    if (!c->have_scene) return set_error(MCRT_ERR_INVALID, "no scene uploaded");
    if (!tri) return set_error(MCRT_ERR_INVALID, "null triangles");
    if (n_tri != c->bvh.n_tri || n_tri == 0) return set_error(MCRT_ERR_INVALID, "the scene has %u triangles, the update has %u", c->bvh.n_tri, n_tri);
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_scene = false;                           // a failed rebuild leaves no scene
    int rc = index_triangles(c, tri, n_tri, pre); if (rc) return rc;
    rc = refresh_soa(c); if (rc) return rc;
    c->have_scene = true;
    return MCRT_OK;
}
extern "C" int mcrt_update_triangles(mcrt_ctx *c, const float *tri, uint32_t n_tri) { return update_triangles(c, tri, n_tri, nullptr); }

extern "C" int mcrt_refit_triangles(mcrt_ctx *c, const float *tri, uint32_t n_tri)
{
    CTX_TRY(c);
    if (!c->have_scene) return set_error(MCRT_ERR_INVALID, "no scene uploaded");
    if (!tri) return set_error(MCRT_ERR_INVALID, "null triangles");
    if (n_tri != c->bvh.n_tri || n_tri == 0) return set_error(MCRT_ERR_INVALID, "the scene has %u triangles, the update has %u", c->bvh.n_tri, n_tri);
    HIP_TRY(hipStreamSynchronize(c->stream));
    float *d_tri = nullptr;
    HIP_TRY(hipMalloc(&d_tri, 36 * (size_t)n_tri));
    int rc = MCRT_OK;
    if (hipMemcpyAsync(d_tri, tri, 36 * (size_t)n_tri, hipMemcpyDefault, c->stream) != hipSuccess) rc = set_error(MCRT_ERR_HIP, "triangle upload failed");
    float pad = 0.0f, lo[3], hi[3];
    if (!rc) rc = mcrt::bvh_refit(d_tri, n_tri, c->d_nodes, c->bvh4.n_nodes, c->d_tris, c->stream, &pad, lo, hi);
    hipFree(d_tri);
    if (!rc) rc = refresh_soa(c);
    if (rc) { c->have_scene = false; return rc; }           // a failed refit leaves no scene
    c->bvh.pad_abs = pad;
    for (int i = 0; i < 3; i++) { c->scene_lo[i] = lo[i]; c->scene_hi[i] = hi[i]; }
    // the host copies (and the host builder's BVH2, which has no refitted counterpart) are out of date: downloaded on demand
    free(c->bvh.nodes); c->bvh.nodes = nullptr; c->bvh.n_nodes = 0;
    free(c->bvh.tri); c->bvh.tri = nullptr;
    free(c->bvh4.nodes); c->bvh4.nodes = nullptr;
    c->host_bvh_stale = true;
    return MCRT_OK;
}

static int upload_scene(mcrt_ctx *c, const float *tri, const uint32_t *tri_mesh, uint32_t n_tri,
                        const mcrt_mesh *meshes, uint32_t n_mesh, const float *mats, uint32_t n_mat,
                        uint32_t start_mat, const float spacing[3], const mcrt::HostTree *pre)
{
    CTX_TRY(c);
    if (!meshes || !mats || n_mesh == 0 || n_mat == 0 || !spacing) return set_error(MCRT_ERR_INVALID, "mcrt_upload_scene: missing tables");
    if (n_tri && (!tri || !tri_mesh)) return set_error(MCRT_ERR_INVALID, "mcrt_upload_scene: missing triangles");
    if (start_mat >= n_mat) return set_error(MCRT_ERR_INVALID, "startingMaterial index %u out of range", start_mat);
    for (uint32_t i = 0; i < n_mesh; i++)
        if (meshes[i].mat_inside >= n_mat || meshes[i].mat_outside >= n_mat) return set_error(MCRT_ERR_INVALID, "mesh %u references a material out of range", i);
    for (uint32_t i = 0; i < n_tri; i++)
        if (tri_mesh[i] >= n_mesh) return set_error(MCRT_ERR_INVALID, "triangle %u references mesh %u out of range", i, tri_mesh[i]);
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_scene(c);
    if (n_tri) {
        c->tri_mesh.assign(tri_mesh, tri_mesh + n_tri);
        int rc = index_triangles(c, tri, n_tri, pre); if (rc) return rc;
        rc = refresh_soa(c); if (rc) return rc;
    }
    HIP_TRY(hipMalloc(&c->d_mats, 32 * (size_t)n_mat));
    HIP_TRY(hipMemcpy(c->d_mats, mats, 32 * (size_t)n_mat, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&c->d_meshes, sizeof(mcrt_mesh) * (size_t)n_mesh));
    HIP_TRY(hipMemcpy(c->d_meshes, meshes, sizeof(mcrt_mesh) * (size_t)n_mesh, hipMemcpyHostToDevice));
    c->n_mesh = n_mesh; c->n_mat = n_mat; c->start_mat = start_mat;
    for (int i = 0; i < 3; i++) c->spacing[i] = spacing[i];
    c->have_scene = true; c->mtab_valid = false;
    return prepare_tables(c);
}
extern "C" int mcrt_upload_scene(mcrt_ctx *c, const float *tri, const uint32_t *tri_mesh, uint32_t n_tri,
                                 const mcrt_mesh *meshes, uint32_t n_mesh, const float *mats, uint32_t n_mat,
                                 uint32_t start_mat, const float spacing[3])
{
    return upload_scene(c, tri, tri_mesh, n_tri, meshes, n_mesh, mats, n_mat, start_mat, spacing, nullptr);
}
// for mcrt_group.cpp: the same calls with a tree the host builder has already made (see index_triangles)
namespace mcrt {
int ctx_bvh_builder(const mcrt_ctx *c) { return c ? c->builder : MCRT_BVH_HOST_SAH; }
int upload_scene_with_tree(mcrt_ctx *c, const float *tri, const uint32_t *tri_mesh, uint32_t n_tri, const mcrt_mesh *meshes, uint32_t n_mesh,
                           const float *mats, uint32_t n_mat, uint32_t start_mat, const float spacing[3], const HostTree *pre)
{
    return upload_scene(c, tri, tri_mesh, n_tri, meshes, n_mesh, mats, n_mat, start_mat, spacing, pre);
}
int update_triangles_with_tree(mcrt_ctx *c, const float *tri, uint32_t n_tri, const HostTree *pre) { return update_triangles(c, tri, n_tri, pre); }
}

extern "C" int mcrt_get_bvh(mcrt_ctx *c, mcrt_bvh *out)
{
    if (!c || !out) return set_error(MCRT_ERR_INVALID, "null argument");
    if (!c->have_scene) return set_error(MCRT_ERR_INVALID, "no scene uploaded");
    { int rc = download_bvh(c); if (rc) return rc; }
    *out = c->bvh;
    return MCRT_OK;
}

extern "C" int mcrt_get_bvh4(mcrt_ctx *c, mcrt_bvh4 *out)
{
    if (!c || !out) return set_error(MCRT_ERR_INVALID, "null argument");
    if (!c->have_scene) return set_error(MCRT_ERR_INVALID, "no scene uploaded");
    HIP_TRY(hipSetDevice(c->device));
    if (c->d_nodes_walk) {
        // the tree AS WALKED: the lane-per-ray walk reads half-float boxes rounded outwards; decoded back into the builders' layout
        if (c->walked_stale || !c->walked_nodes) {
            const size_t bytes = sizeof(mcrt_bvh4_node) * (size_t)c->bvh4.n_nodes;
            float4 *d_tmp = nullptr;
            HIP_TRY(hipMalloc(&d_tmp, bytes));
            hipError_t e = mcrt::launch_nodes_walk_decode(c->d_nodes_walk, c->bvh4.n_nodes, d_tmp, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            free(c->walked_nodes);
            c->walked_nodes = (mcrt_bvh4_node *)malloc(bytes);
            if (e == hipSuccess && c->walked_nodes) e = hipMemcpy(c->walked_nodes, d_tmp, bytes, hipMemcpyDeviceToHost);
            hipFree(d_tmp);
            if (!c->walked_nodes) return set_error(MCRT_ERR_NOMEM, "out of memory");
            if (e != hipSuccess) return set_error(MCRT_ERR_HIP, "mcrt_get_bvh4: %s", hipGetErrorString(e));
            c->walked_stale = false;
        }
        out->n_nodes = c->bvh4.n_nodes; out->max_stack = c->bvh4.max_stack; out->nodes = c->walked_nodes;
        return MCRT_OK;
    }
    { int rc = download_bvh(c); if (rc) return rc; }
    *out = c->bvh4;
    return MCRT_OK;
}

extern "C" int mcrt_upload_texture(mcrt_ctx *c, const float *vox, uint32_t n)
{
    CTX_TRY(c);
    if (n == 0) return set_error(MCRT_ERR_INVALID, "texture size 0");
    const size_t total = (size_t)n * n * n;
    std::vector<float> gen;
    bool finite = true;
    if (!vox) {
        gen.resize(total * 2);
        int rc = mcrt_generate_texture(gen.data(), n);
        if (rc) return rc;
        vox = gen.data();
    } else {
        for (size_t i = 0; i < total * 2; i++) if (!std::isfinite(vox[i])) { finite = false; break; }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    hipFree(c->d_tex); c->d_tex = nullptr;
    HIP_TRY(hipMalloc(&c->d_tex, total * 8));
    HIP_TRY(hipMemcpy(c->d_tex, vox, total * 8, hipMemcpyHostToDevice));
    c->tex_n = n; c->tex_finite = finite;
    return MCRT_OK;
}

extern "C" int mcrt_set_transducer(mcrt_ctx *c, const float *pos, const float *dir, uint32_t n)
{
    CTX_TRY(c);
    if (!pos || !dir || n == 0) return set_error(MCRT_ERR_INVALID, "mcrt_set_transducer: bad arguments");
    if (n != c->n_el) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->d_pos); hipFree(c->d_dir); c->d_pos = c->d_dir = nullptr;
        HIP_TRY(hipMalloc(&c->d_pos, 12 * (size_t)n));
        HIP_TRY(hipMalloc(&c->d_dir, 12 * (size_t)n));
        c->n_el = n;
    }
    HIP_TRY(hipMemcpyAsync(c->d_pos, pos, 12 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_dir, dir, 12 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));   // pos/dir may be pageable host memory owned by the caller
    return MCRT_OK;
}

static int ensure_acc(mcrt_ctx *c, uint32_t ne)
{
    const size_t need = (size_t)ne * c->p.n_rows, needf = (size_t)ne * ((c->p.n_rows + 31u) >> 5);
    if (need > c->acc_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->d_acc); c->d_acc = nullptr; c->acc_cap = 0; c->acc_clean_ne = 0;
        HIP_TRY(hipMalloc(&c->d_acc, need * 8));
        c->acc_cap = need;
    }
    if (needf > c->flag_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->d_flags); c->d_flags = nullptr; c->flag_cap = 0; c->acc_clean_ne = 0;
        HIP_TRY(hipMalloc(&c->d_flags, needf * 4));
        c->flag_cap = needf;
    }
    // k_finalize leaves the bins zeroed; only a shape change (or a failed frame) needs an explicit clear
    if (c->acc_clean_ne != ne || c->acc_clean_rows != c->p.n_rows) {
        HIP_TRY(hipMemsetAsync(c->d_acc, 0, need * 8, c->stream));
        HIP_TRY(hipMemsetAsync(c->d_flags, 0, needf * 4, c->stream));
    }
    c->acc_clean_ne = 0; c->acc_clean_rows = 0;   // dirty until the frame's k_finalize has been enqueued
    return MCRT_OK;
}

static int check_ready(mcrt_ctx *c, uint32_t e0, uint32_t e1)
{
    if (!c->have_scene) return set_error(MCRT_ERR_INVALID, "no scene uploaded");
    if (!c->d_tex) return set_error(MCRT_ERR_INVALID, "no texture uploaded");
    if (c->tex_n != c->p.tex_n) return set_error(MCRT_ERR_INVALID, "texture is %u^3 but params say %u^3", c->tex_n, c->p.tex_n);
    if (!c->pose_pos) {
        if (!c->d_pos) return set_error(MCRT_ERR_INVALID, "no transducer set");
        if (c->n_el != c->p.n_elements) return set_error(MCRT_ERR_INVALID, "transducer has %u elements but params say %u", c->n_el, c->p.n_elements);
    }
    if (e0 >= e1 || e1 > c->p.n_elements) return set_error(MCRT_ERR_INVALID, "scan-line range [%u,%u) invalid for %u elements", e0, e1, c->p.n_elements);
    return MCRT_OK;
}

// out: 0 = RF image only, 1 = + hit indices, 2 = + the segment table (64 B per path and bounce: only allocated when asked for)
static int ensure_work(mcrt_ctx *c, Work &w, uint32_t ne_frame, uint32_t n_frames, int out)
{
    const uint32_t ne = ne_frame * n_frames;
    const size_t np = (size_t)ne * c->p.n_samples;
    const uint32_t B = c->p.max_depth;
    if (out >= 2 && w.segs_cap < np * B) {
        HIP_TRY(hipDeviceSynchronize());
        hipFree(w.d_segs); w.d_segs = nullptr; w.segs_cap = 0;
        HIP_TRY(hipMalloc(&w.d_segs, sizeof(mcrt_segment) * np * B));
        w.segs_cap = np * B;
    }
    if (out >= 1 && w.hits_cap < np * B) {
        HIP_TRY(hipDeviceSynchronize());
        hipFree(w.d_hits); w.d_hits = nullptr; w.hits_cap = 0;
        HIP_TRY(hipMalloc(&w.d_hits, 4 * np * B));
        w.hits_cap = np * B;
    }
    {   // traversal-stack entries beyond the LDS part, one slot per thread of THIS work set's walk launches
        const uint32_t lds_part = mcrt::lane_stack_entries();
        uint32_t blocks = std::max(std::max(c->knobs.trace_blocks, c->knobs.trace_blocks_wide), c->n_cu * 5u);      // (the larger of the walk's two forms)
        if (np <= c->knobs.path_max) blocks = std::max(blocks, mcrt::path_blocks(np));                                   // (... and k_path's grid, when this pass takes the latency form)
        const size_t need = c->bvh4.max_stack > lds_part ? (size_t)(c->bvh4.max_stack - lds_part) * blocks * 256 : 0;
        if (need > w.ovf_cap) {
            HIP_TRY(hipDeviceSynchronize());
            hipFree(w.d_stack_ovf); w.d_stack_ovf = nullptr; w.ovf_cap = 0;
            HIP_TRY(hipMalloc(&w.d_stack_ovf, 4 * need));
            w.ovf_cap = need;
        }
    }
    if (np <= w.paths && B <= w.depth) return MCRT_OK;
    HIP_TRY(hipDeviceSynchronize());
    {   // (the optional tables survive a re-allocation of the rest when they are large enough)
        mcrt_segment *sg = w.d_segs; const size_t sc = w.segs_cap; int32_t *ht = w.d_hits; const size_t hc = w.hits_cap;
        w.d_segs = nullptr; w.d_hits = nullptr;
        free_work_buffers(w);
        w.d_segs = sg; w.segs_cap = sc; w.d_hits = ht; w.hits_cap = hc;
    }
    HIP_TRY(hipMalloc(&w.d_st0, 32 * np)); HIP_TRY(hipMalloc(&w.d_st1, 32 * np)); HIP_TRY(hipMalloc(&w.d_st2, 32 * np));   // two halves: bounce parity
    HIP_TRY(hipMalloc(&w.d_key0, 8 * np)); HIP_TRY(hipMalloc(&w.d_key1, 8 * np));
    HIP_TRY(hipMalloc(&w.d_q, 8 * np)); HIP_TRY(hipMalloc(&w.d_seg_count, 4 * np));
    HIP_TRY(hipMalloc(&w.d_counts, 4 * (MCRT_MAX_BOUNCES + 1)));
    HIP_TRY(hipMalloc(&w.d_cursors, 4 * (size_t)MCRT_MAX_BOUNCES * MCRT_XCDS * MCRT_CURSOR_STRIDE));
    HIP_TRY(hipMalloc(&w.d_mrec, 48 * np * B));
    w.paths = np; w.depth = B;
    return MCRT_OK;
}

// kernel arguments for scan-lines [e0,e1) traced with work set w; acc_e0 = first scan-line of the frame's RF block
static void fill_args(mcrt_ctx *c, const Work &w, mcrt::FrameArgs &a, uint32_t frame, uint32_t n_frames, uint32_t e0, uint32_t e1, uint32_t acc_e0, uint32_t acc_ne)
{
    memset(&a, 0, sizeof a);
    a.nodes_walk = c->d_nodes_walk; a.stack_ovf = w.d_stack_ovf; a.tris = c->d_tris; a.meshes = c->d_meshes; a.mats = c->d_mats; a.tex = c->d_tex;
    a.el_pos = c->pose_pos ? c->pose_pos : c->d_pos; a.el_dir = c->pose_pos ? c->pose_dir : c->d_dir; a.pose_stride = c->pose_pos ? c->p.n_elements : 0u;
    a.row_thr = c->d_row_thr;
    a.acc = c->d_acc; a.flags = c->d_flags;                 // the frame block [n_frames][acc_ne][R]; this group owns columns e0-acc_e0 ...
    a.acc_stride = acc_ne; a.acc_off = e0 - acc_e0;
    a.st0 = w.d_st0; a.st1 = w.d_st1; a.st2 = w.d_st2; a.queue = w.d_q;
    a.key0 = w.d_key0; a.key1 = w.d_key1; a.tri_slot = c->d_tri_slot; a.tris_id = c->d_tris_id; a.counts = w.d_counts; a.cursors = w.d_cursors; a.segs = w.d_segs; a.hits = w.d_hits; a.mrec = w.d_mrec; a.mtab = c->d_mtab; a.seg_count = w.d_seg_count;
    a.stats = c->d_stats; a.error_flag = c->d_error; a.stamps = c->d_stats + 8;
    a.n_mat = c->n_mat; a.n_mesh = c->n_mesh; a.n_nodes = c->bvh4.n_nodes; a.S = c->p.n_samples; a.B = c->p.max_depth; a.R = c->p.n_rows;
    a.e_begin = e0; a.ne_frame = e1 - e0; a.ne = (e1 - e0) * n_frames;   // n_frames consecutive frame ids traced as one pass
    a.ksplit_limit = c->knobs.ksplit_limit;   // bounces with fewer rays than this are cut into pieces (see k_trace)
    if (c->stats_on) a.ksplit_limit = 0;   // counting mode = one walk per ray, so the counts are those of a plain closest-hit walk
    for (int i = 0; i < 3; i++) { a.scene_lo[i] = c->scene_lo[i]; a.scene_hi[i] = c->scene_hi[i]; }
    a.trace_blocks = c->knobs.trace_blocks ? c->knobs.trace_blocks : (c->n_cu - (c->knobs.main_mask ? c->knobs.march_cus : 0u)) * 4u;   // persistent k_trace: 4 four-wave workgroups per CU (1024 on the MI355X's 256 CUs) of the 5 its registers and LDS allow --
                                                                                  // the fifth's registers go to a k_march wavefront beside them (since k_march's fast path: 0.446 -> 0.428 ms per frame on a 20-frame pass, 0.366 -> 0.364 at 128)
    a.trace_blocks_wide = c->knobs.trace_blocks_wide ? c->knobs.trace_blocks_wide : c->n_cu * 5u;      // k_trace_lane_wide: five workgroups per CU
    a.wide_from = c->knobs.wide_from ? c->knobs.wide_from : mcrt::lane_wide_from();
    // ... while the tree is served from the caches: with 16 M triangles (460 MB of walked nodes, past the Infinity Cache) a fifth wavefront per SIMD only
    // adds misses -- 0.667 against 0.638 ms per frame -- where the 1 M-triangle scene (29 MB) gains 3-4 %; the line is drawn at half the Infinity Cache
    if ((uint64_t)c->bvh4.n_nodes * 64ull > (uint64_t)c->knobs.wide_max_tree_mb * 1048576ull) a.trace_blocks_wide = 0;
    if (c->knobs.main_mask) a.trace_blocks_wide = 0;                                                   // (CU-masked streams: the four-wavefront form only)
    a.march_blocks = c->knobs.march_blocks;
    a.packet_mask = (c->stats_on || (uint64_t)a.ne * a.S < c->knobs.packet_from) ? 0u : c->knobs.packet_mask;   // bounces walked a wavefront per ray packet (k_trace_packet); the counting build walks ray by ray
    a.frame = frame; a.seed = c->p.seed; a.start_mat = c->start_mat; a.tex_n = c->tex_n; a.tex_mask = (c->tex_n & (c->tex_n - 1u)) == 0u ? c->tex_n - 1u : 0u;
    a.sanitize = c->p.sanitize_tir; a.tex_finite = c->tex_finite ? 1u : 0u;
    a.freq = c->p.frequency; a.eps = c->p.intensity_epsilon; a.I0 = c->p.initial_intensity; a.offs = c->p.ray_start_offset;
    a.sx = c->spacing[0]; a.sy = c->spacing[1]; a.sz = c->spacing[2]; a.tex_res = c->p.tex_res; a.axial_res_f = c->c.axial_res_f; a.pad_abs = c->bvh.pad_abs; a.tex_rcp = 1.0f / c->p.tex_res; a.fast_div = c->fast_div ? 1u : 0u;
    // k_march's branch-free texture lookup: power-of-two texture, verified division, |x / res| < 2^31
    a.tex_shift = 0; while ((1u << a.tex_shift) < c->tex_n) a.tex_shift++;
    a.lean_bound = 0.0f;
    if (c->fast_div_all && a.tex_mask && a.tex_shift <= 10u) {
        const float lim = 2147483648.0f * c->p.tex_res * (1.0f - 0x1p-20f);
        a.lean_bound = lim < 1e18f ? lim : 1e18f;
        if (!(a.lean_bound > 0.0f)) a.lean_bound = 0.0f;
    }
    a.axial_res_mm = c->c.axial_res_mm; a.time_step = c->c.time_step_us; a.row_dt = c->c.row_dt_us;
    a.max_travel = c->c.max_travel_us; a.sos_d = (double)c->p.speed_of_sound; a.inv_row_dt = 1.0 / c->c.row_dt_us;
    // k_march's fast variant: the reference's 256^3 texture with the branch-free cell, and an LDS image long enough for the row
    // guess of every valid step -- t < max_travel, and rounding is monotone, so (int)(t * inv_row_dt) <= (int)(max_travel * inv_row_dt)
    a.march_rows = 0u;
    {
        const double g = a.max_travel * a.inv_row_dt;
        if (a.lean_bound > 0.0f && c->tex_n == 256u && g >= 0.0 && g < (double)(MCRT_MAX_ROWS + 1)) {
            const uint32_t gmax = (uint32_t)g;
            a.march_rows = (gmax + 2u > c->p.n_rows + 1u) ? gmax + 2u : c->p.n_rows + 1u;
        }
    }
    c->last_lean_bound = a.lean_bound; c->last_march_rows = a.march_rows;
}

static uint32_t side_streams(const mcrt_ctx *c, const mcrt::FrameArgs &a)
{
    if (c->knobs.march_streams) return c->knobs.march_streams;
    // (two only where the walk runs from the caches -- the five-wavefront form's own criterion, fill_args --: on the 16 M-triangle streaming scene the walks are the longer chain and a
    //  second accumulation beside them costs 1.5 %: 0.607 against 0.598 ms per frame)
    return (a.trace_blocks_wide != 0u && (uint64_t)a.ne * a.S >= (uint64_t)MCRT_SIDE_STREAMS_TWO_FROM) ? 2u : 1u;
}

static int timing_events(mcrt_ctx *c, int kind, hipEvent_t *e0, hipEvent_t *e1);

// one bounce of one group: k_trace_lane + k_shade on the group's stream, k_march of the finished segments on its side stream.
// With timing enabled every k_trace launch (the dominant kernel) is bracketed by HIP events on the stream it is launched on.
// (Round 4 tried holding k_march of bounce b back until the walk of bounce b+1 had claimed its last ray -- a device word raised by the walk, waited
//  for with hipStreamWaitValue32, which the command processor releases ~1 us after the store --: 0.360 against 0.343 ms per frame at 128 frames in
//  flight, 0.414 against 0.405 on the driver's pass, and a hang under `rocprofv3 --pmc`.  Removed; DESIGN.md A.6, profiles/round4/exp_round4_kernels.txt.)
static int run_bounce(mcrt_ctx *c, Work &w, hipStream_t st, const mcrt::FrameArgs &a, uint32_t b, bool accumulate, bool overlap)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    { int rc = timing_events(c, 0, &e0, &e1); if (rc) return rc; }
    if (e0) HIP_TRY(hipEventRecord(e0, st));
    HIP_TRY(mcrt::launch_trace(a, b, c->stats_on, st));
    if (e1) HIP_TRY(hipEventRecord(e1, st));
    { int rc = timing_events(c, 1, &e0, &e1); if (rc) return rc; }
    if (e0) HIP_TRY(hipEventRecord(e0, st));
    HIP_TRY(mcrt::launch_shade(a, b, c->stats_on, st));
    if (e1) HIP_TRY(hipEventRecord(e1, st));
    if (accumulate && overlap) {   // the segments of bounce b are final: accumulate them beside the next bounce's walk
        HIP_TRY(hipEventRecord(w.ev_bounce[b], st));
        hipStream_t side;
        { int rc = side_stream(c, w, b % side_streams(c, a), &side); if (rc) return rc; }
        HIP_TRY(hipStreamWaitEvent(side, w.ev_bounce[b], 0));
        { int rc = timing_events(c, 2, &e0, &e1); if (rc) return rc; }
        if (e0) HIP_TRY(hipEventRecord(e0, side));
        HIP_TRY(mcrt::launch_march(a, b, c->stats_on, side));
        if (e1) HIP_TRY(hipEventRecord(e1, side));
    } else if (accumulate) {
        { int rc = timing_events(c, 2, &e0, &e1); if (rc) return rc; }
        if (e0) HIP_TRY(hipEventRecord(e0, st));
        HIP_TRY(mcrt::launch_march(a, b, c->stats_on, st));
        if (e1) HIP_TRY(hipEventRecord(e1, st));
    }
    return MCRT_OK;
}

// a pair of events for a launch of kind 0 (the walk), 1 (k_shade) or 2 (k_march) -- or none when that kind is not being timed
static int timing_events(mcrt_ctx *c, int kind, hipEvent_t *e0, hipEvent_t *e1)
{
    *e0 = *e1 = nullptr;
    if (!c->timing_on || (kind != 0 && c->timing_level < 2)) return MCRT_OK;
    if (c->ev_used == c->ev.size()) {
        if (c->ev.size() >= 65536) return set_error(MCRT_ERR_LIMIT, "timing buffer full: call mcrt_get_kernel_time(reset=1)");
        hipEvent_t x, y;
        HIP_TRY(hipEventCreate(&x)); HIP_TRY(hipEventCreate(&y));
        c->ev.emplace_back(x, y); c->ev_kind.push_back(0);
    }
    *e0 = c->ev[c->ev_used].first; *e1 = c->ev[c->ev_used].second; c->ev_kind[c->ev_used] = (unsigned char)kind; c->ev_used++;
    return MCRT_OK;
}

// scene::cast_rays (scene.cpp:50-183) [+ the accumulation loop] for scan-lines [e0,e1), split into `groups` independent
// scan-line blocks.  Everything is ordered after what is already queued on the context's stream, and the context's stream
// waits for all of it.
// (prepare_frame: buffers and kernel arguments of the groups; enqueue_frame: the launches)
static int prepare_frame(mcrt_ctx *c, uint32_t frame, uint32_t n_frames, uint32_t e0, uint32_t e1, uint32_t &groups, int out,
                         std::vector<mcrt::FrameArgs> &args, std::vector<Work *> &ws)
{
    const uint32_t ne = e1 - e0;
    if (groups > ne) groups = ne;
    if (groups < 1) groups = 1;
    if (groups > 16) groups = 16;
    args.resize(groups); ws.resize(groups);
    for (uint32_t g = 0; g < groups; g++) {
        int rc = get_work(c, g, &ws[g]); if (rc) return rc;
        const uint32_t b0 = e0 + (uint32_t)(((uint64_t)ne * g) / groups), b1 = e0 + (uint32_t)(((uint64_t)ne * (g + 1)) / groups);
        rc = ensure_work(c, *ws[g], b1 - b0, n_frames, out); if (rc) return rc;
        fill_args(c, *ws[g], args[g], frame, n_frames, b0, b1, e0, ne);
        args[g].want_segs = out >= 2 ? 1u : 0u;
        if (out < 1) args[g].hits = nullptr;
    }
    return MCRT_OK;
}

static int enqueue_frame(mcrt_ctx *c, const std::vector<mcrt::FrameArgs> &args, const std::vector<Work *> &ws, bool accumulate)
{
    const uint32_t groups = (uint32_t)args.size();
    const bool overlap = !c->knobs.no_overlap;
    std::vector<hipStream_t> gst(groups);
    for (uint32_t g = 0; g < groups; g++) { int rc = work_stream(c, *ws[g], g == 0, &gst[g]); if (rc) return rc; }
    if (c->scene_pending && c->scene_stream != c->stream) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_scene, 0));   // a scene update issued on another stream
    HIP_TRY(hipEventRecord(c->ev_start, c->stream));
    for (uint32_t g = 0; g < groups; g++) {
        if (gst[g] != c->stream) HIP_TRY(hipStreamWaitEvent(gst[g], c->ev_start, 0));
        HIP_TRY(mcrt::launch_init(args[g], gst[g]));
    }
    // A pass that cannot fill the GPU runs in its LATENCY form: one launch carries every path through all of its bounces (k_path), one more
    // accumulates every bounce's segments -- instead of a walk / shade launch pair per bounce, each as long as its slowest wavefront.
    uint64_t paths = 0;
    for (uint32_t g = 0; g < groups; g++) paths += (uint64_t)args[g].ne * args[g].S;
    const bool fused = !c->stats_on && paths <= c->knobs.path_max;
    for (uint32_t g = 0; g < groups && fused; g++) {          // (every group's k_path first, then the accumulations: the second group must not wait for the host to enqueue the first's k_march)
        hipEvent_t e0 = nullptr, e1 = nullptr;
        { int rc = timing_events(c, 0, &e0, &e1); if (rc) return rc; }
        if (e0) HIP_TRY(hipEventRecord(e0, gst[g]));
        HIP_TRY(mcrt::launch_path(args[g], gst[g]));
        if (e1) HIP_TRY(hipEventRecord(e1, gst[g]));
    }
    for (uint32_t g = 0; g < groups && fused && accumulate; g++) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        { int rc = timing_events(c, 2, &e0, &e1); if (rc) return rc; }
        if (e0) HIP_TRY(hipEventRecord(e0, gst[g]));
        HIP_TRY(mcrt::launch_march(args[g], mcrt::MCRT_ALL_BOUNCES, false, gst[g]));
        if (e1) HIP_TRY(hipEventRecord(e1, gst[g]));
    }
    for (uint32_t b = 0; b < c->p.max_depth && !fused; b++)
        for (uint32_t g = 0; g < groups; g++) {
            int rc = run_bounce(c, *ws[g], gst[g], args[g], b, accumulate, overlap); if (rc) return rc;
        }
    for (uint32_t g = 0; g < groups; g++) {
        if (accumulate && overlap) {
            for (uint32_t i = 0; i < side_streams(c, args[g]); i++) {
                if (!ws[g]->side[i]) continue;
                HIP_TRY(hipEventRecord(ws[g]->ev_join[i], ws[g]->side[i]));
                HIP_TRY(hipStreamWaitEvent(gst[g], ws[g]->ev_join[i], 0));
            }
        }
        if (gst[g] != c->stream) {
            HIP_TRY(hipEventRecord(ws[g]->ev_done, gst[g]));
            HIP_TRY(hipStreamWaitEvent(c->stream, ws[g]->ev_done, 0));
        }
    }
    return MCRT_OK;
}

static int run_frame(mcrt_ctx *c, uint32_t frame, uint32_t n_frames, uint32_t e0, uint32_t e1, bool accumulate, uint32_t groups, int out)
{
    std::vector<mcrt::FrameArgs> args; std::vector<Work *> ws;
    int rc = prepare_frame(c, frame, n_frames, e0, e1, groups, out, args, ws); if (rc) return rc;
    return enqueue_frame(c, args, ws, accumulate);
}

static uint32_t frame_groups(const mcrt_ctx *c)
{
    return c->stats_on ? 1u : c->knobs.groups;
}

extern "C" int mcrt_trace_frames(mcrt_ctx *c, uint32_t frame, uint32_t n_frames, uint32_t e0, uint32_t e1, float *rf_dev)
{
    CTX_TRY(c);
    int rc = check_ready(c, e0, e1); if (rc) return rc;
    if (!rf_dev) return set_error(MCRT_ERR_INVALID, "null rf_dev");
    if (n_frames == 0 || n_frames > 1024) return set_error(MCRT_ERR_LIMIT, "n_frames must be 1..1024");
    if ((uint64_t)(e1 - e0) * n_frames * c->p.n_samples > (1ull << 27))        // (~600 bytes of work buffers per path)
        return set_error(MCRT_ERR_LIMIT, "%u frames x %u scan-lines x %u samples: more than 2^27 paths in one pass", n_frames, e1 - e0, c->p.n_samples);
    const uint32_t lines = (e1 - e0) * n_frames;
    rc = ensure_acc(c, lines); if (rc) return rc;
    uint32_t groups = frame_groups(c);
    if (groups == 1u && !c->stats_on && (uint64_t)lines * c->p.n_samples <= c->knobs.path_max) groups = c->knobs.path_groups;      // the latency form (enqueue_frame)
    rc = run_frame(c, frame, n_frames, e0, e1, true, groups, 0); if (rc) return rc;
    HIP_TRY(mcrt::launch_finalize(c->d_acc, c->d_flags, rf_dev, lines, c->p.n_rows, c->d_error, c->stream));
    c->acc_clean_ne = lines; c->acc_clean_rows = c->p.n_rows;
    return MCRT_OK;
}

extern "C" int mcrt_trace_frame(mcrt_ctx *c, uint32_t frame, uint32_t e0, uint32_t e1, float *rf_dev)
{
    return mcrt_trace_frames(c, frame, 1, e0, e1, rf_dev);
}

// A pass whose frames each have their own probe pose (transducer.h:82-118 update() between the frames of main.cpp:92-152): the element
// tables [n_frames][E][3] are staged in the context (host pointers are copied on the stream) and k_init reads frame f's rows.
extern "C" int mcrt_trace_frames_poses(mcrt_ctx *c, uint32_t frame, uint32_t n_frames, uint32_t e0, uint32_t e1,
                                       const float *pos, const float *dir, float *rf_dev)
{
    CTX_TRY(c);
    if (!pos || !dir) return set_error(MCRT_ERR_INVALID, "mcrt_trace_frames_poses: null pose tables");
    if (n_frames == 0 || n_frames > 1024) return set_error(MCRT_ERR_LIMIT, "n_frames must be 1..1024");
    const uint32_t E = c->p.n_elements;
    const size_t bytes = 12 * (size_t)n_frames * E;
    const float *src[2] = { pos, dir };
    const float *dev[2] = { nullptr, nullptr };
    // A table in HOST memory belongs to the caller and may be pageable: it is copied into pinned memory the context owns before this
    // call returns (the caller may free or rewrite it at once), and goes to the device from there on the stream.  The staging buffers are
    // reused: the copy of the previous call (an early node of the previous pass, not the pass) is waited for first.
    bool staged = false;
    for (int k = 0; k < 2; k++) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, src[k]) == hipSuccess && at.type == hipMemoryTypeDevice) { dev[k] = src[k]; continue; }
        (void)hipGetLastError();
        if (c->pose_copy_pending) { HIP_TRY(hipEventSynchronize(c->ev_pose)); c->pose_copy_pending = false; }
        if (c->pose_cap[k] < bytes) {
            HIP_TRY(hipStreamSynchronize(c->stream));
            hipFree(c->d_pose[k]); c->d_pose[k] = nullptr; c->pose_cap[k] = 0;
            if (c->h_pose[k]) { hipHostFree(c->h_pose[k]); c->h_pose[k] = nullptr; }
            HIP_TRY(hipMalloc(&c->d_pose[k], bytes));
            HIP_TRY(hipHostMalloc((void **)&c->h_pose[k], bytes, hipHostMallocDefault));
            c->pose_cap[k] = bytes;
        }
        memcpy(c->h_pose[k], src[k], bytes);
        HIP_TRY(hipMemcpyAsync(c->d_pose[k], c->h_pose[k], bytes, hipMemcpyHostToDevice, c->stream));
        dev[k] = c->d_pose[k]; staged = true;
    }
    if (staged) {
        if (!c->ev_pose) HIP_TRY(hipEventCreateWithFlags(&c->ev_pose, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->ev_pose, c->stream));
        c->pose_copy_pending = true;
    }
    c->pose_pos = dev[0]; c->pose_dir = dev[1];
    const int rc = mcrt_trace_frames(c, frame, n_frames, e0, e1, rf_dev);
    c->pose_pos = c->pose_dir = nullptr;
    return rc;
}

// copies the per-path tables (work set 0) to the host: segs [ne][S][B], seg_count [ne][S], hits [ne][S][B] (= segment.tri, -2 beyond the path's end)
static int copy_out(mcrt_ctx *c, uint32_t ne, int32_t *hits, mcrt_segment *segs, uint32_t *seg_count)
{
    const size_t np = (size_t)ne * c->p.n_samples, B = c->p.max_depth;
    const Work &w = c->work[0];
    HIP_TRY(hipStreamSynchronize(c->stream));
    int rc = check_device_error(c); if (rc) return rc;
    std::vector<uint32_t> cnt;
    if (!seg_count && (hits || segs)) { cnt.resize(np); seg_count = cnt.data(); }
    if (seg_count) HIP_TRY(hipMemcpy(seg_count, w.d_seg_count, np * 4, hipMemcpyDeviceToHost));
    if (segs) {
        HIP_TRY(hipMemcpy(segs, w.d_segs, np * B * sizeof(mcrt_segment), hipMemcpyDeviceToHost));
        for (size_t p = 0; p < np; p++)                         // slots beyond a path's end are unspecified on the device
            for (size_t b = seg_count[p]; b < B; b++) memset(&segs[p * B + b], 0, sizeof(mcrt_segment));
    }
    if (hits) {
        HIP_TRY(hipMemcpy(hits, w.d_hits, np * B * 4, hipMemcpyDeviceToHost));
        for (size_t p = 0; p < np; p++)
            for (size_t b = seg_count[p]; b < B; b++) hits[p * B + b] = -2;
    }
    return MCRT_OK;
}

extern "C" int mcrt_trace_frame_debug(mcrt_ctx *c, uint32_t frame, uint32_t e0, uint32_t e1, float *rf_dev,
                                      int32_t *hits, mcrt_segment *segs, uint32_t *seg_count)
{
    CTX_TRY(c);
    int rc = check_ready(c, e0, e1); if (rc) return rc;
    if (!rf_dev) return set_error(MCRT_ERR_INVALID, "null rf_dev");
    rc = ensure_acc(c, e1 - e0); if (rc) return rc;
    rc = run_frame(c, frame, 1, e0, e1, true, 1, segs ? 2 : 1); if (rc) return rc;   // one group: the per-path tables are contiguous
    HIP_TRY(mcrt::launch_finalize(c->d_acc, c->d_flags, rf_dev, e1 - e0, c->p.n_rows, c->d_error, c->stream));
    c->acc_clean_ne = e1 - e0; c->acc_clean_rows = c->p.n_rows;
    return copy_out(c, e1 - e0, hits, segs, seg_count);
}

extern "C" int mcrt_cast_rays(mcrt_ctx *c, uint32_t frame, uint32_t e0, uint32_t e1, mcrt_segment *segs, uint32_t *seg_count, int32_t *hits)
{
    CTX_TRY(c);
    int rc = check_ready(c, e0, e1); if (rc) return rc;
    rc = run_frame(c, frame, 1, e0, e1, false, 1, segs ? 2 : 1); if (rc) return rc;
    return copy_out(c, e1 - e0, hits, segs, seg_count);
}

static int ensure_tmp(mcrt_ctx *c, size_t n)
{
    if (n > c->tmp_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->d_tmp); c->d_tmp = nullptr; c->tmp_cap = 0;
        HIP_TRY(hipMalloc(&c->d_tmp, n * 4));
        c->tmp_cap = n;
    }
    return MCRT_OK;
}

extern "C" int mcrt_convolve_frames(mcrt_ctx *c, float *rf_dev, uint32_t n_frames, uint32_t E, uint32_t R, const float *ax, uint32_t n_ax, const float *lat, uint32_t n_lat)
{
    CTX_TRY(c);
    if (!rf_dev || !ax || !lat || E == 0 || R == 0 || n_frames == 0) return set_error(MCRT_ERR_INVALID, "mcrt_convolve: bad arguments");
    if (n_ax == 0 || n_ax > 16 || n_lat == 0 || n_lat > 32) return set_error(MCRT_ERR_LIMIT, "kernel sizes must be 1..16 axial, 1..32 lateral");
    int rc = ensure_tmp(c, (size_t)n_frames * E * R); if (rc) return rc;
    mcrt::ConvTaps t; memset(&t, 0, sizeof t);
    memcpy(t.ax, ax, 4 * n_ax); memcpy(t.lat, lat, 4 * n_lat); t.n_ax = n_ax; t.n_lat = n_lat;
    HIP_TRY(mcrt::launch_convolve(rf_dev, c->d_tmp, n_frames, E, R, t, c->stream));
    return MCRT_OK;
}

extern "C" int mcrt_convolve(mcrt_ctx *c, float *rf_dev, uint32_t E, uint32_t R, const float *ax, uint32_t n_ax, const float *lat, uint32_t n_lat)
{
    return mcrt_convolve_frames(c, rf_dev, 1, E, R, ax, n_ax, lat, n_lat);
}

extern "C" int mcrt_envelope_frames(mcrt_ctx *c, float *rf_dev, uint32_t n_frames, uint32_t E, uint32_t R)
{
    CTX_TRY(c);
    if (!rf_dev || E == 0 || R == 0 || n_frames == 0) return set_error(MCRT_ERR_INVALID, "mcrt_envelope: bad arguments");
    if ((uint64_t)n_frames * E > 0x7fffffffull) return set_error(MCRT_ERR_LIMIT, "mcrt_envelope: too many scan-lines");
    if (R > MCRT_MAX_ROWS) return set_error(MCRT_ERR_LIMIT, "mcrt_envelope: at most %d rows", MCRT_MAX_ROWS);
    HIP_TRY(mcrt::launch_envelope(rf_dev, n_frames * E, R, c->stream));      // the scan-lines of all images are independent columns
    return MCRT_OK;
}

extern "C" int mcrt_envelope(mcrt_ctx *c, float *rf_dev, uint32_t E, uint32_t R)
{
    return mcrt_envelope_frames(c, rf_dev, 1, E, R);
}

extern "C" int mcrt_scan_convert_frames(mcrt_ctx *c, const float *rf_dev, uint32_t n_frames, uint32_t E, uint32_t R, double radius_mm, double total_angle,
                                        float *out_dev, uint32_t orows, uint32_t ocols)
{
    CTX_TRY(c);
    if (!rf_dev || !out_dev || E == 0 || R == 0 || orows == 0 || ocols == 0 || n_frames == 0) return set_error(MCRT_ERR_INVALID, "mcrt_scan_convert: bad arguments");
    if (n_frames > 65535u) return set_error(MCRT_ERR_LIMIT, "mcrt_scan_convert: at most 65535 images per call");
    const uint32_t key[6] = { E, R, orows, ocols, c->p.speed_of_sound, 1u };
    const double keyd[2] = { radius_mm * 1e6 + total_angle, c->c.max_travel_us };
    if (memcmp(key, c->map_key, sizeof key) || memcmp(keyd, c->map_keyd, sizeof keyd)) {
        std::vector<float> mc((size_t)orows * ocols), mr((size_t)orows * ocols);
        // (the rf_image template parameter is max_travel_time.to<unsigned int>(), main.cpp:36 -- the same truncation as max_rows uses)
        { int rc = mcrt_scan_maps(E, R, radius_mm, total_angle, (uint32_t)c->c.max_travel_us, c->p.speed_of_sound, orows, ocols, mr.data(), mc.data()); if (rc) return rc; }
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->d_map_col); hipFree(c->d_map_row); c->d_map_col = c->d_map_row = nullptr;
        HIP_TRY(hipMalloc(&c->d_map_col, mc.size() * 4)); HIP_TRY(hipMalloc(&c->d_map_row, mr.size() * 4));
        HIP_TRY(hipMemcpy(c->d_map_col, mc.data(), mc.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->d_map_row, mr.data(), mr.size() * 4, hipMemcpyHostToDevice));
        memcpy(c->map_key, key, sizeof key); memcpy(c->map_keyd, keyd, sizeof keyd);
    }
    HIP_TRY(mcrt::launch_remap(rf_dev, n_frames, E, R, c->d_map_col, c->d_map_row, out_dev, orows * ocols, c->stream));
    return MCRT_OK;
}

extern "C" int mcrt_scan_convert(mcrt_ctx *c, const float *rf_dev, uint32_t E, uint32_t R, double radius_mm, double total_angle,
                                 float *out_dev, uint32_t orows, uint32_t ocols)
{
    return mcrt_scan_convert_frames(c, rf_dev, 1, E, R, radius_mm, total_angle, out_dev, orows, ocols);
}

extern "C" int mcrt_export_rf(mcrt_ctx *c, const float *rf_dev, uint32_t E, uint32_t R, float *host)
{
    CTX_TRY(c);
    if (!rf_dev || !host || E == 0 || R == 0) return set_error(MCRT_ERR_INVALID, "mcrt_export_rf: bad arguments");
    int rc = ensure_tmp(c, (size_t)E * R); if (rc) return rc;
    HIP_TRY(mcrt::launch_transpose(rf_dev, c->d_tmp, E, R, c->stream));
    HIP_TRY(hipMemcpyAsync(host, c->d_tmp, (size_t)E * R * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

extern "C" int mcrt_import_rf(mcrt_ctx *c, const float *host, uint32_t E, uint32_t R, float *rf_dev)
{
    CTX_TRY(c);
    if (!rf_dev || !host || E == 0 || R == 0) return set_error(MCRT_ERR_INVALID, "mcrt_import_rf: bad arguments");
    int rc = ensure_tmp(c, (size_t)E * R); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_tmp, host, (size_t)E * R * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(mcrt::launch_transpose(c->d_tmp, rf_dev, R, E, c->stream));          // [R][E] -> [E][R]
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

extern "C" int mcrt_alloc(mcrt_ctx *c, size_t bytes, void **dev)
{
    CTX_TRY(c);
    if (!dev) return set_error(MCRT_ERR_INVALID, "null out pointer");
    HIP_TRY(hipMalloc(dev, bytes ? bytes : 1));
    return MCRT_OK;
}
extern "C" int mcrt_free(mcrt_ctx *c, void *dev) { CTX_TRY(c); HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(dev)); return MCRT_OK; }
extern "C" int mcrt_memcpy_d2h(mcrt_ctx *c, void *host, const void *dev, size_t bytes)
{
    CTX_TRY(c);
    HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MCRT_OK;
}
extern "C" int mcrt_memcpy_h2d(mcrt_ctx *c, void *dev, const void *host, size_t bytes)
{
    CTX_TRY(c);
    HIP_TRY(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MCRT_OK;
}

extern "C" int mcrt_enable_stats(mcrt_ctx *c, int on) { CTX_TRY(c); c->stats_on = on != 0; return MCRT_OK; }
extern "C" int mcrt_get_stats(mcrt_ctx *c, mcrt_stats *out, int reset)
{
    CTX_TRY(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long v[6];
    HIP_TRY(hipMemcpy(v, c->d_stats, sizeof v, hipMemcpyDeviceToHost));
    if (out) { out->queries = v[0]; out->nodes_visited = v[1]; out->tris_tested = v[2]; out->segments = v[3]; out->rf_steps = v[4]; out->hits = v[5]; }
    if (reset) { HIP_TRY(hipMemsetAsync(c->d_stats, 0, sizeof v, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); }
    return MCRT_OK;
}

// diagnostic builds (-DMCRT_STAMP): per-phase cycle sums of k_trace [0,16) and its per-bounce launch timeline [16,120), k_march's sections [120,130) (the timeline alone: -DMCRT_STAMP_LITE); zeros otherwise
extern "C" int mcrt_debug_stamps(mcrt_ctx *c, uint64_t out[200], int reset)
{
    CTX_TRY(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->d_stats + 8, 200 * 8, hipMemcpyDeviceToHost));
    if (reset) { HIP_TRY(hipMemsetAsync(c->d_stats + 8, 0, 200 * 8, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); }
    return MCRT_OK;
}

extern "C" int mcrt_debug_tail_histograms(mcrt_ctx *c, uint64_t out[2560], int reset)
{
    CTX_TRY(c);
    if (!out) return set_error(MCRT_ERR_INVALID, "null out pointer");
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, c->d_stats + 256, 2560 * 8, hipMemcpyDeviceToHost));
    if (reset) { HIP_TRY(hipMemsetAsync(c->d_stats + 256, 0, 2560 * 8, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); }
    return MCRT_OK;
}

extern "C" int mcrt_debug_fast_paths(mcrt_ctx *c, uint32_t out[4])
{
    CTX_TRY(c);
    if (!out) return set_error(MCRT_ERR_INVALID, "null out pointer");
    out[0] = c->fast_div ? 1u : 0u; out[1] = c->last_lean_bound > 0.0f ? 1u : 0u; out[2] = c->last_march_rows; out[3] = 0u;
    return MCRT_OK;
}

extern "C" int mcrt_debug_set_error(mcrt_ctx *c, uint32_t bits)
{
    CTX_TRY(c);
    if (!c->knobs.test_hooks) return set_error(MCRT_ERR_INVALID, "mcrt_debug_set_error is a test hook: create the context with MCRT_TEST_HOOKS set in the environment");
    uint32_t e = 0;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(&e, c->d_error, 4, hipMemcpyDeviceToHost));
    e |= bits;
    HIP_TRY(hipMemcpy(c->d_error, &e, 4, hipMemcpyHostToDevice));
    return MCRT_OK;
}

extern "C" int mcrt_enable_timing(mcrt_ctx *c, int on) { CTX_TRY(c); c->timing_on = on != 0; c->timing_level = on; return MCRT_OK; }
extern "C" int mcrt_get_kernel_times(mcrt_ctx *c, double avg_ms[3], uint32_t n[3], int reset)
{
    CTX_TRY(c);
    // level 1: every event was recorded on the context's stream -- wait for that stream only (a caller polling the walk's time must not stall on other
    // contexts of the device: a group's other ranks on the root GPU, a host application's own streams); level 2: k_march's events live on the side streams,
    // so each recorded pair is waited for by itself
    if (c->timing_level < 2) HIP_TRY(hipStreamSynchronize(c->stream));
    else for (size_t i = 0; i < c->ev_used; i++) HIP_TRY(hipEventSynchronize(c->ev[i].second));
    double sum[3] = { 0, 0, 0 }; uint32_t cnt[3] = { 0, 0, 0 };
    for (size_t i = 0; i < c->ev_used; i++) {
        float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, c->ev[i].first, c->ev[i].second));
        const int k = c->ev_kind[i] < 3 ? c->ev_kind[i] : 0;
        sum[k] += ms; cnt[k]++;
    }
    for (int k = 0; k < 3; k++) { if (avg_ms) avg_ms[k] = cnt[k] ? sum[k] / (double)cnt[k] : 0.0; if (n) n[k] = cnt[k]; }
    if (reset) c->ev_used = 0;
    return MCRT_OK;
}
extern "C" int mcrt_get_kernel_time(mcrt_ctx *c, double *avg_ms, uint32_t *n, int reset)
{
    double a[3]; uint32_t k[3];
    const int rc = mcrt_get_kernel_times(c, a, k, reset); if (rc) return rc;
    if (avg_ms) *avg_ms = a[0];
    if (n) *n = k[0];
    return MCRT_OK;
}

extern "C" int mcrt_debug_math(mcrt_ctx *c, int op, const double *x, const double *y, double *out, uint32_t n)
{
    CTX_TRY(c);
    if (!x || !out || n == 0) return set_error(MCRT_ERR_INVALID, "mcrt_debug_math: bad arguments");
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&dx, 8 * (size_t)n)); HIP_TRY(hipMalloc(&dout, 8 * (size_t)n));
    HIP_TRY(hipMemcpy(dx, x, 8 * (size_t)n, hipMemcpyHostToDevice));
    if (y) { HIP_TRY(hipMalloc(&dy, 8 * (size_t)n)); HIP_TRY(hipMemcpy(dy, y, 8 * (size_t)n, hipMemcpyHostToDevice)); }
    HIP_TRY(mcrt::launch_math_probe(op, dx, dy, dout, n, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, dout, 8 * (size_t)n, hipMemcpyDeviceToHost));
    hipFree(dx); hipFree(dy); hipFree(dout);
    return MCRT_OK;
}

extern "C" int mcrt_debug_philox(mcrt_ctx *c, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    CTX_TRY(c);
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc(&d, 16));
    HIP_TRY(mcrt::launch_philox_probe(ctr, key, d, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(out, d, 16, hipMemcpyDeviceToHost));
    hipFree(d);
    return MCRT_OK;
}
