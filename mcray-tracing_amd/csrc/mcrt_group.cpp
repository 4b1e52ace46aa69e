// mcrt_group.cpp -- several GPUs of one node behind the C-ABI (include/mcrt.h, mcrt_group_*): SURVEY 8(e).
//
// The reference's frame loop (main.cpp:92-152) is one thread on one CPU; its paths are independent and deposit only into their own
// scan-line's column (main.cpp:128,139), so a frame shards by contiguous scan-line blocks with nothing exchanged inside the path.
// A group = one tracing context per listed device, each fed by its OWN host thread (a pass is ~35 launches: eight GPUs fed from one
// thread would queue behind each other's launch latency), and a root context on devices[0] that receives the blocks:
//
//   rank g, its trace stream :  [wait: block buffer i free]  mcrt_trace_frames(shard g) -> blk_g[i]          (device g)
//   rank g, its copy stream  :  [wait: traced, staging i free]  hipMemcpyPeerAsync blk_g[i] -> stage[i] + offset_g   (xGMI, one hop)
//   root stream              :  [wait: every rank copied]  k_blocks_to_frames stage[i] -> rf_dev [F][E][R]   (device 0)
//
// i = pass & 1: a rank's next pass starts on the other block buffer without waiting for the root, so the root's post-processing of
// pass k (PSF, envelope, scan conversion: rfimage.h:93-140, they need neighbouring columns) overlaps the ranks' trace of pass k+1.
// The blocks are 60 KB per frame and rank: latency-bound, single-hop -- a peer copy per rank, not a ring collective (the multi-PROCESS
// path, one process per GPU under torch.distributed, gathers the same blocks with one RCCL collective: mcray-tracing_amd/dist.py).
#include "../../include/mcrt.h"
#include "mcrt_internal.h"
#include "mcrt_kernels.h"

#include <hip/hip_runtime.h>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using mcrt::set_error;

namespace mcrt {
hipStream_t ctx_stream(mcrt_ctx *c);      // mcrt_api.cpp
}

#define G_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { (void)hipGetLastError(); return set_error(e_ == hipErrorOutOfMemory ? MCRT_ERR_NOMEM : MCRT_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } } while (0)

namespace {
struct Member {
    int device = 0;
    mcrt_ctx *ctx = nullptr;
    hipStream_t trace = nullptr, copy = nullptr;
    hipEvent_t ev_t0[2] = {}, ev_traced[2] = {}, ev_c0[2] = {}, ev_copied[2] = {};
    bool used[2] = { false, false };                 // the events of slot i have been recorded at least once
    float *blk[2] = { nullptr, nullptr }; size_t blk_cap[2] = { 0, 0 };
    int last_slot = -1;
    uint64_t copied_pass = 0;                        // the pass (1-based) whose peer copy this rank enqueued last
};
}  // namespace

struct mcrt_group {
    std::vector<Member> mem;
    int root_device = 0;
    mcrt_ctx *root = nullptr;
    mcrt_params p{};
    float *stage[2] = { nullptr, nullptr }; size_t stage_cap[2] = { 0, 0 };
    hipEvent_t ev_reordered[2] = {}, ev_root_now[2] = {}; bool reordered_used[2] = { false, false };
    uint64_t pass = 0;
    int builder = MCRT_BVH_HOST_SAH;                 // what mcrt_group_set_bvh_builder last set on every rank
    std::vector<uint32_t> tri_mesh;                  // per-triangle mesh index of the uploaded scene (the host builder of mcrt_group_update_triangles needs it)
    double last_build_s = 0.0, last_upload_s = 0.0;  // host SAH build (once, calling thread) and the ranks' uploads (concurrent) of the last scene call
    // the ranks' host threads
    std::vector<std::thread> threads;
    std::mutex mu; std::condition_variable cv_job, cv_done;
    uint64_t gen = 0; uint32_t pending = 0; bool stop = false;
    std::function<int(uint32_t)> job;
    std::vector<int> rcs; std::vector<std::string> errs;
};

static void worker(mcrt_group *g, uint32_t r)
{
    uint64_t seen = 0;
    (void)hipSetDevice(g->mem[r].device);
    for (;;) {
        std::function<int(uint32_t)> job;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_job.wait(lk, [&] { return g->stop || g->gen != seen; });
            if (g->stop) return;
            seen = g->gen; job = g->job;
        }
        const int rc = job(r);
        std::string msg = rc ? mcrt_last_error() : "";
        {
            std::lock_guard<std::mutex> lk(g->mu);
            g->rcs[r] = rc; g->errs[r] = std::move(msg);
            if (--g->pending == 0) g->cv_done.notify_all();
        }
    }
}

// fn(rank) on every rank's thread, concurrently; the first failing rank's status (its message prefixed with the rank)
static int run_all(mcrt_group *g, std::function<int(uint32_t)> fn)
{
    {
        std::unique_lock<std::mutex> lk(g->mu);
        g->job = std::move(fn); g->pending = (uint32_t)g->mem.size(); g->gen++;
        std::fill(g->rcs.begin(), g->rcs.end(), 0);
        g->cv_job.notify_all();
        g->cv_done.wait(lk, [&] { return g->pending == 0; });
    }
    for (size_t r = 0; r < g->mem.size(); r++)
        if (g->rcs[r]) return set_error(g->rcs[r], "rank %zu: %s", r, g->errs[r].c_str());
    return MCRT_OK;
}

extern "C" int mcrt_group_shard(uint32_t rank, uint32_t n_ranks, uint32_t n_elements, uint32_t *e_begin, uint32_t *e_end)
{
    if (n_ranks == 0 || rank >= n_ranks || !e_begin || !e_end) return set_error(MCRT_ERR_INVALID, "mcrt_group_shard: bad arguments");
    const uint32_t base = n_elements / n_ranks, rem = n_elements % n_ranks;
    *e_begin = rank * base + (rank < rem ? rank : rem);
    *e_end = *e_begin + base + (rank < rem ? 1u : 0u);
    return MCRT_OK;
}

extern "C" int mcrt_group_destroy(mcrt_group *g)
{
    if (!g) return MCRT_OK;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->stop = true;
        g->cv_job.notify_all();
    }
    for (auto &t : g->threads) if (t.joinable()) t.join();
    for (Member &m : g->mem) {
        (void)hipSetDevice(m.device);
        if (m.copy) (void)hipStreamSynchronize(m.copy);
        if (m.trace) (void)hipStreamSynchronize(m.trace);
        if (m.ctx) mcrt_destroy(m.ctx);                      // (waits for the device)
        for (int i = 0; i < 2; i++) {
            if (m.blk[i]) (void)hipFree(m.blk[i]);
            for (hipEvent_t e : { m.ev_t0[i], m.ev_traced[i], m.ev_c0[i], m.ev_copied[i] }) if (e) (void)hipEventDestroy(e);
        }
        if (m.copy) (void)hipStreamDestroy(m.copy);
        if (m.trace) (void)hipStreamDestroy(m.trace);
    }
    (void)hipSetDevice(g->root_device);
    if (g->root) mcrt_destroy(g->root);
    for (int i = 0; i < 2; i++) {
        if (g->stage[i]) (void)hipFree(g->stage[i]);
        if (g->ev_reordered[i]) (void)hipEventDestroy(g->ev_reordered[i]);
        if (g->ev_root_now[i]) (void)hipEventDestroy(g->ev_root_now[i]);
    }
    delete g;
    return MCRT_OK;
}

extern "C" int mcrt_group_create(const int *devices, uint32_t n, mcrt_group **out)
{
    if (!out) return set_error(MCRT_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (!devices || n == 0 || n > 64) return set_error(MCRT_ERR_INVALID, "mcrt_group_create: 1..64 devices");
    mcrt_group *g = new (std::nothrow) mcrt_group();
    if (!g) return set_error(MCRT_ERR_NOMEM, "out of host memory");
    g->mem.resize(n); g->rcs.assign(n, 0); g->errs.assign(n, "");
    g->root_device = devices[0];
    int rc = mcrt_create(devices[0], &g->root);
    for (uint32_t r = 0; r < n && !rc; r++) {
        Member &m = g->mem[r];
        m.device = devices[r];
        rc = mcrt_create(m.device, &m.ctx);
        if (rc) break;
        hipError_t e = hipSetDevice(m.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m.trace, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m.copy, hipStreamNonBlocking);
        for (int i = 0; i < 2 && e == hipSuccess; i++) {
            e = hipEventCreate(&m.ev_t0[i]);
            if (e == hipSuccess) e = hipEventCreate(&m.ev_traced[i]);
            if (e == hipSuccess) e = hipEventCreate(&m.ev_c0[i]);
            if (e == hipSuccess) e = hipEventCreate(&m.ev_copied[i]);
        }
        if (e == hipSuccess && m.device != g->root_device) {          // xGMI peer mapping where the platform offers it (the peer copy works either way)
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, m.device, g->root_device) == hipSuccess && can) {
                const hipError_t pe = hipDeviceEnablePeerAccess(g->root_device, 0);
                if (pe != hipSuccess) (void)hipGetLastError();          // (already enabled by another rank of this process: fine)
            }
        }
        if (e != hipSuccess) { (void)hipGetLastError(); rc = set_error(MCRT_ERR_HIP, "mcrt_group_create: rank %u on device %d: %s", r, m.device, hipGetErrorString(e)); break; }
        rc = mcrt_set_stream(m.ctx, m.trace);
    }
    if (!rc) {
        hipError_t e = hipSetDevice(g->root_device);
        for (int i = 0; i < 2 && e == hipSuccess; i++) {
            e = hipEventCreateWithFlags(&g->ev_reordered[i], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_root_now[i], hipEventDisableTiming);
        }
        if (e != hipSuccess) { (void)hipGetLastError(); rc = set_error(MCRT_ERR_HIP, "mcrt_group_create: %s", hipGetErrorString(e)); }
    }
    if (rc) { const std::string keep = mcrt_last_error(); mcrt_group_destroy(g); return set_error(rc, "%s", keep.c_str()); }
    mcrt_get_params(g->root, &g->p);
    for (uint32_t r = 0; r < n; r++) g->threads.emplace_back(worker, g, r);
    *out = g;
    return MCRT_OK;
}

extern "C" int mcrt_group_size(const mcrt_group *g) { return g ? (int)g->mem.size() : 0; }
extern "C" mcrt_ctx *mcrt_group_root(mcrt_group *g) { return g ? g->root : nullptr; }
extern "C" mcrt_ctx *mcrt_group_member(mcrt_group *g, uint32_t rank) { return (g && rank < g->mem.size()) ? g->mem[rank].ctx : nullptr; }

#define GRP_TRY(g) do { if (!(g)) return set_error(MCRT_ERR_INVALID, "null group"); } while (0)

// The ranks take the new parameters first and the root last: a set of parameters some context refuses leaves the group on the old ones
// (the ranks that had accepted are put back), never half on each.
extern "C" int mcrt_group_set_params(mcrt_group *g, const mcrt_params *p)
{
    GRP_TRY(g);
    if (!p) return set_error(MCRT_ERR_INVALID, "null params");
    const mcrt_params pp = *p, old = g->p;
    int rc = run_all(g, [g, pp](uint32_t r) { return mcrt_set_params(g->mem[r].ctx, &pp); });
    if (!rc) rc = mcrt_set_params(g->root, &pp);
    if (rc) {
        const std::string keep = mcrt_last_error();
        (void)run_all(g, [g, old](uint32_t r) { return mcrt_set_params(g->mem[r].ctx, &old); });
        (void)mcrt_set_params(g->root, &old);
        return set_error(rc, "%s", keep.c_str());
    }
    g->p = pp;
    return MCRT_OK;
}

extern "C" int mcrt_group_set_bvh_builder(mcrt_group *g, int builder)
{
    GRP_TRY(g);
    const int rc = run_all(g, [g, builder](uint32_t r) { return mcrt_set_bvh_builder(g->mem[r].ctx, builder); });
    if (!rc) g->builder = builder;
    return rc;
}

static int host_pointer_only(const void *p, const char *who)
{
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice)
        return set_error(MCRT_ERR_INVALID, "%s: a device pointer belongs to one GPU; the group calls take host memory", who);
    (void)hipGetLastError();
    return MCRT_OK;
}

// THE HOST BUILDER RUNS ONCE.  The SAH build (mcrt_build_bvh + mcrt_build_bvh4: 0.7 s for 1 M triangles, itself multi-threaded) is host work that
// does not depend on the GPU: the calling thread builds the tree, every rank's thread copies it and uploads (rounds 4 ran the single-context
// call on every rank: N identical builds racing for the host's cores).  The device builder (MCRT_BVH_DEVICE_LBVH) needs no host work and runs
// per rank, on its own GPU.
namespace {
struct BuiltOnce {
    mcrt_bvh bvh{}; mcrt_bvh4 bvh4{}; mcrt::HostTree tree{ nullptr, nullptr }; bool have = false;
    ~BuiltOnce() { if (have) { mcrt_free_bvh(&bvh); mcrt_free_bvh4(&bvh4); } }
    int build(const float *tri, const uint32_t *tri_mesh, uint32_t n_tri)
    {
        int rc = mcrt_build_bvh(tri, tri_mesh, n_tri, &bvh);
        if (!rc) { have = true; rc = mcrt_build_bvh4(&bvh, &bvh4); }
        if (!rc) { tree.bvh = &bvh; tree.bvh4 = &bvh4; }
        return rc;
    }
};
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

extern "C" int mcrt_group_upload_scene(mcrt_group *g, const float *tri, const uint32_t *tri_mesh, uint32_t n_tri, const mcrt_mesh *meshes, uint32_t n_mesh,
                                       const float *mats, uint32_t n_mat, uint32_t start_mat, const float spacing[3])
{
    GRP_TRY(g);
    if (n_tri && (!tri || !tri_mesh)) return set_error(MCRT_ERR_INVALID, "mcrt_group_upload_scene: missing triangles");
    if (tri) { int rc = host_pointer_only(tri, "mcrt_group_upload_scene"); if (rc) return rc; }
    BuiltOnce once;
    const double t0 = now_s();
    if (n_tri && g->builder == MCRT_BVH_HOST_SAH) {
        for (uint32_t i = 0; i < n_tri; i++)          // (mcrt_upload_scene's own check, needed before the builder reads the table)
            if (tri_mesh[i] >= n_mesh) return set_error(MCRT_ERR_INVALID, "triangle %u references mesh %u out of range", i, tri_mesh[i]);
        int rc = once.build(tri, tri_mesh, n_tri); if (rc) return rc;
    }
    const double t1 = now_s();
    const mcrt::HostTree *pre = once.tree.bvh ? &once.tree : nullptr;
    const int rc = run_all(g, [=](uint32_t r) { return mcrt::upload_scene_with_tree(g->mem[r].ctx, tri, tri_mesh, n_tri, meshes, n_mesh, mats, n_mat, start_mat, spacing, pre); });
    g->last_build_s = t1 - t0; g->last_upload_s = now_s() - t1;
    if (!rc) { if (n_tri) g->tri_mesh.assign(tri_mesh, tri_mesh + n_tri); else g->tri_mesh.clear(); }
    return rc;
}

extern "C" int mcrt_group_update_triangles(mcrt_group *g, const float *tri, uint32_t n_tri)
{
    GRP_TRY(g);
    if (!tri) return set_error(MCRT_ERR_INVALID, "null triangles");
    { int rc = host_pointer_only(tri, "mcrt_group_update_triangles"); if (rc) return rc; }
    BuiltOnce once;
    const double t0 = now_s();
    if (g->builder == MCRT_BVH_HOST_SAH && n_tri != 0 && n_tri == g->tri_mesh.size()) {      // (a wrong count is the ranks' error to report)
        int rc = once.build(tri, g->tri_mesh.data(), n_tri); if (rc) return rc;
    }
    const double t1 = now_s();
    const mcrt::HostTree *pre = once.tree.bvh ? &once.tree : nullptr;
    const int rc = run_all(g, [=](uint32_t r) { return mcrt::update_triangles_with_tree(g->mem[r].ctx, tri, n_tri, pre); });
    g->last_build_s = t1 - t0; g->last_upload_s = now_s() - t1;
    return rc;
}

extern "C" int mcrt_group_last_scene_seconds(mcrt_group *g, double *build_s, double *upload_s)
{
    GRP_TRY(g);
    if (build_s) *build_s = g->last_build_s;
    if (upload_s) *upload_s = g->last_upload_s;
    return MCRT_OK;
}

extern "C" int mcrt_group_refit_triangles(mcrt_group *g, const float *tri, uint32_t n_tri)
{
    GRP_TRY(g);
    if (!tri) return set_error(MCRT_ERR_INVALID, "null triangles");
    { int rc = host_pointer_only(tri, "mcrt_group_refit_triangles"); if (rc) return rc; }
    return run_all(g, [=](uint32_t r) { return mcrt_refit_triangles(g->mem[r].ctx, tri, n_tri); });
}

extern "C" int mcrt_group_upload_texture(mcrt_group *g, const float *vox, uint32_t n)
{
    GRP_TRY(g);
    if (n == 0) return set_error(MCRT_ERR_INVALID, "texture size 0");
    std::vector<float> gen;
    if (!vox) {                                  // volume<>::volume() (volume.h:19-35) once, not once per rank
        gen.resize((size_t)n * n * n * 2);
        int rc = mcrt_generate_texture(gen.data(), n); if (rc) return rc;
        vox = gen.data();
    }
    return run_all(g, [=](uint32_t r) { return mcrt_upload_texture(g->mem[r].ctx, vox, n); });
}

extern "C" int mcrt_group_set_transducer(mcrt_group *g, const float *pos, const float *dir, uint32_t n)
{
    GRP_TRY(g);
    return run_all(g, [=](uint32_t r) { return mcrt_set_transducer(g->mem[r].ctx, pos, dir, n); });
}

static int ensure_dev(float **buf, size_t *cap, size_t bytes, int device)
{
    if (*cap >= bytes) return MCRT_OK;
    G_HIP(hipSetDevice(device));
    G_HIP(hipDeviceSynchronize());                  // (a growing buffer: rare; everything that may still read the old one is waited for)
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr; *cap = 0;
    G_HIP(hipMalloc((void **)buf, bytes));
    *cap = bytes;
    return MCRT_OK;
}

static int group_trace(mcrt_group *g, uint32_t frame, uint32_t F, const float *pos, const float *dir, float *rf_dev)
{
    GRP_TRY(g);
    if (!rf_dev) return set_error(MCRT_ERR_INVALID, "null rf_dev");
    if (F == 0 || F > 1024) return set_error(MCRT_ERR_LIMIT, "n_frames must be 1..1024");
    const uint32_t G = (uint32_t)g->mem.size(), E = g->p.n_elements, R = g->p.n_rows;
    if (E < G) return set_error(MCRT_ERR_INVALID, "%u scan-lines cannot be cut over %u ranks", E, G);
    const int i = (int)(g->pass & 1u);
    g->pass++;
    const uint64_t this_pass = g->pass;                 // (>= 1: a Member's copied_pass starts at 0)
    std::vector<uint32_t> off(G + 1);
    for (uint32_t r = 0; r < G; r++) { uint32_t b, e; mcrt_group_shard(r, G, E, &b, &e); off[r] = b; off[r + 1] = e; }
    // one frame (or one rank): the blocks are contiguous pieces of the frame and land in place; otherwise they are staged on the root
    // device rank after rank and one kernel interleaves them into frames
    const bool staged = F > 1 && G > 1;
    const size_t frame_bytes = (size_t)E * R * sizeof(float);
    if (staged) { int rc = ensure_dev(&g->stage[i], &g->stage_cap[i], frame_bytes * F, g->root_device); if (rc) return rc; }
    G_HIP(hipSetDevice(g->root_device));
    hipStream_t root_stream = mcrt::ctx_stream(g->root);
    G_HIP(hipEventRecord(g->ev_root_now[i], root_stream));        // what the root's stream holds NOW (earlier readers of rf_dev) precedes the copies into it
    float *const stage = g->stage[i];
    int rc = run_all(g, [=](uint32_t r) -> int {
        Member &m = g->mem[r];
        const uint32_t ne = off[r + 1] - off[r];
        const size_t bytes = (size_t)F * ne * R * sizeof(float);
        { int rc2 = ensure_dev(&m.blk[i], &m.blk_cap[i], bytes, m.device); if (rc2) return rc2; }
        G_HIP(hipSetDevice(m.device));
        if (m.used[i]) G_HIP(hipStreamWaitEvent(m.trace, m.ev_copied[i], 0));          // the copy of two passes ago has left this block buffer
        G_HIP(hipEventRecord(m.ev_t0[i], m.trace));
        const int rc2 = pos ? mcrt_trace_frames_poses(m.ctx, frame, F, off[r], off[r + 1], pos, dir, m.blk[i])
                            : mcrt_trace_frames(m.ctx, frame, F, off[r], off[r + 1], m.blk[i]);
        if (rc2) return rc2;
        G_HIP(hipEventRecord(m.ev_traced[i], m.trace));
        G_HIP(hipStreamWaitEvent(m.copy, m.ev_traced[i], 0));
        if (staged) { if (g->reordered_used[i]) G_HIP(hipStreamWaitEvent(m.copy, g->ev_reordered[i], 0)); }   // the staging buffer's last reader
        else G_HIP(hipStreamWaitEvent(m.copy, g->ev_root_now[i], 0));
        float *dst = staged ? stage + (size_t)F * off[r] * R : rf_dev + (size_t)off[r] * R;
        G_HIP(hipEventRecord(m.ev_c0[i], m.copy));
        if (m.device == g->root_device) G_HIP(hipMemcpyAsync(dst, m.blk[i], bytes, hipMemcpyDeviceToDevice, m.copy));
        else G_HIP(hipMemcpyPeerAsync(dst, g->root_device, m.blk[i], m.device, bytes, m.copy));
        G_HIP(hipEventRecord(m.ev_copied[i], m.copy));
        m.used[i] = true; m.last_slot = i; m.copied_pass = this_pass;
        return MCRT_OK;
    });
    // Also when a rank failed: the ranks that succeeded have peer copies into rf_dev / the staging buffer in flight on their own copy streams.
    // The root's stream is ordered behind every copy that WAS enqueued in this pass, so a caller that handles the error by reusing or freeing
    // rf_dev behind the root's stream (mcrt_synchronize(mcrt_group_root(g)), or more work on that stream) does not race with them.
    G_HIP(hipSetDevice(g->root_device));
    for (uint32_t r = 0; r < G; r++)
        if (g->mem[r].copied_pass == this_pass) G_HIP(hipStreamWaitEvent(root_stream, g->mem[r].ev_copied[i], 0));
    if (rc) return rc;
    if (staged) {
        G_HIP(mcrt::launch_blocks_to_frames(stage, rf_dev, F, E, R, G, off.data(), root_stream));
        G_HIP(hipEventRecord(g->ev_reordered[i], root_stream));
        g->reordered_used[i] = true;
    }
    return MCRT_OK;
}

extern "C" int mcrt_group_trace_frames(mcrt_group *g, uint32_t frame, uint32_t n_frames, float *rf_dev)
{
    return group_trace(g, frame, n_frames, nullptr, nullptr, rf_dev);
}

extern "C" int mcrt_group_trace_frames_poses(mcrt_group *g, uint32_t frame, uint32_t n_frames, const float *pos, const float *dir, float *rf_dev)
{
    if (!pos || !dir) return set_error(MCRT_ERR_INVALID, "mcrt_group_trace_frames_poses: null pose tables");
    { int rc = host_pointer_only(pos, "mcrt_group_trace_frames_poses"); if (rc) return rc; }
    { int rc = host_pointer_only(dir, "mcrt_group_trace_frames_poses"); if (rc) return rc; }
    return group_trace(g, frame, n_frames, pos, dir, rf_dev);
}

extern "C" int mcrt_group_synchronize(mcrt_group *g)
{
    GRP_TRY(g);
    int rc = run_all(g, [g](uint32_t r) -> int {
        Member &m = g->mem[r];
        G_HIP(hipSetDevice(m.device));
        const int rc2 = mcrt_synchronize(m.ctx);                // the rank's trace stream + its device error word
        G_HIP(hipStreamSynchronize(m.copy));
        return rc2;
    });
    const int rc_root = mcrt_synchronize(g->root);
    return rc ? rc : rc_root;
}

extern "C" int mcrt_group_last_pass_ms(mcrt_group *g, float *trace_ms, float *copy_ms)
{
    GRP_TRY(g);
    for (size_t r = 0; r < g->mem.size(); r++) {
        Member &m = g->mem[r];
        if (trace_ms) trace_ms[r] = 0.0f;
        if (copy_ms) copy_ms[r] = 0.0f;
        if (m.last_slot < 0 || !m.used[m.last_slot]) continue;      // (no pass yet, or its events were never recorded: a pass that failed on this rank)
        const int i = m.last_slot;
        G_HIP(hipSetDevice(m.device));
        G_HIP(hipEventSynchronize(m.ev_copied[i]));
        if (trace_ms) G_HIP(hipEventElapsedTime(&trace_ms[r], m.ev_t0[i], m.ev_traced[i]));
        if (copy_ms) G_HIP(hipEventElapsedTime(&copy_ms[r], m.ev_c0[i], m.ev_copied[i]));
    }
    return MCRT_OK;
}
