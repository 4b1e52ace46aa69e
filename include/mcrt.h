/*
 * mcrt.h -- C-ABI of libmcrt_hip.so: the MI355X (gfx950) implementation of the Monte-Carlo
 * ultrasound ray-tracing hot path of thepochynsons/MCRay-Tracing.
 *
 * The reference has no FFI layer; its seam is C++ (paths relative to /root/reference/src):
 *   scene::scene(json, transducer&)            scene.h:24, scene.cpp:16-48   -> mcrt_upload_scene
 *   transducer<N>::element(i)                  transducer.h:64-67           -> mcrt_set_transducer
 *   volume<256,145> texture_volume             main.cpp:52, volume.h:19-35  -> mcrt_upload_texture
 *   rf_image.clear()                           main.cpp:102, rfimage.h:161  -> (inside mcrt_trace_frame)
 *   scene.cast_rays<S,E>(transducer)           main.cpp:104, scene.cpp:50   -> mcrt_trace_frame / mcrt_cast_rays
 *   accumulation loop + rf_image::add_echo     main.cpp:106-144, rfimage.h:33-40 -> (fused in mcrt_trace_frame)
 *   rf_image.convolve(psf)                     main.cpp:146, rfimage.h:93-123 -> mcrt_convolve
 *   rf_image.envelope() / postprocess()        main.cpp:147-148, rfimage.h:54-91,125-140 -> mcrt_envelope / mcrt_scan_convert
 *   transducer<N>::update() between frames     transducer.h:82-118, main.cpp:100 -> mcrt_set_transducer, mcrt_trace_frames_poses
 * A maintainer of the reference replaces main.cpp:102-148 with the calls shown in INTEGRATION.md.
 *
 * Conventions: every function returns 0 on success or a negative mcrt_status; the message is
 * available from mcrt_last_error() (thread-local).  No exceptions cross this boundary.  A
 * context belongs to one GPU and is used from one host thread at a time.  Pointers named
 * *_dev are device pointers on the context's GPU; everything else is host memory.  Uploads copy.
 * There is NO CPU fallback: without a usable GPU mcrt_create fails.
 */
#ifndef MCRT_H
#define MCRT_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCRT_VERSION 106   /* round 3: + mcrt_trace_frames_poses, mcrt_envelope_frames, mcrt_scan_convert_frames; the slab rule of the closest-hit contract is one fma per plane;
                              104: + the test hooks mcrt_debug_set_error, mcrt_debug_fast_paths; RF images are NaN while the device error word is set;
                              105 (round 4): + mcrt_group_* (several GPUs behind one call), mcrt_scan_maps; the scan-conversion maps follow the reference's float promotions;
                              106 (round 5): environment knobs are only read under MCRT_TUNING=1; the HIP-graph replay of passes (MCRT_GRAPH) is gone */

typedef enum {
    MCRT_OK = 0,
    MCRT_ERR_INVALID = -1,      /* bad argument / call order               */
    MCRT_ERR_HIP = -2,          /* HIP runtime error (message has details) */
    MCRT_ERR_NOMEM = -3,
    MCRT_ERR_NO_DEVICE = -4,
    MCRT_ERR_LIMIT = -5         /* a documented capacity was exceeded      */
} mcrt_status;

typedef struct mcrt_ctx mcrt_ctx;

/* Run-time form of the reference's compile-time constants (main.cpp:23-37, ray.h:23-24,
 * scene.h:49, scene.cpp:115).  mcrt_default_params() fills the reference values. */
typedef struct {
    uint32_t n_elements;         /* E: scan-lines = transducer elements = RF columns (512)   */
    uint32_t n_samples;          /* S: Monte-Carlo sample paths per scan-line (5)            */
    uint32_t max_depth;          /* B: ray::max_depth (10), <= 16                            */
    uint32_t n_rows;             /* R: RF rows (465 = max_rows), <= 2048                     */
    float    frequency;          /* MHz (4.5)                                                */
    float    intensity_epsilon;  /* 1e-10                                                    */
    float    initial_intensity;  /* 1.0                                                      */
    float    ray_start_offset;   /* 0.1 (scene.cpp:115)                                      */
    uint32_t speed_of_sound;     /* um/us (1500)                                             */
    double   depth_cm;           /* 15                                                       */
    uint32_t seed;               /* RNG key word 0; key word 1 is the frame id               */
    uint32_t sanitize_tir;       /* 0 = reference behaviour: NaN echo on total internal reflection */
    uint32_t tex_n;              /* texture edge in voxels (256)                             */
    float    tex_res;            /* voxel edge, scene units (0.145)                          */
} mcrt_params;

/* mesh.h:12-20 reduced to what the hot path reads */
typedef struct { uint32_t mat_inside, mat_outside, vascular, _pad; } mcrt_mesh;

/* 64-byte BVH2 node.  child >= 0: inner node index; child < 0: leaf, v = ~child,
 * first triangle = v >> 3, count = (v & 7) + 1 (positions in leaf order). */
typedef struct {
    float lo0[3]; int32_t c0;
    float hi0[3]; int32_t c1;
    float lo1[3]; uint32_t pad0;
    float hi1[3]; uint32_t pad1;
} mcrt_bvh_node;

typedef struct {
    uint32_t n_nodes, n_tri, max_depth;
    float pad_abs;           /* absolute part of the per-triangle bounds padding (4e-6 * scene scale) */
    mcrt_bvh_node *nodes;    /* [n_nodes]                                                     */
    float *tri;              /* [n_tri][12] leaf order: v0.xyz,bits(tri id) | v1.xyz,bits(mesh id) | v2.xyz,0 */
} mcrt_bvh;

/* 128-byte BVH4 node, the BUILDERS' form: four 32-byte child records with float boxes -- what mcrt_build_bvh4 and the device builder
 * emit, what a refit updates and what this ABI exports.  The GPU walk reads a 64-byte copy made from it after every build / refit
 * (child-transposed half-float boxes rounded outwards, csrc/mcrt_kernels.hip k_nodes_walk); mcrt_get_bvh4 hands out that copy decoded
 * back into this form, i.e. the tree exactly as walked.  ref: >= 0 inner node; < 0 leaf (as in mcrt_bvh_node, at most 4 triangles);
 * MCRT_BVH4_EMPTY = unused slot.  Built by collapsing the SAH BVH2. */
#define MCRT_BVH4_EMPTY ((int32_t)0x80000000)
typedef struct { float lo[3]; float hi_x; float hi_y, hi_z; int32_t ref; uint32_t pad; } mcrt_bvh4_child;
typedef struct { mcrt_bvh4_child c[4]; } mcrt_bvh4_node;
typedef struct {
    uint32_t n_nodes, max_stack;   /* max_stack: worst-case traversal stack entries for this tree */
    mcrt_bvh4_node *nodes;
} mcrt_bvh4;

/* ray_physics::segment (ray.h:28-36) as a POD; media is the material INDEX in effect along it */
typedef struct {
    float from[3], to[3], dir[3];
    float reflected_intensity, initial_intensity, attenuation;
    double distance_traveled;
    int32_t media;
    int32_t tri;             /* triangle hit at the end of the segment, -1 = none */
} mcrt_segment;              /* 64 bytes */

typedef struct {
    uint64_t queries, nodes_visited, tris_tested, segments, rf_steps, hits;
} mcrt_stats;

const char *mcrt_last_error(void);
int mcrt_version(void);
int mcrt_device_count(void);

int mcrt_create(int device, mcrt_ctx **out);
int mcrt_destroy(mcrt_ctx *ctx);
/* The stream every asynchronous entry point enqueues on.  Calls are ordered on the stream they were issued on; the one cross-stream
 * guarantee: a trace issued after mcrt_set_stream waits for the device work of the last scene upload / update / refit wherever that ran.
 *  NULL = the context's own (non-blocking) stream -- NOT the legacy
 * default stream: to order the kernels with work on HIP's legacy stream pass hipStreamLegacy explicitly, and with a
 * framework's stream (torch.cuda.Stream().cuda_stream) pass that handle. */
int mcrt_set_stream(mcrt_ctx *ctx, void *hip_stream);
/* Waits for the context's stream AND reads the context's device error word: MCRT_ERR_LIMIT when a launch since the last call was
 * abandoned (a persistent kernel's watchdog expired, a traversal stack ran out).  A caller that synchronises by other means (its own
 * stream, a framework's synchronize) must still call this to learn of such a launch; until it does -- the call clears the word --
 * every RF image the context finalises is NaN throughout, so that a broken frame cannot pass for an image. */
int mcrt_synchronize(mcrt_ctx *ctx);

int mcrt_default_params(mcrt_params *p);
int mcrt_set_params(mcrt_ctx *ctx, const mcrt_params *p);
int mcrt_get_params(mcrt_ctx *ctx, mcrt_params *out);           /* the parameters in effect */

/* Which builder mcrt_upload_scene / mcrt_update_triangles use for the BVH that replaces the per-mesh
 * btBvhTriangleMeshShape of scene.cpp:306-309:
 *   MCRT_BVH_HOST_SAH     binned-SAH build on the host (default; best trees, seconds for 1 M triangles)
 *   MCRT_BVH_DEVICE_LBVH  Morton-order LBVH built on the GPU (milliseconds; for moving geometry, the interactive path
 *                         the reference prepares in inputmanager.cpp:117-121 / transducer.h:82-118).  Needs >= 8 triangles.
 * Images do not depend on the builder: the closest-hit contract is independent of the hierarchy. */
enum { MCRT_BVH_HOST_SAH = 0, MCRT_BVH_DEVICE_LBVH = 1 };
int mcrt_set_bvh_builder(mcrt_ctx *ctx, int builder);

/* Geometry in WORLD space (scene.cpp:313-324 already applied: v*scaling + deltas*scaling^2 + origin),
 * triangles in OBJ face order, meshes in scene order.  Builds the BVH (see mcrt_set_bvh_builder) and uploads it.
 * materials: [n_mat][8] = impedance, attenuation, mu0, mu1, sigma, specularity, shininess, thickness. */
int mcrt_upload_scene(mcrt_ctx *ctx, const float *tri_xyz /*[T][9]*/, const uint32_t *tri_mesh /*[T]*/, uint32_t n_tri,
                      const mcrt_mesh *meshes, uint32_t n_mesh, const float *materials, uint32_t n_mat,
                      uint32_t start_mat, const float spacing[3]);
/* New vertex positions for the uploaded scene's triangles (same count, same order, same mesh / material tables):
 * re-indexes them with the selected builder.  tri_xyz may be a host or a device pointer. */
int mcrt_update_triangles(mcrt_ctx *ctx, const float *tri_xyz /*[T][9]*/, uint32_t n_tri);
/* The same, keeping the uploaded tree: every box is refitted bottom-up on the GPU around the moved triangles (about a
 * millisecond for 1 M triangles).  Right for deformations that keep the neighbourhoods intact; frames are exact either way,
 * only the walk gets slower when the tree no longer matches the geometry.  tri_xyz may be a host or a device pointer. */
int mcrt_refit_triangles(mcrt_ctx *ctx, const float *tri_xyz /*[T][9]*/, uint32_t n_tri);
/* voxels [n^3][2] = {texture_noise, scattering_probability}; NULL => generate the reference's texture */
int mcrt_upload_texture(mcrt_ctx *ctx, const float *voxels, uint32_t n);
int mcrt_set_transducer(mcrt_ctx *ctx, const float *pos /*[E][3]*/, const float *dir /*[E][3]*/, uint32_t n_elements);

/* clear + trace + accumulate for scan-lines [e_begin, e_end).  rf_dev: device float [(e_end-e_begin)][R]
 * (scan-line-major).  Asynchronous on the context's stream. */
int mcrt_trace_frame(mcrt_ctx *ctx, uint32_t frame_id, uint32_t e_begin, uint32_t e_end, float *rf_dev);
/* n_frames consecutive frames (ids frame_id .. frame_id+n_frames-1, same scene and probe pose) traced as ONE pass: every
 * stage of the pipeline then runs over n_frames times the rays, which is what fills the GPU when a single frame is small.
 * rf_dev: device float [n_frames][(e_end-e_begin)][R].  Each image is bit-identical to the one mcrt_trace_frame produces.
 * Limits (MCRT_ERR_LIMIT beyond them): n_frames <= 1024 and n_frames x scan-lines x samples <= 2^27 paths per pass (a path takes
 * about 600 bytes of work buffers). */
int mcrt_trace_frames(mcrt_ctx *ctx, uint32_t frame_id, uint32_t n_frames, uint32_t e_begin, uint32_t e_end, float *rf_dev);
/* The same pass with a probe pose PER FRAME: pos / dir are [n_frames][E][3] tables (host or device memory), frame f of the pass is
 * traced from the elements pos[f], dir[f] -- the moving probe the reference's loop is built for (transducer<N>::update(),
 * transducer.h:82-118; inputmanager.cpp:117-121; the frame loop main.cpp:92-152 reads the transducer anew every frame).  Each image is
 * bit-identical to mcrt_set_transducer(pos[f], dir[f]) followed by mcrt_trace_frame(frame_id + f).  The context's own transducer
 * (mcrt_set_transducer) is neither needed nor changed. */
/* Lifetime: a table in HOST memory is copied before the call returns (free or rewrite it at once); a table in DEVICE memory is read by
 * the pass on the context's stream -- keep it unchanged until that work has finished. */
int mcrt_trace_frames_poses(mcrt_ctx *ctx, uint32_t frame_id, uint32_t n_frames, uint32_t e_begin, uint32_t e_end,
                            const float *pos /*[F][E][3]*/, const float *dir /*[F][E][3]*/, float *rf_dev);
/* same, and additionally returns per-path data to HOST buffers (any may be NULL); synchronous.
 * hits [ne][S][B] int32 (-1 miss, -2 not cast); segs [ne][S][B]; seg_count [ne][S]. */
int mcrt_trace_frame_debug(mcrt_ctx *ctx, uint32_t frame_id, uint32_t e_begin, uint32_t e_end, float *rf_dev,
                           int32_t *hits, mcrt_segment *segs, uint32_t *seg_count);
/* scene::cast_rays (scene.cpp:50-183) alone: segments only, no RF accumulation; synchronous */
int mcrt_cast_rays(mcrt_ctx *ctx, uint32_t frame_id, uint32_t e_begin, uint32_t e_end,
                   mcrt_segment *segs, uint32_t *seg_count, int32_t *hits);

/* rf_image::convolve (rfimage.h:93-123) in place on a device image [E][R]; tmp_dev same size or NULL */
int mcrt_convolve(mcrt_ctx *ctx, float *rf_dev, uint32_t n_elements, uint32_t n_rows,
                  const float *axial, uint32_t n_ax, const float *lateral, uint32_t n_lat);
/* the same on the n_frames images [n_frames][E][R] of an mcrt_trace_frames pass, in one launch per convolution pass */
int mcrt_convolve_frames(mcrt_ctx *ctx, float *rf_dev, uint32_t n_frames, uint32_t n_elements, uint32_t n_rows,
                         const float *axial, uint32_t n_ax, const float *lateral, uint32_t n_lat);
/* rf_image::envelope (rfimage.h:54-91) in place on a device image [E][R].  n_rows <= 2048 (MCRT_ERR_LIMIT beyond: a wavefront holds
 * its scan-line in LDS) -- the limit mcrt_params.n_rows has anyway; an image brought in through mcrt_import_rf is bound by it too. */
int mcrt_envelope(mcrt_ctx *ctx, float *rf_dev, uint32_t n_elements, uint32_t n_rows);
/* the same on the n_frames images [n_frames][E][R] of a pass (main.cpp:147 once per frame), one launch */
int mcrt_envelope_frames(mcrt_ctx *ctx, float *rf_dev, uint32_t n_frames, uint32_t n_elements, uint32_t n_rows);
/* rf_image::postprocess scan conversion (rfimage.h:125-140,183-215), exact bilinear;
 * out_dev float [out_rows][out_cols] */
int mcrt_scan_convert(mcrt_ctx *ctx, const float *rf_dev, uint32_t n_elements, uint32_t n_rows,
                      double radius_mm, double total_angle_rad, float *out_dev, uint32_t out_rows, uint32_t out_cols);
/* rf_image::create_mapping (rfimage.h:183-215) alone, on the host (no GPU needed): map_row = the reference's map_x (ROW coordinate
 * in the RF image), map_col = its map_y (COLUMN coordinate), each [out_rows][out_cols] row-major -- what mcrt_scan_convert gathers
 * with.  max_travel_us / speed_of_sound are rf_image's unsigned template parameters (100, 1500; main.cpp:36). */
int mcrt_scan_maps(uint32_t n_elements, uint32_t n_rows, double radius_mm, double total_angle_rad, uint32_t max_travel_us,
                   uint32_t speed_of_sound, uint32_t out_rows, uint32_t out_cols, float *map_row, float *map_col);
/* the same on the n_frames images of a pass (main.cpp:148 once per frame), one launch; out_dev float [n_frames][out_rows][out_cols] */
int mcrt_scan_convert_frames(mcrt_ctx *ctx, const float *rf_dev, uint32_t n_frames, uint32_t n_elements, uint32_t n_rows,
                             double radius_mm, double total_angle_rad, float *out_dev, uint32_t out_rows, uint32_t out_cols);

/* device [E][R]  ->  host [R][E] row-major (the cv::Mat layout of rfimage.h:217); synchronous */
int mcrt_export_rf(mcrt_ctx *ctx, const float *rf_dev, uint32_t n_elements, uint32_t n_rows, float *host_rows_by_cols);

/* host [R][E] row-major (the cv::Mat layout)  ->  device [E][R]: the inverse of mcrt_export_rf, for callers that deposit
 * echoes on the host (rf_image::add_echo, rfimage.h:33-40) and post-process on the GPU; synchronous */
int mcrt_import_rf(mcrt_ctx *ctx, const float *host_rows_by_cols, uint32_t n_elements, uint32_t n_rows, float *rf_dev);

/* ---------------------------------------------------------------------------------------------------------------------------
 * Several GPUs of one node behind the same calls (the reference's frame loop main.cpp:92-152 is one GPU-less thread).
 * Paths are independent and deposit only into their own scan-line's column (main.cpp:128,139 use ray_i as the column), so the
 * scan-lines are cut into contiguous shards -- rank g of G traces [g*E/G, (g+1)*E/G), the first E % G ranks one more -- with scene,
 * texture and transducer replicated; every rank's [F][E_g][R] block then crosses xGMI once (hipMemcpyPeerAsync on the rank's own copy
 * stream) into GPU devices[0], where one kernel lays the blocks out as the [F][E][R] frames a single context would have produced, bit
 * for bit (RF bins are integer sums: no partition changes them).  PSF, envelope and scan conversion need neighbouring columns
 * (rfimage.h:113-118) and run on the gathered frames: call mcrt_convolve_frames / mcrt_envelope_frames / mcrt_scan_convert_frames on
 * mcrt_group_root().
 * A group owns one tracing context per listed device (each driven by its own host thread, so G GPUs are fed in parallel) and a root
 * context on devices[0] for the gathered frames.  A device may be listed more than once: its contexts then share that GPU (how the
 * one-GPU test box runs a two-rank group).  Passes are DOUBLE-BUFFERED: mcrt_group_trace_frames returns once everything is enqueued,
 * the ranks' next pass does not wait for the root's stream, so post-processing pass k on the root overlaps the trace of pass k+1
 * (alternate two rf_dev buffers to use it).  Errors: the first failing rank's status, its message prefixed with "rank r:". */
typedef struct mcrt_group mcrt_group;
int mcrt_group_create(const int *devices, uint32_t n_devices, mcrt_group **out);
int mcrt_group_destroy(mcrt_group *grp);
int mcrt_group_size(const mcrt_group *grp);                        /* ranks (0 for NULL) */
mcrt_ctx *mcrt_group_root(mcrt_group *grp);                        /* context on devices[0] that owns the gathered frames: post-processing, mcrt_alloc, exports */
mcrt_ctx *mcrt_group_member(mcrt_group *grp, uint32_t rank);       /* the rank's tracing context (statistics, timing, mcrt_cast_rays on one shard) */
/* the contiguous scan-line shard of `rank` when n_elements are cut over n_ranks (no group needed) */
int mcrt_group_shard(uint32_t rank, uint32_t n_ranks, uint32_t n_elements, uint32_t *e_begin, uint32_t *e_end);
/* the replicated set-up calls: the single-context call of the same name on every rank (concurrently), params also on the root (the ranks
 * first: a set of parameters a context refuses leaves the whole group on the old ones).  Scene data is HOST memory in every group call: a
 * device pointer belongs to one GPU and is refused.  With the host SAH builder (the default) the tree is built ONCE, on the calling
 * thread, and every rank uploads a copy (round 4 built it once per rank); the device LBVH builder runs per rank on its own GPU. */
int mcrt_group_set_params(mcrt_group *grp, const mcrt_params *p);
int mcrt_group_set_bvh_builder(mcrt_group *grp, int builder);
int mcrt_group_upload_scene(mcrt_group *grp, const float *tri_xyz_host, const uint32_t *tri_mesh, uint32_t n_tri, const mcrt_mesh *meshes, uint32_t n_mesh,
                            const float *materials, uint32_t n_mat, uint32_t start_mat, const float spacing[3]);
int mcrt_group_update_triangles(mcrt_group *grp, const float *tri_xyz_host, uint32_t n_tri);
/* seconds the last mcrt_group_upload_scene / _update_triangles spent in the host builder (once) and in the ranks' concurrent uploads */
int mcrt_group_last_scene_seconds(mcrt_group *grp, double *build_s, double *upload_s);
int mcrt_group_refit_triangles(mcrt_group *grp, const float *tri_xyz_host, uint32_t n_tri);
int mcrt_group_upload_texture(mcrt_group *grp, const float *voxels, uint32_t n);                  /* NULL: the reference's texture, generated once */
int mcrt_group_set_transducer(mcrt_group *grp, const float *pos, const float *dir, uint32_t n_elements);   /* all E elements */
/* mcrt_trace_frames over the whole group: rf_dev is a device buffer [n_frames][E][R] ON devices[0]; asynchronous -- the frames are
 * complete on the ROOT context's stream (anything enqueued on mcrt_group_root() afterwards sees them; mcrt_group_synchronize waits). */
int mcrt_group_trace_frames(mcrt_group *grp, uint32_t frame_id, uint32_t n_frames, float *rf_dev);
/* ... with a probe pose per frame (mcrt_trace_frames_poses); pos / dir: HOST tables [n_frames][E][3], copied before the call returns */
int mcrt_group_trace_frames_poses(mcrt_group *grp, uint32_t frame_id, uint32_t n_frames, const float *pos, const float *dir, float *rf_dev);
/* waits for every rank and the root; reports a rank's device error word as mcrt_synchronize does */
int mcrt_group_synchronize(mcrt_group *grp);
/* per rank, the device time of its last mcrt_group_trace_frames* pass: trace (k_init .. k_finalize) and its block's peer copy, in ms
 * (HIP events on the rank's streams; synchronises).  trace_ms / copy_ms: [mcrt_group_size()] each, either may be NULL */
int mcrt_group_last_pass_ms(mcrt_group *grp, float *trace_ms, float *copy_ms);

/* device memory helpers for callers without their own allocator */
int mcrt_alloc(mcrt_ctx *ctx, size_t bytes, void **dev);
int mcrt_free(mcrt_ctx *ctx, void *dev);
int mcrt_memcpy_d2h(mcrt_ctx *ctx, void *host, const void *dev, size_t bytes);
int mcrt_memcpy_h2d(mcrt_ctx *ctx, void *dev, const void *host, size_t bytes);

/* instrumentation: counted BVH nodes / triangles / RF steps of the next trace calls (slower build
 * of the kernel); enable=0 returns to the timed kernel */
int mcrt_enable_stats(mcrt_ctx *ctx, int enable);
int mcrt_get_stats(mcrt_ctx *ctx, mcrt_stats *out, int reset);
/* average device time of the trace kernel over the launches since the last reset (HIP events on the
 * context's stream), in milliseconds; n = launches measured */
int mcrt_enable_timing(mcrt_ctx *ctx, int enable);      /* 1: the walk's launches; 2: also k_shade's and k_march's (each on the stream it runs on) */
int mcrt_get_kernel_time(mcrt_ctx *ctx, double *avg_ms, uint32_t *n, int reset);
/* the same per kernel: [0] the walk, [1] k_shade, [2] k_march (the last two only under mcrt_enable_timing(ctx, 2)) */
int mcrt_get_kernel_times(mcrt_ctx *ctx, double avg_ms[3], uint32_t n[3], int reset);

/* ---- host-side pieces of the path (no GPU needed) ---- */
int mcrt_build_bvh(const float *tri_xyz, const uint32_t *tri_mesh, uint32_t n_tri, mcrt_bvh *out);
void mcrt_free_bvh(mcrt_bvh *bvh);
int mcrt_get_bvh(mcrt_ctx *ctx, mcrt_bvh *out /* borrowed pointers, valid until next upload */);
int mcrt_build_bvh4(const mcrt_bvh *bvh2, mcrt_bvh4 *out);
void mcrt_free_bvh4(mcrt_bvh4 *bvh4);
/* The BVH4 as the closest-hit walk reads it (borrowed; valid until the next upload / update / refit).  The default
 * lane-per-ray walk stores node boxes as half floats rounded outwards (64-byte nodes: the walk is bound by the number of
 * 16-byte pieces it fetches); the boxes returned here are those decoded values, so a CPU walk of this tree visits exactly the
 * nodes the GPU walk visits.  Hits never depend on the node boxes (closest hit). */
int mcrt_get_bvh4(mcrt_ctx *ctx, mcrt_bvh4 *out /* borrowed */);
/* the exact row look-up table used instead of the per-echo double division (see DESIGN.md "RF rows"):
 * thr[r] = smallest double t with fl(t / row_dt) >= r, r = 0..n_rows */
int mcrt_row_thresholds(double row_dt_us, uint32_t n_rows, double *thr);
/* volume<n,res>::volume() volume.h:19-35 */
int mcrt_generate_texture(float *voxels, uint32_t n);
/* psf<>::psf psf.h:34-58 */
int mcrt_psf_kernels(float freq, float var_x, float var_y, uint32_t res_um, float *axial, uint32_t n_ax, float *lateral, uint32_t n_lat);
/* transducer<N>::transducer transducer.h:24-62 */
int mcrt_transducer_elements(uint32_t n_elements, double radius_cm, double separation_mm,
                             const float position[3], const float angles_deg[3], float *pos, float *dir);

/* contract-math probe used by the parity tests: evaluates op over n inputs ON THE GPU.
 * op: 0 log_d, 1 exp_d, 2 sin_d, 3 cos_d, 4 sqrt_d, 5 div_d(x,y), 6 logf, 7 expf, 8 powf(x,y),
 *     9 sqrtf, 10 divf(x,y), 11 pow_d(x,y), 12/13 low 31 bits / remaining bits of the fixed-point echo rint(x*2^40).  x,y,out are host double arrays (float ops use the
 * value converted to float). */
int mcrt_debug_math(mcrt_ctx *ctx, int op, const double *x, const double *y, double *out, uint32_t n);
int mcrt_debug_philox(mcrt_ctx *ctx, const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
/* diagnostic builds of the library (-DMCRT_STAMP, or -DMCRT_STAMP_LITE for the timeline alone) only: out[0..15] per-phase cycle
 * sums of k_trace, out[16+4b..] the launch timeline of bounce b (100 MHz clock: ~earliest wave start, ~earliest empty queue,
 * latest wave end, summed wave lifetimes; the first two stored complemented), out[60..119] per-bounce wavefront counts, start
 * times, longest lifetime and node-step iterations, out[120..129] cycle sums of k_march's sections (tools/stamps.py decodes them); all zero otherwise */
int mcrt_debug_stamps(mcrt_ctx *ctx, uint64_t out[200], int reset);
/* the same diagnostic builds: per bounce b < 10, out[256 b + ...]: [0..63] wavefronts of the walk by the time they END, [64..127] by the time they
 * find the ray queue dry (20 us bins on the wavefront's own clock), [128..191] by the node-step iterations (bins of 8) since their last successful claim, [192..255] by the time since their last successful claim; all zero otherwise */
int mcrt_debug_tail_histograms(mcrt_ctx *ctx, uint64_t out[2560], int reset);
/* which of the RF accumulation's fast paths the context's LAST traced frame ran with (they are switched on by checks made on the
 * device, and a check that fails silently costs a third of the frame): out[0] the reciprocal-multiply voxel quotient (verified
 * exhaustively against IEEE division for params.tex_res), out[1] the branch-free voxel cell, out[2] the entries of the padded
 * { threshold, bin } image of k_march's fast variant (0: generic variant), out[3] reserved (0) */
int mcrt_debug_fast_paths(mcrt_ctx *ctx, uint32_t out[4]);
/* TEST HOOK, refused (MCRT_ERR_INVALID) unless the context was created with MCRT_TEST_HOOKS set in the environment: ORs `bits` into the
 * context's device error word on its stream, as an abandoned launch would (bit 0: traversal stack ran out, bit 1: kernel watchdog
 * expired) -- what mcrt_synchronize reports and what turns finalised RF images into NaN until it is asked */
int mcrt_debug_set_error(mcrt_ctx *ctx, uint32_t bits);

#ifdef __cplusplus
}
#endif
#endif
