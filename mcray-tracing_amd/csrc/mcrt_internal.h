// mcrt_internal.h -- shared between the translation units of libmcrt_hip.so
#pragma once
#include <stdint.h>

#define MCRT_BVH_MAX_DEPTH 32      // deepest leaf the builder may emit == traversal stack entries per lane
#define MCRT_STACK 64              // BVH4 traversal stack entries per path (LDS); trees needing more are rejected at upload
#define MCRT_KSPLIT_DEFAULT 262144 // work items a small bounce of k_trace is cut into (pieces x rays); one 128 x 1024 frame at a time is cut in two
#ifndef MCRT_KSPLIT_MAX
#define MCRT_KSPLIT_MAX 1048576
#endif
#define MCRT_GROUPS_DEFAULT 1        // independent scan-line groups a frame is traced as (their kernels overlap)
#define MCRT_SIDE_STREAMS 4         // streams k_march launches rotate over
#define MCRT_SIDE_STREAMS_DEFAULT 0       // 0: by the size of the pass -- one side stream, two from MCRT_SIDE_STREAMS_TWO_FROM paths
#define MCRT_SIDE_STREAMS_TWO_FROM 5242880u   // (40 frames of 128 x 1024 paths.  One box, ms per B-mode frame with one / two side streams: 20 frames 0.338 / 0.350, 32: 0.311 / 0.314, 48: 0.306 / 0.298,
                                              //  64: 0.300 / 0.291, 96: 0.298 / 0.282, 128: 0.295 / 0.283 (three: 0.287; at 20 frames 0.446).  In a large pass the accumulations, stretched threefold beside the
                                              //  walks, are the longer chain: two of them side by side shorten it; in a small pass the second one takes the walk's tail away.)
#define MCRT_LBVH_LEAF 1             // device builder: triangles per leaf (1..4); measured best at 1, like the SAH builder's own leaves
#define MCRT_XCDS 8                   // XCDs of the MI355X = sub-queues of a bounce's ray queue (see k_trace)
#define MCRT_CURSOR_STRIDE 64         // uint32 between two queue cursors: 256 B, so they sit in different L2 lines / channels
#define MCRT_XCD_MIN_ITEMS 262144     // bounces with fewer work items use a single queue
#define MCRT_PACKET_MASK_DEFAULT 2u   // bounces (bit b) walked a wavefront per ray packet (k_trace_packet): bounce 1 -- every pass size from two frames up and every BASELINE
                                      // configuration gains 0.3-6 % (profiles/round5/exp_packet.txt); bounce 2 is a wash, later bounces lose (packet_count_*.json)
#define MCRT_PACKET_FROM 262144u      // ... in passes of at least this many paths (one 128 x 1024 frame at a time keeps the lane walk: its launches are cut into pieces, 1.624 vs 1.634 ms)
#define MCRT_PATH_MAX_DEFAULT 655360u  // passes of at most this many paths take the latency form (k_path: one launch for all bounces): five 128 x 1024 frames (ms per frame at 1 / 2 / 3 / 4 / 5 / 6 frames: 0.87 / 0.77 / 0.72 / 0.69 / 0.67 / 0.65 against the staged 1.62 / 1.12 / 0.88 / 0.76 / 0.68 / 0.60)
#define MCRT_PATH_GROUPS_DEFAULT 2u     // ... traced as this many scan-line groups on their own streams (a group's k_march runs beside the other groups' slowest wavefronts)
#define MCRT_MAX_ROWS 2048
#define MCRT_MAX_BOUNCES 16
#define MCRT_STATS_WORDS (256 + 2560)  // the context's counter block: 8 statistics, 248 stamps (mcrt_debug_stamps), 10 x 256 tail histograms (mcrt_debug_tail_histograms)

#define MCRT_TRI_PIECES 3             // 16-byte pieces of the walk's triangle record: v0|id, v1|mesh, v2|edge tolerance (48 B; the plane is rebuilt from the vertices)

namespace mcrt {
int set_error(int code, const char *fmt, ...);
// A TUNING knob from the environment: nullptr unless the process runs with MCRT_TUNING=1 (mcrt_host.cpp).  The library is meant to be
// linked into someone else's program: left alone it reads ONE environment variable, once; the knobs themselves are
// for tools/ and tests/, which set MCRT_TUNING=1 beside the knob they turn.
const char *tuning_env(const char *name);
}
#ifdef MCRT_H
// mcrt_api.cpp <-> mcrt_group.cpp: a scene (or a triangle update) installed with a tree the HOST builder has already made of exactly
// these triangles -- a group builds once on the calling thread and every rank copies and uploads (the device builder ignores it)
namespace mcrt {
struct HostTree { const mcrt_bvh *bvh; const mcrt_bvh4 *bvh4; };
int ctx_bvh_builder(const mcrt_ctx *c);
int upload_scene_with_tree(mcrt_ctx *c, const float *tri, const uint32_t *tri_mesh, uint32_t n_tri, const mcrt_mesh *meshes, uint32_t n_mesh,
                           const float *mats, uint32_t n_mat, uint32_t start_mat, const float spacing[3], const HostTree *pre);
int update_triangles_with_tree(mcrt_ctx *c, const float *tri, uint32_t n_tri, const HostTree *pre);
}
#endif
