// mcrt_kernels.h -- kernel argument blocks and launchers (mcrt_kernels.hip <-> mcrt_api.cpp)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mcrt.h"
#include "mcrt_internal.h"

namespace mcrt {

struct FrameArgs {
    // scene (HBM-resident, read-only)
    const uint4 *nodes_walk;   // [n_nodes][4]  the walk's 64-byte nodes: child-transposed half-float boxes (rounded outwards) + refs
    int *stack_ovf;            // [max_stack - MCRT_LANE_STACK][trace_blocks * 256] traversal-stack entries beyond the LDS part (this work set's own)
    const float4 *tris;        // [T][3]         48-B triangle records, leaf order: v0|id, v1|mesh, v2|edge tolerance
    const float4 *tris_id;     // [T][3]         the same records in TRIANGLE-ID order (k_shade looks the winning triangle up by its id: one dependent load less than through tri_slot)
    const uint4 *meshes;       // [n_mesh]      mat_inside, mat_outside, vascular, -
    const float4 *mats;        // [n_mat][2]    imp, att, mu0, mu1 | sigma, spec, shine, thick
    const float2 *tex;         // [n^3]         texture_noise, scattering_probability
    const float *el_pos;       // [E][3], or [F][E][3] when the frames of a pass have their own probe poses (pose_stride = E)
    const float *el_dir;       // same shape
    const double *row_thr;     // [R+1] row thresholds (see row_of)
    // per-frame work buffers; np = ne * S paths
    float4 *st0, *st1, *st2;   // [2][np] path state in queue order, two halves by bounce parity: from, ray length factor | dir, media | distance_traveled(f64), outside, intensity
                               //         (the walk reads st0 + st1 and rebuilds the ray from them: ray_of)
    uint32_t *queue;           // [2][np] live path ids of bounce b in buffer b & 1
    unsigned long long *key0, *key1;   // [np] closest hit per ray: fraction bits << 32 | triangle id (atomicMin), ping-pong by bounce parity
    const uint32_t *tri_slot;  // [T] triangle id -> position in the leaf-order triangle array
    uint32_t *counts;          // [MAX_BOUNCES+1] live rays per bounce
    uint32_t *cursors;         // [MAX_BOUNCES][MCRT_XCDS][MCRT_CURSOR_STRIDE] queue cursors of the persistent walk, one per XCD sub-queue
    mcrt_segment *segs;        // [np][B]   written only when want_segs (mcrt_cast_rays / mcrt_trace_frame_debug with a segment buffer)
    int32_t *hits;             // [np][B]   triangle hit at the end of each segment (-1 none); null unless the caller asked for hit indices
    float4 *mrec;              // [B][np][3] what k_march needs of a segment: from,refl | delta,intensity | t_start(f64),steps,media
    const float4 *mtab;        // [M] per material, for k_march: mu0, mu1, sigma, per-step attenuation factor
    uint32_t *seg_count;       // [np]
    long long *acc;            // [ne][R] fixed-point RF accumulators (2^-40 units)
    uint32_t *flags;           // [ne][(R+31)/32] non-finite flags
    unsigned long long *stats; // [6]
    uint32_t *error_flag;      // device word, bit 0: traversal stack overflow
    unsigned long long *stamps; // diagnostic builds only (tools/variants/round6_stamps.patch); unused by the product's kernels
    // sizes / parameters
    uint32_t n_nodes, S, B, R, e_begin, ne, ne_frame, pose_stride, acc_stride, acc_off, trace_blocks, trace_blocks_wide, wide_from, packet_mask, march_blocks, ksplit_limit, frame, seed, start_mat, tex_n, tex_mask, sanitize, tex_finite, fast_div, want_segs, tex_shift, march_rows, n_mat, n_mesh;   // march_rows: entries of k_march's padded LDS image when its fast variant applies, else 0
    float scene_lo[3], scene_hi[3];   // bounds of the whole BVH
    float freq, eps, I0, offs, sx, sy, sz, tex_res, axial_res_f, pad_abs, tex_rcp, lean_bound;
    double axial_res_mm, time_step, row_dt, max_travel, sos_d, inv_row_dt;
};

constexpr uint32_t MCRT_ALL_BOUNCES = 0xffffffffu;   // launch_march: accumulate the segments of every bounce in one launch

struct ConvTaps { float ax[16]; float lat[32]; uint32_t n_ax, n_lat; };

hipError_t launch_init(const FrameArgs &a, hipStream_t st);
hipError_t launch_trace(const FrameArgs &a, uint32_t b, bool stats, hipStream_t st);
hipError_t launch_nodes_walk(const float4 *nodes, uint32_t n_nodes, uint4 *out, hipStream_t st);
hipError_t launch_nodes_walk_decode(const uint4 *walk, uint32_t n_nodes, float4 *out, hipStream_t st);
uint32_t lane_stack_entries();
uint32_t lane_wide_from();
hipError_t launch_shade(const FrameArgs &a, uint32_t b, bool stats, hipStream_t st);
uint32_t path_blocks(size_t np);                                  // workgroups of a k_path launch over np paths (sizes the overflow stacks)
hipError_t launch_path(const FrameArgs &a, hipStream_t st);      // k_path: every bounce of every path in one launch (the latency form)
hipError_t launch_march(const FrameArgs &a, uint32_t b, bool stats, hipStream_t st);
hipError_t launch_finalize(long long *acc, uint32_t *flags, float *rf, uint32_t ne, uint32_t R, const uint32_t *error_flag, hipStream_t st);
hipError_t launch_convolve(float *img, float *tmp, uint32_t n_img, uint32_t E, uint32_t R, const ConvTaps &taps, hipStream_t st);
hipError_t launch_envelope(float *img, uint32_t E, uint32_t R, hipStream_t st);
hipError_t launch_remap(const float *img, uint32_t n_img, uint32_t E, uint32_t R, const float *map_col, const float *map_row, float *out, uint32_t n, hipStream_t st);
hipError_t launch_blocks_to_frames(const float *blocks, float *frames, uint32_t F, uint32_t E, uint32_t R, uint32_t G, const uint32_t *off /*[G+1]*/, hipStream_t st);   // at most 64 ranks
hipError_t launch_transpose(const float *in, float *out, uint32_t E, uint32_t R, hipStream_t st);
hipError_t launch_math_probe(int op, const double *x, const double *y, double *out, uint32_t n, hipStream_t st);
hipError_t launch_verify_div(float res, float rcp, unsigned long long *bad, hipStream_t st);
hipError_t launch_tris_by_id(const float4 *tris, uint32_t n_tri, float4 *out, hipStream_t st);   // leaf-order records -> id order (record i goes to slot id(i))
hipError_t launch_expand_tris(const float4 *in48, uint32_t n_tri, float4 *out, hipStream_t st);   // (v0|id, v1|mesh, v2|-) -> the walk's records (v2.w = the edge tolerance)
hipError_t launch_material_table(const float4 *mats, uint32_t n_mat, float axial_res_f, float freq, float4 *mtab, hipStream_t st);
hipError_t launch_philox_probe(const uint32_t c[4], const uint32_t k[2], uint32_t *out, hipStream_t st);

}  // namespace mcrt
