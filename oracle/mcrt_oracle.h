/*
 * mcrt_oracle.h -- CPU ORACLE for the Monte-Carlo ultrasound ray-tracing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The shipped path is the HIP library
 * (include/mcrt.h); it never links, loads or calls anything in this directory.
 *
 * It is a plain-C restatement of the reference's algorithm (thepochynsons/MCRay-Tracing):
 *   src/scene.cpp:50-183 (cast_rays), :281-298 (distance_in_mm, enlarge), :300-334 (mesh placement)
 *   src/ray.cpp (all), src/ray.h, src/mesh.h
 *   src/main.cpp:23-37 (constants), :106-144 (RF accumulation)
 *   src/rfimage.h:33-51 (add_echo, micros_traveled), :93-123 (convolve), :161-164 (clear)
 *   src/psf.h:34-58,80-92, src/volume.h:19-61, src/transducer.h:24-62
 * plus Bullet's closest-hit triangle arithmetic (btTriangleRaycastCallback::processTriangle),
 * which is NOT under /root/reference (Bullet is un-vendored and un-pinned there, README.md:16).
 *
 * PARITY PIN STATUS
 *   pinned   : psf taps, tissue texture, units-derived constants/time axis, degree->radian
 *              conversion -- checked against values produced by compiling the reference's own
 *              headers (oracle/ref_probe.cpp -> tests/golden/ref_probe.json).
 *   UNPINNED : the triangle-intersection arithmetic (Bullet absent), the RNG stream (the
 *              reference seeds a fresh mt19937 from std::random_device for every draw, so it
 *              has no reproducible stream) and hence hit indices / RF images end-to-end.
 *              "parity unpinned" for those; they are pinned oracle <-> GPU under the
 *              counter-based RNG + deterministic-math contract documented in DESIGN.md.
 */
#ifndef MCRT_ORACLE_H
#define MCRT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- deterministic math contract (DESIGN.md "Deterministic math") ---- */
double orc_log_d(double x);
double orc_exp_d(double x);
void   orc_sincos_d(double a, double *s, double *c);
float  orc_logf(float x);
float  orc_expf(float x);
float  orc_powf(float x, float y);
double orc_pow_d(double x, double y);     /* x in [0,inf), used by power_cosine_variate */
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_u53(uint32_t hi, uint32_t lo);
/* math_mode: 0 = deterministic contract (default), 1 = libm (what the reference calls) */
void   orc_set_math_mode(int mode);

/* ---- static pieces ---- */
/* volume.h:19-35 restated (libstdc++ minstd_rand0 + polar normal). out = [N*N*N][2] floats */
void orc_texture_generate(float *out, uint32_t n);
/* psf.h:34-58 */
void orc_psf(float freq, float var_x, float var_y, uint32_t res_um,
             float *axial, uint32_t n_ax, float *lateral, uint32_t n_lat);
/* transducer.h:24-62.  angles_deg = {x,y,z}.  pos/dir = [E][3] */
void orc_transducer(uint32_t n_elem, double radius_cm, double sep_mm,
                    const float position[3], const float angles_deg[3],
                    float *pos, float *dir);
/* scene.cpp:313-324: v_world = v*scaling + (deltas*scaling*scaling + origin), in place on [n][3] */
void orc_place_vertices(float *v, uint32_t n, float scaling, const float deltas[3], const float origin[3]);

/* units-derived constants of main.cpp:23-37 / rfimage.h:178-180 for a given frequency etc. */
typedef struct {
    double axial_res_mm;      /* (double)(1.45f / f)                               */
    float  axial_res_f;       /* float form used for delta_step and attenuation    */
    double time_step_us;      /* micros_traveled(axial_resolution)                 */
    double row_dt_us;         /* axial_res_um(integer) / speed_of_sound            */
    double max_travel_us;     /* depth / speed of sound                            */
    uint32_t axial_res_um;    /* (unsigned)(axial_res_f*1000.0f)                   */
    uint32_t max_rows;        /* (sos*max_travel)/axial_res_um, integer division   */
} orc_consts;
void orc_constants(float frequency_mhz, uint32_t sos, double depth_cm, orc_consts *out);

/* ---- scene ---- */
typedef struct { uint32_t mat_inside, mat_outside, vascular, _pad; } orc_mesh;

/* 64-byte BVH2 node: the layout the PRODUCT's builder emits (include/mcrt.h mcrt_bvh_node).
 * The oracle only walks it; it never builds one. */
typedef struct {
    float lo0[3]; int32_t c0;
    float hi0[3]; int32_t c1;
    float lo1[3]; uint32_t pad0;
    float hi1[3]; uint32_t pad1;
} orc_bvh_node;

typedef struct {
    uint32_t n_tri;
    const float    *tri;       /* [T][9] world-space vertices, original (OBJ face) order */
    const uint32_t *tri_mesh;  /* [T] mesh index */
    uint32_t n_mesh;
    const orc_mesh *mesh;
    uint32_t n_mat;
    const float *mat;          /* [M][8] impedance, attenuation, mu0, mu1, sigma, specularity, shininess, thickness */
    uint32_t start_mat;
    float spacing[3];
    /* optional acceleration structure built by the product (NULL => brute force) */
    uint32_t n_nodes;
    const orc_bvh_node *nodes;
    const float *bvh_tri;      /* [T][12]: v0.xyz,id | v1.xyz,mesh | v2.xyz,0   (leaf order) */
    float pad_abs;             /* orc_pad_abs(tri, n_tri): absolute part of the per-triangle bounds padding */
    /* optional BVH4 built by the product (include/mcrt.h mcrt_bvh4_node: 4 x {lo[3],hi[3],ref,pad}); walked with use_bvh = 2 */
    uint32_t n_nodes4;
    const void *nodes4;
} orc_scene;
float orc_pad_abs(const float *tri, uint32_t n_tri);

typedef struct {
    uint32_t n_elements, n_samples, max_depth, n_rows;
    float frequency;             /* transducer.frequency, MHz (main.cpp:24)      */
    float intensity_epsilon;     /* ray.h:24                                     */
    float initial_intensity;     /* scene.h:49                                   */
    float ray_start_offset;      /* scene.cpp:115: 0.1f                          */
    uint32_t sos;                /* 1500                                         */
    double depth_cm;             /* 15                                           */
    uint32_t seed;
    uint32_t sanitize_tir;       /* 0 = reference behaviour (NaN echo on TIR)    */
    uint32_t tex_n;              /* 256 */
    float tex_res;               /* 0.145f */
} orc_params;
void orc_default_params(orc_params *p);

typedef struct {
    float from[3], to[3], dir[3];
    float reflected_intensity, initial_intensity, attenuation;
    double distance_traveled;
    int32_t media;      /* material index in effect along the segment (quirk 3: by value) */
    int32_t tri;        /* triangle hit at the end of the segment, -1 = miss */
} orc_segment;          /* 64 bytes */

typedef struct {
    uint64_t queries, nodes_visited, tris_tested, segments, rf_steps, hits;
} orc_stats;

/* closest hit on the segment [from,to]; returns tri id or -1.  use_bvh: 0 brute force, 1 BVH2 walk, 2 BVH4 walk
 * (the GPU's visiting order, so node/triangle counts can be compared). */
int32_t orc_closest_hit(const orc_scene *sc, const float from[3], const float to[3], int use_bvh,
                        float *frac, float normal[3], float point[3], orc_stats *st);

/*
 * One frame for elements [e_begin, e_end).
 *  hits     : [E_range][S][B] int32 triangle ids (-1 miss, -2 no query)        (may be NULL)
 *  segs     : [E_range][S][B] segments, seg_count [E_range][S]                 (may be NULL)
 *  rf_ref   : [R][E_range] float, reference summation order (main.cpp:106-144) (may be NULL)
 *  rf_fix   : [E_range][R] int64 fixed-point accumulators (contract mode)      (may be NULL)
 *  rf_flags : [E_range][R] uint8 non-finite flags for rf_fix                   (may be NULL)
 */
/* test aid: while set, a call that produces rf_ref also adds every echo, in the same order, into this [R][E_range] double image */
void orc_set_ref64(double *buf);

void orc_trace_frame(const orc_scene *sc, const orc_params *p,
                     const float *el_pos, const float *el_dir, const float *texture,
                     uint32_t frame_id, uint32_t e_begin, uint32_t e_end, int use_bvh, int n_threads,
                     int32_t *hits, orc_segment *segs, uint32_t *seg_count,
                     float *rf_ref, int64_t *rf_fix, uint8_t *rf_flags, orc_stats *st);


/* ---- test entry points into the physics (ray.cpp) -- tests/test_oracle_physics.py, tests/ref_reading.py.  They run the same
 * static functions the trace path runs.  rng = { key0 (seed), key1 (frame), element, sample, bounce }: the draws are the contract's
 * (block 0 thickness, block 1 {power-cosine u, choice x}, block 2+attempt {disc a, disc r}), see DESIGN.md 3. */
typedef struct {
    float from[3], dir[3];
    int32_t media;              /* material index (ray.h:17 holds the material by value)                                   */
    int32_t outside;            /* media_outside: -1 = nullptr, -2 = aliases the ray's own media (quirk 2), >= 0 = index   */
    float intensity, frequency;
    double dist_mm;
} orc_ray_state;                /* 48 bytes */
typedef struct {
    float reflected_intensity;  /* hit_result.reflected_intensity (ray.cpp:82,96)                                           */
    uint32_t _pad0;
    orc_ray_state returned;     /* hit_result.returned (ray.cpp:91-94)                                                      */
    float random_angle, random_normal[3], incidence, refr_ratio, refraction_angle, refr_dir[3], refl_dir[3];
    float intensity_refl, intensity_refr, refraction_factor, reflection_factor;
    double u_pc, u_x;           /* the two uniforms of block 1                                                              */
    int32_t tir, chose_reflection, mat_after, after_vasc;
    uint32_t ruv_attempts, _pad1;
} orc_hit_debug;
float    orc_debug_power_cosine(int v, double number);
uint32_t orc_debug_random_unit_vector(const float v[3], float cos_theta, const uint32_t rng[5], float w[3]);
void     orc_debug_hit_boundary(const orc_scene *sc, const orc_params *prm, const orc_ray_state *r, const float hit_point[3],
                                const float normal[3], uint32_t mesh, const uint32_t rng[5], orc_hit_debug *out);
float    orc_debug_ray_segment(const orc_scene *sc, const orc_params *prm, const orc_ray_state *r, float from_off[3], float to[3]);
double   orc_debug_travel(const orc_scene *sc, orc_ray_state *r, const float to_point[3]);
float    orc_debug_thickness(float sigma, const uint32_t rng[5]);
uint64_t orc_debug_accumulate_segment(const orc_scene *sc, const orc_params *prm, const float *tex, const orc_segment *sg, float *rf /*[R]*/);
/* measured zeros: counters of the contract's additions (see mcrt_oracle.c "MEASURED ZEROS"); counting(sc, 1) zeroes and starts,
 * counting(NULL, 0) stops.  out = { pad-rule rejects inside the scene bounds, pad-rule rejects anywhere, echo-guard trips,
 * random_unit_vector retries, ... give-ups, total internal reflections, NaN echoes, 0 } */
void orc_debug_counting(const orc_scene *sc, int on);
void orc_debug_counters(uint64_t out[8]);

/* contract finalisation: float = flag ? NaN : (float)((double)acc * 2^-40); out [R][E] row-major */
void orc_finalize_rf(const int64_t *rf_fix, const uint8_t *rf_flags, uint32_t n_elem, uint32_t n_rows, float *out_rows_by_cols);

/* rfimage.h:93-123 on a row-major [R][E] image, in place (tmp = scratch of same size) */
void orc_convolve(float *img, float *tmp, uint32_t rows, uint32_t cols,
                  const float *axial, uint32_t n_ax, const float *lateral, uint32_t n_lat);
/* rfimage.h:54-91 on a row-major [R][E] image, in place */
void orc_envelope(float *img, uint32_t rows, uint32_t cols);
/* rfimage.h:183-215 create_mapping: map_row = the reference's map_x (row coordinate), map_col = its map_y (column coordinate);
 * [out_rows][out_cols] row-major.  max_travel_us / sos: the rf_image template parameters (unsigned int). */
void orc_scan_maps(uint32_t rows, uint32_t cols, double radius_mm, double total_angle_rad, uint32_t max_travel_us, uint32_t sos,
                   uint32_t out_rows, uint32_t out_cols, float *map_row, float *map_col);
/* rfimage.h:183-215 mapping + exact bilinear remap (BORDER_CONSTANT 0). out = [out_rows][out_cols] */
void orc_scan_convert(const float *img, uint32_t rows, uint32_t cols, double radius_mm, double total_angle_rad,
                      double max_travel_us, double sos, float *out, uint32_t out_rows, uint32_t out_cols);

/* ANALYSIS (tools/bvh_width.py), no parity path uses it: the scene's BVH2 collapsed to W-wide nodes (2..16; quant 8 = boxes snapped
 * outwards to a 256-step grid in the node's own frame), and the closest-hit queries of traced segments counted on it:
 * out[q] = { inner nodes, leaves, triangles, deepest stack }, tri[q] the triangle found */
struct orc_wide;
struct orc_wide *orc_wide_build(const orc_scene *sc, uint32_t W, int quant);
void orc_wide_free(struct orc_wide *w);
uint32_t orc_wide_nodes(const struct orc_wide *w);
uint32_t orc_wide_order(const struct orc_wide *w, int mode /*0 breadth first, 1 largest box area first*/, uint32_t *order /*[nodes]*/);
void orc_wide_visits(struct orc_wide *w, uint32_t *visits /*[nodes] or NULL*/);
void orc_wide_count(const struct orc_wide *w, const orc_scene *sc, const orc_params *prm, const orc_segment *segs, uint64_t n,
                    uint32_t *out, int32_t *tri, int n_threads);

/* ANALYSIS (tools/seed_count.py): the queries of traced segments walked over the BVH4 with no seed (mode 0), the query's own answer as
 * the initial closest hit (mode 1) or triangle seed_tri[q] tested against the query's ray first (mode 2); out[q] = { nodes, triangles } */
void orc_seed_count(const orc_scene *sc, const orc_params *prm, const orc_segment *segs, uint64_t n, int mode, const int32_t *seed_tri,
                    uint32_t *out, int32_t *tri, int n_threads);

/* ANALYSIS (tools/packet_count.py): W consecutive queries walking the BVH4 as ONE packet (shared stack; a node is visited when any ray passes it);
 * out[p] = { nodes, leaves, triangles, sum and max of the rays' solo node visits, live rays }; tri[q] the triangle each ray finds */
void orc_packet_count(const orc_scene *sc, const orc_params *prm, const orc_segment *segs, uint64_t n, uint32_t W, int order,
                      uint32_t *out, int32_t *tri, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
