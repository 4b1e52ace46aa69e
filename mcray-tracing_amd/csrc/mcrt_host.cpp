// mcrt_host.cpp -- host-side pieces of the hot path (no GPU needed):
//   * BVH2 builder (binned SAH) replacing btBvhTriangleMeshShape construction + the DBVT
//     broadphase (scene.cpp:255,309): ONE flattened tree over the triangles of all meshes.
//   * the reference's static tables: tissue texture (volume.h:19-35), PSF taps (psf.h:34-58),
//     transducer element geometry (transducer.h:24-62), scan-conversion maps (rfimage.h:183-215).
//   * the library's error string (mcrt_last_error).
// Plain C++ with no HIP in it: tests/test_host_sanitize.py compiles this file by itself under ASan + UBSan.
#include "../../include/mcrt.h"
#include "mcrt_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace mcrt {
static thread_local std::string g_err;
int set_error(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
}  // namespace mcrt
extern "C" const char *mcrt_last_error(void) { return mcrt::g_err.c_str(); }

// tuning knobs (mcrt_internal.h): only a process started with MCRT_TUNING=1 has any
const char *mcrt::tuning_env(const char *name)
{
    const char *e = getenv("MCRT_TUNING");       // (looked at when a knob is asked for -- mcrt_create, a BVH build --, never on the frame path)
    return (e && e[0] == '1') ? getenv(name) : nullptr;
}

namespace {

struct Prim {
    float lo[3], hi[3];   // padded bounds
    float c[3];           // centroid of the unpadded bounds
    uint32_t id;
};

struct Box {
    float lo[3], hi[3];
    void reset() { for (int i = 0; i < 3; i++) { lo[i] = INFINITY; hi[i] = -INFINITY; } }
    void grow(const float *l, const float *h) { for (int i = 0; i < 3; i++) { lo[i] = std::min(lo[i], l[i]); hi[i] = std::max(hi[i], h[i]); } }
    void grow_pt(const float *p) { for (int i = 0; i < 3; i++) { lo[i] = std::min(lo[i], p[i]); hi[i] = std::max(hi[i], p[i]); } }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0) || !(dy >= 0) || !(dz >= 0)) return 0.f;
        return dx * dy + dy * dz + dz * dx;
    }
};

constexpr int kLeafMaxDefault = 4;     // (leaf size and triangle cost can be set through MCRT_SAH_LEAF_MAX / MCRT_SAH_COST_TRI for experiments: members of the Builder)
constexpr int kBins = 16;
constexpr int kMaxDepth = MCRT_BVH_MAX_DEPTH;   // deepest leaf; the traversal stack holds this many entries
constexpr float kCostNode = 1.0f, kCostTriDefault = 1.2f;

struct Builder {
    int kLeafMax = kLeafMaxDefault; float kCostTri = kCostTriDefault;
    std::vector<Prim> prims;
    std::vector<mcrt_bvh_node> nodes;
    int deepest = 0;

    int levels_needed(uint32_t n) const { int k = 0; uint64_t cap = (uint64_t)kLeafMax; while (cap < n) { cap <<= 1; k++; } return k; }

    static int32_t leaf_ref(uint32_t first, uint32_t cnt) { return ~(int32_t)((first << 3) | (cnt - 1)); }

    // returns child reference, bounds of the subtree in `out`
    int32_t build(uint32_t lo, uint32_t hi, int depth, Box &out)
    {
        const uint32_t n = hi - lo;
        out.reset();
        Box cb; cb.reset();
        for (uint32_t i = lo; i < hi; i++) { out.grow(prims[i].lo, prims[i].hi); cb.grow_pt(prims[i].c); }
        deepest = std::max(deepest, depth);

        uint32_t mid = 0;
        bool have_split = false;
        const bool force_median = depth + levels_needed(n) >= kMaxDepth;
        float best_cost = INFINITY;
        if (n > 1 && !force_median) {
            int best_axis = -1, best_bin = -1;
            const float parent_area = out.half_area();
            for (int axis = 0; axis < 3; axis++) {
                const float cmin = cb.lo[axis], cext = cb.hi[axis] - cb.lo[axis];
                if (!(cext > 0.f)) continue;
                const float scale = (float)kBins / cext;
                Box bb[kBins]; uint32_t cnt[kBins];
                for (int b = 0; b < kBins; b++) { bb[b].reset(); cnt[b] = 0; }
                for (uint32_t i = lo; i < hi; i++) {
                    int b = (int)((prims[i].c[axis] - cmin) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    bb[b].grow(prims[i].lo, prims[i].hi); cnt[b]++;
                }
                float right_area[kBins]; uint32_t right_cnt[kBins];
                Box acc; acc.reset(); uint32_t c = 0;
                for (int b = kBins - 1; b > 0; b--) { acc.grow(bb[b].lo, bb[b].hi); c += cnt[b]; right_area[b] = acc.half_area(); right_cnt[b] = c; }
                acc.reset(); c = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    acc.grow(bb[b].lo, bb[b].hi); c += cnt[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = kCostNode + kCostTri * (acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1]) / std::max(parent_area, 1e-30f);
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                const float cmin = cb.lo[best_axis], scale = (float)kBins / (cb.hi[best_axis] - cb.lo[best_axis]);
                auto it = std::partition(prims.begin() + lo, prims.begin() + hi, [&](const Prim &p) {
                    int b = (int)((p.c[best_axis] - cmin) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    return b <= best_bin;
                });
                mid = (uint32_t)(it - prims.begin());
                have_split = mid > lo && mid < hi;
            }
        }
        if (n <= (uint32_t)kLeafMax && (n == 1 || !have_split || best_cost >= kCostTri * (float)n))
            return leaf_ref(lo, n);
        if (!have_split) {   // median split on the widest centroid axis (also the depth-limit fallback)
            int axis = 0; float ext = -1.f;
            for (int a = 0; a < 3; a++) { float e = cb.hi[a] - cb.lo[a]; if (e > ext) { ext = e; axis = a; } }
            mid = lo + n / 2;
            std::nth_element(prims.begin() + lo, prims.begin() + mid, prims.begin() + hi,
                             [axis](const Prim &a, const Prim &b) { return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.id < b.id); });
        }
        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        Box bl, br;
        int32_t cl = build(lo, mid, depth + 1, bl);
        int32_t cr = build(mid, hi, depth + 1, br);
        mcrt_bvh_node &N = nodes[me];
        for (int i = 0; i < 3; i++) { N.lo0[i] = bl.lo[i]; N.hi0[i] = bl.hi[i]; N.lo1[i] = br.lo[i]; N.hi1[i] = br.hi[i]; }
        N.c0 = cl; N.c1 = cr; N.pad0 = 0; N.pad1 = 0;
        return me;
    }
};

}  // namespace

extern "C" int mcrt_build_bvh(const float *tri, const uint32_t *tri_mesh, uint32_t n_tri, mcrt_bvh *out)
{
    if (!tri || !out || n_tri == 0) return mcrt::set_error(MCRT_ERR_INVALID, "mcrt_build_bvh: no triangles");
    if (n_tri >= (1u << 28)) return mcrt::set_error(MCRT_ERR_LIMIT, "mcrt_build_bvh: more than 2^28 triangles");
    Builder b;
    if (const char *e = mcrt::tuning_env("MCRT_SAH_LEAF_MAX")) { int v = atoi(e); if (v >= 1 && v <= 8) b.kLeafMax = v; }
    if (const char *e = mcrt::tuning_env("MCRT_SAH_COST_TRI")) { float v = (float)atof(e); if (v > 0.0f) b.kCostTri = v; }
    b.prims.resize(n_tri);
    // Padding (DESIGN.md "Closest hit"): Bullet's triangle test accepts points up to 1e-4 of the triangle's
    // height outside an edge, and the slab arithmetic rounds; each triangle's bounds are widened accordingly.
    // A triangle is eligible only while the ray overlaps ITS padded bounds (as in Bullet's per-triangle BVH
    // leaves); node boxes are exact unions of those, and the contract's plane distance fl(plane * inv + c) (one fma, finite
    // reciprocal: DESIGN.md 3) is a monotone function of the plane, so a box that contains another yields the wider interval
    // and node culling can never remove an eligible triangle.
    float scale = 0.f;
    for (size_t i = 0; i < (size_t)n_tri * 9; i++) { float a = std::fabs(tri[i]); if (a > scale && std::isfinite(a)) scale = a; }
    const float abs_pad = 4e-6f * std::max(scale, 1e-3f);
    for (uint32_t t = 0; t < n_tri; t++) {
        const float *v = tri + (size_t)t * 9;
        Prim &p = b.prims[t];
        // must equal k_expand_tris' padded bounds bit for bit: node boxes are unions of exactly these boxes
        float ext = 0.f;
        for (int a = 0; a < 3; a++) {
            float l = fminf(v[a], fminf(v[3 + a], v[6 + a]));
            float h = fmaxf(v[a], fmaxf(v[3 + a], v[6 + a]));
            p.lo[a] = l; p.hi[a] = h; p.c[a] = 0.5f * (l + h);
            ext = fmaxf(ext, h - l);
        }
        const float pad = 2e-4f * ext + abs_pad;
        for (int a = 0; a < 3; a++) { p.lo[a] = p.lo[a] - pad; p.hi[a] = p.hi[a] + pad; }
        p.id = t;
    }
    b.nodes.reserve(n_tri);
    Box root;
    int32_t r = b.build(0, n_tri, 0, root);
    if (r < 0) {   // everything fitted one leaf: wrap it in a root node (both children the same leaf)
        b.nodes.emplace_back();
        mcrt_bvh_node &N = b.nodes[0];
        for (int i = 0; i < 3; i++) { N.lo0[i] = root.lo[i]; N.hi0[i] = root.hi[i]; N.lo1[i] = root.lo[i]; N.hi1[i] = root.hi[i]; }
        N.c0 = r; N.c1 = r; N.pad0 = N.pad1 = 0;
    }
    out->n_nodes = (uint32_t)b.nodes.size();
    out->n_tri = n_tri;
    out->max_depth = (uint32_t)b.deepest;
    out->pad_abs = abs_pad;
    out->nodes = (mcrt_bvh_node *)malloc(sizeof(mcrt_bvh_node) * b.nodes.size());
    out->tri = (float *)malloc(sizeof(float) * 12 * (size_t)n_tri);
    if (!out->nodes || !out->tri) { free(out->nodes); free(out->tri); return mcrt::set_error(MCRT_ERR_NOMEM, "mcrt_build_bvh: out of memory"); }
    memcpy(out->nodes, b.nodes.data(), sizeof(mcrt_bvh_node) * b.nodes.size());
    for (uint32_t i = 0; i < n_tri; i++) {
        const uint32_t id = b.prims[i].id;
        const float *v = tri + (size_t)id * 9;
        float *o = out->tri + (size_t)i * 12;
        uint32_t mesh = tri_mesh ? tri_mesh[id] : 0u, zero = 0u;
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; memcpy(&o[3], &id, 4);
        o[4] = v[3]; o[5] = v[4]; o[6] = v[5]; memcpy(&o[7], &mesh, 4);
        o[8] = v[6]; o[9] = v[7]; o[10] = v[8]; memcpy(&o[11], &zero, 4);
    }
    return MCRT_OK;
}

extern "C" void mcrt_free_bvh(mcrt_bvh *bvh)
{
    if (!bvh) return;
    free(bvh->nodes); free(bvh->tri);
    bvh->nodes = nullptr; bvh->tri = nullptr; bvh->n_nodes = bvh->n_tri = 0;
}

// ---- BVH2 -> BVH4 collapse ----------------------------------------------------------------------
namespace {
struct Collapser {
    const mcrt_bvh *b2;
    std::vector<mcrt_bvh4_node> nodes;
    struct Slot { int32_t ref; float lo[3], hi[3]; };
    static float harea(const Slot &s) { float dx = s.hi[0] - s.lo[0], dy = s.hi[1] - s.lo[1], dz = s.hi[2] - s.lo[2]; return dx * dy + dy * dz + dz * dx; }
    static void kids(const mcrt_bvh_node &n, Slot &a, Slot &b)
    {
        a.ref = n.c0; b.ref = n.c1;
        for (int i = 0; i < 3; i++) { a.lo[i] = n.lo0[i]; a.hi[i] = n.hi0[i]; b.lo[i] = n.lo1[i]; b.hi[i] = n.hi1[i]; }
    }
    // returns {node4 index, worst-case stack entries needed below (and including) this node}
    std::pair<int32_t, uint32_t> build(int32_t n2)
    {
        Slot s[4]; int k = 2;
        kids(b2->nodes[n2], s[0], s[1]);
        if (s[0].ref == s[1].ref && s[0].ref < 0) k = 1;          // degenerate single-leaf root wrapper
        while (k < 4) {
            int pick = -1; float best = -1.f;
            for (int i = 0; i < k; i++) if (s[i].ref >= 0) { float a = harea(s[i]); if (a > best) { best = a; pick = i; } }
            if (pick < 0) break;
            Slot a, b; kids(b2->nodes[s[pick].ref], a, b);
            s[pick] = a; s[k++] = b;
        }
        const int32_t me = (int32_t)nodes.size();
        nodes.emplace_back();
        uint32_t deepest = 0;
        for (int i = 0; i < 4; i++) {
            mcrt_bvh4_child c;
            if (i < k) {
                int32_t ref = s[i].ref;
                if (ref >= 0) { auto r = build(ref); ref = r.first; deepest = std::max(deepest, r.second); }
                c.lo[0] = s[i].lo[0]; c.lo[1] = s[i].lo[1]; c.lo[2] = s[i].lo[2];
                c.hi_x = s[i].hi[0]; c.hi_y = s[i].hi[1]; c.hi_z = s[i].hi[2]; c.ref = ref; c.pad = 0;
            } else {
                c.lo[0] = c.lo[1] = c.lo[2] = INFINITY; c.hi_x = c.hi_y = c.hi_z = -INFINITY; c.ref = MCRT_BVH4_EMPTY; c.pad = 0;
            }
            nodes[me].c[i] = c;
        }
        return { me, (uint32_t)(k - 1) + deepest };
    }
};
}  // namespace

extern "C" int mcrt_build_bvh4(const mcrt_bvh *b2, mcrt_bvh4 *out)
{
    if (!b2 || !out || !b2->nodes || b2->n_nodes == 0) return mcrt::set_error(MCRT_ERR_INVALID, "mcrt_build_bvh4: no BVH2");
    Collapser c; c.b2 = b2;
    c.nodes.reserve(b2->n_nodes / 2 + 4);
    auto r = c.build(0);
    out->n_nodes = (uint32_t)c.nodes.size();
    out->max_stack = r.second;
    out->nodes = (mcrt_bvh4_node *)malloc(sizeof(mcrt_bvh4_node) * c.nodes.size());
    if (!out->nodes) return mcrt::set_error(MCRT_ERR_NOMEM, "mcrt_build_bvh4: out of memory");
    memcpy(out->nodes, c.nodes.data(), sizeof(mcrt_bvh4_node) * c.nodes.size());
    return MCRT_OK;
}

extern "C" void mcrt_free_bvh4(mcrt_bvh4 *b)
{
    if (!b) return;
    free(b->nodes); b->nodes = nullptr; b->n_nodes = 0;
}

// thr[r] = smallest double t with fl(t / dt) >= r (IEEE division is monotone in t), r = 0..R: the kernel's row_of()
// turns the reference's `row = t / dt; if (row < max_rows)` (rfimage.h:35-36) into table look-ups, exactly.
extern "C" int mcrt_row_thresholds(double dt, uint32_t R, double *thr)
{
    if (!thr || !(dt > 0.0)) return mcrt::set_error(MCRT_ERR_INVALID, "mcrt_row_thresholds: bad arguments");
    thr[0] = 0.0;
    for (uint32_t r = 1; r <= R; r++) {
        double t = (double)r * dt;
        while (t / dt >= (double)r) t = std::nextafter(t, -INFINITY);
        while (t / dt < (double)r) t = std::nextafter(t, INFINITY);
        thr[r] = t;
    }
    return MCRT_OK;
}

// ---- volume<n,res>::volume() (volume.h:19-35) ------------------------------------------------
// libstdc++ semantics: std::default_random_engine is minstd_rand0 (multiplier 16807, modulus 2^31-1,
// default seed 1); generate_canonical<double,53> consumes two draws; normal_distribution<double> is
// the Marsaglia polar method, which returns y*m and keeps x*m for the following call.
extern "C" int mcrt_generate_texture(float *vox, uint32_t n)
{
    if (!vox || n == 0) return mcrt::set_error(MCRT_ERR_INVALID, "mcrt_generate_texture: bad arguments");
    uint64_t state = 1;
    auto next = [&state]() -> double { state = (state * 16807ull) % 2147483647ull; return (double)(uint32_t)(state - 1); };
    const double R = 2147483646.0, RR = R * R;
    auto canonical = [&]() -> double {
        double sum = next();
        sum += next() * R;
        double r = sum / RR;
        return r >= 1.0 ? std::nextafter(1.0, 0.0) : r;
    };
    const size_t total = (size_t)n * n * n;
    for (size_t i = 0; i < total; i++) {
        double x, y, r2;
        do {
            x = 2.0 * canonical() - 1.0;
            y = 2.0 * canonical() - 1.0;
            r2 = x * x + y * y;
        } while (r2 > 1.0 || r2 == 0.0);
        const double mult = std::sqrt(-2 * std::log(r2) / r2);
        vox[2 * i] = (float)(y * mult);        // texture_noise: first variate of the pair
        vox[2 * i + 1] = (float)(x * mult);    // scattering_probability: the saved one
    }
    return MCRT_OK;
}

// ---- psf<>::psf (psf.h:34-58; psf.h:9 defines M_PI as 3.14159) --------------------------------
extern "C" int mcrt_psf_kernels(float freq, float var_x, float var_y, uint32_t res_um, float *axial, uint32_t n_ax, float *lateral, uint32_t n_lat)
{
    if (!axial || !lateral) return mcrt::set_error(MCRT_ERR_INVALID, "mcrt_psf_kernels: null output");
    const double pi_psf = 3.14159;
    const float res = (float)res_um / 1000.0f;
    const float half_ax = (float)((size_t)n_ax * res_um) / 1000.0f / 2.0f;
    const float half_lat = (float)((size_t)n_lat * res_um) / 1000.0f / 2.0f;
    for (uint32_t i = 0; i < n_ax; i++) {
        const float x = (float)i * res - half_ax;
        const double x2 = (double)x * (double)x;
        axial[i] = (float)(std::exp(-0.5f * (x2 / (double)var_x)) * std::cos(2 * pi_psf * (double)freq * (double)x));
    }
    for (uint32_t i = 0; i < n_lat; i++) {
        const float y = (float)i * res - half_lat;
        const double y2 = (double)y * (double)y;
        lateral[i] = (float)std::exp(-0.5f * (y2 / (double)var_y));
    }
    return MCRT_OK;
}

// ---- transducer<N>::transducer (transducer.h:24-62) -------------------------------------------
namespace {
struct F3 { float x, y, z; };
inline F3 rot(F3 v, F3 ax, float ang)   // btVector3::rotate
{
    const float d = ax.x * v.x + ax.y * v.y + ax.z * v.z;
    const F3 o{ ax.x * d, ax.y * d, ax.z * d };
    const F3 xx{ v.x - o.x, v.y - o.y, v.z - o.z };
    const F3 yy{ ax.y * v.z - ax.z * v.y, ax.z * v.x - ax.x * v.z, ax.x * v.y - ax.y * v.x };
    const float c = std::cos(ang), s = std::sin(ang);
    return F3{ o.x + xx.x * c + yy.x * s, o.y + xx.y * c + yy.y * s, o.z + xx.z * c + yy.z * s };
}
}  // namespace

extern "C" int mcrt_transducer_elements(uint32_t n, double radius_cm, double sep_mm, const float position[3], const float angles_deg[3], float *pos, float *dir)
{
    if (!pos || !dir || !position || !angles_deg || n == 0) return mcrt::set_error(MCRT_ERR_INVALID, "mcrt_transducer_elements: bad arguments");
    const double pi = 3.14159265358979323846264338327950288419716939937510;   // units.h:360
    const double xa = (angles_deg[0] * pi * 1.0) / 180.0, ya = (angles_deg[1] * pi * 1.0) / 180.0, za = (angles_deg[2] * pi * 1.0) / 180.0;
    const float amp = (float)(((sep_mm / radius_cm) * 1.0) / 10.0);   // mm/cm -> scalar
    const double amplitude = amp;
    double angle = -(amplitude * (double)n / 2.0) + amplitude / 2.0;
    const float rf = (float)radius_cm;
    for (uint32_t t = 0; t < n; t++) {
        const float a = (float)angle;
        F3 d{ std::sin(a), std::cos(a), 0.f };
        d = rot(d, F3{ 0, 0, 1 }, (float)za);
        d = rot(d, F3{ 1, 0, 0 }, (float)xa);
        d = rot(d, F3{ 0, 1, 0 }, (float)ya);
        pos[3 * t] = position[0] + rf * d.x; pos[3 * t + 1] = position[1] + rf * d.y; pos[3 * t + 2] = position[2] + rf * d.z;
        dir[3 * t] = d.x; dir[3 * t + 1] = d.y; dir[3 * t + 2] = d.z;
        angle = angle + amplitude;
    }
    return MCRT_OK;
}

// rfimage.h:183-215 create_mapping, evaluated once per geometry on the host (as the reference does in its constructor).
// Operand types as C++ gives them to the reference's statements (pinned by tests/golden/ref_probe.json "scan_maps_*": the same
// statements evaluated with the reference's own unit types, compiled from its units.h):
//   :186 ratio: `max_travel_time * speed_of_sound * 0.001f` is an unsigned product times a float = FLOAT (150.0f for 100 us x 1500);
//        `+ radius` stays float; `- radius * cos(angle_f / 2.0)` is double; `/ rows` double; rounded once to float
//   :189 shift_y: millimeter_t (double) * cosf(angle_f / 2.0f)
//   :201-205 fi, fj, r: float throughout            :208 angle: atan2f, widened
//   :211 map_x (row coordinate): float throughout, the divisor the same float depth as in :186
//   :212 map_y (column coordinate): radian_t arithmetic in double, * (float)rf_width, rounded once
// (Rounds 1-3 held the depth as a double -- 150.0000071 -- and divided in double: ratio 0.385048121 instead of 0.385048091.)
extern "C" int mcrt_scan_maps(uint32_t E, uint32_t R, double radius_mm, double total_angle, uint32_t max_travel_us, uint32_t speed_of_sound,
                              uint32_t orows, uint32_t ocols, float *map_row, float *map_col)
{
    if (!map_row || !map_col || E == 0 || R == 0 || orows == 0 || ocols == 0 || !(total_angle > 0.0)) return mcrt::set_error(MCRT_ERR_INVALID, "mcrt_scan_maps: bad arguments");
    const float radius_f = (float)radius_mm, ta_f = (float)total_angle;
    const float depth_mm_f = (float)(uint32_t)(max_travel_us * speed_of_sound) * 0.001f;
    const float ratio = (float)(((double)(depth_mm_f + radius_f) - (double)radius_f * std::cos((double)ta_f / 2.0)) / (double)(int)orows);
    const double shift_y = radius_mm * (double)std::cos(ta_f / 2.0f);
    const float half_width = (float)(int)ocols / 2.0f;
    for (uint32_t j = 0; j < ocols; j++)
        for (uint32_t i = 0; i < orows; i++) {
            const float fi = (float)(int)i + (float)shift_y / ratio;
            const float fj = (float)(int)j - half_width;
            const float r = std::sqrt(fi * fi + fj * fj);
            const double angle = (double)std::atan2(fj, fi);
            map_row[(size_t)i * ocols + j] = (r * ratio - radius_f) / depth_mm_f * (float)R;
            map_col[(size_t)i * ocols + j] = (float)(((angle - (-total_angle / 2)) / total_angle) * (double)(float)E);
        }
    return MCRT_OK;
}
