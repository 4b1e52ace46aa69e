// mcrt_kernels.hip -- gfx950 kernels of the hot path.
//
//   k_init/k_trace_lane/k_shade   scene::cast_rays (scene.cpp:50-183) + ray_physics (ray.cpp) as a wavefront pipeline:
//                lane-per-ray BVH4 closest hit, then the interface physics, one launch each per bounce
//   k_march      the RF accumulation loop (main.cpp:106-144, rfimage.h:33-40, volume.h:46-61), a lane pair / quad per segment
//                (a generic variant, and a fast one for the reference's 256^3 texture and time axis)
//   k_finalize   fixed-point RF bins -> float image (+ clears the bins: rf_image::clear, rfimage.h:161)
//   k_conv_*     rf_image::convolve (rfimage.h:93-123)
//   k_envelope   rf_image::envelope (rfimage.h:54-91)
//   k_remap      rf_image::postprocess scan conversion (rfimage.h:125-140)
//
// Everything is scalar fp32/fp64 VALU + integer work; there is no dense contraction, hence no MFMA.
// Arithmetic follows the parity contract expression by expression (compiled -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "../../include/mcrt.h"
#include "mcrt_internal.h"
#include "mcrt_detmath.h"
#include "mcrt_kernels.h"

#ifndef MCRT_SHADE_WAVES
#define MCRT_SHADE_WAVES 5             // k_shade wavefronts per SIMD the register budget is set for
#endif
#ifndef MCRT_MARCH_PAIRS_FROM
#define MCRT_MARCH_PAIRS_FROM 1048576    // k_march: lane pairs per segment for passes with at least this many paths, quads below
#endif

namespace mcrt {

struct f3 { float x, y, z; };
MCRT_DEV f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
MCRT_DEV f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
MCRT_DEV f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
MCRT_DEV f3 neg(f3 a) { return mk(-a.x, -a.y, -a.z); }
MCRT_DEV f3 scale(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
MCRT_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }                 // btVector3::dot, scalar path
MCRT_DEV f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MCRT_DEV f3 normalized(f3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return scale(a, inv); }    // btVector3::normalized
MCRT_DEV f3 xyz(float4 q) { return mk(q.x, q.y, q.z); }

enum { M_IMP = 0, M_ATT, M_MU0, M_MU1, M_SIGMA, M_SPEC, M_SHINE, M_THICK };
constexpr int OUT_NONE = -1;   // media_outside == nullptr
constexpr int OUT_SELF = -2;   // media_outside aliases the ray's own media (ray.cpp:38 + scene.cpp:154)

struct Hit { float frac; int tri; int mesh; f3 n; float da; };

// ray parameter interval [tmin,tmax] (clamped to [tlow,tcap]) in which o + t*d lies inside the box
MCRT_DEV bool slab(f3 lo, f3 hi, f3 o, f3 inv, float tlow, float tcap, float &tmin_o, float &tmax_o)
{
    float t0x = (lo.x - o.x) * inv.x, t1x = (hi.x - o.x) * inv.x;
    float t0y = (lo.y - o.y) * inv.y, t1y = (hi.y - o.y) * inv.y;
    float t0z = (lo.z - o.z) * inv.z, t1z = (hi.z - o.z) * inv.z;
    float tmin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tlow));
    float tmax = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tcap));
    tmin_o = tmin; tmax_o = tmax;
    return tmin <= tmax;
}

// The contract's plane distance (round 3): ONE fused multiply-add per plane, t = fl(plane * inv + c) with c = -(o * inv) rounded once
// per ray and axis.  For a fixed ray it is a monotone function of the plane, which is all the order-independence argument needs
// (DESIGN.md 3).  The reciprocal direction is kept FINITE: 1/0 (a ray parallel to an axis) and overflowing quotients become
// +-2^100, so every distance is a finite number (|plane| < 2^20 in any scene) and the argument needs no special cases; the sign of
// the huge distance still says on which side of the plane the origin lies.
MCRT_DEV float rcp_dir(float d) { const float r = 1.0f / d; return r > 0x1p+100f ? 0x1p+100f : (r < -0x1p+100f ? -0x1p+100f : r); }
MCRT_DEV f3 ray_c(f3 o, f3 inv) { return mk(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z)); }
// ray parameter interval of a box under that rule (the triangles' padded bounds: the eligibility test of the contract)
MCRT_DEV bool slab_c(f3 lo, f3 hi, f3 c, f3 inv, float tlow, float tcap, float &tmin_o, float &tmax_o)
{
    const float t0x = fmaf(lo.x, inv.x, c.x), t1x = fmaf(hi.x, inv.x, c.x);
    const float t0y = fmaf(lo.y, inv.y, c.y), t1y = fmaf(hi.y, inv.y, c.y);
    const float t0z = fmaf(lo.z, inv.z, c.z), t1z = fmaf(hi.z, inv.z, c.z);
    float lo3 = fminf(t0z, t1z), hi3 = fmaxf(t0z, t1z), tmin, tmax;
    const float lo1 = fminf(t0x, t1x), lo2 = fminf(t0y, t1y), hi1 = fmaxf(t0x, t1x), hi2 = fmaxf(t0y, t1y);
    asm("v_max_f32 %0, %1, %2" : "=v"(lo3) : "v"(lo3), "v"(tlow));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmin) : "v"(lo1), "v"(lo2), "v"(lo3));
    asm("v_min_f32 %0, %1, %2" : "=v"(hi3) : "v"(hi3), "v"(tcap));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(hi1), "v"(hi2), "v"(hi3));
    tmin_o = tmin; tmax_o = tmax;
    return tmin <= tmax;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// the walk only needs (fraction, triangle): k_shade looks the plane of the winner up again
struct Best { float frac; int tri; };

// The walk's triangle record, 48 bytes = three 16-byte pieces in leaf order:
//   v0 | id, v1 | mesh     the vertices (edge tests; the plane and the triangle's own padded bounds are rebuilt from them: tri_plane, tri_padded_bounds)
//   v2 | -1e-4 |n|^2       ... and processTriangle's edge tolerance
// Rounds 1-3 stored plane AND padded bounds (96 bytes, six pieces per triangle tested), round 4 tried the plane as a fourth piece.  The walk is bound by
// the cache accesses it makes: what a few register instructions rebuild -- with the contract's own expressions, so bit for bit -- is not
// fetched.
// the walk's leaf-order triangle records once more in triangle-id order, for k_shade (refresh_soa: after every build, update and refit)
__global__ void k_tris_by_id(const float4 *tris, uint32_t n_tri, float4 *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tri) return;
    const float4 r0 = tris[MCRT_TRI_PIECES * (size_t)i], r1 = tris[MCRT_TRI_PIECES * (size_t)i + 1], r2 = tris[MCRT_TRI_PIECES * (size_t)i + 2];
    const uint32_t id = __float_as_uint(r0.w);
    if (id >= n_tri) return;
    out[MCRT_TRI_PIECES * (size_t)id] = r0; out[MCRT_TRI_PIECES * (size_t)id + 1] = r1; out[MCRT_TRI_PIECES * (size_t)id + 2] = r2;
}

__global__ void k_expand_tris(const float4 *in, uint32_t n_tri, float4 *out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tri) return;
    const float4 t0 = in[3 * (size_t)t], t1 = in[3 * (size_t)t + 1], t2 = in[3 * (size_t)t + 2];
    const f3 v0 = xyz(t0), v1 = xyz(t1), v2 = xyz(t2);
    const f3 n = cross(v1 - v0, v2 - v0);
    float4 *o = out + MCRT_TRI_PIECES * (size_t)t;
    o[0] = make_float4(v0.x, v0.y, v0.z, t0.w);
    o[1] = make_float4(v1.x, v1.y, v1.z, t1.w);
    o[2] = make_float4(v2.x, v2.y, v2.z, dot(n, n) * -0.0001f);           // processTriangle's edge tolerance, -1e-4 |n|^2
}
// the plane of a triangle, n = (v1 - v0) x (v2 - v0) and dot(v0, n) (Bullet's processTriangle)
MCRT_DEV float4 tri_plane(f3 v0, f3 v1, f3 v2)
{
    const f3 n = cross(v1 - v0, v2 - v0);
    return make_float4(n.x, n.y, n.z, dot(v0, n));
}
// the triangle's own padded bounds (contract: pad = 2e-4 * largest extent + pad_abs), bit for bit what the builders put around the
// leaves (mcrt_build_bvh, k_prims): min / max are exact, the three roundings (extent, pad, the six sums) are the builders' own
MCRT_DEV void tri_padded_bounds(f3 v0, f3 v1, f3 v2, float pad_abs, f3 &lo_o, f3 &hi_o)
{
    const f3 lo = mk(fminf(v0.x, fminf(v1.x, v2.x)), fminf(v0.y, fminf(v1.y, v2.y)), fminf(v0.z, fminf(v1.z, v2.z)));
    const f3 hi = mk(fmaxf(v0.x, fmaxf(v1.x, v2.x)), fmaxf(v0.y, fmaxf(v1.y, v2.y)), fmaxf(v0.z, fmaxf(v1.z, v2.z)));
    const float ext = fmaxf(fmaxf(fmaxf(0.0f, hi.x - lo.x), hi.y - lo.y), hi.z - lo.z);
    const float pad = 2e-4f * ext + pad_abs;
    lo_o = mk(lo.x - pad, lo.y - pad, lo.z - pad);
    hi_o = mk(hi.x + pad, hi.y + pad, hi.z + pad);
}

struct Rng { uint32_t k0, k1, element, sample, bounce; };
MCRT_DEV void rng_block(const Rng &g, uint32_t block, double &a, double &b)
{
    uint32_t o[4];
    philox4x32_10(g.element, g.sample, g.bounce, block, g.k0, g.k1, o);
    a = u53(o[0], o[1]);
    b = u53(o[2], o[3]);
}

// ray.cpp:167-211
MCRT_DEV f3 random_unit_vector(f3 v, float cos_theta, const Rng &g)
{
    bool flag = false;
    float px, py, p;
    uint32_t attempt = 0;
    do {
        double ua, ur;
        rng_block(g, 2u + attempt, ua, ur);
        double a = ua * 2 * PI_D;
        double r = 0.5 * sqrt(ur);
        double sa, ca;
        det_sincos(a, sa, ca);
        px = (float)(r * ca);
        py = (float)(r * sa);
        p = px * px + py * py;
        attempt++;
    } while (!(p <= 0.25f) && attempt < 8u);
    float vx = v.x, vy = v.y, vz = v.z;
    if (fabsf(vx) > fabsf(vy)) { vx = vy; vy = v.x; flag = true; }
    float b = 1 - vx * vx;
    float radicando = 1 - cos_theta * cos_theta;
    radicando = radicando / (p * b);
    float c = sqrtf(radicando);
    px = px * c;
    py = py * c;
    float d = cos_theta - vx * px;
    float wx = vx * cos_theta - b * px;
    float wy = vy * d + vz * py;
    float wz = vz * d - vy * py;
    if (flag) { float aux = wy; wy = wx; wx = aux; }
    return mk(wx, wy, wz);
}

MCRT_DEV float std_max(float a, float b) { return (a < b) ? b : a; }

MCRT_DEV uint32_t vox_index(float q, uint32_t n, uint32_t mask)
{
    long long i;
    if (!(fabsf(q) < 9.2233720368547758e18f)) i = (long long)0x8000000000000000ull;
    else if (fabsf(q) < 2147483648.0f) i = (long long)(int)q;
    else i = (long long)q;
    return mask ? ((uint32_t)i & mask) : ((uint32_t)i) % n;   // mask = n-1 when n is a power of two (the reference's 256)
}

MCRT_DEV uint32_t steps_from(double q)
{
    if (!(fabs(q) < 9.2233720368547758e18)) return 0u;
    return (uint32_t)(long long)q;
}

MCRT_DEV long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- quad (4-lane) exchanges on the DPP path: no LDS, VALU rate.  Control flow around them is quad-uniform (the four
// lanes of a path hold identical state), so the source lanes are always active.
template <int CTRL> MCRT_DEV int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
template <int CTRL> MCRT_DEV float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }


#define QP_BCAST(k) ((k) * 0x55)

// The LDS image of a scan-line in k_march: entry r = { thr[r], bin[r] }, 16 bytes -- a step's two thresholds and its bin are then
// three constant offsets from ONE address (row << 4), with no base register.
struct RowBin { double thr; long long bin; };

// row = (int)(t / row_dt) if that quotient is < R, else -1 (rfimage.h:33-40), WITHOUT the double division:
// thr[r] (host-computed, mcrt_row_thresholds) is the smallest double t whose IEEE quotient fl(t/row_dt) is >= r, so the
// row is the largest r with thr[r] <= t.  Exactly equivalent to the division for every double t >= 0.
MCRT_DEV int row_of(double t, const RowBin *rb, uint32_t R, double inv_dt, double thr_end)
{
    if (!(t < thr_end) || !(t >= 0.0)) return -1;
    int r = (int)(t * inv_dt);                                   // within one row of the answer
    r = r < 0 ? 0 : (r > (int)R - 1 ? (int)R - 1 : r);
    const double lo = rb[r].thr, hi = rb[r + 1].thr;
    if ((t < lo) | !(t < hi)) {                                  // the estimate missed by a rounding: walk to the row
        while (t < rb[r].thr) r--;
        while (t >= rb[r + 1].thr) r++;
    }
    return r;
}

// the same row when a good guess is at hand: two threshold reads confirm it.  (k_march's guess is t * inv_dt itself, which misses
// only by a rounding: a guess from the lane's previous row + its stride misses whenever the row advances by one more than the
// stride -- every tenth step or so, i.e. in EVERY step of a wavefront some lane would take the search below, for all 64.)
// PADDED: the image holds entries up to the largest guess a valid step can make ((int)(max_travel * inv_dt), + 1), those beyond
// thr[R] filled with -inf -- the guess needs no clamp, and a time beyond the image fails "t < hi" and is sorted out by row_of.
template <bool PADDED>
MCRT_DEV int row_near(double t, int guess, const RowBin *rb, uint32_t R, double inv_dt, double thr_end)
{
    const int r = PADDED ? guess : (guess < 0 ? 0 : (guess > (int)R - 1 ? (int)R - 1 : guess));
    const double lo = rb[r].thr, hi = rb[r + 1].thr;
    if ((t >= lo) & (t < hi)) return r;                          // (false for NaN, negative times and times beyond the image)
    return row_of(t, rb, R, inv_dt, thr_end);
}

// x / tex_res, correctly rounded, as two fmas around a multiply by the rounded reciprocal (Markstein's correction).
// Used only when the GPU itself has verified (k_verify_div, exhaustive over the gated range) that the sequence
// equals IEEE division for this tex_res; otherwise, and outside the gate, the division instruction sequence is used.
MCRT_DEV float div_res(float x, const FrameArgs &a)
{
    const float ax = fabsf(x);
    if (a.fast_div && ((ax > 1e-18f && ax < 1e18f) || x == 0.0f)) {
        const float q0 = x * a.tex_rcp;
        const float r = fmaf(-q0, a.tex_res, x);
        return fmaf(r, a.tex_rcp, q0);
    }
    return x / a.tex_res;
}

// texture cell of a point, volume.h:46-61 (x / resolution, (int) cast, modulo), for any texture size and magnitude
MCRT_DEV size_t vox_cell(f3 p, const FrameArgs &a)
{
    const uint32_t vx = vox_index(div_res(p.x, a), a.tex_n, a.tex_mask), vy = vox_index(div_res(p.y, a), a.tex_n, a.tex_mask), vz = vox_index(div_res(p.z, a), a.tex_n, a.tex_mask);
    return ((size_t)vx * a.tex_n + vy) * a.tex_n + vz;
}
// the same cell when every coordinate is below lean_bound in magnitude and the size is a power of two: branch-free.
// |x / res| < 2^31 there, and the corrected reciprocal multiply is the verified quotient for |x| > 1e-18 and x == 0; for
// the tiny values in between both it and the true quotient are below 1 in magnitude (tex_res > 1e-16), so the cell is 0
// either way.
MCRT_DEV uint32_t vox_lean1(float x, const FrameArgs &a)
{
    const float q0 = x * a.tex_rcp;
    const float r = fmaf(-q0, a.tex_res, x);
    return (uint32_t)(int)fmaf(r, a.tex_rcp, q0) & a.tex_mask;
}
MCRT_DEV uint32_t vox_cell_lean(f3 p, const FrameArgs &a)
{
    return (((vox_lean1(p.x, a) << a.tex_shift) | vox_lean1(p.y, a)) << a.tex_shift) | vox_lean1(p.z, a);
}
// ... and when the texture is the reference's 256^3 (volume.h:19): the three low bytes packed by two v_perm_b32.
// (The device copy keeps the reference's cell order, (x * 256 + y) * 256 + z.  Round 6 counted and measured other orders -- x fastest, 128-byte
//  lines as 4 x 2 x 2 bricks or 4 x 1 x 4 tiles -- and the quotients as packed fp32: all slower, the kernel is bound by the instructions it issues,
//  not by its gathers.  DESIGN.md A.8, profiles/round6/exp_march_layout.txt, tools/variants/round6_march_layout.patch.)
MCRT_DEV uint32_t vox_q(float x, const FrameArgs &a)
{
    const float q0 = x * a.tex_rcp;
    const float r = fmaf(-q0, a.tex_res, x);
    return (uint32_t)(int)fmaf(r, a.tex_rcp, q0);
}
MCRT_DEV uint32_t vox_cell_lean256(f3 p, const FrameArgs &a)
{
    const uint32_t yz = __builtin_amdgcn_perm(vox_q(p.y, a), vox_q(p.z, a), 0x0c0c0400u);      // { z.b0, y.b0, 0, 0 }
    return __builtin_amdgcn_perm(vox_q(p.x, a), yz, 0x0c040100u);                                // { z.b0, y.b0, x.b0, 0 }
}
// The same with the reciprocal and the resolution held in VECTOR registers.  gfx950 issues v_mul_f32 / v_fma_f32 / v_add_f32 in 2 cycles per wavefront when every register
// operand is a vector register, and in 4 as soon as one is a SCALAR register (profiles/round6/valu_classes.json: the same for v_add_u32, v_and_b32 ...; min / max / compare /
// convert / shift / packed / f64 / fma_mix instructions take 4 either way).  The three instructions of a quotient read the wave-uniform constants: as scalar operands -- what the
// compiler picks by itself -- the 36 of an iteration of k_march cost twice what they need to.  (vgpr(): an empty asm the compiler cannot see through.)
MCRT_DEV float vgpr(float s) { float v = s; asm volatile("" : "+v"(v)); return v; }
MCRT_DEV uint32_t vox_q_v(float x, float rcp_v, float res_v)
{
    const float q0 = x * rcp_v;
    const float r = fmaf(-q0, res_v, x);
    return (uint32_t)(int)fmaf(r, rcp_v, q0);
}
MCRT_DEV uint32_t vox_cell_lean256_v(f3 p, float rcp_v, float res_v)
{
    const uint32_t yz = __builtin_amdgcn_perm(vox_q_v(p.y, rcp_v, res_v), vox_q_v(p.z, rcp_v, res_v), 0x0c0c0400u);
    return __builtin_amdgcn_perm(vox_q_v(p.x, rcp_v, res_v), yz, 0x0c040100u);
}
MCRT_DEV float abs_sum(f3 p) { return (fabsf(p.x) + fabsf(p.y)) + fabsf(p.z); }   // >= every |coordinate|; NaN/inf propagate

// one echo into the scan-line's fixed-point LDS bins (2^-40 units; integer adds commute, so the image does not depend
// on the order lanes, waves or workgroups arrive in)
// rint(echo * 2^40) for |echo| < 1024 (so |echo * 2^40| < 2^50), round to nearest even -- in TWO floating-point instructions: the
// product echo * 2^40 is exact in double, and adding 1.5 * 2^52 to it rounds the sum to an integer (ulp = 1 in [2^52, 2^53))
// whose low mantissa bits ARE that integer, offset by 2^51; one fma does both, an integer subtract removes the offset.
MCRT_DEV long long fix40(float echo)
{
    const double x = fma((double)echo, 0x1p40, 0x1.8p52);
    return (long long)__double_as_longlong(x) - (long long)__double_as_longlong(0x1.8p52);
}

MCRT_DEV void rf_add(RowBin *rb, uint32_t *lflags, int row, float echo)
{
    if (row < 0) return;
    if (!(fabsf(echo) < 1024.0f)) { atomicOr(&lflags[row >> 5], 1u << (row & 31)); return; }
    const long long v = fix40(echo);
    if (v != 0) atomicAdd((unsigned long long *)&rb[row].bin, (unsigned long long)v);
}

// =============================================================================================================
// The frame is a WAVEFRONT pipeline that mirrors the reference's own structure (scene::cast_rays produces segments,
// main.cpp:106-144 consumes them), one launch per stage and bounce, queues of live paths in HBM between stages:
//
//   k_init            first_ray of every (scan-line, sample) path, scene.cpp:83-101            1 lane  / path
//   for bounce b:
//     k_trace_lane    closest hit of every live ray: BVH4 walk                                  1 lane  / ray
//     k_shade         thickness draw, travel, hit_boundary, segment record, next ray;           1 lane  / ray
//                     survivors are compacted into the next bounce's queue (wave ballot + prefix)
//   k_march           RF accumulation of every segment (main.cpp:112-140)                       2 lanes / segment
//
// Every stage therefore runs with full wavefronts of lanes doing the same thing: dead paths cost nothing after the
// bounce they die in, the fp64-heavy interface physics is not replicated, and the lean walk kernel keeps 5 waves/SIMD.
// Paths draw random numbers from their own (scan-line, sample, bounce) counter and RF bins are integer sums, so the
// image does not depend on queue order.  Path state, rays and closest-hit words live in QUEUE ORDER and are compacted with
// the queue every bounce (ping-pong halves by bounce parity): every launch reads and writes them densely and coalesced.
// (Round 3 measured the alternative the sample loop of scene.cpp:102-110 suggests -- queues SORTED into bundles of the sample
// paths of a scan-line with the same reflect / refract history, path state in place by path id: 58 vs 56 % of the walk's lanes
// active, the pass 8 % slower; DESIGN.md A.4, profiles/round3/exp_*.)
// =============================================================================================================

struct Ray { f3 f2, to; };

#define MCRT_KEY_MISS ((0x3f800000ull << 32) | 0xffffffffull)   // fraction 1.0, no triangle

// pieces per ray for a bounce with n rays: the largest power of two <= limit / n, at most 16 (1 when the bounce is large)
MCRT_DEV uint32_t ksplit(uint32_t n, uint32_t limit)
{
    uint32_t k = 1u;
    while (k < 16u && n * (k * 2u) <= limit) k *= 2u;
    return k;
}

// max_ray_length (ray.cpp:110-113) + enlarge (scene.cpp:292-298) + the 0.1 start offset (scene.cpp:115).
// The segment a ray is tested on is a pure function of the path state -- origin, direction and the length factor L / 100 -- so the
// state carries that ONE float (ray_len, evaluated once per bounce where intensity and medium are at hand) and both the walk and
// k_shade rebuild the end points from it with the same expressions (ray_of): rounds 1-3 wrote a 32-byte ray record per ray and
// bounce in k_shade and read it back twice.
MCRT_DEV float ray_len(float intensity, float att, const FrameArgs &a)
{
    const float L = 10.f * det_logf(a.eps / intensity) / -att * a.freq;
    return L / 100.0f;
}
MCRT_DEV Ray ray_of(f3 from, f3 dir, float Ls, const FrameArgs &a)
{
    Ray r;
    r.to = mk(from.x + Ls * (a.sx * dir.x), from.y + Ls * (a.sy * dir.y), from.z + Ls * (a.sz * dir.z));
    r.f2 = mk(from.x + a.offs * dir.x, from.y + a.offs * dir.y, from.z + a.offs * dir.z);
    return r;
}

#define MCRT_STATE0_AT(pos, S) ((pos) - (pos) % (S))      /* queue position of the bounce-0 state of the path queued at pos: its scan-line's first sample (k_init) */
__global__ void __launch_bounds__(256) k_init(FrameArgs a)
{
    const uint32_t pos = blockIdx.x * blockDim.x + threadIdx.x;          // this thread fills queue position `pos`
    const uint32_t np = a.ne * a.S;
    if (pos == 0) { a.counts[0] = np; for (uint32_t b = 1; b <= a.B; b++) a.counts[b] = 0u; }
    if (pos < MCRT_MAX_BOUNCES * MCRT_XCDS) a.cursors[(size_t)pos * MCRT_CURSOR_STRIDE] = 0u;   // k_trace_lane's queue cursors (relative, see there)
    if (pos >= np) return;
    // Queue position -> path.  Paths are numbered frame-major (pid = (frame * ne_frame + scan-line) * S + sample) but QUEUED
    // scan-line-major: the F frames of a scan-line sit next to each other.  The queue is swept in order, so the rays in flight
    // then belong to a few scan-lines (times all frames) and walk the same part of the BVH; later bounces inherit the order
    // from the order-preserving compaction of k_shade.
    const uint32_t F = a.ne / a.ne_frame;
    const uint32_t qline = pos / a.S, sample = pos % a.S;
    const uint32_t scan = qline / F, fr = qline % F;
    const uint32_t pid = (fr * a.ne_frame + scan) * a.S + sample;
    const size_t pe = (size_t)fr * a.pose_stride + a.e_begin + scan;      // pose_stride = 0: one probe pose for every frame of the pass (transducer.h:64-67)
    const f3 from = mk(a.el_pos[3 * pe], a.el_pos[3 * pe + 1], a.el_pos[3 * pe + 2]);
    const f3 dir = mk(a.el_dir[3 * pe], a.el_dir[3 * pe + 1], a.el_dir[3 * pe + 2]);
    const float intensity = a.I0 / (float)a.S;
    // Every sample path of a scan-line starts as a copy of the same first_ray (scene.cpp:83-101): the state of bounce 0 is written ONCE per queued
    // (scan-line, frame), at its first sample's position -- where the walk reads it (ray_stride) and where k_shade / k_path look it up for all S samples
    // (MCRT_STATE0_AT) -- instead of S times (48 B x 2.6 M paths per 20-frame pass written here and read back by k_shade).
    if (sample == 0u) {
        a.st0[pos] = make_float4(from.x, from.y, from.z, ray_len(intensity, a.mats[2 * a.start_mat].y, a));   // origin | length factor of the ray (ray_of)
        a.st1[pos] = make_float4(dir.x, dir.y, dir.z, __int_as_float((int)a.start_mat));
        a.st2[pos] = make_float4(0.0f, 0.0f, __int_as_float(OUT_NONE), intensity);  // distance_traveled (double) | outside | intensity
    }
    a.queue[pos] = pid;                                  // queue of bounce 0 (buffer 0 of two)
    a.seg_count[pid] = 0u;
    if (pos < a.ne) a.key0[pos] = MCRT_KEY_MISS;          // bounce 0: one closest-hit word per queued (scan-line, frame)
}

// =============================================================================================================
// k_trace_lane -- the closest-hit walk (Bullet's rayTest, scene.cpp:115-126), ONE LANE per ray, 64 rays per wavefront.
//
// (Round 1 walked a ray with a quad of four lanes, one child of the BVH4 node each: 16 rays in flight per wavefront, with the
// counters showing its wavefronts parked on memory for more than half of their life -- latency-bound.  One lane per ray puts
// four times as many rays behind every wavefront and spends fewer instructions per ray: no quad ranking exchanges, and the
// slab planes of two children at a time go through the packed-f32 pipe.)
//
// Nodes are read from a COMPACT copy of the BVH4 (k_nodes_walk): 64 bytes per node instead of 128 -- the walk is bound by the
// vector memory pipe (tools/fetch_roof.hip: a scattered 16-byte-per-lane load costs the compute unit's TCP ~0.75 lanes per
// clock, whatever the cache level), so what counts is the number of 16-byte pieces a lane fetches per node: four
//     lo.x[4] lo.y[4] | lo.z[4] hi.x[4] | hi.y[4] hi.z[4] | ref[4]          (boxes as IEEE half floats, child-transposed)
// instead of seven.  The halves are rounded OUTWARDS (lo down, hi up), so every stored box contains the builder's box: node
// boxes only ever cull, and the contract's closest hit does not depend on them as long as they contain their triangles'
// padded bounds -- hits stay bit-identical, the walk visits ~2.5 % more nodes (measured on the 1 M-triangle
// scene).  Unused slots are stored as the point box at +infinity, which no slab test hits (so the walk needs no EMPTY test).
// mcrt_get_bvh4 hands out the tree AS WALKED (the decoded boxes), so a CPU walk of it counts exactly this walk's visits.
// Per ray the arithmetic is the contract's slab test, (plane - origin) * reciprocal with the min / max combination of slab();
// the next node is the nearest hit child (key: t_near bits with the slot number in the two low bits), the other hit children
// are stacked in slot order -- the order, and therefore the visit counts, of a sequential walk.
// Traversal stacks: MCRT_LANE_STACK entries per lane in LDS ([entry][thread], conflict-free); deeper entries (only reachable on
// degenerate paths of deep trees) go to a global overflow array.
// =============================================================================================================
#ifndef MCRT_LANE_STACK
#define MCRT_LANE_STACK 32           // (the headline workload's deepest walk stacks 16 entries, 14 at the 99.9th percentile: profiles/round4/bvh_width.json)
#endif
// Persistent kernels carry a WATCHDOG: every 4096 iterations of its outer loop a wavefront compares the 100 MHz wall clock
// with its start, and a kernel that is still running after MCRT_WATCHDOG_SECONDS sets bit 1 of the device error word and leaves
// -- a logic error then surfaces as MCRT_ERR_LIMIT from the next synchronising call instead of a hung GPU.
#ifndef MCRT_WATCHDOG_SECONDS
#define MCRT_WATCHDOG_SECONDS 20
#endif
#define MCRT_WATCHDOG_DECL() const unsigned long long wd_start = wall_clock64(); uint32_t wd_iter = 0;
#define MCRT_WATCHDOG_CHECK() { if ((++wd_iter & 4095u) == 0u && wall_clock64() - wd_start > (unsigned long long)MCRT_WATCHDOG_SECONDS * 100000000ull) { \
        if ((threadIdx.x & 63) == 0) atomicOr(a.error_flag, 2u); break; } }
#ifndef MCRT_LANE_VGPRS
#define MCRT_LANE_VGPRS 104          // register budget of k_trace_lane: four of its wavefronts per SIMD (1024 persistent workgroups, 4 per CU) take 416 of the
#endif                               // SIMD's 512 registers and leave 96 for a k_march wavefront (80) beside them; at 96 (five wavefronts' worth, rounds 2-3)
                                     // the refill that rebuilds the ray from the path state spilled four registers
#ifndef MCRT_LANE_REFILL
#define MCRT_LANE_REFILL 16          // fetch and set up new rays once this many of a wavefront's 64 lanes are without one
#endif
#ifndef MCRT_LANE_LEAF_BATCH
#define MCRT_LANE_LEAF_BATCH 20      // leave the inner-node phase once this many lanes are parked on a leaf
#endif
#ifndef MCRT_LANE_POLL
#define MCRT_LANE_POLL 1             // walkers of one ray exchange their closest hit through the ray's word (k_trace_lane)
#endif
#ifndef MCRT_MARCH_WAVES
#define MCRT_MARCH_WAVES 6           // waves per SIMD the register budget of k_march is set for (7: 14 spilled registers, 789 vs 750 us per launch; 5: 777)
#endif
#ifndef MCRT_MARCH_TILE
#define MCRT_MARCH_TILE 256          // slots a wavefront of k_march sorts by segment length at a time (a multiple of 64, at most 256: one byte per slot)
#endif
#ifndef MCRT_SHADE_TABLE
#define MCRT_SHADE_TABLE 32          // rows of the material / mesh tables k_shade keeps in LDS (larger scenes read them from memory)
#endif
#ifndef MCRT_MARCH_LDS_TABLES
#define MCRT_MARCH_LDS_TABLES 1      // k_march keeps the per-material table and a tile's length classes in LDS (0: rounds 2-3, both re-read through the vector memory pipe)
#endif
#ifndef MCRT_MARCH_MTAB
#define MCRT_MARCH_MTAB 32           // rows of the per-material table k_march keeps in LDS (scenes with more materials read it from memory)
#endif
#ifndef MCRT_LANE_ADOPT_STEPS
#define MCRT_LANE_ADOPT_STEPS 4      // while idle lanes wait for a subtree, the inner-node phase returns to the hand-over after this many steps
#endif
#ifndef MCRT_LANE_FETCH
#define MCRT_LANE_FETCH 128          // queue positions a wavefront claims per atomic in a LARGE launch (>= MCRT_LANE_FETCH_FROM items), 64 below: measured
#endif                               // 0.434 / 0.426 / 0.426 ms per frame with 64 / 128 / 256 at 128 frames in flight (16.7 M items), 0.523 / 0.531 with 64 / 128 on
#ifndef MCRT_LANE_FETCH_SMALL
#define MCRT_LANE_FETCH_SMALL 64
#endif
#ifndef MCRT_LANE_FETCH_FROM         // a 20-frame pass (2.6 M items: what a wavefront holds back at the end of the queue weighs more there)
#define MCRT_LANE_FETCH_FROM 4194304
#endif

// float -> half, rounded towards -infinity / +infinity (integer steps on the half's bit pattern from the nearest-even conversion)
MCRT_DEV uint32_t half_towards(float x, bool up)
{
    __half h = __float2half_rn(x);
    uint32_t b = (uint32_t)__half_as_ushort(h);
    const float back = __half2float(h);
    if (x != x) return 0x7e00u;                                  // NaN stays NaN (never produced by the builders)
    if (up ? (back < x) : (back > x)) {                          // the nearest half lies on the wrong side: one step towards the target
        const bool neg = (b & 0x8000u) != 0u;
        if ((b & 0x7fffu) == 0u) b = up ? 0x0001u : 0x8001u;     // +-0 -> the smallest subnormal of the right sign
        else if (neg == up) b -= 1u;                             // magnitude shrinks: negative going up, positive going down
        else b += 1u;                                            // magnitude grows (0x7bff + 1 = 0x7c00 = infinity: still an outward bound)
    }
    b &= 0xffffu;
    // no subnormal halves (the walk's arithmetic then never depends on a denormal mode): snap outwards to 0 or +-2^-14
    if ((b & 0x7c00u) == 0u && (b & 0x03ffu) != 0u) {
        const bool neg = (b & 0x8000u) != 0u;
        b = up ? (neg ? 0x8000u : 0x0400u) : (neg ? 0x8400u : 0x0000u);
    }
    return b;
}
MCRT_DEV float half_bits_to_float(uint32_t b) { return __half2float(__ushort_as_half((unsigned short)b)); }

// the walk's 64-byte nodes from the builders' 128-byte ones
__global__ void k_nodes_walk(const float4 *in, uint32_t n_nodes, uint4 *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    uint32_t lo[3][4], hi[3][4]; int ref[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float4 A = in[8 * (size_t)i + 2 * c], B = in[8 * (size_t)i + 2 * c + 1];
        ref[c] = __float_as_int(B.z);
        const bool empty = ref[c] == MCRT_BVH4_EMPTY;
        const float l[3] = { A.x, A.y, A.z }, h[3] = { A.w, B.x, B.y };
#pragma unroll
        for (int k = 0; k < 3; k++) { lo[k][c] = empty ? 0x7c00u : half_towards(l[k], false); hi[k][c] = empty ? 0x7c00u : half_towards(h[k], true); }
    }
    uint4 *o = out + 4 * (size_t)i;
#define MCRT_PACK4(v) (v)[0] | ((v)[1] << 16), (v)[2] | ((v)[3] << 16)
    o[0] = make_uint4(MCRT_PACK4(lo[0]), MCRT_PACK4(lo[1]));
    o[1] = make_uint4(MCRT_PACK4(lo[2]), MCRT_PACK4(hi[0]));
    o[2] = make_uint4(MCRT_PACK4(hi[1]), MCRT_PACK4(hi[2]));
    o[3] = make_uint4((uint32_t)ref[0], (uint32_t)ref[1], (uint32_t)ref[2], (uint32_t)ref[3]);
#undef MCRT_PACK4
}
// ... and back: the tree as the walk sees it, in the builders' layout (for mcrt_get_bvh4)
__global__ void k_nodes_walk_decode(const uint4 *in, uint32_t n_nodes, float4 *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const uint4 q0 = in[4 * (size_t)i], q1 = in[4 * (size_t)i + 1], q2 = in[4 * (size_t)i + 2], q3 = in[4 * (size_t)i + 3];
    const uint32_t w[12] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w };      // lo.x lo.y lo.z hi.x hi.y hi.z, two words each
    uint32_t ref[4] = { q3.x, q3.y, q3.z, q3.w };
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float v[6];
#pragma unroll
        for (int k = 0; k < 6; k++) v[k] = half_bits_to_float((w[2 * k + (c >> 1)] >> ((c & 1) * 16)) & 0xffffu);
        const bool empty = (int)ref[c] == MCRT_BVH4_EMPTY;
        if (empty) { v[0] = v[1] = v[2] = INFINITY; v[3] = v[4] = v[5] = -INFINITY; }                 // the builders' own form of an unused slot
        out[8 * (size_t)i + 2 * c] = make_float4(v[0], v[1], v[2], v[3]);
        out[8 * (size_t)i + 2 * c + 1] = make_float4(v[4], v[5], __int_as_float((int)ref[c]), 0.0f);
    }
}

// the four children's plane distances from a packed pair of half-float words: the contract's t = fl(plane * inv + c), c = -(o * inv),
// ONE mixed-precision fma per plane (v_fma_mix_f32 reads the half operand directly; op_sel picks the half of the word)
struct Planes4 { float a0, a1, b0, b1; };
MCRT_DEV Planes4 planes4(uint32_t w01, uint32_t w23, float c, float inv)
{
    Planes4 r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r.a0) : "v"(w01), "v"(inv), "v"(c));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.a1) : "v"(w01), "v"(inv), "v"(c));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r.b0) : "v"(w23), "v"(inv), "v"(c));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.b1) : "v"(w23), "v"(inv), "v"(c));
    return r;
}

// the same with the node's packed word in a SCALAR register (k_trace_packet: the node is wave-uniform)
MCRT_DEV Planes4 planes4_s(uint32_t w01, uint32_t w23, float c, float inv)
{
    Planes4 r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r.a0) : "s"(w01), "v"(inv), "v"(c));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.a1) : "s"(w01), "v"(inv), "v"(c));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r.b0) : "s"(w23), "v"(inv), "v"(c));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r.b1) : "s"(w23), "v"(inv), "v"(c));
    return r;
}

// ---- the lane-per-ray walk's two steps ---------------------------------------------------------------------------------
// A lane's traversal stack: entries [sb, sp), entry e of thread t at lds[e * 256 + t] while e < MCRT_LANE_STACK, beyond that in the
// global overflow array (only reachable on degenerate paths of deep trees).
template <int STACK> struct LaneStackT { int *lds; int *ovf; size_t ovf_stride; int tid; static constexpr int depth = STACK; };     // depth: entries in LDS
constexpr int CUR_IDLE = (int)0x80000000;      // walk state: cur >= 0 inner node, cur < 0 ~(leaf descriptor), CUR_IDLE = no walk in progress
template <class LS> MCRT_DEV void lane_pop(const LS &S, int &cur, int &sp, int sb)
{
    if (sp > sb) {
        sp--;
        // (the overflow part is asked for the whole wavefront first: the general form alone computes the 64-bit overflow address in every popping lane and reads through a flat load)
        if (__builtin_expect(__any(sp >= LS::depth), 0)) cur = (sp < LS::depth) ? S.lds[sp * 256 + S.tid] : S.ovf[(size_t)(sp - LS::depth) * S.ovf_stride];
        else cur = S.lds[sp * 256 + S.tid];
    }
    else cur = CUR_IDLE;
}
template <class LS> MCRT_DEV void lane_push(const LS &S, int &sp, int v)
{
    if (sp < LS::depth) S.lds[sp * 256 + S.tid] = v; else S.ovf[(size_t)(sp - LS::depth) * S.ovf_stride] = v;
    sp++;
}
struct LaneRay { float cx, cy, cz, ix, iy, iz; bool nx, ny, nz; };     // c = -(origin * reciprocal direction) and the reciprocal direction; reciprocal negative?

// slab interval of one child from the distances of its three NEAR and three FAR planes
MCRT_DEV bool slab_near_far(float nx, float ny, float nz, float fx, float fy, float fz, float tlow, float tcap, float &tmin_o)
{
    float tmin, tmax;
    asm("v_max_f32 %0, %1, %2" : "=v"(nz) : "v"(nz), "v"(tlow));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmin) : "v"(nx), "v"(ny), "v"(nz));
    asm("v_min_f32 %0, %1, %2" : "=v"(fz) : "v"(fz), "v"(tcap));
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(fx), "v"(fy), "v"(fz));
    tmin_o = tmin;
    return tmin <= tmax;
}

// one inner node: the four children's slab tests, the nearest hit child next, the other hit children stacked in slot order
template <class LS> MCRT_DEV void lane_node_compute(const LS &S, const LaneRay &r, float t_lo, float tcap, const uint4 Q0, const uint4 Q1, const uint4 Q2, const uint4 RF, int &cur, int &sp, int sb);
template <class LS> MCRT_DEV void lane_node_step(const FrameArgs &a, const LS &S, const LaneRay &r, float t_lo, float tcap, int &cur, int &sp, int sb)
{
    const uint4 *N = (const uint4 *)((const char *)a.nodes_walk + ((uint32_t)cur << 6));
    const uint4 Q0 = N[0], Q1 = N[1], Q2 = N[2], RF = N[3];            // (as eight 8-byte pieces instead: 0.492 vs 0.427 ms per frame, round 3)
    lane_node_compute(S, r, t_lo, tcap, Q0, Q1, Q2, RF, cur, sp, sb);
}

template <class LS> MCRT_DEV void lane_node_compute(const LS &S, const LaneRay &r, float t_lo, float tcap, const uint4 Q0, const uint4 Q1, const uint4 Q2, const uint4 RF, int &cur, int &sp, int sb)
{
    // six plane distances of the four children
    // Which plane of a slab the ray meets first follows from the SIGN of the reciprocal direction (low plane for a positive one):
    // the packed words of the near and far planes are picked per axis (12 selects) instead of ordering the 24 distances afterwards
    // (24 min / max).  With low <= high and a monotone distance function the picked distances ARE the minimum and maximum whenever
    // both are numbers; where one is not (rays parallel to an axis) the interval comes out wider, never narrower -- a node may be
    // entered that min/max would have skipped, the triangle tests decide as before.
    const Planes4 XN = planes4(r.nx ? Q1.z : Q0.x, r.nx ? Q1.w : Q0.y, r.cx, r.ix), XF = planes4(r.nx ? Q0.x : Q1.z, r.nx ? Q0.y : Q1.w, r.cx, r.ix);
    const Planes4 YN = planes4(r.ny ? Q2.x : Q0.z, r.ny ? Q2.y : Q0.w, r.cy, r.iy), YF = planes4(r.ny ? Q0.z : Q2.x, r.ny ? Q0.w : Q2.y, r.cy, r.iy);
    const Planes4 ZN = planes4(r.nz ? Q2.z : Q1.x, r.nz ? Q2.w : Q1.y, r.cz, r.iz), ZF = planes4(r.nz ? Q1.x : Q2.z, r.nz ? Q1.y : Q2.w, r.cz, r.iz);
    float tn0, tn1, tn2, tn3;
    const bool h0 = slab_near_far(XN.a0, YN.a0, ZN.a0, XF.a0, YF.a0, ZF.a0, t_lo, tcap, tn0);
    const bool h1 = slab_near_far(XN.a1, YN.a1, ZN.a1, XF.a1, YF.a1, ZF.a1, t_lo, tcap, tn1);
    const bool h2 = slab_near_far(XN.b0, YN.b0, ZN.b0, XF.b0, YF.b0, ZF.b0, t_lo, tcap, tn2);
    const bool h3 = slab_near_far(XN.b1, YN.b1, ZN.b1, XF.b1, YF.b1, ZF.b1, t_lo, tcap, tn3);
    // nearest hit child first (key unique per node: t_near bits with the slot number in the two low bits), the others are
    // stacked in slot order -- exactly the quad walk's order
    const uint32_t k0 = h0 ? ((__float_as_uint(tn0) & ~3u) | 0u) : 0xffffffffu, k1 = h1 ? ((__float_as_uint(tn1) & ~3u) | 1u) : 0xffffffffu;
    const uint32_t k2 = h2 ? ((__float_as_uint(tn2) & ~3u) | 2u) : 0xffffffffu, k3 = h3 ? ((__float_as_uint(tn3) & ~3u) | 3u) : 0xffffffffu;
    const uint32_t kmin = min(min(k0, k1), min(k2, k3));
    typedef int vi4 __attribute__((ext_vector_type(4)));
    vi4 RV = { (int)RF.x, (int)RF.y, (int)RF.z, (int)RF.w };
    asm volatile("" : "+v"(RV));                                     // (the child references are fetched WITH the boxes, not after the tests in a second round trip:
                                                                     //  with the references only for nodes that have a hit child 0.440 vs 0.429 ms per frame, round 3)
    const int r0 = RV.x, r1 = RV.y, r2 = RV.z, r3 = RV.w;
    if (kmin == 0xffffffffu) { lane_pop(S, cur, sp, sb); return; }
    const bool e0 = k0 == kmin, e1 = k1 == kmin, e2 = k2 == kmin, e3 = k3 == kmin;
    const bool p0 = h0 && !e0, p1 = h1 && !e1, p2 = h2 && !e2, p3 = h3 && !e3;
    if (__builtin_expect(__any(sp + 4 > LS::depth), 0)) {       // (some lane may leave the LDS part: the general form)
        if (p0) lane_push(S, sp, r0);
        if (p1) lane_push(S, sp, r1);
        if (p2) lane_push(S, sp, r2);
        if (p3) lane_push(S, sp, r3);
    } else {
        // four UNCONDITIONAL stores instead of four branches: a reference that is not kept is overwritten by the next one (its
        // offset does not advance), and the last lands above the new top of the stack (inside the lane's column: sp + 3 < 32)
        // (offsets as 0 / 1 counts shifted into the address -- v_lshl_add_u32 with inline constants --: with 0 / 256 the step also paid for the literal and for a shift of the sum)
        char *top = (char *)&S.lds[sp * 256 + S.tid];
        int c0 = p0 ? 1 : 0, c1 = p1 ? 1 : 0, c2 = p2 ? 1 : 0, c3 = p3 ? 1 : 0;
        asm("" : "+v"(c0), "+v"(c1), "+v"(c2));      // (opaque: seen through, every shifted count becomes a second select on a literal)
        char *t1 = top + (c0 << 10), *t2 = t1 + (c1 << 10), *t3 = t2 + (c2 << 10);
        *(int *)top = r0;
        *(int *)t1 = r1;
        *(int *)t2 = r2;
        *(int *)t3 = r3;
        sp += c0 + c1 + c2 + c3;
    }
    cur = e0 ? r0 : e1 ? r1 : e2 ? r2 : r3;      // (on the comparisons the pushes made already)
}

// one leaf: the contract's triangle test (btTriangleRaycastCallback::processTriangle behind the padded-bounds rule) on each of its
// triangles, then the next stack entry.  helper: the lane walks an adopted subtree (see k_trace_lane): a triangle at exactly the
// owner's closest fraction is a candidate.  Returns the number of triangles of the leaf.
template <class LS> MCRT_DEV uint32_t lane_leaf_test(const FrameArgs &a, const LS &S, f3 f2, f3 to, f3 inv, f3 rc, float t_lo, bool helper, Best &best, int &cur, int &sp, int sb)
{
    const uint32_t v = (uint32_t)~cur;
    const uint32_t first = v >> 3, cnt = (v & 7u) + 1u;
    for (uint32_t k = 0; k < cnt; k++) {
        const float4 *T = (const float4 *)((const char *)a.tris + (first + k) * (uint32_t)(16 * MCRT_TRI_PIECES));
        // the record's pieces are fetched TOGETHER, not stage by stage behind the early exits: a leaf phase then costs one
        // memory round trip (the pieces of a rejected triangle are wasted loads; staged: 0.349 against 0.343 ms per frame, round 4)
        typedef float vf4 __attribute__((ext_vector_type(4)));
        vf4 W0 = ((const vf4 *)T)[0], W1 = ((const vf4 *)T)[1], W2 = ((const vf4 *)T)[2];
        asm volatile("" : "+v"(W0), "+v"(W1), "+v"(W2));      // (pinned as three register tuples: pinned word by word the compiler copied seven of them out of the tuples first)
        const float4 V0 = make_float4(W0.x, W0.y, W0.z, W0.w), V1 = make_float4(W1.x, W1.y, W1.z, W1.w), V2 = make_float4(W2.x, W2.y, W2.z, W2.w);
        const float4 P = tri_plane(xyz(V0), xyz(V1), xyz(V2));
        const f3 nrm = xyz(P);
        const float da = dot(nrm, f2) - P.w;
        const float db = dot(nrm, to) - P.w;
        if (da * db >= 0.0f) continue;
        const int id = __float_as_int(V0.w);
        const float proj = da - db;
        const float frac = da / proj;
        if (!(frac < best.frac || (frac == best.frac && (id < best.tri || (helper && best.tri < 0)))) || !(frac >= t_lo)) continue;
        float tmin, tmax;
        f3 plo, phi;
        tri_padded_bounds(xyz(V0), xyz(V1), xyz(V2), a.pad_abs, plo, phi);
        if (!(slab_c(plo, phi, rc, inv, 0.0f, 1.0f, tmin, tmax) && frac >= tmin && frac <= tmax)) continue;
        const float edge_tol = V2.w;
        const float s = 1.0f - frac;
        const f3 p = mk(s * f2.x + frac * to.x, s * f2.y + frac * to.y, s * f2.z + frac * to.z);
        const f3 p0 = xyz(V0) - p, p1 = xyz(V1) - p, p2 = xyz(V2) - p;
        if (!(dot(cross(p0, p1), nrm) >= edge_tol)) continue;
        if (!(dot(cross(p1, p2), nrm) >= edge_tol)) continue;
        if (!(dot(cross(p2, p0), nrm) >= edge_tol)) continue;
        best.frac = frac; best.tri = id;
    }
    lane_pop(S, cur, sp, sb);
    return cnt;
}

// population count of a wave mask as a 32-bit SCALAR (a comparison of __popcll's 64-bit result is compiled to a vector instruction)
MCRT_DEV uint32_t popc_mask(unsigned long long m)
{
    uint32_t n = (uint32_t)__builtin_popcount((uint32_t)m) + (uint32_t)__builtin_popcount((uint32_t)(m >> 32));
    asm volatile("" : "+s"(n));
    return n;
}

// position of the r-th (0-based) set bit of a 64-bit mask (r < popcount): binary search on popcounts
MCRT_DEV int nth_set_bit(unsigned long long m, uint32_t r)
{
    int base = 0;
#pragma unroll
    for (int w = 32; w >= 1; w >>= 1) {
        const unsigned long long low = m & ((1ull << w) - 1ull);
        const uint32_t c = (uint32_t)__popcll(low);
        if (r >= c) { r -= c; m >>= w; base += w; } else m = low;
    }
    return base;
}

// x = taken ? nx : x, in place (the hand-over of a subtree replaces a lane's ray state: trace_lane_body)
MCRT_DEV void take_if(float &x, float nx, unsigned long long m) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(nx), "s"(m)); }
MCRT_DEV void take_if(int &x, int nx, unsigned long long m) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(nx), "s"(m)); }
MCRT_DEV void take_if(uint32_t &x, uint32_t nx, unsigned long long m) { asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(nx), "s"(m)); }

template <bool STATS, int STACK, bool DYN>
MCRT_DEV void trace_lane_body(const FrameArgs &a, const uint32_t b)
{
    // the traversal stacks in LDS, [STACK][256]: entry sp of thread t at sp*256 + t -> conflict-free (DYN: sized at the launch, see k_trace_lane_wide)
    extern __shared__ int stack_dyn[];
    __shared__ int stack_fix[DYN ? 1 : STACK * 256];
    int *const stack = DYN ? stack_dyn : stack_fix;
    const int tid = threadIdx.x, lane = tid & 63;
    // Bounce 0 is special: every sample path of a scan-line starts as a copy of the same first_ray (scene.cpp:83-101), so only
    // ONE ray per (frame, scan-line) is walked -- the first sample's -- and k_shade hands its hit to all S samples.
    const uint32_t n_rays = (b == 0u) ? a.ne : a.counts[b];
    // When a bounce has far fewer rays than the GPU has lanes, each ray is cut into K sub-ranges of its parameter interval inside
    // the scene bounds and the K pieces are walked by K different lanes: the launch then lasts as long as the longest PIECE
    // instead of the longest ray.  Sub-ranges are half-open and partition [0,1), and every find goes through the ray's atomicMin
    // word, so the result is exactly the single-walk answer.
    const uint32_t K = ksplit(n_rays, a.ksplit_limit);
    const uint32_t n = n_rays * K;
    const size_t st_half = (size_t)(b & 1u) * a.ne * a.S;      // path state and closest-hit words in queue order, ping-pong by bounce parity
    const float4 *st0 = a.st0 + st_half, *st1 = a.st1 + st_half;   // (the ray is rebuilt from origin | length factor and direction: ray_of)
    const uint32_t ray_stride = (b == 0u) ? a.S : 1u;          // bounce 0: the first sample of each queued scan-line stands for all
    unsigned long long *keys = (b & 1u) ? a.key1 : a.key0;
#define MCRT_KEYP(p) (&keys[p])
    unsigned long long st_nodes = 0, st_tris = 0, st_q = 0;
    // (overflow entries of this lane: [entry - MCRT_LANE_STACK][grid thread])
    const LaneStackT<STACK> S = { stack, a.stack_ovf + ((size_t)blockIdx.x * 256 + tid), (size_t)gridDim.x * 256, tid };

    // WORK DISTRIBUTION, XCD-aware.  Workgroups are dealt round-robin to the 8 XCDs (workgroup w runs on XCD w % 8), each with
    // its own L2.  The queue is cut into 8 contiguous sub-queues, one per XCD, each with its own cursor: an XCD sweeps ITS part
    // of the queue in order, so the rays in flight on it belong to a few scan-lines (small L2 working set), and the returning
    // atomics that hand out the work go to 8 addresses instead of one (same-address atomics serialise in L2 at ~6 ns each).  A
    // wavefront whose sub-queue has run dry moves on to the next one, so the XCDs finish together.  Bounces with few items use
    // one queue.  The kernel is PERSISTENT over the bounce's queue: a lane whose ray is finished writes its hit word and takes
    // the next unclaimed item (from a wave-private pool refilled with one atomic on its XCD's cursor).
    const uint32_t X = (n >= (uint32_t)MCRT_XCD_MIN_ITEMS) ? (uint32_t)MCRT_XCDS : 1u;
    if (X == 1u && blockIdx.x * 256u >= n) return;
    const uint32_t x_shift = (X == 1u) ? 0u : 3u;
    uint32_t cur_x = blockIdx.x & (X - 1u), visited = 0;
#define MCRT_SUB_LO(sq) ((uint32_t)(((unsigned long long)n * (sq)) >> x_shift))
#define MCRT_SUB_STATIC(sq) (((gridDim.x - (sq) + X - 1u) >> x_shift) * 256u)
    uint32_t *cursors = a.cursors + (size_t)b * MCRT_XCDS * MCRT_CURSOR_STRIDE;
    uint32_t i = MCRT_SUB_LO(cur_x) + (blockIdx.x >> x_shift) * 256u + (uint32_t)tid;      // the first item of each lane is assigned statically
    if (i >= MCRT_SUB_LO(cur_x + 1u)) i = 0xffffffffu;
    uint32_t ray_id = 0;                         // queue position of the ray (and of its closest-hit word)
    bool exhausted = false, fresh = true;
    f3 f2 = mk(0, 0, 0), to = mk(1, 1, 1), inv = mk(1, 1, 1);
    float t_lo = 0.0f;
    Best best; best.frac = 1.0f; best.tri = -1;
    int sp = 0, sb = 0, cur = CUR_IDLE;          // the lane's stack entries live in [sb, sp): sb moves up when the bottom entry is given away (see below)
    bool shared = false;                         // another lane of the wavefront works on a subtree of this lane's ray: results meet in the ray's word
    bool helper = false;                         // this lane walks an adopted subtree: it starts from the owner's closest fraction WITHOUT the owner's
                                                 // triangle, so a triangle at exactly that fraction is a candidate (the word's atomicMin applies the id rule)
#define MCRT_ON_INNER(c) __builtin_amdgcn_sicmp((c), -1, 38)
#define MCRT_ON_LEAF(c) __builtin_amdgcn_uicmp((uint32_t)(c), 0x80000000u, 34)
#define MCRT_WALKING(c) __builtin_amdgcn_sicmp((c), CUR_IDLE, 33)
    uint32_t pool_next = 0, pool_end = 0; bool queue_empty = false;   // wave-uniform
    const uint32_t fetch = n >= (uint32_t)MCRT_LANE_FETCH_FROM ? (uint32_t)MCRT_LANE_FETCH : (uint32_t)MCRT_LANE_FETCH_SMALL;
    unsigned long long poll_old = 0; uint32_t poll_ray = 0; bool poll_pending = false;      // (see the end of the loop)
    MCRT_WATCHDOG_DECL()
    for (;;) {
        MCRT_WATCHDOG_CHECK()
        // ---- finished rays report and idle lanes take new ones, once enough of them wait (the code runs for the whole wavefront) ----
        // (once the queue has run dry, finished lanes report at once: they are the helpers of the donation step below)
        const bool do_refill = popc_mask(__ballot(cur == CUR_IDLE && !exhausted)) >= (uint32_t)MCRT_LANE_REFILL || MCRT_WALKING(cur) == 0ull ||
                               (queue_empty && __any(cur == CUR_IDLE && !fresh));
        if (do_refill) {
            if (cur == CUR_IDLE && !fresh) {
                if (best.tri >= 0) {
                    const unsigned long long word = ((unsigned long long)__float_as_uint(best.frac) << 32) | (unsigned long long)(uint32_t)best.tri;
                    if (K == 1u && !shared) *MCRT_KEYP(ray_id) = word;   // the only walker of this ray: a plain store
                    else atomicMin(MCRT_KEYP(ray_id), word);
                }
                fresh = true; shared = false; helper = false; i = 0xffffffffu;
            }
            // (Publishing finished rays WHILE the launch runs -- so that k_shade could start on them in the launch's tail -- needs a device-scope
            //  release here: the XCDs' L2s are not coherent with each other inside a launch.  Measured, round 4: __threadfence() + one atomic per
            //  refill round make a launch of the 20-frame pass 3.11 ms instead of 0.67, of a 128-frame pass 13.6 instead of 3.08.  Not done.)
            const bool need = fresh && !exhausted;
            const unsigned long long dynm = __ballot(need && i == 0xffffffffu);
            if (dynm) {
                while (pool_next >= pool_end && !queue_empty) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(&cursors[(size_t)cur_x * MCRT_CURSOR_STRIDE], fetch);
                    base = __shfl(base, 0, 64);
                    const uint32_t hi = MCRT_SUB_LO(cur_x + 1u);
                    const unsigned long long start = (unsigned long long)MCRT_SUB_LO(cur_x) + MCRT_SUB_STATIC(cur_x) + base;
                    if (start < hi) {
                        pool_next = (uint32_t)start; pool_end = min((uint32_t)start + fetch, hi);
                    }
                    else if (++visited >= X) queue_empty = true;      // (the launch enters its TAIL: it only finishes the rays in flight from here on.  Round 4 let the
                                                                      //  accumulation's stream wait for this moment -- a device word + hipStreamWaitValue32 --: slower)
                    else cur_x = (cur_x + 1u) & (X - 1u);
                }
                if (need && i == 0xffffffffu) {
                    const uint32_t mine = pool_next + (uint32_t)__popcll(dynm & ((1ull << lane) - 1ull));
                    if (queue_empty) i = n;
                    else if (mine < pool_end) i = mine;
                }
                const uint32_t taken = (uint32_t)__popcll(dynm);
                pool_next = (pool_next + taken < pool_end) ? pool_next + taken : pool_end;
            }
            if (need && i != 0xffffffffu) {
                if (i < n) {
                    uint32_t piece = 0u;                                 // the pieces of one ray land in different wavefronts
                    if (K == 1u) ray_id = i;                             // (no division on the common path)
                    else { piece = i / n_rays; ray_id = i - piece * n_rays; }
                    const float4 s0 = st0[(size_t)ray_id * ray_stride], s1 = st1[(size_t)ray_id * ray_stride];
                    const Ray ry = ray_of(mk(s0.x, s0.y, s0.z), mk(s1.x, s1.y, s1.z), s0.w, a);
                    f2 = ry.f2; to = ry.to;
                    const f3 d = to - f2;
                    inv = mk(rcp_dir(d.x), rcp_dir(d.y), rcp_dir(d.z));
                    t_lo = 0.0f;
                    float t_hi = 1.0f;
                    if (K > 1u) {
                        float tin, tout;
                        if (slab(mk(a.scene_lo[0], a.scene_lo[1], a.scene_lo[2]), mk(a.scene_hi[0], a.scene_hi[1], a.scene_hi[2]), f2, inv, 0.0f, 1.0f, tin, tout)) {
                            const float w = tout - tin;
                            if (piece > 0u) t_lo = tin + w * ((float)piece / (float)K);
                            if (piece + 1u < K) t_hi = tin + w * ((float)(piece + 1u) / (float)K);
                        } else if (piece > 0u) t_hi = 0.0f;
                    }
                    best.frac = t_hi; best.tri = -1;
                    sp = 0; sb = 0; shared = false; helper = false; cur = (a.n_nodes != 0u && t_lo < t_hi) ? 0 : CUR_IDLE; fresh = false;
                    if (STATS && piece == 0u) st_q++;
                } else exhausted = true;
            }
        }
        // (This block stands BEFORE the test below for a reason of code generation only -- without walking lanes there are no donors --: behind it the compiler
        //  kept a second copy of the ray's state and moved 16 registers over at the top of every round and back at its end; here it needs 73 registers, not 80.)
        // ---- the END of a launch (and one frame at a time, where a bounce has fewer rays than the GPU has lanes): the queue is
        // empty, lanes run out of rays while a few long walks go on.  Idle lanes then TAKE OVER SUBTREES: the k-th idle lane adopts
        // the bottom stack entry (the farthest, usually largest pending subtree) of the k-th lane that has one, with a copy of its
        // ray and its current closest fraction, walks it on its own stack, and reports through the ray's closest-hit word, whose
        // atomicMin is exactly the contract's (smaller fraction, then smaller triangle id) rule -- the answer is the single walk's.
        // A launch then ends after its wavefronts' remaining WORK, not after their longest walk.  (Not in the counting build, whose
        // visit counts are those of one walk per ray.)
        // Measured and not kept: the same hand-over BETWEEN wavefronts through tickets and entries in global memory
        // (the heaviest wavefront's walks are chains with little to give away: its 230-odd node steps stayed, the pushes' round
        // trips were added); one ray per four lanes at the start of a small launch; rays dealt out across the wavefronts.
        if (!STATS && __builtin_amdgcn_readfirstlane((int)queue_empty)) {      // (a SCALAR branch: as a lane condition the compiler copied the whole ray state, 16 registers, at the top of every round)
            const bool thief = cur == CUR_IDLE && fresh;
            const bool donor = cur != CUR_IDLE && sp > sb && sb < STACK;
            const unsigned long long tm = __ballot(thief), dm = __ballot(donor);
            if (tm != 0ull && dm != 0ull) {
                const unsigned long long below = (1ull << lane) - 1ull;
                const uint32_t pairs = (uint32_t)min(__popcll(tm), __popcll(dm));
                const uint32_t trank = (uint32_t)__popcll(tm & below), drank = (uint32_t)__popcll(dm & below);
                const bool take = thief && trank < pairs, give = donor && drank < pairs;
                const int src = take ? nth_set_bit(dm, trank) : lane;       // (k_path posts the donors' lanes in LDS instead: worth 4 % there, nothing here -- the hand-over only runs in a launch's tail)
                const int d_sb = __shfl(sb, src, 64);
                const float c0 = __shfl(f2.x, src, 64), c1 = __shfl(f2.y, src, 64), c2 = __shfl(f2.z, src, 64);
                const float c3 = __shfl(to.x, src, 64), c4 = __shfl(to.y, src, 64), c5 = __shfl(to.z, src, 64);
                const float c6 = __shfl(inv.x, src, 64), c7 = __shfl(inv.y, src, 64), c8 = __shfl(inv.z, src, 64);
                const float c9 = __shfl(t_lo, src, 64), c10 = __shfl(best.frac, src, 64);
                const uint32_t c11 = (uint32_t)__shfl((int)ray_id, src, 64);
                const int c12 = __shfl(best.tri, src, 64), c13 = __shfl((int)helper, src, 64);
                // (the state is replaced IN PLACE, one select per register on the takers' mask: written as assignments under `if (take)` the compiler kept
                //  a second copy of the ray's state for the branch and moved 16 registers over at the top of EVERY round of the walk, and back at its end)
                const unsigned long long tk = __ballot(take);
                const int got = stack[take ? d_sb * 256 + (tid & ~63) + src : tid];      // the donor's bottom entry (same wavefront, read before the donor moves on)
                take_if(cur, got, tk);
                take_if(f2.x, c0, tk); take_if(f2.y, c1, tk); take_if(f2.z, c2, tk);
                take_if(to.x, c3, tk); take_if(to.y, c4, tk); take_if(to.z, c5, tk);
                take_if(inv.x, c6, tk); take_if(inv.y, c7, tk); take_if(inv.z, c8, tk);
                take_if(t_lo, c9, tk); take_if(best.frac, c10, tk); take_if(best.tri, -1, tk); take_if(ray_id, c11, tk);
                take_if(sp, 0, tk); take_if(sb, 0, tk);
                fresh = fresh && !take;
                helper = take ? (c12 >= 0 || c13 != 0) : helper;         // (an owner without a find so far passes on the ray's own bound, which stays exclusive;
                                                                         // a lane that is itself a helper passes its owner's fraction on)
                sb += give ? 1 : 0;
                shared = shared || take || give;
            }
        }

        // (ONE way round the loop: with a second back edge from here -- `continue` -- the compiler kept two copies of the ray's state, one across the
        //  refill and one across the walk, and moved 16 registers over at the top of every round and back at its end)
        if (MCRT_WALKING(cur) == 0ull) { if (!__any(!exhausted)) break; }
        else {

        // ---- phase 1: inner nodes, until enough lanes are parked on a leaf ----
        const float tcap = fminf(1.0f, best.frac);               // best only changes in phase 2
        // (then phase 1 is cut short: see MCRT_LANE_ADOPT_STEPS; held as scalars -- as a per-lane condition it made the whole loop a divergent one)
        const int thieves_wait = __builtin_amdgcn_readfirstlane((!STATS && queue_empty && __any(cur == CUR_IDLE && fresh)) ? 1 : 0);
        int steps_left = thieves_wait ? MCRT_LANE_ADOPT_STEPS : 0x7fffffff;      // (one counter, no second condition in the loop)
        const f3 rc = ray_c(f2, inv);
        const LaneRay lr = { rc.x, rc.y, rc.z, inv.x, inv.y, inv.z, inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f };
        for (;;) {
            const unsigned long long inner = MCRT_ON_INNER(cur);
            if (inner == 0ull) break;
            if (popc_mask(MCRT_ON_LEAF(cur)) >= (uint32_t)MCRT_LANE_LEAF_BATCH) break;     // (as 32-bit scalars: a 64-bit comparison is a vector instruction)
            if (--steps_left < 0) break;
            if (cur >= 0) {
                if (STATS) st_nodes++;
                lane_node_step(a, S, lr, t_lo, tcap, cur, sp, sb);
            }
        }
        // ---- phase 2: the parked leaves; the triangle test of the contract (btTriangleRaycastCallback::processTriangle behind
        // the padded-bounds rule), one lane per ray, same expressions as the quad walk's shared test ----
        if ((uint32_t)cur > 0x80000000u) {
            const uint32_t cnt = lane_leaf_test(a, S, f2, to, inv, rc, t_lo, helper, best, cur, sp, sb);
            if (STATS) st_tris += cnt;
        }

        // ---- walkers of ONE ray (the pieces of a cut ray, an owner and the lanes that took over its subtrees) meet in the ray's
        // closest-hit word: each publishes its find there and takes the smallest word back as its own closest hit, so a subtree or
        // piece behind another walker's hit is left as the single walk would leave it.  The word only ever holds real finds, and the
        // smallest of them is the answer, so cutting by it cannot cut the answer.  The returned word is looked at ONE round later
        // (its latency is then behind the node fetches of the round in between).
        if (!STATS && MCRT_LANE_POLL && (K > 1u || queue_empty)) {
            if (poll_pending) {
                poll_pending = false;
                const unsigned long long mine = ((unsigned long long)__float_as_uint(best.frac) << 32) | (unsigned long long)(uint32_t)best.tri;   // (no find: id 0xffffffff)
                if (poll_ray == ray_id && cur != CUR_IDLE && poll_old < mine) {
                    best.frac = __uint_as_float((uint32_t)(poll_old >> 32)); best.tri = (int)(uint32_t)poll_old; helper = false;
                }
            }
            if ((shared || K > 1u) && cur != CUR_IDLE) {
                const unsigned long long word = (best.tri >= 0) ? (((unsigned long long)__float_as_uint(best.frac) << 32) | (unsigned long long)(uint32_t)best.tri) : ~0ull;
                poll_old = atomicMin(MCRT_KEYP(ray_id), word); poll_ray = ray_id; poll_pending = true;
            }
        }
        }
    }
#undef MCRT_SUB_LO
#undef MCRT_SUB_STATIC
#undef MCRT_ON_INNER
#undef MCRT_ON_LEAF
#undef MCRT_WALKING
#undef MCRT_KEYP
    if (STATS) {
        unsigned long long v[3] = { st_q, st_nodes, st_tris };
#pragma unroll
        for (int k = 0; k < 3; k++) {
            long long x = wave_sum_i64((long long)v[k]);
            if (lane == 0 && x) atomicAdd(&a.stats[k], (unsigned long long)x);
        }
    }
}

// The walk comes as TWO kernels around one body.  k_trace_lane is compiled for 96 registers (budget 104): four of its wavefronts per SIMD (1024
// persistent workgroups) beside one k_march wavefront, nothing spilled -- the form for small launches, whose time is a chain of dependent
// steps.  k_trace_lane_wide is compiled for FIVE wavefronts per SIMD beside that k_march wavefront (5 x 80 + 80 registers; 1280 workgroups;
// MCRT_LANE_WIDE_STACK LDS stack entries so that five workgroups and k_march's LDS fit a CU): the compiler spills a dozen registers, all of them
// in the refill, hand-over and reporting code outside the node and leaf loops.  Sensitivity builds (profiles/round4/exp_sensitivity.txt) had shown the
// walk at the knee of its two pipes with four wavefronts to hide latency behind; the fifth is worth 3-4 % of a 128-frame pass (0.330 against
// 0.342 ms per frame; 1.8 % at 96 frames, 1.4 % at 48, 0.5 % at 32), costs a 20-frame pass 1 % and one frame at a time 6 % -- so launch_trace
// took the wide form from 4 Mi queued rays (32 frames of the headline workload) upwards through round 5.  Round 6 (kernels built without machine LICM: the
// wide form spills 20 bytes per lane instead of 48, and the small passes that the narrow form was kept for run as k_path): the wide form wins at EVERY staged
// pass size -- 5 / 6 / 8 / 12 / 20 frames: 0.678 / 0.614 / 0.512 / 0.426 / 0.359 against 0.699 / 0.628 / 0.530 / 0.446 / 0.381 ms per frame -- so it is the
// default from the first ray (MCRT_LANE_WIDE_FROM), for trees the caches hold (mcrt_api.cpp: fill_args); the narrow form stays for larger trees, CU-masked
// streams and the counting build.  (Its stack is sized at the launch: with a static LDS array the compiler caps the kernel's occupancy
// by LDS and hands the registers back.)  Late round 6: with the hand-over in front of the walking test (trace_lane_body) the body needs 67 registers and no scratch in
// either form; a SIXTH walk wavefront per SIMD then fits (MCRT_LANE_WIDE_WAVES 7, 1536 workgroups): 0.345-0.348 against 0.340 ms on the 20-frame pass, 0.295-0.298
// against 0.300 at 128 frames -- five stay.
#ifndef MCRT_LANE_WIDE_STACK
#define MCRT_LANE_WIDE_STACK 24          // (28: 0.332 against 0.3295 ms per frame; deeper walks go on in the overflow array, as in the other form)
#endif
#ifndef MCRT_LANE_WIDE_FROM
#define MCRT_LANE_WIDE_FROM 1u
#endif
template <bool STATS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(MCRT_LANE_VGPRS))) k_trace_lane(FrameArgs a, uint32_t b)
{
    trace_lane_body<STATS, MCRT_LANE_STACK, false>(a, b);
}
#ifndef MCRT_LANE_WIDE_WAVES
#define MCRT_LANE_WIDE_WAVES 6           // wavefronts per SIMD k_trace_lane_wide's registers are budgeted for: five of its own + one of k_march (512 / 6 -> 80 registers)
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MCRT_LANE_WIDE_WAVES, MCRT_LANE_WIDE_WAVES))) k_trace_lane_wide(FrameArgs a, uint32_t b)
{
    trace_lane_body<false, MCRT_LANE_WIDE_STACK, true>(a, b);
}

// =============================================================================================================
// k_trace_packet -- north_star's literal traversal: ONE WAVEFRONT PER RAY PACKET.  The 64 lanes of a wavefront hold 64 consecutive rays
// of the queue (neighbours: the sample paths of one scan-line with one reflect / refract history) and walk the BVH4 TOGETHER: one
// traversal stack for the wavefront (64 entries in ONE vector register, entry e in lane e), the current node wave-uniform and fetched
// through the SCALAR cache (one s_load_dwordx16 per node and wavefront instead of 64 lanes x four 16-byte pieces through the vector memory
// pipe -- the pipe that binds the lane walk), every lane tests the four child boxes against ITS ray with ITS closest fraction (the lane
// walk's arithmetic), a child is entered when ANY lane passes it, nearest first by the first passing lane's t_near; a leaf's triangles are
// fetched the same way and tested by every lane.  Legal under the contract: the closest hit (smaller fraction, then smaller triangle id, of
// the triangles whose padded bounds the ray passes) does not depend on the visiting order, boxes only cull, and a lane that does not pass a
// box passes nothing inside it -- so every lane gets exactly the lane walk's answer, bit for bit (the parity tests do not know which kernel ran).
// What it costs is counted in profiles/round5/packet_count_*.json: the packet visits the UNION of its rays' nodes -- 1.1 x the longest ray's at
// bounce 1, 1.7 x at bounce 2, 7 x at bounce 9 (a wavefront's 64 neighbours then belong to several histories) -- so launch_trace takes it only
// for the bounces named in FrameArgs::packet_mask.
// =============================================================================================================
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
MCRT_DEV u32x16 sload16(const void *p) { u32x16 r; asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory"); return r; }
MCRT_DEV u32x8 sload8(const void *p) { u32x8 r; asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory"); return r; }
// entry `l` (wave-uniform) of the wavefront's stack register becomes the wave-uniform value `v`: a vector compare and select (v_writelane_b32 wants the lane
// number in M0 and its moves on the scalar ALU, the pipe this kernel is short of)
MCRT_DEV int writelane(int v, int l, int old) { return (int)(threadIdx.x & 63u) == l ? v : old; }
MCRT_DEV u32x4 sload4(const void *p) { u32x4 r; asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory"); return r; }

// the packet's walk.  SGN: 0..7 = every ray of the packet runs the same way along every axis, bit 0 / 1 / 2 = towards -x / -y / -z (a bundle's rays differ by a
// fraction of a degree: the common case) -- which plane of a slab is the near one is then a COMPILE-TIME pick and the 24 plane distances read the node's words
// straight from scalar registers; 8 = mixed directions, picked per lane.  (The scalar ALU is this kernel's scarce pipe -- one per CU, and every step of every
// wavefront needs its ballots, keys and stack moves there: 12 selects per step are worth eight copies of the loop.)
template <int SGN>
MCRT_DEV void packet_walk(const FrameArgs &a, const LaneRay &lr, const f3 f2, const f3 to, const f3 inv, const f3 rc, Best &best)
{
    constexpr bool UNI = SGN < 8, NX = (SGN & 1) != 0, NY = (SGN & 2) != 0, NZ = (SGN & 4) != 0;
    int stk = 0;                                                          // the wavefront's traversal stack: entry e in lane e
    int sp = 0, cur = 0;                                                  // wave-uniform
    const unsigned long long wd_start = wall_clock64(); uint32_t wd_iter = 0;
    for (;;) {
        if ((++wd_iter & 4095u) == 0u && wall_clock64() - wd_start > (unsigned long long)MCRT_WATCHDOG_SECONDS * 100000000ull) { if ((threadIdx.x & 63) == 0) atomicOr(a.error_flag, 2u); break; }
        if (cur >= 0) {
            // the node: through the scalar cache, one load per wavefront
            // (words: lo.x[4] lo.y[4] | lo.z[4] hi.x[4] | hi.y[4] hi.z[4] | ref[4]; two halves per word)
            const u32x16 N = sload16((const char *)a.nodes_walk + ((size_t)(uint32_t)cur << 6));
            const uint32_t lox0 = N[0], lox1 = N[1], loy0 = N[2], loy1 = N[3], loz0 = N[4], loz1 = N[5], hix0 = N[6], hix1 = N[7], hiy0 = N[8], hiy1 = N[9], hiz0 = N[10], hiz1 = N[11];
            const float tcap = fminf(1.0f, best.frac);
            float tn0, tn1, tn2, tn3;
            bool h0, h1, h2, h3;
            if (UNI) {
                const Planes4 XN = planes4_s(NX ? hix0 : lox0, NX ? hix1 : lox1, lr.cx, lr.ix), XF = planes4_s(NX ? lox0 : hix0, NX ? lox1 : hix1, lr.cx, lr.ix);
                const Planes4 YN = planes4_s(NY ? hiy0 : loy0, NY ? hiy1 : loy1, lr.cy, lr.iy), YF = planes4_s(NY ? loy0 : hiy0, NY ? loy1 : hiy1, lr.cy, lr.iy);
                const Planes4 ZN = planes4_s(NZ ? hiz0 : loz0, NZ ? hiz1 : loz1, lr.cz, lr.iz), ZF = planes4_s(NZ ? loz0 : hiz0, NZ ? loz1 : hiz1, lr.cz, lr.iz);
                h0 = slab_near_far(XN.a0, YN.a0, ZN.a0, XF.a0, YF.a0, ZF.a0, 0.0f, tcap, tn0);
                h1 = slab_near_far(XN.a1, YN.a1, ZN.a1, XF.a1, YF.a1, ZF.a1, 0.0f, tcap, tn1);
                h2 = slab_near_far(XN.b0, YN.b0, ZN.b0, XF.b0, YF.b0, ZF.b0, 0.0f, tcap, tn2);
                h3 = slab_near_far(XN.b1, YN.b1, ZN.b1, XF.b1, YF.b1, ZF.b1, 0.0f, tcap, tn3);
            } else {
                const Planes4 XN = planes4(lr.nx ? hix0 : lox0, lr.nx ? hix1 : lox1, lr.cx, lr.ix), XF = planes4(lr.nx ? lox0 : hix0, lr.nx ? lox1 : hix1, lr.cx, lr.ix);
                const Planes4 YN = planes4(lr.ny ? hiy0 : loy0, lr.ny ? hiy1 : loy1, lr.cy, lr.iy), YF = planes4(lr.ny ? loy0 : hiy0, lr.ny ? loy1 : hiy1, lr.cy, lr.iy);
                const Planes4 ZN = planes4(lr.nz ? hiz0 : loz0, lr.nz ? hiz1 : loz1, lr.cz, lr.iz), ZF = planes4(lr.nz ? loz0 : hiz0, lr.nz ? loz1 : hiz1, lr.cz, lr.iz);
                h0 = slab_near_far(XN.a0, YN.a0, ZN.a0, XF.a0, YF.a0, ZF.a0, 0.0f, tcap, tn0);
                h1 = slab_near_far(XN.a1, YN.a1, ZN.a1, XF.a1, YF.a1, ZF.a1, 0.0f, tcap, tn1);
                h2 = slab_near_far(XN.b0, YN.b0, ZN.b0, XF.b0, YF.b0, ZF.b0, 0.0f, tcap, tn2);
                h3 = slab_near_far(XN.b1, YN.b1, ZN.b1, XF.b1, YF.b1, ZF.b1, 0.0f, tcap, tn3);
            }
            // WHICH children: any lane's.  In WHICH ORDER: nearest first by the t_near of the FIRST lane that passes each (one v_readlane per entered child; bits
            // order like the value, t_near >= 0; the slot in the two low bits makes the keys distinct).  Sorted pushes cost scalar work but save visits: with the
            // lane walk's rule instead (nearest first, the others in slot order: one v_readlane, no sort) the packet ran 1 % slower (profiles/round5/exp_packet.txt).
            const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
            const int r0 = (int)N[12], r1 = (int)N[13], r2 = (int)N[14], r3 = (int)N[15];
            const uint32_t nh = (m0 ? 1u : 0u) + (m1 ? 1u : 0u) + (m2 ? 1u : 0u) + (m3 ? 1u : 0u);
            if (nh == 1u) { cur = m0 ? r0 : m1 ? r1 : m2 ? r2 : r3; continue; }          // one child entered: no order to work out, nothing to stack
            if (nh >= 2u) {
                uint32_t k0 = 0xffffffffu, k1 = 0xffffffffu, k2 = 0xffffffffu, k3 = 0xffffffffu;
                if (m0) k0 = ((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tn0), __ffsll((long long)m0) - 1) & ~3u) | 0u;
                if (m1) k1 = ((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tn1), __ffsll((long long)m1) - 1) & ~3u) | 1u;
                if (m2) k2 = ((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tn2), __ffsll((long long)m2) - 1) & ~3u) | 2u;
                if (m3) k3 = ((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tn3), __ffsll((long long)m3) - 1) & ~3u) | 3u;
                // a sorting network on the four (key, reference) pairs: five compare-and-swaps
                int a0 = r0, a1 = r1, a2 = r2, a3 = r3;
#define MCRT_PK_CAS(ka, ra, kb, rb) { const bool sw_ = kb < ka; const uint32_t kl_ = sw_ ? kb : ka, kh_ = sw_ ? ka : kb; const int rl_ = sw_ ? rb : ra, rh_ = sw_ ? ra : rb; ka = kl_; kb = kh_; ra = rl_; rb = rh_; }
                MCRT_PK_CAS(k0, a0, k1, a1) MCRT_PK_CAS(k2, a2, k3, a3) MCRT_PK_CAS(k0, a0, k2, a2) MCRT_PK_CAS(k1, a1, k3, a3) MCRT_PK_CAS(k1, a1, k2, a2)
#undef MCRT_PK_CAS
                if (sp + nh - 1u > (uint32_t)MCRT_STACK) { if ((threadIdx.x & 63) == 0) atomicOr(a.error_flag, 1u); break; }      // nh - 1 entries go onto the 64-lane stack register; trees whose worst case needs more than MCRT_STACK are refused at upload, so this guards the register, never silently
                if (k3 != 0xffffffffu) { stk = writelane(a3, sp, stk); sp++; }     // farthest first: the nearest pops first
                if (k2 != 0xffffffffu) { stk = writelane(a2, sp, stk); sp++; }
                stk = writelane(a1, sp, stk); sp++;
                cur = a0;
                continue;
            }
        } else {
            const uint32_t v = (uint32_t)~cur;
            const uint32_t first = v >> 3, cnt = (v & 7u) + 1u;
            for (uint32_t k = 0; k < cnt; k++) {
                const char *T = (const char *)a.tris + (size_t)(first + k) * (16u * MCRT_TRI_PIECES);
                const u32x8 A = sload8(T); const u32x4 C2 = sload4(T + 32);
                const f3 v0 = mk(__uint_as_float(A[0]), __uint_as_float(A[1]), __uint_as_float(A[2])), v1 = mk(__uint_as_float(A[4]), __uint_as_float(A[5]), __uint_as_float(A[6]));
                const f3 v2 = mk(__uint_as_float(C2[0]), __uint_as_float(C2[1]), __uint_as_float(C2[2]));
                const int id = (int)A[3];
                const float edge_tol = __uint_as_float(C2[3]);
                const float4 P = tri_plane(v0, v1, v2);
                const f3 nrm = xyz(P);
                const float da = dot(nrm, f2) - P.w;
                const float db = dot(nrm, to) - P.w;
                bool ok = da * db < 0.0f;
                if (!__any(ok)) continue;
                const float proj = da - db;
                const float frac = da / proj;
                ok = ok && (frac < best.frac || (frac == best.frac && id < best.tri)) && frac >= 0.0f;
                if (!__any(ok)) continue;
                float tmin, tmax;
                f3 plo, phi;
                tri_padded_bounds(v0, v1, v2, a.pad_abs, plo, phi);
                ok = ok && slab_c(plo, phi, rc, inv, 0.0f, 1.0f, tmin, tmax) && frac >= tmin && frac <= tmax;
                if (!__any(ok)) continue;
                const float s = 1.0f - frac;
                const f3 p = mk(s * f2.x + frac * to.x, s * f2.y + frac * to.y, s * f2.z + frac * to.z);
                const f3 p0 = v0 - p, p1 = v1 - p, p2 = v2 - p;
                ok = ok && dot(cross(p0, p1), nrm) >= edge_tol && dot(cross(p1, p2), nrm) >= edge_tol && dot(cross(p2, p0), nrm) >= edge_tol;
                if (ok) { best.frac = frac; best.tri = id; }
            }
        }
        if (sp == 0) break;
        sp = __builtin_amdgcn_readfirstlane(sp - 1);
        cur = __builtin_amdgcn_readlane(stk, sp);
    }
}

__global__ void __launch_bounds__(64) k_trace_packet(FrameArgs a, uint32_t b)
{
    const uint32_t n_rays = a.counts[b];
    const uint32_t lane = threadIdx.x;
    const uint32_t base = blockIdx.x * 64u;
    if (base >= n_rays || a.n_nodes == 0u) return;
    const uint32_t i = base + lane;
    const bool live = i < n_rays;
    const uint32_t ray_id = live ? i : n_rays - 1u;
    const size_t st_half = (size_t)(b & 1u) * a.ne * a.S;
    const float4 s0 = a.st0[st_half + ray_id], s1 = a.st1[st_half + ray_id];
    const Ray ry = ray_of(mk(s0.x, s0.y, s0.z), mk(s1.x, s1.y, s1.z), s0.w, a);
    const f3 f2 = ry.f2, to = ry.to;
    const f3 d = to - f2;
    const f3 inv = mk(rcp_dir(d.x), rcp_dir(d.y), rcp_dir(d.z));
    const f3 rc = ray_c(f2, inv);
    const LaneRay lr = { rc.x, rc.y, rc.z, inv.x, inv.y, inv.z, inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f };
    Best best; best.frac = live ? 1.0f : -1.0f; best.tri = -1;            // (a lane beyond the queue passes no box: its closest fraction is negative)
    // do all the packet's rays run the same way along every axis?  (lane 0 is live: base < n_rays)
    const int sgn = (lr.nx ? 1 : 0) | (lr.ny ? 2 : 0) | (lr.nz ? 4 : 0);
    const int sgn0 = __builtin_amdgcn_readfirstlane(sgn);
    const int which = __all(!live || sgn == sgn0) ? sgn0 : 8;               // wave-uniform
    switch (which) {
    case 0: packet_walk<0>(a, lr, f2, to, inv, rc, best); break;
    case 1: packet_walk<1>(a, lr, f2, to, inv, rc, best); break;
    case 2: packet_walk<2>(a, lr, f2, to, inv, rc, best); break;
    case 3: packet_walk<3>(a, lr, f2, to, inv, rc, best); break;
    case 4: packet_walk<4>(a, lr, f2, to, inv, rc, best); break;
    case 5: packet_walk<5>(a, lr, f2, to, inv, rc, best); break;
    case 6: packet_walk<6>(a, lr, f2, to, inv, rc, best); break;
    case 7: packet_walk<7>(a, lr, f2, to, inv, rc, best); break;
    default: packet_walk<8>(a, lr, f2, to, inv, rc, best); break;
    }
    if (live && best.tri >= 0) {
        unsigned long long *keys = (b & 1u) ? a.key1 : a.key0;
        keys[i] = ((unsigned long long)__float_as_uint(best.frac) << 32) | (unsigned long long)(uint32_t)best.tri;
    }
}

// ---- interface interaction (scene.cpp:122-165, ray.cpp:11-97) of ONE path at bounce b, given its ray and the closest-hit
// word of the walk: thickness draw, travel, hit_boundary, the segment's records, the continuing ray's state.  Returns whether
// the path goes on.
struct PathState { f3 from, dir; float intensity; int media, outside; double dist_mm; };

// the scene's material and mesh tables as k_shade reads them: its LDS copies when they fit (MCRT_SHADE_TABLE rows each), else memory.
// (`lds` is wave-uniform: a scalar branch picks the load, so the LDS side compiles to ds_read, not to a flat load)
struct ShadeTables {
    const float4 *mats_g; const uint4 *meshes_g; const float4 *mats_l; const uint4 *meshes_l; bool lds;
    // (the empty asm keeps the two sides different instructions: otherwise the compiler merges them into ONE load through a selected flat pointer)
    MCRT_DEV float4 mat(uint32_t r) const { float4 v; if (lds) { v = mats_l[r]; asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); } else v = mats_g[r]; return v; }
    MCRT_DEV uint4 mesh(uint32_t r) const { uint4 v; if (lds) { v = meshes_l[r]; asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); } else v = meshes_g[r]; return v; }
};
template <bool STATS>
MCRT_DEV bool shade_path(const FrameArgs &a, const ShadeTables &tb, uint32_t b, uint32_t pid, PathState &ps, f3 f2, f3 to, unsigned long long key, bool &reflected,
                         unsigned long long &st_seg, unsigned long long &st_hits)
{
    bool alive = false;
    f3 from = ps.from, dir = ps.dir;
    float intensity = ps.intensity; int media = ps.media, outside = ps.outside; double dist_mm = ps.dist_mm;
    reflected = false;
    {
        Hit best; best.frac = __uint_as_float((uint32_t)(key >> 32)); best.tri = (int)(uint32_t)key; best.da = 0.0f; best.mesh = 0; best.n = mk(0, 0, 0);
        if (best.tri >= 0) {
            // plane normal, mesh and the origin-side value of the winning triangle, as the walk's test evaluated them
            const float4 *T = a.tris_id + MCRT_TRI_PIECES * (size_t)best.tri;      // (the id-ordered copy: through tri_slot the fetch was a chain of two dependent random reads)
            const float4 t2 = T[1];                                     // (v1, mesh)
            const float4 P = tri_plane(xyz(T[0]), xyz(t2), xyz(T[2]));
            best.n = xyz(P);
            best.da = dot(best.n, f2) - P.w;
            best.mesh = __float_as_int(t2.w);
        }
        const uint32_t line = pid / a.S, fr = line / a.ne_frame;      // (two divisions; the remainders by multiply-subtract)
        const uint32_t e_abs = a.e_begin + (line - fr * a.ne_frame);
        Rng g; g.k0 = a.seed; g.k1 = a.frame + fr; g.element = e_abs; g.sample = pid - line * a.S; g.bounce = b;
        const float4 m0 = tb.mat(2 * media);   // imp, att, mu0, mu1  (second half: sigma, spec, shine, thick)
        const float att = m0.y;

        f3 seg_to = to;
        float seg_refl = 0.0f; const float seg_init = intensity; const double seg_dist = dist_mm;
        const f3 seg_from = from, seg_dir = dir; const int seg_media = media; int seg_tri = -1;
        if (best.tri >= 0) {
            if (STATS) st_hits++;
            f3 nn = normalized(best.n);
            if (best.da <= 0.0f) nn = neg(nn);
            const float sfr = 1.0f - best.frac;
            const f3 hp = mk(sfr * f2.x + best.frac * to.x, sfr * f2.y + best.frac * to.y, sfr * f2.z + best.frac * to.z);
            const uint4 organ = tb.mesh(best.mesh);   // mat_inside, mat_outside, vascular
            // thickness penetration scene.cpp:132-139 (Box-Muller on block 0)
            const float sigma_t = tb.mat(2 * organ.x + 1).w;
            float qpen = 0.0f;
            if (sigma_t != 0.0f) {
                double n1, n2, sn, cs;
                rng_block(g, 0u, n1, n2);
                det_sincos(n2 * 2 * PI_D, sn, cs);
                const double z = sqrt(-2.0 * det_log(1.0 - n1)) * cs;
                qpen = (float)fabs(z * (double)sigma_t + 0.0);
            }
            const f3 inside = mk(qpen * dir.x + hp.x, qpen * dir.y + hp.y, qpen * dir.z + hp.z);
            // travel ray.cpp:99-103, distance_in_mm scene.cpp:281-290
            const float xd = fabsf(from.x - inside.x) * a.sx, yd = fabsf(from.y - inside.y) * a.sy, zd = fabsf(from.z - inside.z) * a.sz;
            const double mm = sqrt((double)xd * (double)xd + (double)yd * (double)yd + (double)zd * (double)zd) * 10;
            dist_mm = dist_mm + mm;
            intensity = intensity * det_expf(-att * ((float)mm * 0.01f) * a.freq);

            // hit_boundary: material transition logic ray.cpp:14-47 (bug-compatible)
            int after_vasc, mat_after;
            if (outside != OUT_NONE) {
                if (organ.z) { after_vasc = OUT_NONE; mat_after = (outside == OUT_SELF) ? media : outside; }
                else { after_vasc = (outside == (int)organ.x) ? (int)organ.y : (int)organ.x; mat_after = media; }
            } else {
                if (organ.z) { after_vasc = OUT_SELF; mat_after = (int)organ.x; }
                else { after_vasc = OUT_NONE; mat_after = (int)organ.x; }
            }
            const float4 a0 = tb.mat(2 * mat_after), a1 = tb.mat(2 * mat_after + 1);
            double u_pc, u_x;
            rng_block(g, 1u, u_pc, u_x);
            // power_cosine_variate ray.cpp:213-224
            const int indice = (int)a1.z + 1;
            const float exponente = (float)((double)1.0 / indice);
            const float random_angle = (float)det_pow_pos(u_pc, (double)exponente);
            const f3 rn = random_unit_vector(nn, random_angle, g);

            float inc = dot(dir, neg(rn));
            if (inc < 0) inc = dot(dir, rn);
            const float rr = m0.x / a0.x;
            float refa = 1 - rr * rr * (1 - inc * inc);
            const bool tir = refa < 0;
            refa = sqrtf(refa);
            const float kk = rr * inc - refa;
            f3 refr = mk(rr * dir.x + kk * rn.x, rr * dir.y + kk * rn.y, rr * dir.z + kk * rn.z);
            refr = normalized(refr);
            const float two_c = 2 * inc;
            f3 refl = mk(dir.x + two_c * rn.x, dir.y + two_c * rn.y, dir.z + two_c * rn.z);
            refl = normalized(refl);

            float i_refl;
            if (tir) i_refl = intensity;
            else {
                const float num = m0.x * inc - a0.x * refa;
                const float den = m0.x * inc + a0.x * refa;
                const float qq = num / den;
                i_refl = (float)((double)intensity * ((double)qq * (double)qq));
            }
            const float i_refr = intensity - i_refl;

            const float ra = dot(dir, refr);
            float refraction_factor = det_powf(ra, a1.y);
            const float rb = dot(dir, refl);
            const float reflection_factor = det_powf(rb, a1.y);
            if (a.sanitize && tir) refraction_factor = 0.0f;
            seg_refl = (std_max(refraction_factor, 0.0f) + std_max(reflection_factor, 0.0f)) * random_angle;
            seg_to = inside;
            seg_tri = best.tri;

            const float x = (float)u_x;
            const float prob = i_refl / intensity;
            float i_new;
            from = hp;
            if (prob > x) { dir = refl; i_new = i_refl > a.eps ? i_refl : 0.0f; reflected = true; }
            else { dir = refr; media = mat_after; outside = after_vasc; i_new = i_refr > a.eps ? i_refr : 0.0f; }
            if (i_new > a.eps) { intensity = i_new; alive = true; }
        }
        if (STATS) st_seg++;

        // what the accumulation loop needs of this segment (main.cpp:112-121), computed once here by one lane instead of by
        // every lane of k_march's quad: start time, step count, the per-step advance
        {
            const f3 df = seg_to - seg_from;
            const float dist_f = sqrtf(dot(df, df)) * 10.0f;
            const uint32_t steps = steps_from((double)dist_f / a.axial_res_mm);
            const double t_start = (seg_dist * 1000.0) / a.sos_d;
            float4 *mr = a.mrec + 3 * ((size_t)b * a.ne * a.S + pid);    // [bounce][path]: neighbouring paths are neighbours in memory
            mr[0] = make_float4(seg_from.x, seg_from.y, seg_from.z, seg_refl);
            mr[1] = make_float4(a.axial_res_f * seg_dir.x, a.axial_res_f * seg_dir.y, a.axial_res_f * seg_dir.z, seg_init);
            mr[2] = make_float4(__int_as_float(__double2loint(t_start)), __int_as_float(__double2hiint(t_start)), __uint_as_float(steps), __int_as_float(seg_media));
        }
        // ray_physics::segment (ray.h:28-36) -> slot [path][bounce], for the callers that ask for the segments themselves
        if (a.want_segs) {
            mcrt_segment sg;
            sg.from[0] = seg_from.x; sg.from[1] = seg_from.y; sg.from[2] = seg_from.z;
            sg.to[0] = seg_to.x; sg.to[1] = seg_to.y; sg.to[2] = seg_to.z;
            sg.dir[0] = seg_dir.x; sg.dir[1] = seg_dir.y; sg.dir[2] = seg_dir.z;
            sg.reflected_intensity = seg_refl; sg.initial_intensity = seg_init; sg.attenuation = att;
            sg.distance_traveled = seg_dist; sg.media = seg_media; sg.tri = seg_tri;
            a.segs[(size_t)pid * a.B + b] = sg;
        }
        if (a.hits) a.hits[(size_t)pid * a.B + b] = seg_tri;
        a.seg_count[pid] = b + 1u;
        alive = alive && (b + 1u < a.B);
    }
    ps.from = from; ps.dir = dir; ps.intensity = intensity; ps.media = media; ps.outside = outside; ps.dist_mm = dist_mm;
    return alive;
}


// ---- interface interaction of a bounce's live rays: one lane per ray ----
template <bool STATS>
__global__ void __launch_bounds__(256, MCRT_SHADE_WAVES) k_shade(FrameArgs a, uint32_t b)
{
    const uint32_t n = a.counts[b];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x * blockDim.x >= n) return;
    const int lane = threadIdx.x & 63;
    // the scene's material and mesh tables in LDS when they fit (they nearly always do: the reference's scenes have 9 materials and <= 11 meshes):
    // a ray looks up five material rows and one mesh row -- a quarter of this kernel's cache accesses, and the frame is bound by their sum
    __shared__ float4 mats_l[2 * MCRT_SHADE_TABLE];
    __shared__ uint4 meshes_l[MCRT_SHADE_TABLE];
    const bool tables_in_lds = a.n_mat <= (uint32_t)MCRT_SHADE_TABLE && a.n_mesh <= (uint32_t)MCRT_SHADE_TABLE;
    if (tables_in_lds) {
        for (uint32_t r = threadIdx.x; r < 2u * a.n_mat; r += blockDim.x) mats_l[r] = a.mats[r];
        for (uint32_t r = threadIdx.x; r < a.n_mesh; r += blockDim.x) meshes_l[r] = a.meshes[r];
        __syncthreads();
    }
    const ShadeTables tb = { a.mats, a.meshes, mats_l, meshes_l, tables_in_lds };
    // two queue buffers, ping-pong by bounce parity (like the path state)
    const uint32_t *q_in = a.queue + (size_t)(b & 1u) * a.ne * a.S;
    uint32_t *q_out = a.queue + (size_t)((b + 1u) & 1u) * a.ne * a.S;
    const bool valid = i < n;
    bool alive = false, reflected = false;
    uint32_t pid = 0;
    PathState ps; ps.from = mk(0, 0, 0); ps.dir = mk(0, 0, 1); ps.intensity = 0.0f; ps.media = 0; ps.outside = OUT_NONE; ps.dist_mm = 0.0;
    unsigned long long st_seg = 0, st_hits = 0;
    if (valid) {
        pid = q_in[i];
        // path state lives in queue order (ping-pong halves by bounce parity), so a wavefront reads and writes it coalesced
        const size_t sin = (size_t)(b & 1u) * a.ne * a.S + (b == 0u ? MCRT_STATE0_AT(i, a.S) : i);
        const float4 s0 = a.st0[sin], s1 = a.st1[sin], s2 = a.st2[sin];
        ps.from = mk(s0.x, s0.y, s0.z); ps.intensity = s2.w;
        ps.dir = mk(s1.x, s1.y, s1.z); ps.media = __float_as_int(s1.w);
        ps.dist_mm = __hiloint2double(__float_as_int(s2.y), __float_as_int(s2.x));
        ps.outside = __float_as_int(s2.z);
        const Ray ry = ray_of(ps.from, ps.dir, s0.w, a);          // the segment the walk tested (same expressions, same bits)
        const f3 f2 = ry.f2, to = ry.to;
        const size_t hi = (b == 0u) ? (size_t)(i / a.S) : (size_t)i;            // bounce 0: one walk per queued (scan-line, frame) (see k_trace_lane, k_init)
        const unsigned long long key = ((b & 1u) ? a.key1 : a.key0)[hi];
        alive = shade_path<STATS>(a, tb, b, pid, ps, f2, to, key, reflected, st_seg, st_hits);
    }
    const f3 from = ps.from, dir = ps.dir; const float intensity = ps.intensity; const int media = ps.media, outside = ps.outside; const double dist_mm = ps.dist_mm;

    // survivors -> next bounce's queue (ballot + prefix; ONE atomic per workgroup: tens of thousands of returning atomics on the
    // single counter would serialise in L2 and bound the kernel).  Inside a workgroup's block the reflected rays
    // come first, then the refracted ones, each in queue order: the samples of a scan-line that took the same decisions stay
    // adjacent, so the rays of a k_trace_lane wavefront mostly belong to a few tight bundles (same nodes, similar walk length).
    __shared__ uint32_t wave_live[4], wave_refl[4], block_base;
    const unsigned long long live = __ballot(alive);
    const unsigned long long live_refl = __ballot(alive && reflected);
    const int wv = threadIdx.x >> 6;
    if (lane == 0) { wave_live[wv] = (uint32_t)__popcll(live); wave_refl[wv] = (uint32_t)__popcll(live_refl); }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t total = wave_live[0] + wave_live[1] + wave_live[2] + wave_live[3];
        block_base = total ? atomicAdd(&a.counts[b + 1u], total) : 0u;
    }
    __syncthreads();
    if (live) {
        // (round 5: reflected-first over the WORKGROUP's 256 rays, not per wavefront -- three of the next bounce's four 64-ray blocks are then of one history,
        //  which is what a ray packet wants, k_trace_packet)
        const uint32_t refl_all = wave_refl[0] + wave_refl[1] + wave_refl[2] + wave_refl[3];
        uint32_t refl_before = 0, refr_before = 0;
        for (int w = 0; w < wv; w++) { refl_before += wave_refl[w]; refr_before += wave_live[w] - wave_refl[w]; }
        if (alive) {
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t pos = block_base + (reflected ? refl_before + (uint32_t)__popcll(live_refl & below)
                                                         : refl_all + refr_before + (uint32_t)__popcll(live & ~live_refl & below));
            q_out[pos] = pid;
            ((b & 1u) ? a.key0 : a.key1)[pos] = MCRT_KEY_MISS;            // the next bounce's closest-hit word of this ray
            const size_t so = (size_t)((b + 1u) & 1u) * a.ne * a.S + pos;
            a.st0[so] = make_float4(from.x, from.y, from.z, ray_len(intensity, tb.mat(2 * media).y, a));   // origin | the next ray's length factor
            a.st1[so] = make_float4(dir.x, dir.y, dir.z, __int_as_float(media));
            a.st2[so] = make_float4(__int_as_float(__double2loint(dist_mm)), __int_as_float(__double2hiint(dist_mm)), __int_as_float(outside), intensity);
        }
    }
    if (STATS) {
        long long x = wave_sum_i64((long long)st_seg), y = wave_sum_i64((long long)st_hits);
        if (lane == 0) { if (x) atomicAdd(&a.stats[3], (unsigned long long)x); if (y) atomicAdd(&a.stats[5], (unsigned long long)y); }
    }
}

// =============================================================================================================
// k_path -- the LATENCY form of cast_rays (scene.cpp:50-183), for passes that cannot fill the GPU: ONE launch carries every path through ALL of its
// bounces.  The reference traces one frame at a time (main.cpp:92-152); at 128 x 1024 paths that is 2048 wavefronts on 1024 SIMDs, and the staged
// pipeline above spends it on ten dependent walk -> shade launch pairs, each as long as its slowest wavefront plus two launch gaps (round 5: 1.48 ms
// per frame against 0.32 ms of throughput).  Here a lane OWNS a path: walk (the lane walk's own node and leaf steps), shade_path, next bounce -- no queue, no
// compaction, no grid-wide dependency; a wavefront goes on to its next bounce as soon as ITS 64 walks are done.  Lanes whose path has died (and lanes that
// finish a walk early) are what the throughput form would compact away; here the GPU is under-filled anyway, so they HELP: they adopt the bottom stack
// entries of the lanes still walking (the same hand-over as at the end of a k_trace_lane launch) and report through the owner's closest-hit word, which
// lives in LDS (the walkers of a ray are lanes of one wavefront).  Bit-exact by the contract's order-independence: the word's minimum is the single walk's
// answer.  Round 2's fused kernel lost in THROUGHPUT mode (fp64 physics at a quarter of the lanes, 173 registers: DESIGN.md A.4); that argument does not
// hold where registers are free.  k_march(MCRT_ALL_BOUNCES) accumulates the segments afterwards.
// =============================================================================================================
#ifndef MCRT_PATH_WAVES
#define MCRT_PATH_WAVES 4            // wavefronts per SIMD k_path's registers are budgeted for (109 registers without machine LICM, see the Makefile)
#endif
#ifndef MCRT_PATH_ADOPT_STEPS
#define MCRT_PATH_ADOPT_STEPS 1       // node steps between two hand-overs while idle lanes wait (k_trace_lane: 4; here, at 32 owners per wavefront, 1 / 2 / 3: 0.946 / 0.953 / 0.980 ms per frame)
#endif
#ifndef MCRT_PATH_LEAF_BATCH
#define MCRT_PATH_LEAF_BATCH MCRT_LANE_LEAF_BATCH        // lanes parked on a leaf that end the inner-node phase (8 / 20 / 32: 1.20 / 1.14 / 1.19 ms per frame)
#endif
#ifndef MCRT_PATH_LEAF_GATE
#define MCRT_PATH_LEAF_GATE 16       // the triangle tests of an iteration wait until this many lanes are parked on a leaf (or no lane has an inner node left): none / 4 / 8 / 16: 0.913 / 0.916 / 0.907 / 0.900 ms per frame
#endif
static_assert(MCRT_PATH_LEAF_GATE <= MCRT_PATH_LEAF_BATCH, "k_path: with more parked lanes than MCRT_PATH_LEAF_BATCH the inner-node phase stops stepping, so the leaf phase must have started by then");
#ifndef MCRT_PATH_OWNERS
#define MCRT_PATH_OWNERS 32          // paths per wavefront: the first MCRT_PATH_OWNERS lanes own one each, the others only ever help -- one 128 x 1024 frame is then 4096
                                     // wavefronts = four per SIMD, each walk shared by twice the lanes (64 owners at two per SIMD: 0.88 ms per launch; 32 at four: 0.70)
#endif
__global__ void __launch_bounds__(256, MCRT_PATH_WAVES) k_path(FrameArgs a)
{
    __shared__ int stack[MCRT_LANE_STACK * 256];
    __shared__ unsigned long long wbest[256];         // closest-hit word of the ray the lane OWNS in this bounce; helpers (lanes of the same wavefront) publish here
    __shared__ unsigned char donor_of[256];           // hand-over: the k-th donor of a wavefront posts its lane here, the k-th idle lane reads it (instead of a 6-round search of the donors' ballot)
    __shared__ float4 mats_l[2 * MCRT_SHADE_TABLE];
    __shared__ uint4 meshes_l[MCRT_SHADE_TABLE];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t np = a.ne * a.S, pos = (blockIdx.x * 4u + (uint32_t)(tid >> 6)) * (uint32_t)MCRT_PATH_OWNERS + (uint32_t)lane;
    const bool tables_in_lds = a.n_mat <= (uint32_t)MCRT_SHADE_TABLE && a.n_mesh <= (uint32_t)MCRT_SHADE_TABLE;
    if (tables_in_lds) {
        for (uint32_t r = tid; r < 2u * a.n_mat; r += blockDim.x) mats_l[r] = a.mats[r];
        for (uint32_t r = tid; r < a.n_mesh; r += blockDim.x) meshes_l[r] = a.meshes[r];
        __syncthreads();
    }
    const ShadeTables tb = { a.mats, a.meshes, mats_l, meshes_l, tables_in_lds };
    const LaneStackT<MCRT_LANE_STACK> S = { stack, a.stack_ovf + ((size_t)blockIdx.x * 256 + tid), (size_t)gridDim.x * 256, tid };
    // the path of this lane, as k_init left it (queue position -> path id, state of bounce 0)
    bool alive = lane < MCRT_PATH_OWNERS && pos < np;
    uint32_t pid = 0; float Ls = 0.0f;
    PathState ps; ps.from = mk(0, 0, 0); ps.dir = mk(0, 0, 1); ps.intensity = 0.0f; ps.media = 0; ps.outside = OUT_NONE; ps.dist_mm = 0.0;
    if (alive) {
        pid = a.queue[pos];
        const uint32_t p0 = MCRT_STATE0_AT(pos, a.S);
        const float4 s0 = a.st0[p0], s1 = a.st1[p0], s2 = a.st2[p0];
        ps.from = mk(s0.x, s0.y, s0.z); Ls = s0.w; ps.dir = mk(s1.x, s1.y, s1.z); ps.media = __float_as_int(s1.w);
        ps.dist_mm = __hiloint2double(__float_as_int(s2.y), __float_as_int(s2.x)); ps.outside = __float_as_int(s2.z); ps.intensity = s2.w;
    }
#define MCRT_ON_INNER(c) __builtin_amdgcn_sicmp((c), -1, 38)
#define MCRT_ON_LEAF(c) __builtin_amdgcn_uicmp((uint32_t)(c), 0x80000000u, 34)
#define MCRT_WALKING(c) __builtin_amdgcn_sicmp((c), CUR_IDLE, 33)
#define MCRT_WORD(bst) (((unsigned long long)__float_as_uint((bst).frac) << 32) | (unsigned long long)(uint32_t)(bst).tri)
    MCRT_WATCHDOG_DECL()
    bool abandoned = false;
    for (uint32_t b = 0; b < a.B && !abandoned; b++) {
        if (!__any(alive)) break;
        // ---- the walk: every live lane starts its own ray at the root; the others start as helpers-in-waiting ----
        f3 f2 = mk(0, 0, 0), to = mk(1, 1, 1), inv = mk(1, 1, 1);
        float t_lo = 0.0f;
        Best best; best.frac = 1.0f; best.tri = -1;
        int sp = 0, sb = 0, cur = CUR_IDLE, owner = tid;      // owner: the lane (index in the workgroup) whose ray this lane is walking
        bool fresh = true, shared = false, helper = false;
        unsigned long long poll_old = 0; int poll_owner = -1; bool poll_pending = false;
        wbest[tid] = MCRT_KEY_MISS;
        // Bounce 0: every sample path of a scan-line starts as a copy of the same first_ray (scene.cpp:83-101), so ONE lane per scan-line of the wavefront
        // walks it -- its first -- and the others start as its helpers (k_trace_lane walks one ray per scan-line at bounce 0 for the same reason); each lane
        // then takes its leader's word.  (64 lanes walking the same ray in lockstep cost a tenth of the launch: 140 of 1480 k cycles per wavefront.)
        const bool leads = b != 0u || lane == 0 || pos % a.S == 0u;
        const unsigned long long lead_mask = __ballot(alive && leads);
        if (alive && leads) {
            const Ray ry = ray_of(ps.from, ps.dir, Ls, a);
            f2 = ry.f2; to = ry.to;
            const f3 d = to - f2;
            inv = mk(rcp_dir(d.x), rcp_dir(d.y), rcp_dir(d.z));
            cur = a.n_nodes != 0u ? 0 : CUR_IDLE; fresh = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        for (;;) {
            { if ((++wd_iter & 4095u) == 0u && wall_clock64() - wd_start > (unsigned long long)MCRT_WATCHDOG_SECONDS * 100000000ull) { if (lane == 0) atomicOr(a.error_flag, 2u); abandoned = true; break; } }
            // a lane that has finished a walk -- its own ray's or an adopted subtree -- reports to the ray's word and is free to help
            if (cur == CUR_IDLE && !fresh) {
                if (best.tri >= 0) atomicMin(&wbest[owner], MCRT_WORD(best));
                fresh = true; shared = false; helper = false;
            }
            if (MCRT_WALKING(cur) == 0ull) break;
            // ---- idle lanes take over subtrees (see k_trace_lane: the same hand-over, from the first step on) ----
            {
                const bool thief = cur == CUR_IDLE && fresh;
                const bool donor = cur != CUR_IDLE && sp > sb && sb < MCRT_LANE_STACK;
                const unsigned long long tm = __ballot(thief), dm = __ballot(donor);
                if (tm != 0ull && dm != 0ull) {
                    const unsigned long long below = (1ull << lane) - 1ull;
                    const uint32_t pairs = (uint32_t)min(__popcll(tm), __popcll(dm));
                    const uint32_t trank = (uint32_t)__popcll(tm & below), drank = (uint32_t)__popcll(dm & below);
                    const bool take = thief && trank < pairs, give = donor && drank < pairs;
                    if (give) donor_of[(tid & ~63) + (int)drank] = (unsigned char)lane;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                    const int src = take ? (int)donor_of[(tid & ~63) + (int)trank] : lane;
                    const int d_sb = __shfl(sb, src, 64);
                    const float c0 = __shfl(f2.x, src, 64), c1 = __shfl(f2.y, src, 64), c2 = __shfl(f2.z, src, 64);
                    const float c3 = __shfl(to.x, src, 64), c4 = __shfl(to.y, src, 64), c5 = __shfl(to.z, src, 64);
                    const float c6 = __shfl(inv.x, src, 64), c7 = __shfl(inv.y, src, 64), c8 = __shfl(inv.z, src, 64);
                    const float c10 = __shfl(best.frac, src, 64);
                    const int c11 = __shfl(owner, src, 64);
                    const int c12 = __shfl(best.tri, src, 64), c13 = __shfl((int)helper, src, 64);
                    if (take) {
                        cur = stack[d_sb * 256 + (tid & ~63) + src];        // the donor's bottom entry (same wavefront, read before the donor moves on)
                        f2 = mk(c0, c1, c2); to = mk(c3, c4, c5); inv = mk(c6, c7, c8);
                        best.frac = c10; best.tri = -1; owner = c11;
                        sp = 0; sb = 0; fresh = false; shared = true;
                        helper = c12 >= 0 || c13 != 0;
                    }
                    if (give) { sb++; shared = true; }      // (trace_lane_body's in-place selects before the exit test, tried here: 0.826-0.829 against 0.809 ms per frame -- the hand-over runs every iteration here)
                }
            }
            // ---- phase 1: inner nodes, until enough lanes are parked on a leaf (cut short while idle lanes wait for a subtree) ----
            const float tcap = fminf(1.0f, best.frac);
            const int thieves_wait = __builtin_amdgcn_readfirstlane(__any(cur == CUR_IDLE && fresh) ? 1 : 0);
            int steps_left = thieves_wait ? MCRT_PATH_ADOPT_STEPS : 0x7fffffff;
            const f3 rc = ray_c(f2, inv);
            const LaneRay lr = { rc.x, rc.y, rc.z, inv.x, inv.y, inv.z, inv.x < 0.0f, inv.y < 0.0f, inv.z < 0.0f };
            for (;;) {
                const unsigned long long inner = MCRT_ON_INNER(cur);
                if (inner == 0ull) break;
                if (popc_mask(MCRT_ON_LEAF(cur)) >= (uint32_t)MCRT_PATH_LEAF_BATCH) break;
                if (--steps_left < 0) break;
                if (cur >= 0) lane_node_step(a, S, lr, t_lo, tcap, cur, sp, sb);
            }
            // ---- phase 2: the parked leaves ----
            // (the triangle tests wait until MCRT_PATH_LEAF_GATE lanes are parked on a leaf, or no lane has an inner node left: with a hand-over every node step the leaf
            //  phase -- the whole wavefront executes it -- would otherwise run in nearly every iteration for a lane or two)
            if (popc_mask(MCRT_ON_LEAF(cur)) >= (uint32_t)MCRT_PATH_LEAF_GATE || MCRT_ON_INNER(cur) == 0ull)
            if ((uint32_t)cur > 0x80000000u) lane_leaf_test(a, S, f2, to, inv, rc, t_lo, helper, best, cur, sp, sb);
            // ---- walkers of one ray meet in its word: publish the find, take the smallest word back one round later ----
            if (poll_pending) {
                poll_pending = false;
                const unsigned long long mine = MCRT_WORD(best);      // (no find: id 0xffffffff)
                if (poll_owner == owner && cur != CUR_IDLE && poll_old < mine) {
                    best.frac = __uint_as_float((uint32_t)(poll_old >> 32)); best.tri = (int)(uint32_t)poll_old; helper = false;
                }
            }
            if (shared && cur != CUR_IDLE) {
                const unsigned long long word = (best.tri >= 0) ? MCRT_WORD(best) : ~0ull;
                poll_old = atomicMin(&wbest[owner], word); poll_owner = owner; poll_pending = true;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (abandoned) break;
        // ---- the interface: thickness draw, travel, hit_boundary, the segment's records (k_shade's own code) ----
        if (alive) {
            const int leader = 63 - __clzll((long long)(lead_mask & ((2ull << lane) - 1ull)));        // (bounce 0: the nearest leading lane at or below this one; later: the lane itself)
            const unsigned long long key = wbest[(tid & ~63) + leader];
            const Ray ry = ray_of(ps.from, ps.dir, Ls, a);            // the segment the walk tested (same expressions, same bits)
            bool reflected = false; unsigned long long st_seg = 0, st_hits = 0;
            alive = shade_path<false>(a, tb, b, pid, ps, ry.f2, ry.to, key, reflected, st_seg, st_hits);
            if (alive) Ls = ray_len(ps.intensity, tb.mat(2 * ps.media).y, a);
        }
    }
#undef MCRT_ON_INNER
#undef MCRT_ON_LEAF
#undef MCRT_WALKING
#undef MCRT_WORD
}

// ---- RF accumulation (main.cpp:112-140) of the segments produced in bounce b.  A workgroup owns a range of the sample
// slots of ONE scan-line ("line" = frame * ne_frame + scan-line), so its fixed-point bins live in LDS and are flushed once
// with global integer atomics.  Inside it every wavefront runs its slots as a task pool: a group of G lanes (template parameter: 2 or 4) per
// segment; a group that has finished (or found a dead path's empty slot) takes the next slot, so short, long and missing
// segments do not wait for each other.  Eight consecutive steps per iteration, lane j of the group owns steps j, j+G, ...
// (that many texture gathers per lane in flight).
// Tried on top of this and measured slower (MI355X, 32 frames per pass, per launch alone): one task pool per workgroup instead of
// a quarter of its slots per wavefront (an LDS cursor: 891 vs 829 us -- the extra scalar work outweighs the better balance); a
// one-read row stepper for steps after a segment's first (the row advances by G or G + 1: no gain, the two threshold reads were
// never the cost); fewer resident workgroups per CU so that k_shade / the next k_trace find registers at once (LDS padding: no gain);
// one wavefront per scan-line, four lines and four bin arrays per workgroup (a pool of S slots per wavefront instead of S/4: the GPU is
// then a quarter as finely cut and the heaviest lines set the pace -- 1714 vs 827 us).
// With the sorted tiles (750 us): the steps' echoes as straight-line code (row confirmed by two reads, zero adds into a spare bin, the
// rest left to a general path entered when any lane needs it) -- the loop issues two scalar instructions for three vector ones, but
// the straight line keeps four echoes and rows alive: 20 spilled registers at 6 waves/SIMD (1255 us), 784 us at 5; the zero adds
// all meet in one LDS word (869 / 757 us).  Round 3, the same idea with the add itself the only masked instruction (common case = estimate
// confirmed and echo small; the rest collected in a bit mask for a general path entered when any lane needs it): 2341 vs 2263 us per
// 128-frame launch -- the work done for steps that are not valid costs more than the branches it replaces.  Taking parts of the step out (wrong images, timing only): no gathers 693, no row
// search 679, no adds 707, none of the three 570 us, no loop at all 8 us (cycle stamps of the full kernel: hand-out 17 %, advance 4 %, voxel + gathers 27 %,
// rows and bins with the wait for the gathers 51 %).
// With the fast path (1397 us per 128-frame launch alone): one lane per segment instead of a pair (four steps per lane and iteration, no
// redundant advance): 1420 us; the row guessed from the lane's previous row + G instead of from its time (two double operations less per
// step): 1426 us.
// FAST (round 3): the reference's 256^3 texture with the branch-free cell, and an LDS image padded to the largest row guess of a
// valid step -- no texture size, shift, row count or LDS base in the step's instructions (the generic kernel had spilled those
// scalars: ~13 v_readlane per four steps).
template <bool STATS, int G, bool FAST>
__global__ void __launch_bounds__(256, MCRT_MARCH_WAVES) k_march(FrameArgs a, uint32_t b, uint32_t chunks)
{
    constexpr int H = 8 / G;                 // RF steps per lane and iteration: a group does G*H = 8 consecutive steps
#ifndef MCRT_MARCH_REFILL_DIV
#define MCRT_MARCH_REFILL_DIV 4
#endif
    constexpr int REFILL = (64 / G) / MCRT_MARCH_REFILL_DIV;   // new segments are handed out while at least a quarter of the wavefront's groups are idle or finished
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = tid >> 6, j = tid & (G - 1);
    const uint32_t R = a.R, nf = (R + 31u) >> 5, nrt = FAST ? a.march_rows : R + 1u;      // entries of the LDS image (march_lds_bytes)
    RowBin *rb = (RowBin *)smem;
    uint32_t *lflags = (uint32_t *)(rb + nrt);
    uint32_t *sort_cnt = lflags + ((nf + 3u) & ~3u) + wv * 64;                              // this wavefront's 64 length classes ...
    unsigned char *sort_list = (unsigned char *)(lflags + ((nf + 3u) & ~3u) + 4 * 64) + wv * MCRT_MARCH_TILE;   // ... its tile's slots, longest first ...
    unsigned char *sort_cls = (unsigned char *)(lflags + ((nf + 3u) & ~3u) + 4 * 64) + 4 * MCRT_MARCH_TILE + wv * MCRT_MARCH_TILE;   // ... and their classes (worked out once)
    // the per-material table (a few 16-byte rows) in LDS: the tile sort and every segment load look it up -- as reads of the vector memory pipe
    // they were a tenth of this kernel's cache accesses, and the frame is bound by the sum of its kernels' accesses
    float4 *mtab_l = (float4 *)((unsigned char *)(lflags + ((nf + 3u) & ~3u) + 4 * 64) + 8 * MCRT_MARCH_TILE);
    const bool mtab_in_lds = MCRT_MARCH_LDS_TABLES && a.n_mat <= (uint32_t)MCRT_MARCH_MTAB;
    for (uint32_t r = tid; r < nrt; r += nthr) { rb[r].thr = r <= R ? a.row_thr[r] : -__builtin_inf(); rb[r].bin = 0; }
    for (uint32_t r = tid; r < nf; r += nthr) lflags[r] = 0u;
    if (mtab_in_lds) for (uint32_t r = tid; r < a.n_mat; r += nthr) mtab_l[r] = a.mtab[r];
    __syncthreads();
    // (a scalar branch picks the load; the empty asm keeps the LDS side a ds_read -- merged, the compiler emits ONE flat load through a selected pointer)
    auto mtab_row = [&](int m) -> float4 { float4 v; if (mtab_in_lds) { v = mtab_l[m]; asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); } else v = a.mtab[m]; return v; };
#define MCRT_MTAB(m) mtab_row(m)

    // XCD-aware numbering: workgroup w runs on XCD w % 8; give every XCD a CONTIGUOUS range of scan-lines, so that the texture
    // cells its segments touch (neighbouring scan-lines cross the same tissue) are shared in ITS L2
    uint32_t bid = blockIdx.x;
    if (gridDim.x % MCRT_XCDS == 0u) bid = (blockIdx.x % MCRT_XCDS) * (gridDim.x / MCRT_XCDS) + blockIdx.x / MCRT_XCDS;
    // ... and within it the F frames of a scan-line one after the other (they cross exactly the same tissue)
    const uint32_t F = a.ne / a.ne_frame, ol = bid / chunks, chunk = bid % chunks;
    const uint32_t line = (ol % F) * a.ne_frame + ol / F;
    // this wavefront's slot range: the line's S slots are cut into chunks*4 contiguous pieces
    const uint32_t per = (a.S + chunks * 4u - 1u) / (chunks * 4u);
    const uint32_t s_begin = min(a.S, (chunk * 4u + (uint32_t)wv) * per), s_end = min(a.S, s_begin + per);
    const size_t pid0 = (size_t)line * a.S;
    unsigned long long st_steps = 0;

    // The wavefront takes its slots in TILES of MCRT_MARCH_TILE, and the segments of a tile LONGEST FIRST (counting sort by the
    // number of 8-step iterations a segment needs, dead paths' slots left out): groups that start together then finish together,
    // so the hand-out code below -- which the whole wavefront executes -- runs for many groups at once and seldom, and the lanes
    // of a wavefront step in lockstep.  (RF bins are integer sums: the order is free.)
    uint32_t tile0 = s_begin, list_base = s_begin, list_n = 0, list_pos = 0;      // wave-uniform: first slot of the next tile; of the tile in hand: first slot, live segments, the next one to hand out
    bool tiles_left = s_begin < s_end;
    const double thr_end = a.row_thr[R];
    bool busy = false;
    // A GROUP of G lanes (a DPP quad, or half of one) owns a segment.  Lane j of the group carries the segment's running
    // state (point, time, intensity) j steps AHEAD of the group's base step: every lane does the same sequential updates the
    // reference does, shifted, and owns steps j, j+G, j+2G, ...
    f3 point = mk(0, 0, 0), delta = mk(0, 0, 0);
    double t = 0.0, t_start = 0.0;
    float inten = 0.0f, k_att = 0.0f, seg_refl = 0.0f, m_dens = 0.0f, m_sigma = 0.0f, m_mu = 0.0f;
    uint32_t sidx = 0, steps = 0;
    bool more = false;
    // b == MCRT_ALL_BOUNCES: the launch accumulates EVERY bounce's segments; a group then walks its path's segments one after
    // the other (seg_b = the one in progress, seg_n = how many the path has) before it takes the next slot
    const bool all_b = b == MCRT_ALL_BOUNCES;
    uint32_t seg_b = all_b ? 0u : b, seg_n = 0; size_t seg_pid = 0;
    const float rcp_v = vgpr(a.tex_rcp), res_v = vgpr(a.tex_res);      // (vector-register copies: see vox_cell_lean256_v)
#define MCRT_LOAD_SEGMENT() { \
        const float4 *mr = a.mrec + 3 * ((size_t)seg_b * a.ne * a.S + seg_pid); \
        const float4 g0 = mr[0], g1 = mr[1], g2 = mr[2]; \
        const float4 mt = MCRT_MTAB(__float_as_int(g2.w)); \
        point = mk(g0.x, g0.y, g0.z); seg_refl = g0.w; \
        delta = mk(g1.x, g1.y, g1.z); inten = g1.w; \
        t_start = __hiloint2double(__float_as_int(g2.y), __float_as_int(g2.x)); \
        steps = __float_as_uint(g2.z); \
        m_mu = mt.x; m_dens = mt.y; m_sigma = mt.z; k_att = mt.w; \
        t = t_start; sidx = (uint32_t)j; \
        /* scattering is exactly +0 for every voxel when mu0 == sigma == 0 (finite texture): the adds are no-ops */ \
        const bool silent = a.tex_finite && m_mu == 0.0f && m_sigma == 0.0f; \
        more = !silent && steps > 0u && t < a.max_travel; \
        _Pragma("unroll") for (int u = 1; u < G; u++) if (j >= u) MCRT_ADVANCE() \
        busy = true; }
#define MCRT_ADVANCE() { point = point + delta; t = t + a.time_step; inten *= k_att; }
    for (;;) {
        // ---- finished segments and idle quads.  The boundary echo of a finished segment (main.cpp:139) and the probing of new
        // slots are code the whole wavefront runs however few quads need it, so both wait until REFILL quads are
        // finished or idle (or nothing is left to step) ----
        const bool fin = busy && !more;
        if (popc_mask(__ballot((!busy || fin) && j == 0)) >= (uint32_t)REFILL || !__any(busy && more)) {
            if (fin) {
                if (j == 0) {
                    const double te = t_start + a.time_step * (double)(uint32_t)(steps - 1u);
                    rf_add(rb, lflags, row_of(te, rb, R, a.inv_row_dt, thr_end), seg_refl / (float)a.S);
                }
                busy = false;
                if (all_b && seg_b + 1u < seg_n) { seg_b++; MCRT_LOAD_SEGMENT() }      // the path's next segment
            }
            while (list_pos < list_n || tiles_left) {
                if (list_pos >= list_n) {
                    // ---- the next tile: classes 0 (longest) .. 63, a slot's class from its segment's step count ----
                    const uint32_t t1 = min(s_end, tile0 + (uint32_t)MCRT_MARCH_TILE);
                    sort_cnt[lane] = 0u;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
                    // (the class of a slot is worked out once, while counting, and kept in LDS for the placing pass: worked out twice -- rounds 2-3 --
                    //  it cost the placing pass the same three global reads per slot again)
                    auto slot_class = [&](uint32_t slot) -> uint32_t {
                        if (slot >= t1) return 0xffffffffu;
                        const uint32_t sn = a.seg_count[pid0 + slot], sb0 = all_b ? 0u : b;
                        if (sb0 >= sn) return 0xffffffffu;
                        uint32_t its = 0u;
                        if (!all_b) {
                            const float4 g2 = a.mrec[3 * ((size_t)sb0 * a.ne * a.S + pid0 + slot) + 2];
                            const float4 mt = MCRT_MTAB(__float_as_int(g2.w));
                            const bool silent = a.tex_finite && mt.x == 0.0f && mt.z == 0.0f;
                            its = silent ? 0u : (__float_as_uint(g2.z) + 7u) >> 3;
                        }
                        return 63u - (its < 63u ? its : 63u);
                    };
                    for (int k = 0; k < MCRT_MARCH_TILE / 64; k++) {
                        const uint32_t c = slot_class(tile0 + (uint32_t)(k * 64 + lane));
                        sort_cls[k * 64 + lane] = (unsigned char)c;                      // (0xff: no live segment in this slot)
                        if (c != 0xffffffffu) atomicAdd(&sort_cnt[c], 1u);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                    const uint32_t mine_cnt = sort_cnt[lane];
                    uint32_t incl = mine_cnt;                                    // inclusive prefix sum over the 64 classes
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                    sort_cnt[lane] = incl - mine_cnt;                            // now the class's next free position in the list
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                    for (int k = 0; k < MCRT_MARCH_TILE / 64; k++) {
                        const uint32_t c = MCRT_MARCH_LDS_TABLES ? (uint32_t)sort_cls[k * 64 + lane] : (slot_class(tile0 + (uint32_t)(k * 64 + lane)) & 0xffu);
                        if (c != 0xffu) sort_list[atomicAdd(&sort_cnt[c], 1u)] = (unsigned char)(k * 64 + lane);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
                    list_n = (uint32_t)__shfl((int)incl, 63, 64); list_pos = 0u;
                    list_base = tile0; tile0 = t1; tiles_left = t1 < s_end;
                    continue;
                }
                const unsigned long long want = __ballot(!busy && j == 0);
                if (popc_mask(want) < (uint32_t)REFILL) break;
                const uint32_t mine = list_pos + (uint32_t)__popcll(want & ((1ull << (lane & ~(G - 1))) - 1ull));
                if (!busy && mine < list_n) {
                    seg_pid = pid0 + list_base + (uint32_t)sort_list[mine];
                    seg_b = all_b ? 0u : b;
                    seg_n = all_b ? a.seg_count[seg_pid] : b + 1u;           // (the list holds live slots only: no need to ask again)
                    if (seg_b < seg_n) MCRT_LOAD_SEGMENT()
                }
                const uint32_t nw = (uint32_t)__popcll(want);
                list_pos = (list_pos + nw < list_n) ? list_pos + nw : list_n;
            }
        }
        if (!__any(busy)) { if (list_pos >= list_n && !tiles_left) break; else continue; }

        // ---- G*H steps of every running segment ----
        if (busy && more) {
            f3 myp[H]; double myt[H]; float myin[H]; bool myv[H];
            float reach = 0.0f;
#pragma unroll
            for (int h = 0; h < H; h++) {
                myp[h] = point; myt[h] = t; myin[h] = inten; myv[h] = sidx < steps && t < a.max_travel;           // the reference's loop test
                if (h == 0 || h == H - 1) reach += abs_sum(point);   // every coordinate moves monotonically: first and last bound them all
#pragma unroll
                for (int u = 0; u < G; u++) MCRT_ADVANCE()
                sidx += (uint32_t)G;
            }
            // the quad goes on while its base step (lane 0's) passes the loop test
            more = dpp_i<G == 4 ? QP_BCAST(0) : 0xA0>((sidx < steps && t < a.max_travel) ? 1 : 0) != 0;   // (0xA0: quad_perm [0,0,2,2])
            float2 vox[H];
            if (reach < a.lean_bound) {
#pragma unroll
                for (int h = 0; h < H; h++) vox[h] = a.tex[FAST ? vox_cell_lean256_v(myp[h], rcp_v, res_v) : vox_cell_lean(myp[h], a)];
            } else {
#pragma unroll
                for (int h = 0; h < H; h++) vox[h] = myv[h] ? a.tex[vox_cell(myp[h], a)] : make_float2(0.0f, 0.0f);
            }
            // the steps' rows while the gathers are in flight (LDS reads do not wait for them, and the times are dead afterwards:
            // 1407 -> 1382 us per 128-frame launch against looking each row up just before its add); a step's row is guessed from its
            // time, which misses only by a rounding -- the lane's previous row + its stride misses whenever the row advances by one more
            int rows[H];
#pragma unroll
            for (int h = 0; h < H; h++)
                rows[h] = myv[h] ? row_near<FAST>(myt[h], (int)(myt[h] * a.inv_row_dt), rb, R, a.inv_row_dt, thr_end) : -1;
#pragma unroll
            for (int h = 0; h < H; h++) {
                if (myv[h]) {
                    const float scattering = vox[h].y >= m_dens ? vox[h].x * m_sigma + m_mu : 0.0f;
                    rf_add(rb, lflags, rows[h], myin[h] * scattering);
                    if (STATS) st_steps++;
                }
            }
        }
    }
#undef MCRT_ADVANCE
#undef MCRT_LOAD_SEGMENT
#undef MCRT_MTAB
    if (STATS) {
        long long x = wave_sum_i64((long long)st_steps);
        if (lane == 0 && x) atomicAdd(&a.stats[4], (unsigned long long)x);
    }
    __syncthreads();
    // row of the frame's RF block: [frame][scan-line of the whole block]; this launch covers scan-lines [acc_off, acc_off+ne_frame)
    const size_t row = (size_t)(line / a.ne_frame) * a.acc_stride + a.acc_off + line % a.ne_frame;
    for (uint32_t r = tid; r < R; r += nthr) {
        const long long v = rb[r].bin;
        if (v != 0) atomicAdd((unsigned long long *)&a.acc[row * R + r], (unsigned long long)v);
    }
    for (uint32_t r = tid; r < nf; r += nthr) { const uint32_t f = lflags[r]; if (f) atomicOr(&a.flags[row * nf + r], f); }
}

// fixed-point bins -> float RF image [ne][R]; clears the bins for the next frame.  A frame whose launches set the context's device
// error word (a persistent kernel abandoned by its watchdog, a traversal stack that ran out) is written as NaN throughout: a caller that
// synchronises on its own stream and never asks mcrt_synchronize cannot mistake it for an image.
__global__ void k_finalize(long long *acc, uint32_t *flags, float *rf, uint32_t ne, uint32_t R, const uint32_t *error_flag)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ne * R) return;
    const uint32_t e = (uint32_t)(i / R), r = (uint32_t)(i % R);
    const uint32_t nf = (R + 31u) >> 5;
    const bool bad = ((flags[(size_t)e * nf + (r >> 5)] >> (r & 31)) & 1u) || *error_flag != 0u;
    const long long v = acc[i];
    rf[i] = bad ? __uint_as_float(0x7fc00000u) : (float)((double)v * 0x1p-40);
    acc[i] = 0;
}
__global__ void k_clear_flags(uint32_t *flags, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = 0u;
}

// rfimage.h:96-108 on the scan-line-major image: tmp[e][row] = sum_k img[e][row+k]*ax[k], row in [na, R-na)
// (both passes take a stack of n_img images [n_img][E][R] at once: one launch for all the frames of a pass)
__global__ void k_conv_axial(const float *img, float *tmp, uint32_t n_img, uint32_t E, uint32_t R, ConvTaps taps)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_img * E * R) return;
    const int row = (int)(i % R), na = (int)taps.n_ax;
    if (row < na || row >= (int)R - na) return;
    float conv = 0;
    for (int k = 0; k < na; k++) conv += img[i + k] * taps.ax[k];
    tmp[i] = conv;
}
// rfimage.h:111-122: img[col][row] = sum_k tmp[col+k][row]*lat[k], row in [na,R-na), col in [nl/2, E-nl)
__global__ void k_conv_lateral(const float *tmp, float *img, uint32_t n_img, uint32_t E, uint32_t R, ConvTaps taps)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_img * E * R) return;
    const int row = (int)(i % R), col = (int)((i / R) % E), na = (int)taps.n_ax, nl = (int)taps.n_lat;
    if (row < na || row >= (int)R - na) return;
    if (col < nl / 2 || col >= (int)E - nl) return;
    float conv = 0;
    for (int k = 0; k < nl; k++) conv += tmp[i + (size_t)k * R] * taps.lat[k];
    img[i] = conv;
}

// rfimage.h:54-91, one WAVEFRONT per scan-line.  The reference walks a column once: whenever the signal stops ascending at row i
// (a concave peak), the rows [last peak, i) are overwritten with the line from |last peak| to |c[i]|.  The comparisons only ever
// read rows the walk has not overwritten yet, so the peaks are a pure function of the input column:
//     peak(i) = (c[i-1] < c[i]) && !(c[i] < c[i+1]),  1 <= i <= R-2      (`ascending` after step j is exactly c[j] < c[j+1])
// and row j becomes  last*(1-alpha) + next*alpha  with last / next the peaks around it (prev <= j < next; before the first peak
// `last` is the signed c[0] at row 0, rfimage.h:64), rows after the last peak stay.  Same float expressions as the sequential loop,
// evaluated by 64 lanes from an LDS copy of the column: previous / next peak by a wave-wide max / min scan over lane-contiguous chunks.
// (One lane per column, the round-2 kernel, is a chain of 465 dependent global loads: 650 us per call however few the columns.)
__global__ void __launch_bounds__(64) k_envelope(float *img, uint32_t E, uint32_t R)
{
    __shared__ float col[MCRT_MAX_ROWS];
    __shared__ unsigned short prv[MCRT_MAX_ROWS], nxt[MCRT_MAX_ROWS];
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x >= E || R < 2) return;
    float *c = img + (size_t)blockIdx.x * R;
    for (uint32_t r = lane; r < R; r += 64u) col[r] = c[r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    const uint32_t per = (R + 63u) / 64u, r0 = min(R, lane * per), r1 = min(R, r0 + per);
    constexpr uint32_t NONE = 0xffffu;
    auto peak = [&](uint32_t i) { return i >= 1u && i + 1u < R && (col[i - 1u] < col[i]) && !(col[i] < col[i + 1u]); };
    // last peak at or before each row (0 = the start of the column), first peak after it
    uint32_t last = 0u, first = NONE;
    for (uint32_t i = r0; i < r1; i++) if (peak(i)) { last = i; if (first == NONE) first = i; }
    uint32_t before = last, after = first;                    // inclusive scans over the lanes' chunks ...
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)before, d, 64), dn = (uint32_t)__shfl_down((int)after, d, 64);
        if ((int)lane >= d) before = max(before, up);
        if ((int)lane + d < 64) after = min(after, dn);
    }
    uint32_t run_prev = (uint32_t)__shfl_up((int)before, 1, 64), run_next = (uint32_t)__shfl_down((int)after, 1, 64);      // ... made exclusive
    if (lane == 0u) run_prev = 0u;
    if (lane == 63u) run_next = NONE;
    for (uint32_t i = r0; i < r1; i++) { if (peak(i)) run_prev = i; prv[i] = (unsigned short)run_prev; }
    for (uint32_t i = r1; i > r0; i--) { nxt[i - 1u] = (unsigned short)run_next; if (peak(i - 1u)) run_next = i - 1u; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    for (uint32_t j = lane; j < R; j += 64u) {
        const uint32_t p = prv[j], q = nxt[j];
        if (q == NONE) continue;                               // past the last peak: untouched
        const float last_peak = p == 0u ? col[0] : fabsf(col[p]), new_peak = fabsf(col[q]);
        const float alpha = ((float)j - (float)p) / ((float)q - (float)p);
        c[j] = last_peak * (1 - alpha) + new_peak * alpha;
    }
}

// cv::remap(src, dst, map_y, map_x, INTER_LINEAR, BORDER_CONSTANT 0) with precomputed maps (rfimage.h:139):
// mx = column coordinate (scan-line), my = row coordinate.  src is [E][R] scan-line-major.
__global__ void k_remap(const float *img, uint32_t E, uint32_t R, const float *map_col, const float *map_row, float *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    img += (size_t)blockIdx.y * E * R; out += (size_t)blockIdx.y * n;          // image blockIdx.y of a stack [n_img][E][R] -> [n_img][n]
    const float mx = map_col[i], my = map_row[i];
    const float fx = floorf(mx), fy = floorf(my);
    const float ax = mx - fx, ay = my - fy;
    const long long x0 = (long long)fx, y0 = (long long)fy;
    float v[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; dy++)
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
            const long long xx = x0 + dx, yy = y0 + dy;
            const bool in = (mx == mx) && (my == my) && xx >= 0 && yy >= 0 && xx < (long long)E && yy < (long long)R;
            v[dy][dx] = in ? img[(size_t)xx * R + (size_t)yy] : 0.0f;
        }
    const float top = v[0][0] * (1.0f - ax) + v[0][1] * ax;
    const float bot = v[1][0] * (1.0f - ax) + v[1][1] * ax;
    out[i] = top * (1.0f - ay) + bot * ay;
}

// [E][R] -> [R][E]
__global__ void k_transpose(const float *in, float *out, uint32_t E, uint32_t R)
{
    __shared__ float tile[32][33];
    const uint32_t r0 = blockIdx.x * 32, e0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const uint32_t e = e0 + j, r = r0 + threadIdx.x;
        if (e < E && r < R) tile[j][threadIdx.x] = in[(size_t)e * R + r];
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const uint32_t r = r0 + j, e = e0 + threadIdx.x;
        if (e < E && r < R) out[(size_t)r * E + e] = tile[threadIdx.x][j];
    }
}

// The ranks' blocks, as they arrive from the other GPUs -- rank g's [F][ne_g][R] one after the other -- laid out as the frames
// [F][E][R] a single context would have written (mcrt_group_trace_frames): one float4 per lane where R allows, else scalars.
// off[g] = first scan-line of rank g (off[G] = E); the block of rank g starts at float offset F * off[g] * R of `blocks`.
struct GroupOffsets { uint32_t off[65]; };
template <typename V>
__global__ void k_blocks_to_frames(const V *blocks, V *frames, uint32_t F, uint32_t E, uint32_t Rv, uint32_t G, GroupOffsets o)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // element of `frames`
    if (i >= (size_t)F * E * Rv) return;
    const uint32_t r = (uint32_t)(i % Rv), e = (uint32_t)((i / Rv) % E), f = (uint32_t)(i / ((size_t)Rv * E));
    uint32_t g = 0;
    while (g + 1u < G && e >= o.off[g + 1u]) g++;
    const uint32_t ne = o.off[g + 1u] - o.off[g];
    frames[i] = blocks[((size_t)F * o.off[g] + (size_t)f * ne + (e - o.off[g])) * Rv + r];
}

__global__ void k_math_probe(int op, const double *x, const double *y, double *out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i], b = y ? y[i] : 0.0;
    double r = 0.0, s, c;
    switch (op) {
    case 0: r = det_log(a); break;
    case 1: r = det_exp(a); break;
    case 2: det_sincos(a, s, c); r = s; break;
    case 3: det_sincos(a, s, c); r = c; break;
    case 4: r = sqrt(a); break;
    case 5: r = a / b; break;
    case 6: r = (double)det_logf((float)a); break;
    case 7: r = (double)det_expf((float)a); break;
    case 8: r = (double)det_powf((float)a, (float)b); break;
    case 9: r = (double)sqrtf((float)a); break;
    case 10: r = (double)((float)a / (float)b); break;
    case 11: r = det_pow_pos(a, b); break;
    case 12: r = (double)(fix40((float)a) & 0x7fffffffll); break;          // low 31 bits of the fixed-point echo
    case 13: r = (double)(fix40((float)a) >> 31); break;                    // the rest (arithmetic shift)
    default: break;
    }
    out[i] = r;
}

// exhaustive check that the fmaf-corrected reciprocal multiply equals IEEE division by `res` for every float in
// the gate of div_res(); mismatches are counted
__global__ void k_verify_div(float res, float rcp, unsigned long long *bad)
{
    const uint64_t n = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long local = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < (1ull << 32); b += n) {
        const float x = __uint_as_float((uint32_t)b);
        const float ax = fabsf(x);
        if (!((ax > 1e-18f && ax < 1e18f) || x == 0.0f)) continue;
        const float q0 = x * rcp;
        const float r = fmaf(-q0, res, x);
        const float q = fmaf(r, rcp, q0);
        if (!(q == x / res)) local++;          // as VALUES: for x = -0 the sequence gives +0 where the division gives -0, and both are cell 0 (the one
                                               // bit pattern in the gate where the two differ for 0.145 -- a bitwise comparison here kept the whole fast path switched off)
    }
    if (local) atomicAdd(bad, local);
}

// per material, what k_march reads: mu0, mu1, sigma and the per-step attenuation factor of main.cpp:118-119
// (the segment's attenuation is its medium's, so the factor depends on the material only)
__global__ void k_material_table(const float4 *mats, uint32_t n_mat, float axial_res_f, float freq, float4 *mtab)
{
    const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n_mat) return;
    const float4 s0 = mats[2 * m], s1 = mats[2 * m + 1];
    mtab[m] = make_float4(s0.z, s0.w, s1.x, det_expf(-s0.y * axial_res_f * 0.01f * freq * 1.0f));
}

__global__ void k_philox_probe(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t *out)
{
    uint32_t o[4];
    philox4x32_10(c0, c1, c2, c3, k0, k1, o);
    for (int i = 0; i < 4; i++) out[i] = o[i];
}

// ---------------------------------------------------------------------------------------------
// launchers (called from mcrt_api.cpp, which is plain C++)
// ---------------------------------------------------------------------------------------------
size_t march_lds_bytes(uint32_t R, uint32_t rows)               // rows: entries of the { threshold, bin } image (R + 1, or FrameArgs::march_rows)
{
    const size_t img = (size_t)rows * 16, flg = (size_t)((((R + 31u) >> 5) + 3u) & ~3u) * 4;
    return img + flg + 4 * (64 * 4 + 2 * MCRT_MARCH_TILE) + 16 * MCRT_MARCH_MTAB;        // + per wavefront: 64 length-class counters, one tile of slot numbers and of their classes; + the material table
}

hipError_t launch_init(const FrameArgs &a, hipStream_t st)
{
    const uint32_t np = a.ne * a.S;
    hipLaunchKernelGGL(k_init, dim3((np + 255u) / 256u), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_nodes_walk(const float4 *nodes, uint32_t n_nodes, uint4 *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_nodes_walk, dim3((n_nodes + 255u) / 256u), dim3(256), 0, st, nodes, n_nodes, out);
    return hipGetLastError();
}
hipError_t launch_nodes_walk_decode(const uint4 *walk, uint32_t n_nodes, float4 *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_nodes_walk_decode, dim3((n_nodes + 255u) / 256u), dim3(256), 0, st, walk, n_nodes, out);
    return hipGetLastError();
}

uint32_t lane_stack_entries() { return MCRT_LANE_STACK < MCRT_LANE_WIDE_STACK ? MCRT_LANE_STACK : MCRT_LANE_WIDE_STACK; }   // (the smaller of the two forms' LDS parts: sizes the overflow array)
uint32_t lane_wide_from() { return MCRT_LANE_WIDE_FROM; }

hipError_t launch_trace(const FrameArgs &a, uint32_t b, bool stats, hipStream_t st)
{
    // persistent over the bounce's queue: at most trace_blocks workgroups (the rest of the queue is fetched dynamically);
    // the live-ray count is only known on the device, surplus blocks read it and leave
    uint32_t np = (b == 0u) ? a.ne : a.ne * a.S;
    if (np < a.ksplit_limit) np = a.ksplit_limit;          // small bounces are cut into up to ksplit_limit pieces
    const uint32_t blocks = (np + 255u) / 256u;
    const dim3 grid(blocks < a.trace_blocks ? blocks : a.trace_blocks), blk(256);
    if (!stats && a.trace_blocks_wide != 0u && np >= a.wide_from && !(b >= 1u && b < 32u && ((a.packet_mask >> b) & 1u))) {          // a large launch: five wavefronts per SIMD (k_trace_lane_wide)
        const dim3 gridw(blocks < a.trace_blocks_wide ? blocks : a.trace_blocks_wide);
        hipLaunchKernelGGL(k_trace_lane_wide, gridw, blk, (size_t)MCRT_LANE_WIDE_STACK * 256 * sizeof(int), st, a, b);
        return hipGetLastError();
    }
    if (!stats && b >= 1u && b < 32u && ((a.packet_mask >> b) & 1u)) {       // a bounce walked a wavefront per ray packet (k_trace_packet)
        hipLaunchKernelGGL(k_trace_packet, dim3((a.ne * a.S + 63u) / 64u), dim3(64), 0, st, a, b);
        return hipGetLastError();
    }
    if (stats) hipLaunchKernelGGL((k_trace_lane<true>), grid, blk, 0, st, a, b);
    else hipLaunchKernelGGL((k_trace_lane<false>), grid, blk, 0, st, a, b);
    return hipGetLastError();
}

hipError_t launch_shade(const FrameArgs &a, uint32_t b, bool stats, hipStream_t st)
{
    const uint32_t np = a.ne * a.S;
    const dim3 grid((np + 255u) / 256u), blk(256);
    if (stats) hipLaunchKernelGGL((k_shade<true>), grid, blk, 0, st, a, b);
    else hipLaunchKernelGGL((k_shade<false>), grid, blk, 0, st, a, b);
    return hipGetLastError();
}

// every bounce of every path in one launch (k_path): the latency form, for passes of at most path_max paths (mcrt_api.cpp)
uint32_t path_blocks(size_t np) { return (uint32_t)((np + 4u * MCRT_PATH_OWNERS - 1u) / (4u * MCRT_PATH_OWNERS)); }      // workgroups of a k_path launch over np paths
hipError_t launch_path(const FrameArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(k_path, dim3(path_blocks((size_t)a.ne * a.S)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// RF accumulation of the segments of bounce b
hipError_t launch_march(const FrameArgs &a, uint32_t b, bool stats, hipStream_t st)
{
    // chunks per scan-line (every chunk zeroes and flushes its own copy of the line's bins): about ONE round of resident workgroups
    // (6 per CU).  Measured on the MI355X, 128 x 1024 paths per frame, ms per frame with 1024 / 2048 / 4096 workgroups aimed at: one
    // frame at a time (128 lines) 1.72 / 1.86 / 1.86; 4 frames in flight (512 lines) 0.96 / 0.90 / 0.95; 20 frames (2560 lines, so at
    // least that many workgroups) 0.512 / 0.512 / 0.525; from 16 frames on a line is one chunk either way.
    // The time is (work + workgroups x fixed cost) / throughput + the last workgroup's own length (work / workgroups): the best count
    // grows with the SQUARE ROOT of the work -- 1024 per 131072 paths fits all of the above (chunks rounded down).
    uint32_t target = a.march_blocks;
    if (!target) { target = (uint32_t)(1024.0 * sqrt((double)a.ne * a.S / 131072.0)); if (target < 1024u) target = 1024u; }
    uint32_t chunks = a.ne >= target ? 1u : (a.march_blocks ? (target + a.ne - 1u) / a.ne : target / a.ne);
    const uint32_t max_chunks = (a.S + 63u) / 64u;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1u) chunks = 1u;
    const dim3 grid(a.ne * chunks), blk(256);
    const bool fast = !stats && a.march_rows != 0u;            // (FrameArgs::march_rows: set when the fast kernel's conditions hold)
    const size_t lds = march_lds_bytes(a.R, fast ? a.march_rows : a.R + 1u);
    // lanes per segment: pairs give the higher throughput when there is plenty of work (515 vs 524 us per launch with 16 frames in
    // flight), quads the shorter iterations that matter when one frame at a time is traced (2.19 vs 2.37 ms per frame)
    const bool pairs = (size_t)a.ne * a.S >= (size_t)MCRT_MARCH_PAIRS_FROM;
    if (stats) { if (pairs) hipLaunchKernelGGL((k_march<true, 2, false>), grid, blk, lds, st, a, b, chunks); else hipLaunchKernelGGL((k_march<true, 4, false>), grid, blk, lds, st, a, b, chunks); }
    else if (fast) { if (pairs) hipLaunchKernelGGL((k_march<false, 2, true>), grid, blk, lds, st, a, b, chunks); else hipLaunchKernelGGL((k_march<false, 4, true>), grid, blk, lds, st, a, b, chunks); }
    else { if (pairs) hipLaunchKernelGGL((k_march<false, 2, false>), grid, blk, lds, st, a, b, chunks); else hipLaunchKernelGGL((k_march<false, 4, false>), grid, blk, lds, st, a, b, chunks); }
    return hipGetLastError();
}

hipError_t launch_finalize(long long *acc, uint32_t *flags, float *rf, uint32_t ne, uint32_t R, const uint32_t *error_flag, hipStream_t st)
{
    const size_t n = (size_t)ne * R;
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, acc, flags, rf, ne, R, error_flag);
    const size_t nf = (size_t)ne * ((R + 31u) >> 5);
    hipLaunchKernelGGL(k_clear_flags, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, flags, nf);
    return hipGetLastError();
}

hipError_t launch_convolve(float *img, float *tmp, uint32_t n_img, uint32_t E, uint32_t R, const ConvTaps &taps, hipStream_t st)
{
    const size_t n = (size_t)n_img * E * R;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    hipLaunchKernelGGL(k_conv_axial, grid, blk, 0, st, (const float *)img, tmp, n_img, E, R, taps);
    hipLaunchKernelGGL(k_conv_lateral, grid, blk, 0, st, (const float *)tmp, img, n_img, E, R, taps);
    return hipGetLastError();
}

hipError_t launch_envelope(float *img, uint32_t E, uint32_t R, hipStream_t st)
{
    hipLaunchKernelGGL(k_envelope, dim3(E), dim3(64), 0, st, img, E, R);
    return hipGetLastError();
}

hipError_t launch_remap(const float *img, uint32_t n_img, uint32_t E, uint32_t R, const float *map_col, const float *map_row, float *out, uint32_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_remap, dim3((n + 255) / 256, n_img), dim3(256), 0, st, img, E, R, map_col, map_row, out, n);
    return hipGetLastError();
}

hipError_t launch_blocks_to_frames(const float *blocks, float *frames, uint32_t F, uint32_t E, uint32_t R, uint32_t G, const uint32_t *off, hipStream_t st)
{
    GroupOffsets o;
    for (uint32_t g = 0; g <= G && g < 65u; g++) o.off[g] = off[g];
    const bool vec = (R % 4u) == 0u && ((uintptr_t)blocks % 16u) == 0u && ((uintptr_t)frames % 16u) == 0u;
    const uint32_t Rv = vec ? R / 4u : R;
    const size_t n = (size_t)F * E * Rv;
    if (vec) hipLaunchKernelGGL((k_blocks_to_frames<float4>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float4 *)blocks, (float4 *)frames, F, E, Rv, G, o);
    else hipLaunchKernelGGL((k_blocks_to_frames<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, blocks, frames, F, E, Rv, G, o);
    return hipGetLastError();
}

hipError_t launch_transpose(const float *in, float *out, uint32_t E, uint32_t R, hipStream_t st)
{
    hipLaunchKernelGGL(k_transpose, dim3((R + 31) / 32, (E + 31) / 32), dim3(32, 8), 0, st, in, out, E, R);
    return hipGetLastError();
}

hipError_t launch_math_probe(int op, const double *x, const double *y, double *out, uint32_t n, hipStream_t st)
{
    hipLaunchKernelGGL(k_math_probe, dim3((n + 255) / 256), dim3(256), 0, st, op, x, y, out, n);
    return hipGetLastError();
}

hipError_t launch_verify_div(float res, float rcp, unsigned long long *bad, hipStream_t st)
{
    hipLaunchKernelGGL(k_verify_div, dim3(256 * 16), dim3(256), 0, st, res, rcp, bad);
    return hipGetLastError();
}

hipError_t launch_tris_by_id(const float4 *tris, uint32_t n_tri, float4 *out, hipStream_t st)
{
    if (n_tri == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tris_by_id, dim3((n_tri + 255u) / 256u), dim3(256), 0, st, tris, n_tri, out);
    return hipGetLastError();
}

hipError_t launch_expand_tris(const float4 *in48, uint32_t n_tri, float4 *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_expand_tris, dim3((n_tri + 255u) / 256u), dim3(256), 0, st, in48, n_tri, out);
    return hipGetLastError();
}

hipError_t launch_material_table(const float4 *mats, uint32_t n_mat, float axial_res_f, float freq, float4 *mtab, hipStream_t st)
{
    hipLaunchKernelGGL(k_material_table, dim3((n_mat + 63u) / 64u), dim3(64), 0, st, mats, n_mat, axial_res_f, freq, mtab);
    return hipGetLastError();
}

hipError_t launch_philox_probe(const uint32_t c[4], const uint32_t k[2], uint32_t *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_philox_probe, dim3(1), dim3(1), 0, st, c[0], c[1], c[2], c[3], k[0], k[1], out);
    return hipGetLastError();
}

}  // namespace mcrt
