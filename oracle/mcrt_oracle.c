/*
 * mcrt_oracle.c -- CPU ORACLE (test infrastructure only; see mcrt_oracle.h for the rules
 * and the parity-pin status).  Plain C99.  Build: see oracle/Makefile
 *   gcc -O2 -std=gnu99 -ffp-contract=off -mfma -fopenmp -fPIC -shared
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference/src unless noted).  Arithmetic types follow the reference expression by
 * expression (float vs double promotion, operand order); "no contraction" is part of the
 * contract, fused multiply-adds appear only where written as fma().
 */
#include "mcrt_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ===================================================================================== */
/*  Deterministic math contract.  The reference calls libm (std::log/exp/pow/sin/cos);     */
/*  libm results are not reproducible on a GPU, so oracle and HIP kernels share THIS       */
/*  specification instead: double-precision argument reduction + Taylor/Horner polynomials */
/*  evaluated with fma(), IEEE +,-,*,/ and sqrt only.  Results are within ~2 ulp (double)  */
/*  of libm, i.e. identical after the float rounding the reference applies, except in      */
/*  ~1e-8 of cases (tests/test_oracle_math.py measures this against libm).                 */
/* ===================================================================================== */

static int g_math_mode = 0;
void orc_set_math_mode(int mode) { g_math_mode = mode; }

#define LN2_HI   0.6931471803691238      /* 0x3fe62e42fee00000: ln2 with 21 low bits clear */
#define LN2_LO   1.9082149292705877e-10  /* ln2 - LN2_HI                                  */
#define INV_LN2  1.4426950408889634
#define PIO2_HI  1.5707963267948966
#define PIO2_LO  6.123233995736766e-17
#define TWO_OVER_PI 0.6366197723675814
#define SQRT2_D  1.4142135623730951
#define PI_D     3.141592653589793       /* glibc M_PI, what ray.cpp:178 sees              */

static inline uint64_t d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double   u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
static inline uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float    u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

double orc_log_d(double x)
{
    if (g_math_mode) return log(x);
    if (x != x) return x;
    if (x < 0.0) return u2d(0x7ff8000000000000ull);
    if (x == 0.0) return -INFINITY;
    if (x == INFINITY) return x;
    int k = 0;
    uint64_t u = d2u(x);
    if ((u >> 52) == 0) { x *= 18014398509481984.0 /* 2^54 */; k = -54; u = d2u(x); }
    k += (int)(u >> 52) - 1023;
    double m = u2d((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull);   /* [1,2) */
    if (m > SQRT2_D) { m *= 0.5; k += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    /* log(m) = 2 atanh(s) = 2s + s * sum_{n>=1} 2/(2n+1) z^n ,  |s| <= 0.1716 */
    double p = 2.0 / 23.0;
    p = fma(p, z, 2.0 / 21.0);
    p = fma(p, z, 2.0 / 19.0);
    p = fma(p, z, 2.0 / 17.0);
    p = fma(p, z, 2.0 / 15.0);
    p = fma(p, z, 2.0 / 13.0);
    p = fma(p, z, 2.0 / 11.0);
    p = fma(p, z, 2.0 / 9.0);
    p = fma(p, z, 2.0 / 7.0);
    p = fma(p, z, 2.0 / 5.0);
    p = fma(p, z, 2.0 / 3.0);
    p = p * z;
    double r = fma(s, p, 2.0 * s);
    double kd = (double)k;
    return fma(kd, LN2_HI, fma(kd, LN2_LO, r));
}

double orc_exp_d(double x)
{
    if (g_math_mode) return exp(x);
    if (x != x) return x;
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.1332191019412) return 0.0;
    double kd = rint(x * INV_LN2);
    double r = fma(-kd, LN2_HI, x);
    r = fma(-kd, LN2_LO, r);
    /* Taylor, degree 13, |r| <= 0.3466 */
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int k = (int)kd;
    int k1 = k / 2, k2 = k - k1;
    double s1 = u2d((uint64_t)(k1 + 1023) << 52);
    double s2 = u2d((uint64_t)(k2 + 1023) << 52);
    return (p * s1) * s2;
}

void orc_sincos_d(double a, double *sn, double *cs)
{
    if (g_math_mode) { *sn = sin(a); *cs = cos(a); return; }
    double kd = rint(a * TWO_OVER_PI);
    double r = fma(-kd, PIO2_HI, a);
    r = fma(-kd, PIO2_LO, r);
    double z = r * r;
    /* sin r = r + r^3 * S(z), S = -1/3! + z/5! - ... (up to r^17) */
    double s = 1.0 / 355687428096000.0;              /* 1/17! */
    s = fma(s, z, -1.0 / 1307674368000.0);           /* 1/15! */
    s = fma(s, z, 1.0 / 6227020800.0);               /* 1/13! */
    s = fma(s, z, -1.0 / 39916800.0);                /* 1/11! */
    s = fma(s, z, 1.0 / 362880.0);                   /* 1/9!  */
    s = fma(s, z, -1.0 / 5040.0);
    s = fma(s, z, 1.0 / 120.0);
    s = fma(s, z, -1.0 / 6.0);
    double sr = fma(r * z, s, r);
    /* cos r = 1 + z * C(z), C = -1/2! + z/4! - ... (up to r^18) */
    double c = -1.0 / 6402373705728000.0;            /* 1/18! */
    c = fma(c, z, 1.0 / 20922789888000.0);           /* 1/16! */
    c = fma(c, z, -1.0 / 87178291200.0);             /* 1/14! */
    c = fma(c, z, 1.0 / 479001600.0);                /* 1/12! */
    c = fma(c, z, -1.0 / 3628800.0);                 /* 1/10! */
    c = fma(c, z, 1.0 / 40320.0);
    c = fma(c, z, -1.0 / 720.0);
    c = fma(c, z, 1.0 / 24.0);
    c = fma(c, z, -0.5);
    double cr = fma(z, c, 1.0);
    long long q = (long long)kd & 3;
    switch (q) {
    case 0:  *sn = sr;  *cs = cr;  break;
    case 1:  *sn = cr;  *cs = -sr; break;
    case 2:  *sn = -sr; *cs = -cr; break;
    default: *sn = -cr; *cs = sr;  break;
    }
}

float orc_logf(float x) { return g_math_mode ? logf(x) : (float)orc_log_d((double)x); }
float orc_expf(float x) { return g_math_mode ? expf(x) : (float)orc_exp_d((double)x); }

/* pow for x >= 0 as used by power_cosine_variate (ray.cpp:297: pow(double, float->double)) */
double orc_pow_d(double x, double y)
{
    if (g_math_mode) return pow(x, y);
    if (y == 1.0) return x;
    if (y == 0.0) return 1.0;
    if (x == 0.0) return y > 0.0 ? 0.0 : INFINITY;
    return orc_exp_d(y * orc_log_d(x));
}

/* std::pow(float,float) as used by ray.cpp:232,234 */
float orc_powf(float x, float y)
{
    if (g_math_mode) return powf(x, y);
    if (y == 1.0f) return x;                 /* every loadable scene has specularity 1.0 */
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    double ax = fabs((double)x);
    int y_is_int = (floorf(y) == y);
    int y_is_odd = y_is_int && fabsf(y) < 16777216.0f && (((long long)y) & 1);
    double r;
    if (ax == 0.0) r = (y > 0.0f) ? 0.0 : INFINITY;
    else r = orc_exp_d((double)y * orc_log_d(ax));
    if (x < 0.0f || (x == 0.0f && signbit(x))) {
        if (!y_is_int) return (x == 0.0f) ? (float)r : u2f(0x7fc00000u);
        if (y_is_odd) r = -r;
    }
    return (float)r;
}

/* Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11) */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int i = 0; i < 10; i++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

double orc_u53(uint32_t hi, uint32_t lo)
{
    uint64_t v = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6);
    return (double)v * 0x1p-53;
}

/* ===================================================================================== */
/*  Static pieces                                                                         */
/* ===================================================================================== */

/* volume.h:19-35.  libstdc++: std::default_random_engine = minstd_rand0 (x <- 16807 x mod
 * 2^31-1, seed 1); generate_canonical<double,53> draws twice; normal_distribution<double>
 * is Marsaglia polar, returning y*m first and the saved x*m on the next call. */
static uint32_t lcg_next(uint32_t *s) { *s = (uint32_t)(((uint64_t)*s * 16807u) % 2147483647u); return *s; }
static double lcg_canonical(uint32_t *s)
{
    const double range = 2147483646.0;   /* max - min + 1 */
    double sum = (double)(lcg_next(s) - 1u);
    sum += (double)(lcg_next(s) - 1u) * range;
    double r = sum / (range * range);
    if (r >= 1.0) r = nextafter(1.0, 0.0);
    return r;
}
void orc_texture_generate(float *out, uint32_t n)
{
    uint32_t st = 1u;
    size_t total = (size_t)n * n * n;
    for (size_t i = 0; i < total; i++) {
        double x, y, r2;
        do {
            x = 2.0 * lcg_canonical(&st) - 1.0;
            y = 2.0 * lcg_canonical(&st) - 1.0;
            r2 = x * x + y * y;
        } while (r2 > 1.0 || r2 == 0.0);
        double mult = sqrt(-2.0 * log(r2) / r2);
        out[2 * i + 0] = (float)(y * mult * 1.0 + 0.0);   /* texture_noise            */
        out[2 * i + 1] = (float)(x * mult * 1.0 + 0.0);   /* scattering_probability   */
    }
}

/* psf.h:34-58, 80-92 (note psf.h:9 redefines M_PI as 3.14159) */
void orc_psf(float freq, float var_x, float var_y, uint32_t res_um,
             float *axial, uint32_t n_ax, float *lateral, uint32_t n_lat)
{
    const float half_axial = (float)((size_t)n_ax * res_um) / 1000.0f / 2.0f;
    const float half_lateral = (float)((size_t)n_lat * res_um) / 1000.0f / 2.0f;
    const float resolution = (float)res_um / 1000.0f;
    for (uint32_t i = 0; i < n_ax; i++) {
        const float x = (float)i * resolution - half_axial;
        double g = exp(-0.5f * (((double)x * (double)x) / (double)var_x));
        double c = cos(2 * 3.14159 * (double)freq * (double)x);
        axial[i] = (float)(g * c);
    }
    for (uint32_t i = 0; i < n_lat; i++) {
        const float y = (float)i * resolution - half_lateral;
        lateral[i] = (float)exp(-0.5f * (((double)y * (double)y) / (double)var_y));
    }
}

typedef struct { float x, y, z; } v3;
static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* btVector3::dot (scalar path): x*x' + y*y' + z*z', left to right */
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* btVector3::cross */
static inline v3 vcross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
/* btVector3::normalized(): *this / length() == *this * (1/length()) */
static inline v3 vnormalized(v3 a) { float inv = 1.0f / sqrtf(vdot(a, a)); return vscale(a, inv); }

/* btVector3::rotate(axis, angle): o + (v-o) cos + (axis x v) sin, o = axis (axis.v) */
static v3 vrotate(v3 v, v3 axis, float angle)
{
    v3 o = vscale(axis, vdot(axis, v));
    v3 xx = vsub(v, o);
    v3 yy = vcross(axis, v);
    return vadd(vadd(o, vscale(xx, cosf(angle))), vscale(yy, sinf(angle)));
}

/* transducer.h:24-62; units conversions per include/units/units.h:1375 (deg->rad = v*pi*1/180) */
void orc_transducer(uint32_t n_elem, double radius_cm, double sep_mm,
                    const float position[3], const float angles_deg[3], float *pos, float *dir)
{
    const double PI_VAL = 3.14159265358979323846264338327950288419716939937510;
    double xa = ((double)angles_deg[0] * PI_VAL * 1.0) / 180.0;
    double ya = ((double)angles_deg[1] * PI_VAL * 1.0) / 180.0;
    double za = ((double)angles_deg[2] * PI_VAL * 1.0) / 180.0;
    /* amp = sep/radius is mm/cm; .to<float>() converts to scalar: value*1/10 (units.h:1365,1921) */
    float amp_f = (float)(((sep_mm / radius_cm) * 1.0) / 10.0);
    double amplitude = (double)amp_f;
    double center = amplitude / 2.0;
    double angle = -(amplitude * (double)n_elem / 2.0) + center;
    v3 p0 = V(position[0], position[1], position[2]);
    float rad_f = (float)radius_cm;
    for (uint32_t t = 0; t < n_elem; t++) {
        float af = (float)angle;
        v3 d = V(sinf(af), cosf(af), 0.0f);
        d = vrotate(d, V(0, 0, 1), (float)za);
        d = vrotate(d, V(1, 0, 0), (float)xa);
        d = vrotate(d, V(0, 1, 0), (float)ya);
        v3 p = vadd(p0, V(rad_f * d.x, rad_f * d.y, rad_f * d.z));
        pos[3 * t + 0] = p.x; pos[3 * t + 1] = p.y; pos[3 * t + 2] = p.z;
        dir[3 * t + 0] = d.x; dir[3 * t + 1] = d.y; dir[3 * t + 2] = d.z;
        angle = angle + amplitude;
    }
}

/* scene.cpp:313-324 (+ Bullet applying the mesh's local scaling to each vertex component) */
void orc_place_vertices(float *v, uint32_t n, float scaling, const float deltas[3], const float origin[3])
{
    float pos[3];
    for (int i = 0; i < 3; i++) pos[i] = deltas[i] * scaling * scaling + origin[i];
    for (uint32_t k = 0; k < n; k++)
        for (int i = 0; i < 3; i++) v[3 * k + i] = v[3 * k + i] * scaling + pos[i];
}

/* absolute part of the bounds padding: 4e-6 * largest finite |coordinate| (>= 1e-3) */
float orc_pad_abs(const float *tri, uint32_t n_tri)
{
    float scale = 0.0f;
    for (size_t i = 0; i < (size_t)n_tri * 9; i++) { float a = fabsf(tri[i]); if (a > scale && isfinite(a)) scale = a; }
    return 4e-6f * fmaxf(scale, 1e-3f);
}

/* main.cpp:23-37, rfimage.h:178-180 */
void orc_constants(float f_mhz, uint32_t sos, double depth_cm, orc_consts *o)
{
    o->axial_res_f = 1.45f / f_mhz;
    o->axial_res_mm = (double)o->axial_res_f;
    o->axial_res_um = (uint32_t)(o->axial_res_f * 1000.0f);
    /* micros_traveled(mm): mm -> um is value*1000/1 (units.h:1365), then / sos */
    o->time_step_us = ((o->axial_res_mm * 1000.0) / 1.0) / (double)sos;
    o->row_dt_us = (double)o->axial_res_um / (double)sos;
    /* microsecond_t(cm / (m/s)): (15/1500) [cm s/m] -> us: *10000/1 */
    o->max_travel_us = ((depth_cm / (double)sos) * 10000.0) / 1.0;
    o->max_rows = (uint32_t)((sos * (uint32_t)o->max_travel_us) / o->axial_res_um);
}

void orc_default_params(orc_params *p)
{
    memset(p, 0, sizeof *p);
    p->n_elements = 512; p->n_samples = 5; p->max_depth = 10; p->n_rows = 465;
    p->frequency = 4.5f; p->intensity_epsilon = 1e-10f; p->initial_intensity = 1.0f;
    p->ray_start_offset = 0.1f; p->sos = 1500; p->depth_cm = 15.0; p->seed = 0x5EED;
    p->sanitize_tir = 0; p->tex_n = 256; p->tex_res = 0.145f;
}

/* ===================================================================================== */
/*  Closest hit  (Bullet rayTest + ClosestRayResultCallback, scene.cpp:115-126)            */
/* ===================================================================================== */

typedef struct { float frac; int32_t tri; v3 n; float da; } hit_t;

/* Per-triangle padded bounds (contract, DESIGN.md "Closest hit"): Bullet only hands a triangle to
 * processTriangle after the ray has passed that triangle's own (quantised, slightly enlarged) AABB in the BVH
 * leaf.  The float restatement: lo/hi of the three vertices widened by 2e-4*extent + pad_abs. */
static inline void tri_bounds(const float *t9, float pad_abs, float lo[3], float hi[3])
{
    float ext = 0.0f;
    for (int a = 0; a < 3; a++) {
        float l = fminf(t9[a], fminf(t9[3 + a], t9[6 + a]));
        float h = fmaxf(t9[a], fmaxf(t9[3 + a], t9[6 + a]));
        lo[a] = l; hi[a] = h;
        ext = fmaxf(ext, h - l);
    }
    const float pad = 2e-4f * ext + pad_abs;
    for (int a = 0; a < 3; a++) { lo[a] = lo[a] - pad; hi[a] = hi[a] + pad; }
}

/* The contract's plane distance (round 3): ONE fused multiply-add per plane, t = fl(plane * inv + c) with c = -(o * inv) rounded
 * once per ray and axis -- instead of (plane - o) * inv.  For a fixed ray it is a monotone function of the plane (the exact
 * plane * inv + c is, and rounding is monotone), which is all the order-independence argument needs: a box that
 * contains another yields the wider interval.  The reciprocal direction is kept finite (rcp_dir), so no distance is ever undefined. */
static inline v3 ray_c(v3 o, v3 inv) { return V(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z)); }
/* the reciprocal direction, kept FINITE: 1/0 (a ray parallel to an axis) and overflowing quotients become +-2^100, so every plane
 * distance is a finite number (|plane| < 2^20 in any scene) and the monotonicity argument needs no special cases; the sign of the
 * huge distance still says on which side of the plane the origin lies (to within the rounding of o * inv) */
#define ORC_INV_MAX 0x1p+100f
static inline float rcp_dir(float d) { float r = 1.0f / d; return r > ORC_INV_MAX ? ORC_INV_MAX : (r < -ORC_INV_MAX ? -ORC_INV_MAX : r); }

/* ray parameter interval [tmin,tmax] (clamped to [0,tcap]) in which o + t*d lies inside the box */
static inline int slab(const float lo[3], const float hi[3], v3 c, v3 inv, float tcap, float *tmin_o, float *tmax_o)
{
    float t0x = fmaf(lo[0], inv.x, c.x), t1x = fmaf(hi[0], inv.x, c.x);
    float t0y = fmaf(lo[1], inv.y, c.y), t1y = fmaf(hi[1], inv.y, c.y);
    float t0z = fmaf(lo[2], inv.z, c.z), t1z = fmaf(hi[2], inv.z, c.z);
    float tmin = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
    float tmax = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tcap));
    *tmin_o = tmin; *tmax_o = tmax;
    return tmin <= tmax;
}

/* the same interval for a NODE of the flattened BVH4 (walk_bvh4, which mirrors the GPU walk's visit order so that node and
 * triangle counts can be compared): the near and far plane of each slab are picked by the sign of the reciprocal direction
 * instead of ordering the two distances.  Identical to slab() whenever both distances are numbers (lo <= hi, monotone rounding);
 * where one is not (origin exactly on a plane the ray runs parallel to) fmaxf / fminf drop it and the interval is the wider one.
 * Node tests only steer the walk: the answer is decided by tri_test, which keeps slab(). */
static inline int slab_node(const float lo[3], const float hi[3], v3 c, v3 inv, float tcap, float *tmin_o)
{
    float nx = fmaf(inv.x < 0.0f ? hi[0] : lo[0], inv.x, c.x), fx = fmaf(inv.x < 0.0f ? lo[0] : hi[0], inv.x, c.x);
    float ny = fmaf(inv.y < 0.0f ? hi[1] : lo[1], inv.y, c.y), fy = fmaf(inv.y < 0.0f ? lo[1] : hi[1], inv.y, c.y);
    float nz = fmaf(inv.z < 0.0f ? hi[2] : lo[2], inv.z, c.z), fz = fmaf(inv.z < 0.0f ? lo[2] : hi[2], inv.z, c.z);
    float tmin = fmaxf(fmaxf(nx, ny), fmaxf(nz, 0.0f));
    float tmax = fminf(fminf(fx, fy), fminf(fz, tcap));
    *tmin_o = tmin;
    return tmin <= tmax;
}

/* btTriangleRaycastCallback::processTriangle restated (Bullet, not under /root/reference), preceded by the
 * triangle's own bounds test.  Rules of the contract that make the answer independent of visiting order:
 *   - a hit counts only if its fraction lies inside the ray's overlap with the triangle's padded bounds
 *     (float noise far from the triangle can otherwise pass the three edge tests on a 1e9-long segment);
 *   - smaller fraction wins; equal fraction -> smaller triangle id. */
/* MEASURED ZEROS (tests/test_oracle_physics.py): while counting is on, the three additions the contract makes to the
 * reference's behaviour are counted where they act, so that "documented and improbable" can be replaced by a number per frame:
 *   [0] candidates the padded-bounds rule turned away that the bare processTriangle tests (plane crossing, smaller fraction, three
 *       edge tests) accept AND whose hit point lies inside the scene's own bounds;  [1] the same without the bounds condition;
 *   [2] echoes refused by the |e| < 1024 guard of the fixed-point bins (rfimage.h:38 would add them);
 *   [3] random_unit_vector attempts beyond the first (ray.cpp:170-185 loops while p > 0.25);  [4] ... that gave up after 8;
 *   [5] boundary hits with total internal reflection (ray.cpp:62), [6] NaN echoes (what they lead to, quirk 5: the reference's own
 *       `+= NaN`) -- not additions, counted for scale. */
static int g_counting = 0;
static uint64_t g_count[8];
static float g_scene_lo[3], g_scene_hi[3];
static inline void count_add(int i) { __atomic_fetch_add(&g_count[i], 1ull, __ATOMIC_RELAXED); }
void orc_debug_counting(const orc_scene *sc, int on)
{
    g_counting = 0;
    memset(g_count, 0, sizeof g_count);
    if (!on || !sc) return;
    for (int a = 0; a < 3; a++) { g_scene_lo[a] = INFINITY; g_scene_hi[a] = -INFINITY; }
    for (size_t i = 0; i < (size_t)sc->n_tri * 9; i++) {
        const float v = sc->tri[i]; const int a = (int)(i % 3);
        if (v < g_scene_lo[a]) g_scene_lo[a] = v;
        if (v > g_scene_hi[a]) g_scene_hi[a] = v;
    }
    g_counting = 1;
}
void orc_debug_counters(uint64_t out[8]) { for (int i = 0; i < 8; i++) out[i] = __atomic_load_n(&g_count[i], __ATOMIC_RELAXED); }

/* the three edge tests of processTriangle at fraction frac; *p_out = the interpolated point they were made at */
static inline int tri_edges(v3 v0, v3 v1, v3 v2, v3 n, v3 from, v3 to, float frac, v3 *p_out)
{
    float edge_tol = vdot(n, n) * -0.0001f;
    float s = 1.0f - frac;
    v3 p = V(s * from.x + frac * to.x, s * from.y + frac * to.y, s * from.z + frac * to.z);
    *p_out = p;
    v3 v0p = vsub(v0, p), v1p = vsub(v1, p);
    v3 cp0 = vcross(v0p, v1p);
    if (vdot(cp0, n) >= edge_tol) {
        v3 v2p = vsub(v2, p);
        v3 cp1 = vcross(v1p, v2p);
        if (vdot(cp1, n) >= edge_tol) {
            v3 cp2 = vcross(v2p, v0p);
            if (vdot(cp2, n) >= edge_tol) return 1;
        }
    }
    return 0;
}
static void count_rule_reject(v3 v0, v3 v1, v3 v2, v3 n, v3 from, v3 to, float frac)
{
    v3 p;
    if (!tri_edges(v0, v1, v2, n, from, to, frac, &p)) return;
    count_add(1);
    if (p.x >= g_scene_lo[0] && p.x <= g_scene_hi[0] && p.y >= g_scene_lo[1] && p.y <= g_scene_hi[1] &&
        p.z >= g_scene_lo[2] && p.z <= g_scene_hi[2]) count_add(0);
}

static inline void tri_test(const float *t9, int32_t id, v3 from, v3 to, v3 inv, v3 rc, float pad_abs, hit_t *best)
{
    v3 v0 = V(t9[0], t9[1], t9[2]), v1 = V(t9[3], t9[4], t9[5]), v2 = V(t9[6], t9[7], t9[8]);
    v3 v10 = vsub(v1, v0), v20 = vsub(v2, v0);
    v3 n = vcross(v10, v20);
    float dist = vdot(v0, n);
    float da = vdot(n, from) - dist;
    float db = vdot(n, to) - dist;
    if (da * db >= 0.0f) return;
    float proj = da - db;
    float frac = da / proj;
    if (frac < best->frac || (frac == best->frac && id < best->tri)) {
        float lo[3], hi[3], tmin, tmax;
        tri_bounds(t9, pad_abs, lo, hi);
        if (!slab(lo, hi, rc, inv, 1.0f, &tmin, &tmax) || !(frac >= tmin && frac <= tmax)) {
            if (g_counting) count_rule_reject(v0, v1, v2, n, from, to, frac);
            return;
        }
        v3 p;
        if (tri_edges(v0, v1, v2, n, from, to, frac, &p)) {
            best->frac = frac; best->tri = id; best->n = n; best->da = da;
        }
    }
}

#define ORC_STACK 96

static void walk_bvh(const orc_scene *sc, v3 from, v3 to, hit_t *best, orc_stats *st)
{
    v3 d = vsub(to, from);
    v3 inv = V(rcp_dir(d.x), rcp_dir(d.y), rcp_dir(d.z));
    const v3 rc = ray_c(from, inv);
    int32_t stack[ORC_STACK];
    int sp = 0;
    int32_t cur = 0;
    uint64_t nn = 0, nt = 0;
    for (;;) {
        if (cur >= 0) {
            const orc_bvh_node *N = &sc->nodes[cur];
            nn++;
            float tn0, tn1, tx0, tx1;
            int h0 = slab(N->lo0, N->hi0, rc, inv, fminf(1.0f, best->frac), &tn0, &tx0);
            int h1 = slab(N->lo1, N->hi1, rc, inv, fminf(1.0f, best->frac), &tn1, &tx1);
            if (h0 && h1) {
                int32_t nearc = N->c0, farc = N->c1;
                if (tn1 < tn0) { nearc = N->c1; farc = N->c0; }
                if (sp < ORC_STACK) stack[sp++] = farc;
                cur = nearc;
                continue;
            } else if (h0) { cur = N->c0; continue; }
            else if (h1) { cur = N->c1; continue; }
        } else {
            uint32_t v = (uint32_t)~cur;
            uint32_t first = v >> 3, cnt = (v & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; i++) {
                const float *t = sc->bvh_tri + (size_t)(first + i) * 12;
                float t9[9] = { t[0], t[1], t[2], t[4], t[5], t[6], t[8], t[9], t[10] };
                tri_test(t9, (int32_t)f2u(t[3]), from, to, inv, rc, sc->pad_abs, best);
                nt++;
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    if (st) { st->nodes_visited += nn; st->tris_tested += nt; }
}

/* BVH4 walk in the GPU kernel's order: children hit are visited nearest-first -- by the bit pattern of t_near with its two
 * lowest bits replaced by the slot number (the kernel's unique per-quad key) --, the others are stacked so that they pop in
 * ascending slot order; a leaf's triangles are all tested. */
typedef struct { float lo[3]; float hix, hiy, hiz; int32_t ref; uint32_t pad; } orc_bvh4_child;
#define ORC_BVH4_EMPTY ((int32_t)0x80000000)
static void walk_bvh4(const orc_scene *sc, v3 from, v3 to, hit_t *best, orc_stats *st)
{
    const orc_bvh4_child *nodes = (const orc_bvh4_child *)sc->nodes4;
    v3 d = vsub(to, from);
    v3 inv = V(rcp_dir(d.x), rcp_dir(d.y), rcp_dir(d.z));
    const v3 rc = ray_c(from, inv);
    int32_t stack[ORC_STACK];
    int sp = 0;
    int32_t cur = 0;
    uint64_t nn = 0, nt = 0;
    for (;;) {
        if (cur >= 0) {
            const orc_bvh4_child *N = nodes + 4 * (size_t)cur;
            nn++;
            uint32_t key[4]; int32_t ref[4]; int nh = 0;
            float tcap = fminf(1.0f, best->frac);
            for (int k = 0; k < 4; k++) {
                float hi[3] = { N[k].hix, N[k].hiy, N[k].hiz }, tn;
                int h = slab_node(N[k].lo, hi, rc, inv, tcap, &tn) && N[k].ref != ORC_BVH4_EMPTY;
                key[k] = h ? ((f2u(tn) & ~3u) | (uint32_t)k) : 0xffffffffu; ref[k] = N[k].ref; nh += h;   /* t_near >= 0: bits order like the value */
            }
            if (nh > 0) {
                /* the kernel's order: smallest key next, the others stacked in slot order */
                int jn = -1;
                for (int k = 0; k < 4; k++) if (key[k] != 0xffffffffu && (jn < 0 || key[k] < key[jn])) jn = k;
                int32_t next = ref[jn];
                int pos = 0;
                for (int k = 0; k < 4; k++) {
                    if (key[k] == 0xffffffffu || k == jn) continue;
                    if (sp + pos < ORC_STACK) stack[sp + pos] = ref[k];
                    pos++;
                }
                sp += nh - 1;
                cur = next;
                continue;
            }
        } else {
            uint32_t v = (uint32_t)~cur;
            uint32_t first = v >> 3, cnt = (v & 7u) + 1u;
            for (uint32_t i = 0; i < cnt; i++) {
                const float *t = sc->bvh_tri + (size_t)(first + i) * 12;
                float t9[9] = { t[0], t[1], t[2], t[4], t[5], t[6], t[8], t[9], t[10] };
                tri_test(t9, (int32_t)f2u(t[3]), from, to, inv, rc, sc->pad_abs, best);
                nt++;
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    if (st) { st->nodes_visited += nn; st->tris_tested += nt; }
}

int32_t orc_closest_hit(const orc_scene *sc, const float from_[3], const float to_[3], int use_bvh,
                        float *frac, float normal[3], float point[3], orc_stats *st)
{
    v3 from = V(from_[0], from_[1], from_[2]), to = V(to_[0], to_[1], to_[2]);
    hit_t best; best.frac = 1.0f; best.tri = -1; best.n = V(0, 0, 0); best.da = 0;
    if (st) st->queries++;
    if (use_bvh == 2 && sc->nodes4 && sc->n_nodes4) {
        walk_bvh4(sc, from, to, &best, st);
    } else if (use_bvh && sc->nodes && sc->n_nodes) {
        walk_bvh(sc, from, to, &best, st);
    } else {
        v3 d = vsub(to, from);
        v3 inv = V(rcp_dir(d.x), rcp_dir(d.y), rcp_dir(d.z));
        const v3 rc = ray_c(from, inv);
        for (uint32_t i = 0; i < sc->n_tri; i++) tri_test(sc->tri + (size_t)i * 9, (int32_t)i, from, to, inv, rc, sc->pad_abs, &best);
        if (st) st->tris_tested += sc->n_tri;
    }
    if (best.tri < 0) return -1;
    /* triangleNormal.normalize(); flipped to face the ray origin when dist_a <= 0 */
    v3 nn = vnormalized(best.n);
    if (best.da <= 0.0f) nn = vneg(nn);
    /* ClosestRayResultCallback: m_hitPointWorld.setInterpolate3(from, to, fraction) */
    float s = 1.0f - best.frac;
    if (frac) *frac = best.frac;
    if (normal) { normal[0] = nn.x; normal[1] = nn.y; normal[2] = nn.z; }
    if (point) {
        point[0] = s * from.x + best.frac * to.x;
        point[1] = s * from.y + best.frac * to.y;
        point[2] = s * from.z + best.frac * to.z;
    }
    if (st) st->hits++;
    return best.tri;
}

/* ===================================================================================== */
/*  Ray physics (ray.cpp)                                                                 */
/* ===================================================================================== */

enum { M_IMP = 0, M_ATT, M_MU0, M_MU1, M_SIGMA, M_SPEC, M_SHINE, M_THICK };
#define OUT_NONE (-1)   /* media_outside == nullptr                                         */
#define OUT_SELF (-2)   /* media_outside aliases the ray's own media (quirk 2, ray.cpp:38)   */

typedef struct {
    v3 from, dir;
    int32_t media, outside;
    float intensity, frequency;
    double dist_mm;
    int alive;
} ray_t;

typedef struct {
    uint32_t key[2];
    uint32_t element, sample, bounce;
} rng_t;

static inline void rng_block(const rng_t *g, uint32_t block, double *a, double *b)
{
    uint32_t ctr[4] = { g->element, g->sample, g->bounce, block }, out[4];
    orc_philox4x32_10(ctr, g->key, out);
    *a = orc_u53(out[0], out[1]);
    *b = orc_u53(out[2], out[3]);
}

/* ray.cpp:213-224 power_cosine_variate(int v) */
static float power_cosine_variate(int v, double number)
{
    int indice = v + 1;
    float exponente = (float)((double)1.0 / indice);
    return (float)orc_pow_d(number, (double)exponente);
}

/* ray.cpp:167-211 random_unit_vector; draws come from blocks 2,3,... (one block per attempt) */
static v3 random_unit_vector(v3 v, float cos_theta, const rng_t *g, uint32_t *attempts_out)
{
    int flag = 0;
    float px, py, p;
    uint32_t attempt = 0;
    do {
        double ua, ur;
        rng_block(g, 2u + attempt, &ua, &ur);
        double a = ua * 2 * PI_D;
        double r = 0.5 * sqrt(ur);
        double sa, ca;
        orc_sincos_d(a, &sa, &ca);
        px = (float)(r * ca);
        py = (float)(r * sa);
        p = px * px + py * py;
        attempt++;
    } while (!(p <= 0.25f) && attempt < 8u);
    if (attempts_out) *attempts_out = attempt;
    if (g_counting && attempt > 1u) { for (uint32_t i = 1; i < attempt; i++) count_add(3); if (!(p <= 0.25f)) count_add(4); }
    float vx = v.x, vy = v.y, vz = v.z;
    if (fabsf(vx) > fabsf(vy)) { vx = vy; vy = v.x; flag = 1; }
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
    return V(wx, wy, wz);
}

static inline float std_max(float a, float b) { return (a < b) ? b : a; }   /* std::max operand order */

typedef struct { float reflected_intensity; ray_t returned; } hit_result;

/* ray.cpp:11-97.  dbg (tests only, NULL on the trace path): the intermediate values, see orc_hit_debug */
static hit_result hit_boundary(const ray_t *r, v3 hit_point, v3 surface_normal, const orc_mesh *cm,
                               const orc_scene *sc, const orc_params *prm, const rng_t *g, orc_hit_debug *dbg)
{
    int32_t after_vasc, mat_after;
    if (r->outside != OUT_NONE) {
        if (cm->vascular) { after_vasc = OUT_NONE; mat_after = (r->outside == OUT_SELF) ? r->media : r->outside; }
        else {
            after_vasc = (r->outside == (int32_t)cm->mat_inside) ? (int32_t)cm->mat_outside : (int32_t)cm->mat_inside;
            mat_after = r->media;
        }
    } else {
        if (cm->vascular) { after_vasc = OUT_SELF; mat_after = (int32_t)cm->mat_inside; }
        else { after_vasc = OUT_NONE; mat_after = (int32_t)cm->mat_inside; /* quirk 1: &r.media never equals a map element */ }
    }
    const float *ma = sc->mat + (size_t)mat_after * 8;
    const float *mr = sc->mat + (size_t)r->media * 8;

    double u_pc, u_x;
    rng_block(g, 1u, &u_pc, &u_x);
    float random_angle = power_cosine_variate((int)ma[M_SHINE], u_pc);
    uint32_t ruv_attempts = 0;
    v3 random_normal = random_unit_vector(surface_normal, random_angle, g, &ruv_attempts);

    float incidence = vdot(r->dir, vneg(random_normal));
    if (incidence < 0) incidence = vdot(r->dir, random_normal);
    const float refr_ratio = mr[M_IMP] / ma[M_IMP];
    float refraction_angle = 1 - refr_ratio * refr_ratio * (1 - incidence * incidence);
    const int tir = refraction_angle < 0;
    if (g_counting && tir) count_add(5);
    refraction_angle = sqrtf(refraction_angle);

    /* snells_law ray.cpp:115-124: r*l + (r*c - c2)*n */
    float k = refr_ratio * incidence - refraction_angle;
    v3 refr = V(refr_ratio * r->dir.x + k * random_normal.x,
                refr_ratio * r->dir.y + k * random_normal.y,
                refr_ratio * r->dir.z + k * random_normal.z);
    refr = vnormalized(refr);
    float two_c = 2 * incidence;
    v3 refl = V(r->dir.x + two_c * random_normal.x, r->dir.y + two_c * random_normal.y, r->dir.z + two_c * random_normal.z);
    refl = vnormalized(refl);

    float intensity_refl;
    if (tir) intensity_refl = r->intensity;
    else {
        /* reflection_intensity ray.cpp:126-132: pow(num/denom, 2) promotes to double */
        float num = mr[M_IMP] * incidence - ma[M_IMP] * refraction_angle;
        float den = mr[M_IMP] * incidence + ma[M_IMP] * refraction_angle;
        float q = num / den;
        intensity_refl = (float)((double)r->intensity * ((double)q * (double)q));
    }
    const float intensity_refr = r->intensity - intensity_refl;

    /* Eq. 8, ray.cpp:154-164 */
    float ra = vdot(r->dir, refr);
    float refraction_factor = orc_powf(ra, ma[M_SPEC]);
    float rb = vdot(r->dir, refl);
    float reflection_factor = orc_powf(rb, ma[M_SPEC]);
    if (prm->sanitize_tir && tir) refraction_factor = 0.0f;
    float back = (std_max(refraction_factor, 0.0f) + std_max(reflection_factor, 0.0f)) * random_angle;

    float x = (float)u_x;
    float prob = intensity_refl / r->intensity;
    hit_result res;
    res.reflected_intensity = back;
    res.returned = *r;
    res.returned.from = hit_point;
    if (prob > x) {
        res.returned.dir = refl;
        res.returned.intensity = intensity_refl > prm->intensity_epsilon ? intensity_refl : 0.0f;
    } else {
        res.returned.dir = refr;
        res.returned.media = mat_after;
        res.returned.outside = after_vasc;
        res.returned.intensity = intensity_refr > prm->intensity_epsilon ? intensity_refr : 0.0f;
    }
    if (dbg) {
        dbg->random_angle = random_angle;
        dbg->random_normal[0] = random_normal.x; dbg->random_normal[1] = random_normal.y; dbg->random_normal[2] = random_normal.z;
        dbg->incidence = incidence; dbg->refr_ratio = refr_ratio; dbg->refraction_angle = refraction_angle;
        dbg->refr_dir[0] = refr.x; dbg->refr_dir[1] = refr.y; dbg->refr_dir[2] = refr.z;
        dbg->refl_dir[0] = refl.x; dbg->refl_dir[1] = refl.y; dbg->refl_dir[2] = refl.z;
        dbg->intensity_refl = intensity_refl; dbg->intensity_refr = intensity_refr;
        dbg->refraction_factor = refraction_factor; dbg->reflection_factor = reflection_factor;
        dbg->u_pc = u_pc; dbg->u_x = u_x;
        dbg->tir = tir; dbg->chose_reflection = prob > x; dbg->mat_after = mat_after; dbg->after_vasc = after_vasc;
        dbg->ruv_attempts = ruv_attempts;
    }
    return res;
}

/* scene.cpp:281-290 */
static double distance_in_mm(const orc_scene *sc, v3 a, v3 b)
{
    float xd = fabsf(a.x - b.x) * sc->spacing[0];
    float yd = fabsf(a.y - b.y) * sc->spacing[1];
    float zd = fabsf(a.z - b.z) * sc->spacing[2];
    return sqrt((double)xd * (double)xd + (double)yd * (double)yd + (double)zd * (double)zd) * 10;
}

/* volume.h:46-61; quirk 4: float -> unsigned of a possibly negative value wraps (x86-64) */
static inline uint32_t vox_index(float q, uint32_t n)
{
    int64_t i;
    if (!(fabsf(q) < 9.2233720368547758e18f)) i = (int64_t)0x8000000000000000ull; /* cvttss2si indefinite */
    else i = (int64_t)q;
    return ((uint32_t)i) % n;
}
static inline float get_scattering(const float *tex, uint32_t n, float res, float density, float mu, float sigma, v3 p)
{
    uint32_t x = vox_index(p.x / res, n), y = vox_index(p.y / res, n), z = vox_index(p.z / res, n);
    const float *vx = tex + 2 * (((size_t)x * n + y) * n + z);
    return vx[1] >= density ? vx[0] * sigma + mu : 0.0f;
}

static inline uint32_t steps_from(double q)
{
    /* (unsigned int)(double): x86-64 cvttsd2si r64 then low 32 bits */
    if (!(fabs(q) < 9.2233720368547758e18)) return 0u;
    return (uint32_t)(int64_t)q;
}

#define FIX_SCALE 1099511627776.0   /* 2^40: |echo| < 1024 keeps every term below 2^50 */
static inline void fix_add(int64_t *acc, uint8_t *flag, float echo)
{
    if (!(fabsf(echo) < 1024.0f)) { if (flag) *flag = 1; if (g_counting) count_add(echo != echo ? 6 : 2); return; }
    *acc += (int64_t)rint((double)echo * FIX_SCALE);
}

/* Optional companion of rf_ref: the SAME echoes added in the SAME (reference) order, but into doubles.  The float image is what
 * the reference's `cv::Mat += echo` produces; the distance between the two is that running float sum's own rounding error,
 * which grows with the number of samples per scan-line.  Set by orc_set_ref64 around a call that requests rf_ref (tests only;
 * one call at a time: orc_trace_frame reads the pointer once, before its parallel region, and hands it to the sinks). */
static double *g_ref64 = NULL;
void orc_set_ref64(double *buf) { g_ref64 = buf; }

typedef struct {
    float *rf_ref; uint32_t ref_cols, ref_col;       /* [R][cols] */
    double *rf_ref64;                                 /* the same image carried in double (same shape), or NULL */
    int64_t *rf_fix; uint8_t *rf_flags;               /* [R] of this element */
} rf_sink;

static inline void add_echo(const rf_sink *k, const orc_consts *c, uint32_t n_rows, float echo, double t_us)
{
    /* rfimage.h:33-40 */
    double row = t_us / c->row_dt_us;
    if (row < (double)n_rows) {
        int r = (int)row;
        if (k->rf_ref) {
            k->rf_ref[(size_t)r * k->ref_cols + k->ref_col] += echo;
            if (k->rf_ref64) k->rf_ref64[(size_t)r * k->ref_cols + k->ref_col] += (double)echo;
        }
        if (k->rf_fix) fix_add(&k->rf_fix[r], k->rf_flags ? &k->rf_flags[r] : NULL, echo);
    }
}

/* main.cpp:106-144 for one segment */
static void accumulate_segment(const orc_scene *sc, const orc_params *prm, const orc_consts *c, const float *tex,
                               const orc_segment *sg, const rf_sink *k, orc_stats *st)
{
    const float *m = sc->mat + (size_t)sg->media * 8;
    const double starting_micros = ((sg->distance_traveled * 1000.0) / 1.0) / (double)prm->sos;
    /* scene::distance scene.cpp:342-346: from.distance(to)*10.0f */
    v3 from = V(sg->from[0], sg->from[1], sg->from[2]), to = V(sg->to[0], sg->to[1], sg->to[2]);
    v3 df = vsub(to, from);
    float dist_f = sqrtf(vdot(df, df)) * 10.0f;
    uint32_t steps = steps_from((double)dist_f / c->axial_res_mm);
    v3 dir = V(sg->dir[0], sg->dir[1], sg->dir[2]);
    v3 delta = V(c->axial_res_f * dir.x, c->axial_res_f * dir.y, c->axial_res_f * dir.z);
    v3 point = from;
    double t = starting_micros;
    float intensity = sg->initial_intensity;
    const float k_att = orc_expf(-sg->attenuation * c->axial_res_f * 0.01f * prm->frequency * 1.0f);
    uint64_t n = 0;
    for (uint32_t step = 0; step < steps && t < c->max_travel_us; step++) {
        float scattering = get_scattering(tex, prm->tex_n, prm->tex_res, m[M_MU1], m[M_MU0], m[M_SIGMA], point);
        add_echo(k, c, prm->n_rows, intensity * scattering, t);
        point = vadd(point, delta);
        t = t + c->time_step_us;
        intensity *= k_att;
        n++;
    }
    if (st) st->rf_steps += n;
    add_echo(k, c, prm->n_rows, sg->reflected_intensity / (float)prm->n_samples,
             starting_micros + c->time_step_us * (double)(uint32_t)(steps - 1u));
}

/* scene.cpp:50-183 for one path (element e, sample s) */
static void trace_path(const orc_scene *sc, const orc_params *prm, const orc_consts *c, const float *tex,
                       v3 el_pos, v3 el_dir, uint32_t frame_id, uint32_t e_abs, uint32_t s, int use_bvh,
                       int32_t *hits, orc_segment *segs, uint32_t *seg_count, const rf_sink *k, orc_stats *st)
{
    ray_t r;
    r.from = el_pos; r.dir = el_dir; r.media = (int32_t)sc->start_mat; r.outside = OUT_NONE;
    r.intensity = prm->initial_intensity / (float)prm->n_samples;
    r.frequency = prm->frequency; r.dist_mm = 0.0; r.alive = 1;
    rng_t g; g.key[0] = prm->seed; g.key[1] = frame_id; g.element = e_abs; g.sample = s;
    uint32_t nseg = 0;
    for (uint32_t b = 0; b < prm->max_depth; b++) {
        if (hits) hits[b] = -2;
        if (!r.alive) continue;
        g.bounce = b;
        const float *mr = sc->mat + (size_t)r.media * 8;
        /* max_ray_length ray.cpp:110-113 */
        float L = 10.f * orc_logf(prm->intensity_epsilon / r.intensity) / -mr[M_ATT] * r.frequency;
        /* enlarge scene.cpp:292-298 */
        float Ls = L / 100.0f;
        v3 to = V(r.from.x + Ls * (sc->spacing[0] * r.dir.x), r.from.y + Ls * (sc->spacing[1] * r.dir.y), r.from.z + Ls * (sc->spacing[2] * r.dir.z));
        v3 f2 = V(r.from.x + prm->ray_start_offset * r.dir.x, r.from.y + prm->ray_start_offset * r.dir.y, r.from.z + prm->ray_start_offset * r.dir.z);
        float ff[3] = { f2.x, f2.y, f2.z }, tt[3] = { to.x, to.y, to.z }, frac, nrm[3], pt[3];
        int32_t tri = orc_closest_hit(sc, ff, tt, use_bvh, &frac, nrm, pt, st);
        if (hits) hits[b] = tri;
        orc_segment sg;
        if (tri >= 0) {
            double dist_before = r.dist_mm;
            float i_before = r.intensity;
            const orc_mesh *organ = &sc->mesh[sc->tri_mesh[tri]];
            /* scene.cpp:132-139: q = |N(0, thickness)|, Box-Muller on block 0 */
            float sigma = sc->mat[(size_t)organ->mat_inside * 8 + M_THICK];
            float q = 0.0f;
            if (sigma != 0.0f) {
                double n1, n2, sn, cs;
                rng_block(&g, 0u, &n1, &n2);
                orc_sincos_d(n2 * 2 * PI_D, &sn, &cs);
                double z = sqrt(-2.0 * orc_log_d(1.0 - n1)) * cs;
                q = (float)fabs(z * (double)sigma + 0.0);
            }
            v3 hp = V(pt[0], pt[1], pt[2]);
            v3 inside = V(q * r.dir.x + hp.x, q * r.dir.y + hp.y, q * r.dir.z + hp.z);
            /* travel ray.cpp:99-103 */
            double mm = distance_in_mm(sc, r.from, inside);
            r.dist_mm = r.dist_mm + mm;
            r.intensity = r.intensity * orc_expf(-mr[M_ATT] * ((float)mm * 0.01f) * r.frequency);
            hit_result hr = hit_boundary(&r, hp, V(nrm[0], nrm[1], nrm[2]), organ, sc, prm, &g, NULL);
            sg.from[0] = r.from.x; sg.from[1] = r.from.y; sg.from[2] = r.from.z;
            sg.to[0] = inside.x; sg.to[1] = inside.y; sg.to[2] = inside.z;
            sg.dir[0] = r.dir.x; sg.dir[1] = r.dir.y; sg.dir[2] = r.dir.z;
            sg.reflected_intensity = hr.reflected_intensity; sg.initial_intensity = i_before;
            sg.attenuation = mr[M_ATT]; sg.distance_traveled = dist_before; sg.media = r.media; sg.tri = tri;
            if (hr.returned.intensity > prm->intensity_epsilon) r = hr.returned; else r.alive = 0;
        } else {
            sg.from[0] = r.from.x; sg.from[1] = r.from.y; sg.from[2] = r.from.z;
            sg.to[0] = to.x; sg.to[1] = to.y; sg.to[2] = to.z;
            sg.dir[0] = r.dir.x; sg.dir[1] = r.dir.y; sg.dir[2] = r.dir.z;
            sg.reflected_intensity = 0.0f; sg.initial_intensity = r.intensity;
            sg.attenuation = mr[M_ATT]; sg.distance_traveled = r.dist_mm; sg.media = r.media; sg.tri = -1;
            r.alive = 0;
        }
        if (segs) segs[nseg] = sg;
        nseg++;
        if (st) st->segments++;
        if (k && (k->rf_ref || k->rf_fix)) accumulate_segment(sc, prm, c, tex, &sg, k, st);
    }
    if (seg_count) *seg_count = nseg;
}

/* ===================================================================================== */
/*  Test entry points into the static physics above (tests/test_oracle_physics.py and the  */
/*  independent reading tests/ref_reading.py).  They call the SAME static functions the     */
/*  trace path calls; nothing here is a second implementation.                              */
/* ===================================================================================== */
static ray_t ray_from_state(const orc_ray_state *q)
{
    ray_t r;
    r.from = V(q->from[0], q->from[1], q->from[2]); r.dir = V(q->dir[0], q->dir[1], q->dir[2]);
    r.media = q->media; r.outside = q->outside; r.intensity = q->intensity; r.frequency = q->frequency;
    r.dist_mm = q->dist_mm; r.alive = 1;
    return r;
}
static void state_from_ray(const ray_t *r, orc_ray_state *q)
{
    q->from[0] = r->from.x; q->from[1] = r->from.y; q->from[2] = r->from.z;
    q->dir[0] = r->dir.x; q->dir[1] = r->dir.y; q->dir[2] = r->dir.z;
    q->media = r->media; q->outside = r->outside; q->intensity = r->intensity; q->frequency = r->frequency; q->dist_mm = r->dist_mm;
}
static rng_t rng_from(const uint32_t c[5]) { rng_t g; g.key[0] = c[0]; g.key[1] = c[1]; g.element = c[2]; g.sample = c[3]; g.bounce = c[4]; return g; }

float orc_debug_power_cosine(int v, double number) { return power_cosine_variate(v, number); }

uint32_t orc_debug_random_unit_vector(const float v[3], float cos_theta, const uint32_t rng[5], float w[3])
{
    const rng_t g = rng_from(rng);
    uint32_t attempts = 0;
    v3 r = random_unit_vector(V(v[0], v[1], v[2]), cos_theta, &g, &attempts);
    w[0] = r.x; w[1] = r.y; w[2] = r.z;
    return attempts;
}

void orc_debug_hit_boundary(const orc_scene *sc, const orc_params *prm, const orc_ray_state *rs, const float hit_point[3],
                            const float normal[3], uint32_t mesh, const uint32_t rng[5], orc_hit_debug *out)
{
    const ray_t r = ray_from_state(rs);
    const rng_t g = rng_from(rng);
    hit_result hr = hit_boundary(&r, V(hit_point[0], hit_point[1], hit_point[2]), V(normal[0], normal[1], normal[2]),
                                 &sc->mesh[mesh], sc, prm, &g, out);
    out->reflected_intensity = hr.reflected_intensity;
    state_from_ray(&hr.returned, &out->returned);
}

/* max_ray_length ray.cpp:110-113 + enlarge scene.cpp:292-298 + the 0.1 start offset scene.cpp:115, as trace_path applies them */
float orc_debug_ray_segment(const orc_scene *sc, const orc_params *prm, const orc_ray_state *rs, float from_off[3], float to[3])
{
    const float *mr = sc->mat + (size_t)rs->media * 8;
    float L = 10.f * orc_logf(prm->intensity_epsilon / rs->intensity) / -mr[M_ATT] * rs->frequency;
    float Ls = L / 100.0f;
    for (int a = 0; a < 3; a++) {
        to[a] = rs->from[a] + Ls * (sc->spacing[a] * rs->dir[a]);
        from_off[a] = rs->from[a] + prm->ray_start_offset * rs->dir[a];
    }
    return L;
}

/* travel ray.cpp:99-103 over distance_in_mm(from, to_point) scene.cpp:281-290; returns the millimetres */
double orc_debug_travel(const orc_scene *sc, orc_ray_state *rs, const float to_point[3])
{
    const float *mr = sc->mat + (size_t)rs->media * 8;
    double mm = distance_in_mm(sc, V(rs->from[0], rs->from[1], rs->from[2]), V(to_point[0], to_point[1], to_point[2]));
    rs->dist_mm = rs->dist_mm + mm;
    rs->intensity = rs->intensity * orc_expf(-mr[M_ATT] * ((float)mm * 0.01f) * rs->frequency);
    return mm;
}

/* the thickness draw of scene.cpp:132-139 as the contract makes it (Box-Muller on block 0) */
float orc_debug_thickness(float sigma, const uint32_t rng[5])
{
    const rng_t g = rng_from(rng);
    if (sigma == 0.0f) return 0.0f;
    double n1, n2, sn, cs;
    rng_block(&g, 0u, &n1, &n2);
    orc_sincos_d(n2 * 2 * PI_D, &sn, &cs);
    double z = sqrt(-2.0 * orc_log_d(1.0 - n1)) * cs;
    return (float)fabs(z * (double)sigma + 0.0);
}

/* main.cpp:106-144 for ONE segment into a single RF line rf[R] (float, reference order); returns the RF steps taken */
uint64_t orc_debug_accumulate_segment(const orc_scene *sc, const orc_params *prm, const float *tex, const orc_segment *sg, float *rf)
{
    orc_consts c;
    orc_constants(prm->frequency, prm->sos, prm->depth_cm, &c);
    rf_sink k; memset(&k, 0, sizeof k);
    k.rf_ref = rf; k.ref_cols = 1; k.ref_col = 0;
    orc_stats st; memset(&st, 0, sizeof st);
    accumulate_segment(sc, prm, &c, tex, sg, &k, &st);
    return st.rf_steps;
}

#define ORC_PRIV_ROWS 2048      /* rows of a sample block's private bins (on the task's stack); more rows: one task per scan-line */
void orc_trace_frame(const orc_scene *sc, const orc_params *p,
                     const float *el_pos, const float *el_dir, const float *texture,
                     uint32_t frame_id, uint32_t e_begin, uint32_t e_end, int use_bvh, int n_threads,
                     int32_t *hits, orc_segment *segs, uint32_t *seg_count,
                     float *rf_ref, int64_t *rf_fix, uint8_t *rf_flags, orc_stats *st_out)
{
    orc_consts c;
    orc_constants(p->frequency, p->sos, p->depth_cm, &c);
    const uint32_t ne = e_end - e_begin, S = p->n_samples, B = p->max_depth, R = p->n_rows;
    orc_stats total; memset(&total, 0, sizeof total);
    if (n_threads < 1) n_threads = 1;
    double *const ref64 = rf_ref ? g_ref64 : NULL;     /* read ONCE, before the parallel region: the sinks carry it from here on */
    /* Tasks = (scan-line, block of samples).  The reference-order float image adds a scan-line's echoes strictly in
     * sample order (main.cpp:106-144), so it is only produced with one task per scan-line; the fixed-point image is an
     * integer sum and may be cut into sample blocks, which is what keeps every core of a many-core host busy
     * (block sums are merged with integer adds: the result does not depend on the cut). */
    uint32_t chunks = 1;
    if (!rf_ref && ne > 0 && (uint32_t)n_threads > ne / 2u && R <= ORC_PRIV_ROWS) {
        chunks = (4u * (uint32_t)n_threads + ne - 1u) / ne;
        if (chunks > S) chunks = S;
        if (chunks < 1u) chunks = 1u;
    }
    const uint32_t per = (S + chunks - 1u) / chunks;
    const long long n_tasks = (long long)ne * chunks;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
#endif
    for (long long task = 0; task < n_tasks; task++) {
        orc_stats st; memset(&st, 0, sizeof st);
        const uint32_t ei = (uint32_t)(task / chunks), ck = (uint32_t)(task % chunks);
        const uint32_t s0 = ck * per, s1 = (s0 + per < S) ? s0 + per : S;
        if (s0 >= s1) continue;
        uint32_t e = e_begin + ei;
        v3 pos = V(el_pos[3 * e], el_pos[3 * e + 1], el_pos[3 * e + 2]);
        v3 dir = V(el_dir[3 * e], el_dir[3 * e + 1], el_dir[3 * e + 2]);
        rf_sink k;
        k.rf_ref = rf_ref; k.ref_cols = ne; k.ref_col = ei; k.rf_ref64 = ref64;
        k.rf_fix = rf_fix ? rf_fix + (size_t)ei * R : NULL;
        k.rf_flags = rf_flags ? rf_flags + (size_t)ei * R : NULL;
        int64_t *priv_fix = NULL; uint8_t *priv_flags = NULL;
        int64_t stack_fix[ORC_PRIV_ROWS]; uint8_t stack_flags[ORC_PRIV_ROWS];      /* (R <= 2048 in every caller: no allocation that could fail) */
        if (chunks > 1u && rf_fix) {                    /* private bins of this block, merged below */
            priv_fix = stack_fix; priv_flags = stack_flags;
            memset(priv_fix, 0, sizeof(int64_t) * R); memset(priv_flags, 0, R);
            k.rf_fix = priv_fix; k.rf_flags = rf_flags ? priv_flags : NULL;
        }
        for (uint32_t s = s0; s < s1; s++) {
            size_t pi = (size_t)ei * S + s;
            trace_path(sc, p, &c, texture, pos, dir, frame_id, e, s, use_bvh,
                       hits ? hits + pi * B : NULL, segs ? segs + pi * B : NULL,
                       seg_count ? seg_count + pi : NULL, &k, &st);
        }
        if (priv_fix) {
            int64_t *dst = rf_fix + (size_t)ei * R;
            for (uint32_t r = 0; r < R; r++) {
                if (priv_fix[r]) {
#ifdef _OPENMP
#pragma omp atomic
#endif
                    dst[r] += priv_fix[r];
                }
                if (rf_flags && priv_flags[r]) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
                    rf_flags[(size_t)ei * R + r] = 1;
                }
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            total.queries += st.queries; total.nodes_visited += st.nodes_visited; total.tris_tested += st.tris_tested;
            total.segments += st.segments; total.rf_steps += st.rf_steps; total.hits += st.hits;
        }
    }
    if (st_out) *st_out = total;
}

void orc_finalize_rf(const int64_t *rf_fix, const uint8_t *rf_flags, uint32_t n_elem, uint32_t n_rows, float *out)
{
    for (uint32_t e = 0; e < n_elem; e++)
        for (uint32_t r = 0; r < n_rows; r++) {
            size_t i = (size_t)e * n_rows + r;
            float v = (rf_flags && rf_flags[i]) ? u2f(0x7fc00000u) : (float)((double)rf_fix[i] * 0x1p-40);
            out[(size_t)r * n_elem + e] = v;
        }
}

/* rfimage.h:93-123 */
void orc_convolve(float *img, float *tmp, uint32_t rows_, uint32_t cols_,
                  const float *axial, uint32_t n_ax, const float *lateral, uint32_t n_lat)
{
    const int rows = (int)rows_, cols = (int)cols_, na = (int)n_ax, nl = (int)n_lat;
    for (int col = 0; col < cols; col++)
        for (int row = na; row < rows - na; row++) {
            float conv = 0;
            for (int k = 0; k < na; k++) conv += img[(size_t)(row + k) * cols + col] * axial[k];
            tmp[(size_t)row * cols + col] = conv;
        }
    for (int row = na; row < rows - na; row++)
        for (int col = nl / 2; col < cols - nl; col++) {
            float conv = 0;
            for (int k = 0; k < nl; k++) conv += tmp[(size_t)row * cols + col + k] * lateral[k];
            img[(size_t)row * cols + col] = conv;
        }
}

/* rfimage.h:54-91 */
void orc_envelope(float *img, uint32_t rows, uint32_t cols)
{
#define AT(r, c) img[(size_t)(r) * cols + (c)]
    for (uint32_t column = 0; column < cols; column++) {
        int ascending = AT(0, column) < AT(1, column);
        size_t last_peak_pos = 0;
        float last_peak = AT(last_peak_pos, column);
        for (size_t i = 1; i + 1 < rows; i++) {
            if (AT(i, column) < AT(i + 1, column)) ascending = 1;
            else if (ascending) {
                ascending = 0;
                const float new_peak = fabsf(AT(i, column));
                for (size_t j = last_peak_pos; j < i; j++) {
                    const float alpha = ((float)j - (float)last_peak_pos) / ((float)i - (float)last_peak_pos);
                    AT(j, column) = last_peak * (1 - alpha) + new_peak * alpha;
                }
                last_peak_pos = i;
                last_peak = new_peak;
            }
        }
    }
#undef AT
}

/* rfimage.h:183-215 (create_mapping).  map_row = the reference's map_x (ROW coordinate in the RF image), map_col = its map_y
 * (COLUMN coordinate) -- cv::remap(src, dst, map_y, map_x) at rfimage.h:139 swaps them back.  [out_rows][out_cols], row-major.
 *
 * Operand types, expression by expression (rfimage.h line -> what C++ makes of it; pinned by oracle/ref_probe.cpp, which evaluates
 * the same statements with the reference's own units.h types -> tests/golden/ref_probe.json "scan_maps_*"):
 *   :186 ratio   = (max_travel_time * speed_of_sound * 0.001f      unsigned * unsigned (wraps mod 2^32) -> float * float: 150.0f
 *                   + radius.to<float>()                           float + float = float (180.0f)
 *                   - radius.to<float>() * std::cos(ta_f / 2.0))   float / double -> cos(double) -> float * double = double; float - double = double
 *                  / scan_converted.rows                           double / int = double, rounded ONCE to float by the declaration
 *   :189 shift_y = radius * std::cos(ta_f / 2.0f)                  float / float -> std::cos(float) = cosf; millimeter_t (double) * float = double
 *   :192 half_width = (float)cols / 2.0f                           float
 *   :201 fi      = (float)i + shift_y.to<float>() / ratio          float throughout
 *   :202 fj      = (float)j - half_width                           float
 *   :205 r       = std::sqrt(std::pow(fi,2.0f) + std::pow(fj,2.0f))  float overloads.  GCC expands pow(x, 2.0f) to x * x from -O1 on; at -O0 (the reference's
 *                                                                  CMakeLists.txt sets no optimisation level) glibc's powf is called: the probe built
 *                                                                  at -O0 and at -O1 prints identical maps for all three golden shapes (checked when
 *                                                                  the fixture was generated), so x * x it is
 *   :208 angle   = radian_t(std::atan2(fj, fi))                    atan2f, widened to double
 *   :211 map_x   = (r*ratio - radius.to<float>()) / (max_travel_time*speed_of_sound*0.001f) * (float)rf_height       float throughout
 *   :212 map_y   = ((angle - (-total_angle/2)) / total_angle) * (float)rf_width   radian_t arithmetic in double, * float -> double, rounded once by the store
 * (Rounds 1-3 carried the depth of :186/:211 as a double, `depth_um * 0.001f` = 150.0000071, and divided in double: ratio 0.385048121
 * instead of 0.385048091, 72 % of the row coordinates off by up to 9e-5 -- found by the round-3 review.) */
void orc_scan_maps(uint32_t rows, uint32_t cols, double radius_mm, double total_angle, uint32_t max_travel_us, uint32_t sos,
                   uint32_t out_rows, uint32_t out_cols, float *map_row, float *map_col)
{
    const float radius_f = (float)radius_mm, ta_f = (float)total_angle;
    const float depth_mm_f = (float)(uint32_t)(max_travel_us * sos) * 0.001f;          /* unsigned product, then float */
    const float ratio = (float)(((double)(depth_mm_f + radius_f) - (double)radius_f * cos((double)ta_f / 2.0)) / (double)(int)out_rows);
    const double shift_y = radius_mm * (double)cosf(ta_f / 2.0f);
    const float half_width = (float)(int)out_cols / 2.0f;
    for (uint32_t j = 0; j < out_cols; j++)
        for (uint32_t i = 0; i < out_rows; i++) {
            const float fi = (float)(int)i + (float)shift_y / ratio;
            const float fj = (float)(int)j - half_width;
            const float r = sqrtf(fi * fi + fj * fj);
            const double angle = (double)atan2f(fj, fi);
            map_row[(size_t)i * out_cols + j] = (r * ratio - radius_f) / depth_mm_f * (float)rows;
            map_col[(size_t)i * out_cols + j] = (float)(((angle - (-total_angle / 2)) / total_angle) * (double)(float)cols);
        }
}

/* rfimage.h:125-140 postprocess: cv::remap(src, dst, map_y, map_x, INTER_LINEAR, BORDER_CONSTANT 0) over the maps above.
 * OpenCV is absent: the bilinear kernel is restated as EXACT bilinear in float (OpenCV
 * quantises fractions to 1/32) -- parity unpinned for the interpolation; the maps are pinned (above).
 * max_travel_us / sos are the rf_image template parameters (unsigned int: main.cpp:36 hands over max_travel_time.to<unsigned int>()). */
void orc_scan_convert(const float *img, uint32_t rows, uint32_t cols, double radius_mm, double total_angle,
                      double max_travel_us, double sos, float *out, uint32_t out_rows, uint32_t out_cols)
{
    const size_t n = (size_t)out_rows * out_cols;
    float *map_row = (float *)malloc(n * sizeof(float)), *map_col = (float *)malloc(n * sizeof(float));
    if (!map_row || !map_col) { free(map_row); free(map_col); return; }
    orc_scan_maps(rows, cols, radius_mm, total_angle, (uint32_t)max_travel_us, (uint32_t)sos, out_rows, out_cols, map_row, map_col);
    for (size_t p = 0; p < n; p++) {
        const float my = map_row[p], mx = map_col[p];
        /* remap: dst(i,j) = src(y=my, x=mx) bilinear, constant 0 border */
        float fx = floorf(mx), fy = floorf(my);
        float ax = mx - fx, ay = my - fy;
        long x0 = (long)fx, y0 = (long)fy;
        float v[2][2];
        for (int dy = 0; dy < 2; dy++)
            for (int dx = 0; dx < 2; dx++) {
                long xx = x0 + dx, yy = y0 + dy;
                v[dy][dx] = (mx == mx && my == my && xx >= 0 && yy >= 0 && xx < (long)cols && yy < (long)rows) ? img[(size_t)yy * cols + xx] : 0.0f;
            }
        float top = v[0][0] * (1.0f - ax) + v[0][1] * ax;
        float bot = v[1][0] * (1.0f - ax) + v[1][1] * ax;
        out[p] = top * (1.0f - ay) + bot * ay;
    }
    free(map_row); free(map_col);
}

/* ===================================================================================== */
/*  ANALYSIS (tools/bvh_width.py): the same closest-hit queries walked over the product's  */
/*  BVH2 collapsed to W-wide nodes, W = 2..16, counted.  Not part of any parity path.      */
/* ===================================================================================== */
typedef struct { float lo[3], hi[3]; int32_t ref; } wide_child;
struct orc_wide { uint32_t W, n_nodes; wide_child *c; uint32_t *visits; /* optional: visits per node (orc_wide_visits) */ };

static float wc_harea(const wide_child *s) { float dx = s->hi[0] - s->lo[0], dy = s->hi[1] - s->lo[1], dz = s->hi[2] - s->lo[2]; return dx * dy + dy * dz + dz * dx; }
static void wc_kids(const orc_bvh_node *n, wide_child *a, wide_child *b)
{
    a->ref = n->c0; b->ref = n->c1;
    for (int i = 0; i < 3; i++) { a->lo[i] = n->lo0[i]; a->hi[i] = n->hi0[i]; b->lo[i] = n->lo1[i]; b->hi[i] = n->hi1[i]; }
}
/* the product's collapse rule (csrc/mcrt_host.cpp, Collapser::build), for any width: the inner child with the largest half area is
 * replaced by its two children until the node is full.  quant: 0 = float boxes; 8 = the children's boxes snapped OUTWARDS to a
 * 256-step grid spanning the node's own box per axis (what an 8-bit node in the parent's frame could store) */
static uint32_t wide_build(struct orc_wide *w, const orc_scene *sc, int32_t n2, int quant, uint32_t *cap)
{
    wide_child s[16]; uint32_t k = 2;
    wc_kids(&sc->nodes[n2], &s[0], &s[1]);
    if (s[0].ref == s[1].ref && s[0].ref < 0) k = 1;
    while (k < w->W) {
        int pick = -1; float best = -1.f;
        for (uint32_t i = 0; i < k; i++) if (s[i].ref >= 0) { float a = wc_harea(&s[i]); if (a > best) { best = a; pick = (int)i; } }
        if (pick < 0) break;
        wide_child a, b; wc_kids(&sc->nodes[s[pick].ref], &a, &b);
        s[pick] = a; s[k++] = b;
    }
    if (quant == 8) {
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (uint32_t i = 0; i < k; i++) for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], s[i].lo[a]); hi[a] = fmaxf(hi[a], s[i].hi[a]); }
        for (int a = 0; a < 3; a++) {
            const float step = (hi[a] - lo[a]) / 255.0f;
            if (!(step > 0.0f)) continue;
            for (uint32_t i = 0; i < k; i++) {
                float ql = floorf((s[i].lo[a] - lo[a]) / step), qh = ceilf((s[i].hi[a] - lo[a]) / step);
                float nl = lo[a] + ql * step, nh = lo[a] + qh * step;
                while (nl > s[i].lo[a]) { ql -= 1.0f; nl = lo[a] + ql * step; }
                while (nh < s[i].hi[a]) { qh += 1.0f; nh = lo[a] + qh * step; }
                s[i].lo[a] = nl; s[i].hi[a] = nh;
            }
        }
    }
    if (w->n_nodes == *cap) { *cap *= 2; w->c = (wide_child *)realloc(w->c, sizeof(wide_child) * (size_t)*cap * w->W); }
    const uint32_t me = w->n_nodes++;
    for (uint32_t i = 0; i < w->W; i++) {
        wide_child c;
        if (i < k) {
            c = s[i];
            if (c.ref >= 0) c.ref = (int32_t)wide_build(w, sc, c.ref, quant, cap);
        } else { c.lo[0] = c.lo[1] = c.lo[2] = INFINITY; c.hi[0] = c.hi[1] = c.hi[2] = -INFINITY; c.ref = ORC_BVH4_EMPTY; }
        w->c[(size_t)me * w->W + i] = c;
    }
    return me;
}

struct orc_wide *orc_wide_build(const orc_scene *sc, uint32_t W, int quant)
{
    if (!sc || !sc->nodes || sc->n_nodes == 0 || W < 2 || W > 16) return NULL;
    struct orc_wide *w = (struct orc_wide *)calloc(1, sizeof *w);
    uint32_t cap = sc->n_nodes / (W - 1) + 16;
    w->W = W; w->c = (wide_child *)malloc(sizeof(wide_child) * (size_t)cap * W);
    wide_build(w, sc, 0, quant, &cap);
    return w;
}
void orc_wide_free(struct orc_wide *w) { if (w) { free(w->c); free(w); } }
uint32_t orc_wide_nodes(const struct orc_wide *w) { return w ? w->n_nodes : 0; }
/* visits[node] is incremented for every inner-node visit of the following orc_wide_count calls (NULL: off) */
void orc_wide_visits(struct orc_wide *w, uint32_t *visits) { if (w) w->visits = visits; }

/* an order of the inner nodes, root first, in which a node always comes after its parent: mode 0 = breadth first (by depth), mode 1 =
 * always the pending node with the largest box area next (a ray-independent guess at "most visited").  order[k] = node; returns the count */
uint32_t orc_wide_order(const struct orc_wide *w, int mode, uint32_t *order)
{
    typedef struct { float key; uint32_t node; } item;
    item *heap = (item *)malloc(sizeof(item) * ((size_t)w->n_nodes + 1));
    uint32_t hn = 0, n = 0, seq = 0;
#define HEAP_PUSH(K, N) { uint32_t i_ = hn++; heap[i_].key = (K); heap[i_].node = (N); \
        while (i_ > 0 && heap[(i_ - 1) / 2].key < heap[i_].key) { item t_ = heap[i_]; heap[i_] = heap[(i_ - 1) / 2]; heap[(i_ - 1) / 2] = t_; i_ = (i_ - 1) / 2; } }
    HEAP_PUSH(INFINITY, 0u)
    while (hn > 0) {
        const uint32_t me = heap[0].node;
        heap[0] = heap[--hn];
        for (uint32_t i = 0;;) {
            uint32_t l = 2 * i + 1, r = l + 1, m = i;
            if (l < hn && heap[l].key > heap[m].key) m = l;
            if (r < hn && heap[r].key > heap[m].key) m = r;
            if (m == i) break;
            item t = heap[i]; heap[i] = heap[m]; heap[m] = t; i = m;
        }
        order[n++] = me;
        for (uint32_t k = 0; k < w->W; k++) {
            const wide_child *c = &w->c[(size_t)me * w->W + k];
            if (c->ref < 0) continue;                       /* leaf or empty */
            seq++;
            HEAP_PUSH(mode == 0 ? -(float)seq : wc_harea(c), (uint32_t)c->ref)
        }
    }
#undef HEAP_PUSH
    free(heap);
    return n;
}

/* one query: nearest hit child first (key = t_near bits with the slot in the low 4 bits), the others stacked in slot order;
 * returns the triangle; *steps = inner nodes + leaves visited (the ray's chain of dependent fetches), *nn / *nt nodes and triangles */
static int32_t wide_walk(const struct orc_wide *w, const orc_scene *sc, v3 from, v3 to, uint32_t *nn_o, uint32_t *nt_o, uint32_t *nl_o, uint32_t *maxsp_o)
{
    hit_t best; best.frac = 1.0f; best.tri = -1; best.n = V(0, 0, 0); best.da = 0;
    v3 d = vsub(to, from);
    v3 inv = V(rcp_dir(d.x), rcp_dir(d.y), rcp_dir(d.z));
    const v3 rc = ray_c(from, inv);
    int32_t stack[256]; int sp = 0, maxsp = 0;
    int32_t cur = 0;
    uint32_t nn = 0, nt = 0, nl = 0;
    const uint32_t W = w->W;
    for (;;) {
        if (cur >= 0) {
            const wide_child *N = w->c + (size_t)cur * W;
            nn++;
            if (w->visits) __atomic_fetch_add(&w->visits[cur], 1u, __ATOMIC_RELAXED);
            uint32_t key[16]; int nh = 0;
            const float tcap = fminf(1.0f, best.frac);
            for (uint32_t k = 0; k < W; k++) {
                float tn;
                const int h = N[k].ref != ORC_BVH4_EMPTY && slab_node(N[k].lo, N[k].hi, rc, inv, tcap, &tn);
                key[k] = h ? ((f2u(tn) & ~15u) | k) : 0xffffffffu; nh += h;
            }
            if (nh > 0) {
                int jn = -1;
                for (uint32_t k = 0; k < W; k++) if (key[k] != 0xffffffffu && (jn < 0 || key[k] < key[jn])) jn = (int)k;
                for (uint32_t k = 0; k < W; k++) { if (key[k] == 0xffffffffu || (int)k == jn) continue; if (sp < 256) stack[sp++] = N[k].ref; }
                if (sp > maxsp) maxsp = sp;
                cur = N[jn].ref;
                continue;
            }
        } else {
            const uint32_t v = (uint32_t)~cur, first = v >> 3, cnt = (v & 7u) + 1u;
            nl++;
            for (uint32_t i = 0; i < cnt; i++) {
                const float *t = sc->bvh_tri + (size_t)(first + i) * 12;
                float t9[9] = { t[0], t[1], t[2], t[4], t[5], t[6], t[8], t[9], t[10] };
                tri_test(t9, (int32_t)f2u(t[3]), from, to, inv, rc, sc->pad_abs, &best);
                nt++;
            }
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    *nn_o = nn; *nt_o = nt; *nl_o = nl; *maxsp_o = (uint32_t)maxsp;
    return best.tri;
}

/* the closest-hit queries of `n` segments (from, dir, the intensity and attenuation they started with: what orc_trace_frame returns),
 * rebuilt as trace_path builds them, walked over `w`.  out[q] = { inner nodes, leaves, triangles, deepest stack } of query q;
 * tri[q] = the triangle found (to be compared with the segment's own). */
void orc_wide_count(const struct orc_wide *w, const orc_scene *sc, const orc_params *prm, const orc_segment *segs, uint64_t n,
                    uint32_t *out /*[n][4]*/, int32_t *tri /*[n]*/, int n_threads)
{
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t q = 0; q < (int64_t)n; q++) {
        const orc_segment *sg = &segs[q];
        const v3 from = V(sg->from[0], sg->from[1], sg->from[2]), dir = V(sg->dir[0], sg->dir[1], sg->dir[2]);
        const float L = 10.f * orc_logf(prm->intensity_epsilon / sg->initial_intensity) / -sg->attenuation * prm->frequency;
        const float Ls = L / 100.0f;
        const v3 to = V(from.x + Ls * (sc->spacing[0] * dir.x), from.y + Ls * (sc->spacing[1] * dir.y), from.z + Ls * (sc->spacing[2] * dir.z));
        const v3 f2 = V(from.x + prm->ray_start_offset * dir.x, from.y + prm->ray_start_offset * dir.y, from.z + prm->ray_start_offset * dir.z);
        uint32_t nn, nt, nl, ms;
        tri[q] = wide_walk(w, sc, f2, to, &nn, &nt, &nl, &ms);
        out[4 * q] = nn; out[4 * q + 1] = nl; out[4 * q + 2] = nt; out[4 * q + 3] = ms;
    }
}

/* ANALYSIS (tools/seed_count.py): what would a closest-hit bound known BEFORE the walk save?  The queries of traced segments
 * walked over the product's BVH4 (walk_bvh4, the GPU's order) three ways, nodes and triangles counted per query:
 *   mode 0  as today (best = 1.0, no triangle);
 *   mode 1  PERFECT seed: best = the query's own answer (the lower bound of what any seed can give);
 *   mode 2  LEADER seed: best = triangle seed_tri[q] (the answer of the bundle's representative ray) tested against THIS ray with the
 *           contract's tri_test -- exact by construction (a real hit is an upper bound, the id rule still decides ties); -1 = no seed.
 * out[q] = { nodes, triangles }, tri[q] = the triangle found (must equal the segment's own in every mode). */
void orc_seed_count(const orc_scene *sc, const orc_params *prm, const orc_segment *segs, uint64_t n, int mode, const int32_t *seed_tri,
                    uint32_t *out /*[n][2]*/, int32_t *tri /*[n]*/, int n_threads)
{
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t q = 0; q < (int64_t)n; q++) {
        const orc_segment *sg = &segs[q];
        const v3 from0 = V(sg->from[0], sg->from[1], sg->from[2]), dir = V(sg->dir[0], sg->dir[1], sg->dir[2]);
        const float L = 10.f * orc_logf(prm->intensity_epsilon / sg->initial_intensity) / -sg->attenuation * prm->frequency;
        const float Ls = L / 100.0f;
        const v3 to = V(from0.x + Ls * (sc->spacing[0] * dir.x), from0.y + Ls * (sc->spacing[1] * dir.y), from0.z + Ls * (sc->spacing[2] * dir.z));
        const v3 from = V(from0.x + prm->ray_start_offset * dir.x, from0.y + prm->ray_start_offset * dir.y, from0.z + prm->ray_start_offset * dir.z);
        hit_t best; best.frac = 1.0f; best.tri = -1; best.n = V(0, 0, 0); best.da = 0;
        v3 d = vsub(to, from);
        v3 inv = V(rcp_dir(d.x), rcp_dir(d.y), rcp_dir(d.z));
        const v3 rc = ray_c(from, inv);
        int32_t s = -1;
        if (mode == 1) s = sg->tri; else if (mode == 2 && seed_tri) s = seed_tri[q];
        if (s >= 0) tri_test(sc->tri + (size_t)s * 9, s, from, to, inv, rc, sc->pad_abs, &best);
        orc_stats st; memset(&st, 0, sizeof st);
        walk_bvh4(sc, from, to, &best, &st);
        out[2 * q] = (uint32_t)st.nodes_visited; out[2 * q + 1] = (uint32_t)st.tris_tested + (s >= 0 ? 1u : 0u);
        tri[q] = best.tri;
    }
}

/* ANALYSIS (tools/packet_count.py): north_star's literal design -- "one wavefront per ray packet" -- counted before it is built.  W consecutive
 * queries (queue order: the sample paths of a scan-line with one history are neighbours) walk the BVH4 TOGETHER: one shared stack, a node is
 * visited when ANY ray of the packet passes the box of that child with its OWN current closest fraction, every ray tests every triangle of a
 * visited leaf.  Legal under the contract (the closest hit does not depend on the visiting order; boxes only cull).  Children are visited
 * nearest first by the smallest t_near among the rays that hit them (order 0) or by the FIRST hitting ray's t_near (order 1: what a wavefront
 * can do with one readfirstlane).  out[p] = { nodes the packet visits, leaves, triangles (each tested by all W lanes), sum over its rays of the nodes
 * a ray walking alone visits, the largest of those, live rays }; tri[q] must equal the segments' own triangles. */
void orc_packet_count(const orc_scene *sc, const orc_params *prm, const orc_segment *segs, uint64_t n, uint32_t W, int order,
                      uint32_t *out /*[packets][6]*/, int32_t *tri /*[n]*/, int n_threads)
{
    const orc_bvh4_child *nodes = (const orc_bvh4_child *)sc->nodes4;
    if (W == 0 || W > 64 || !nodes) {                     /* the per-ray arrays below hold 64 rays: anything else is a caller's error, reported in the results */
        for (uint64_t i = 0; i < n; i++) tri[i] = -3;
        return;
    }
    const int64_t n_pack = (int64_t)((n + W - 1) / W);
    if (n_threads > 0) omp_set_num_threads(n_threads);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t p = 0; p < n_pack; p++) {
        v3 from[64], to[64], inv[64], rc[64]; hit_t best[64]; uint32_t alone[64];
        const uint32_t m = (uint32_t)((uint64_t)(p + 1) * W <= n ? W : n - (uint64_t)p * W);
        for (uint32_t r = 0; r < m; r++) {
            const orc_segment *sg = &segs[(uint64_t)p * W + r];
            const v3 f0 = V(sg->from[0], sg->from[1], sg->from[2]), dir = V(sg->dir[0], sg->dir[1], sg->dir[2]);
            const float L = 10.f * orc_logf(prm->intensity_epsilon / sg->initial_intensity) / -sg->attenuation * prm->frequency;
            const float Ls = L / 100.0f;
            to[r] = V(f0.x + Ls * (sc->spacing[0] * dir.x), f0.y + Ls * (sc->spacing[1] * dir.y), f0.z + Ls * (sc->spacing[2] * dir.z));
            from[r] = V(f0.x + prm->ray_start_offset * dir.x, f0.y + prm->ray_start_offset * dir.y, f0.z + prm->ray_start_offset * dir.z);
            const v3 d = vsub(to[r], from[r]);
            inv[r] = V(rcp_dir(d.x), rcp_dir(d.y), rcp_dir(d.z)); rc[r] = ray_c(from[r], inv[r]);
            best[r].frac = 1.0f; best[r].tri = -1; best[r].n = V(0, 0, 0); best[r].da = 0;
            hit_t b1 = best[r]; orc_stats st; memset(&st, 0, sizeof st);
            walk_bvh4(sc, from[r], to[r], &b1, &st); alone[r] = (uint32_t)st.nodes_visited;
        }
        int32_t stack[256]; int sp = 0; int32_t cur = 0; int overflow = 0;
        uint32_t nn = 0, nl = 0, nt = 0;
        for (;;) {
            if (cur >= 0) {
                const orc_bvh4_child *N = nodes + 4 * (size_t)cur;
                nn++;
                float key[4]; int hitc[4]; int nh = 0;
                for (int k = 0; k < 4; k++) {
                    hitc[k] = 0; key[k] = INFINITY;
                    if (N[k].ref == ORC_BVH4_EMPTY) continue;
                    const float hi[3] = { N[k].hix, N[k].hiy, N[k].hiz };
                    for (uint32_t r = 0; r < m; r++) {
                        float tn;
                        if (slab_node(N[k].lo, hi, rc[r], inv[r], fminf(1.0f, best[r].frac), &tn)) {
                            if (!hitc[k] || (order == 0 && tn < key[k])) key[k] = tn;
                            hitc[k] = 1;
                            if (order == 1) break;            /* the first hitting ray decides the order */
                        }
                    }
                    if (order == 1 && hitc[k]) { /* the remaining rays are not consulted for the ORDER; whether the child is visited is already decided */ }
                    nh += hitc[k];
                }
                if (nh > 0) {
                    int ord[4], c = 0;
                    for (int k = 0; k < 4; k++) if (hitc[k]) ord[c++] = k;
                    for (int i = 1; i < c; i++) { int x = ord[i], j = i - 1; while (j >= 0 && key[ord[j]] > key[x]) { ord[j + 1] = ord[j]; j--; } ord[j + 1] = x; }
                    for (int i = c - 1; i >= 1; i--) { if (sp < 256) stack[sp++] = N[ord[i]].ref; else overflow = 1; }        /* farthest first: the nearest pops first */
                    cur = N[ord[0]].ref;
                    continue;
                }
            } else {
                const uint32_t v = (uint32_t)~cur, first = v >> 3, cnt = (v & 7u) + 1u;
                nl++;
                for (uint32_t i = 0; i < cnt; i++) {
                    const float *t = sc->bvh_tri + (size_t)(first + i) * 12;
                    float t9[9] = { t[0], t[1], t[2], t[4], t[5], t[6], t[8], t[9], t[10] };
                    for (uint32_t r = 0; r < m; r++) tri_test(t9, (int32_t)f2u(t[3]), from[r], to[r], inv[r], rc[r], sc->pad_abs, &best[r]);
                    nt++;
                }
            }
            if (sp == 0) break;
            cur = stack[--sp];
        }
        uint32_t sum = 0, mx = 0;
        for (uint32_t r = 0; r < m; r++) { sum += alone[r]; if (alone[r] > mx) mx = alone[r]; tri[(uint64_t)p * W + r] = overflow ? -3 : best[r].tri; }      /* a dropped stack entry would under-count: the packet's results are marked instead */
        uint32_t *o = out + 6 * p;
        o[0] = nn; o[1] = nl; o[2] = nt; o[3] = sum; o[4] = mx; o[5] = m;
    }
}
