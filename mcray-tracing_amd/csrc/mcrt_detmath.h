// mcrt_detmath.h -- deterministic math + counter-based RNG of the parity contract (DESIGN.md
// "Deterministic math", "RNG contract"), device side.
//
// The reference calls libm (std::log ray.cpp:112, std::exp ray.cpp:102 / main.cpp:135, pow
// ray.cpp:132,158,160,223, sin/cos ray.cpp:181-182) and seeds a fresh std::mt19937 from
// std::random_device for every draw (ray.cpp:85,175,216; scene.cpp:132).  Neither is reproducible
// on a GPU, so kernels implement this specification: double-precision range reduction +
// Taylor/Horner polynomials evaluated with fma, IEEE + - * / sqrt only, Philox4x32-10 keyed by
// (seed, frame) with counter (element, sample, bounce, block).  Compiled with -ffp-contract=off:
// an fma happens exactly where fma() is written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MCRT_DEV __device__ __forceinline__

namespace mcrt {

constexpr double LN2_HI = 0.6931471803691238;
constexpr double LN2_LO = 1.9082149292705877e-10;
constexpr double INV_LN2 = 1.4426950408889634;
constexpr double PIO2_HI = 1.5707963267948966;
constexpr double PIO2_LO = 6.123233995736766e-17;
constexpr double TWO_OVER_PI = 0.6366197723675814;
constexpr double SQRT2_D = 1.4142135623730951;
constexpr double PI_D = 3.141592653589793;

MCRT_DEV uint64_t d2u(double x) { return (uint64_t)__double_as_longlong(x); }
MCRT_DEV double u2d(uint64_t u) { return __longlong_as_double((long long)u); }
MCRT_DEV double dinf() { return u2d(0x7ff0000000000000ull); }

// fma(a, b, k) with the CONSTANT k read from a scalar register pair: one vector instruction (v_fma_f64) and two scalar moves per Horner step.
// Written as fma(a, b, literal) the compiler materialises the literal in a vector register pair and accumulates into it (2 x v_mov_b32 +
// v_fmac_f64): every third vector instruction of the physics was such a move.  Same operation, same bits.
MCRT_DEV double fma_k(double a, double b, double k)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}

MCRT_DEV double det_log(double x)
{
    if (x != x) return x;
    if (x < 0.0) return u2d(0x7ff8000000000000ull);
    if (x == 0.0) return -dinf();
    if (x == dinf()) return x;
    int k = 0;
    uint64_t u = d2u(x);
    if ((u >> 52) == 0) { x *= 18014398509481984.0; k = -54; u = d2u(x); }
    k += (int)(u >> 52) - 1023;
    double m = u2d((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > SQRT2_D) { m *= 0.5; k += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 2.0 / 23.0;
    p = fma_k(p, z, 2.0 / 21.0);
    p = fma_k(p, z, 2.0 / 19.0);
    p = fma_k(p, z, 2.0 / 17.0);
    p = fma_k(p, z, 2.0 / 15.0);
    p = fma_k(p, z, 2.0 / 13.0);
    p = fma_k(p, z, 2.0 / 11.0);
    p = fma_k(p, z, 2.0 / 9.0);
    p = fma_k(p, z, 2.0 / 7.0);
    p = fma_k(p, z, 2.0 / 5.0);
    p = fma_k(p, z, 2.0 / 3.0);
    p = p * z;
    double r = fma(s, p, 2.0 * s);
    double kd = (double)k;
    return fma(kd, LN2_HI, fma(kd, LN2_LO, r));
}

MCRT_DEV double det_exp(double x)
{
    if (x != x) return x;
    if (x > 709.782712893384) return dinf();
    if (x < -745.1332191019412) return 0.0;
    double kd = rint(x * INV_LN2);
    double r = fma(-kd, LN2_HI, x);
    r = fma(-kd, LN2_LO, r);
    double p = 1.0 / 6227020800.0;
    p = fma_k(p, r, 1.0 / 479001600.0);
    p = fma_k(p, r, 1.0 / 39916800.0);
    p = fma_k(p, r, 1.0 / 3628800.0);
    p = fma_k(p, r, 1.0 / 362880.0);
    p = fma_k(p, r, 1.0 / 40320.0);
    p = fma_k(p, r, 1.0 / 5040.0);
    p = fma_k(p, r, 1.0 / 720.0);
    p = fma_k(p, r, 1.0 / 120.0);
    p = fma_k(p, r, 1.0 / 24.0);
    p = fma_k(p, r, 1.0 / 6.0);
    p = fma_k(p, r, 0.5);
    p = fma_k(p, r, 1.0);
    p = fma_k(p, r, 1.0);
    int k = (int)kd;
    int k1 = k / 2, k2 = k - k1;
    double s1 = u2d((uint64_t)(k1 + 1023) << 52);
    double s2 = u2d((uint64_t)(k2 + 1023) << 52);
    return (p * s1) * s2;
}

MCRT_DEV void det_sincos(double a, double &sn, double &cs)
{
    double kd = rint(a * TWO_OVER_PI);
    double r = fma(-kd, PIO2_HI, a);
    r = fma(-kd, PIO2_LO, r);
    double z = r * r;
    double s = 1.0 / 355687428096000.0;
    s = fma_k(s, z, -1.0 / 1307674368000.0);
    s = fma_k(s, z, 1.0 / 6227020800.0);
    s = fma_k(s, z, -1.0 / 39916800.0);
    s = fma_k(s, z, 1.0 / 362880.0);
    s = fma_k(s, z, -1.0 / 5040.0);
    s = fma_k(s, z, 1.0 / 120.0);
    s = fma_k(s, z, -1.0 / 6.0);
    double sr = fma(r * z, s, r);
    double c = -1.0 / 6402373705728000.0;
    c = fma_k(c, z, 1.0 / 20922789888000.0);
    c = fma_k(c, z, -1.0 / 87178291200.0);
    c = fma_k(c, z, 1.0 / 479001600.0);
    c = fma_k(c, z, -1.0 / 3628800.0);
    c = fma_k(c, z, 1.0 / 40320.0);
    c = fma_k(c, z, -1.0 / 720.0);
    c = fma_k(c, z, 1.0 / 24.0);
    c = fma_k(c, z, -0.5);
    double cr = fma_k(z, c, 1.0);
    long long q = (long long)kd & 3;
    if (q == 0) { sn = sr; cs = cr; }
    else if (q == 1) { sn = cr; cs = -sr; }
    else if (q == 2) { sn = -sr; cs = -cr; }
    else { sn = -cr; cs = sr; }
}

MCRT_DEV float det_logf(float x) { return (float)det_log((double)x); }
MCRT_DEV float det_expf(float x) { return (float)det_exp((double)x); }

MCRT_DEV double det_pow_pos(double x, double y)   // x >= 0 (ray.cpp:223)
{
    if (y == 1.0) return x;
    if (y == 0.0) return 1.0;
    if (x == 0.0) return y > 0.0 ? 0.0 : dinf();
    return det_exp(y * det_log(x));
}

MCRT_DEV float det_powf(float x, float y)          // std::pow(float,float) ray.cpp:158,160
{
    if (y == 1.0f) return x;
    if (y == 0.0f) return 1.0f;
    if (x != x || y != y) return x + y;
    double ax = fabs((double)x);
    bool y_is_int = (floorf(y) == y);
    bool y_is_odd = y_is_int && fabsf(y) < 16777216.0f && (((long long)y) & 1);
    double r;
    if (ax == 0.0) r = (y > 0.0f) ? 0.0 : dinf();
    else r = det_exp((double)y * det_log(ax));
    if (x < 0.0f || (x == 0.0f && (__float_as_uint(x) >> 31))) {
        if (!y_is_int) return (x == 0.0f) ? (float)r : __uint_as_float(0x7fc00000u);
        if (y_is_odd) r = -r;
    }
    return (float)r;
}

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11)
MCRT_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int i = 0; i < 10; i++) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

MCRT_DEV double u53(uint32_t hi, uint32_t lo)
{
    uint64_t v = ((uint64_t)(hi >> 5) << 26) | (uint64_t)(lo >> 6);
    return (double)v * 0x1p-53;
}

}  // namespace mcrt
