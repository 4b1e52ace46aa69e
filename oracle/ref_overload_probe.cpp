// ref_overload_probe.cpp -- ORACLE-SIDE TOOL (test infrastructure).
//
// Which functions do the UNQUALIFIED calls `abs(vx)` and `sqrt(radicando)` of the reference's ray.cpp:188,197 (float arguments)
// resolve to?  ray.cpp sees: "ray.h" (-> <LinearMath/btVector3.h> [Bullet, absent], <units/units.h>, "mesh.h"), <cmath>, <iostream>,
// <random>, and no `using namespace std`.  This translation unit includes the same headers of the reference, from where they lie,
// minus Bullet's; what Bullet's LinearMath/btScalar.h adds to the picture is its DIRECT `#include <math.h>` and `#include <stdlib.h>`
// [upstream-memory: Bullet is not under /root/reference], which oracle/gen_golden.py supplies through -DPROBE_PRELUDE_MATH_H /
// -DPROBE_PRELUDE_STDLIB_H.  Printed as JSON; the three variants are stored in tests/golden/overloads.json.
#ifdef PROBE_PRELUDE_MATH_H
#include <math.h>
#endif
#ifdef PROBE_PRELUDE_STDLIB_H
#include <stdlib.h>
#endif
#include <units/units.h>
#include "mesh.h"
#include <cmath>
#include <iostream>
#include <random>
#include <type_traits>
#include <cstdio>

int main()
{
    float vx = -0.7f, radicando = 2.0f;
    auto a = abs(vx);
    auto s = sqrt(radicando);
    std::printf("{\"abs_returns_float\":%d,\"abs_of_minus_0p7\":%.9g,\"sqrt_returns_float\":%d}\n",
                (int)std::is_same<decltype(a), float>::value, (double)a, (int)std::is_same<decltype(s), float>::value);
    return 0;
}
