"""An INDEPENDENT SECOND READING of the reference's ray physics (test infrastructure).

Written from the reference's source text -- /root/reference/src/ray.cpp:11-224, src/scene.cpp:102-169,281-298,342-346,
src/main.cpp:106-144, src/rfimage.h:33-40, src/volume.h:46-61 -- statement by statement, NOT from oracle/mcrt_oracle.c: its purpose is
to catch a transcription slip in the oracle (an operand promoted differently, an expression re-associated, a branch swapped) that a
reader's eye passes over.  tests/test_oracle_physics.py runs both on the same random states and compares bit for bit.

How C++ is mirrored:
  * every float variable is an np.float32 scalar, every double an np.float64 (F(), D()); an expression mixing the two is promoted
    by numpy exactly as C++ promotes it (float op double -> double); integer literals adopt the other operand's type in both
    languages.  No np.float32 expression is ever contracted into a fused multiply-add (the reference is built without -march).
  * libm is called through ctypes on the C library itself (powf / expf / logf / pow / sin / cos): numpy's own vectorised
    transcendental functions are NOT glibc's and differ in the last bit.  The oracle is switched to libm (orc.set_math_mode(1))
    for the comparison, so both sides call the same libm the reference calls.
  * pointers are Python object identities.  `material` is a small mutable class; the materials map holds one object per name, a ray
    holds its media BY VALUE (ray.h:17) = its own object, refreshed in place by assignment -- so `&r.media == &mesh.material_inside`
    (ray.cpp:44) is `is` between two different objects, false as in C++, and `material_after_vascularities = &r.media` (ray.cpp:38)
    keeps following the ray's own slot (quirks 1 and 2 of SURVEY 8(a) fall out of the language semantics instead of being encoded).
  * the random draws are ARGUMENTS (the reference seeds a fresh mt19937 from random_device for each; the contract feeds
    counter-based uniforms instead): u_pc (ray.cpp:220), the disc pairs (ray.cpp:178-179), x (ray.cpp:88).
  * btVector3 (Bullet, absent from /root/reference) is the scalar path [upstream-memory]: dot = x*x' + y*y' + z*z' left to right,
    normalized() = v * (1 / sqrt(dot(v, v))), s * v = (v.x*s, v.y*s, v.z*s), distance = length(to - from).
  * unqualified abs(float) / sqrt(float) at ray.cpp:188,197 are the float overloads given Bullet's direct <math.h> include
    (tests/golden/overloads.json, variant "btscalar").
"""
import ctypes as C
import ctypes.util
import numpy as np

F = np.float32
D = np.float64

_libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
for _n in ("powf", "expf", "logf"):
    getattr(_libm, _n).restype = C.c_float
_libm.powf.argtypes = [C.c_float, C.c_float]
_libm.expf.argtypes = [C.c_float]
_libm.logf.argtypes = [C.c_float]
for _n in ("pow", "sin", "cos", "sqrt"):
    getattr(_libm, _n).restype = C.c_double
_libm.pow.argtypes = [C.c_double, C.c_double]
_libm.sin.argtypes = [C.c_double]
_libm.cos.argtypes = [C.c_double]


def powf(x, y): return F(_libm.powf(float(x), float(y)))
def expf(x): return F(_libm.expf(float(x)))
def logf(x): return F(_libm.logf(float(x)))
def pow_d(x, y): return D(_libm.pow(float(x), float(y)))
def sin_d(x): return D(_libm.sin(float(x)))
def cos_d(x): return D(_libm.cos(float(x)))


M_PI = D(3.14159265358979323846)     # glibc <math.h>; ray.cpp does not include psf.h, whose 3.14159 macro it therefore never sees
INTENSITY_EPSILON = F(1e-10)         # ray.h:24: static constexpr float intensity_epsilon = 1e-10


class material:                      # mesh.h:7-10
    __slots__ = ("impedance", "attenuation", "mu0", "mu1", "sigma", "specularity", "shininess", "thickness")

    def __init__(self, *v):
        for k, x in zip(self.__slots__, v):
            setattr(self, k, F(x))

    def assign(self, other):         # C++ copy-assignment into an existing object: the address stays, the contents change
        for k in self.__slots__:
            setattr(self, k, getattr(other, k))

    def values(self):
        return tuple(getattr(self, k) for k in self.__slots__)


class mesh:                          # mesh.h:12-20
    def __init__(self, is_vascular, material_inside, material_outside):
        self.is_vascular = bool(is_vascular)
        self.material_inside = material_inside       # references into the materials map (scene.cpp:239-240)
        self.material_outside = material_outside


class vec3:                          # btVector3, scalar path
    __slots__ = ("x", "y", "z")

    def __init__(self, x, y, z):
        self.x, self.y, self.z = F(x), F(y), F(z)

    def dot(self, o): return self.x * o.x + self.y * o.y + self.z * o.z
    def __neg__(self): return vec3(-self.x, -self.y, -self.z)
    def __add__(self, o): return vec3(self.x + o.x, self.y + o.y, self.z + o.z)
    def __sub__(self, o): return vec3(self.x - o.x, self.y - o.y, self.z - o.z)
    def scaled(self, s): return vec3(self.x * s, self.y * s, self.z * s)           # v * s and s * v
    def length(self): return np.sqrt(self.dot(self))
    def normalized(self): return self.scaled(F(1.0) / self.length())
    def distance(self, o): return (o - self).length()
    def tuple(self): return (self.x, self.y, self.z)


class ray:                           # ray.h:13-26 (depth, parent_collision, null left out: they do not enter the arithmetic)
    def __init__(self, origin, direction, media, media_outside, intensity, frequency, distance_traveled):
        self.origin, self.direction = origin, direction
        self.media = material(*media.values())       # BY VALUE: the ray's own object
        self.media_outside = media_outside           # pointer: None, a map element, or some ray's own media object
        self.intensity, self.frequency = F(intensity), F(frequency)
        self.distance_traveled = D(distance_traveled)


# ray.cpp:213-224
def power_cosine_variate(v, number):
    assert isinstance(v, int)
    number = D(number)
    indice = v + 1
    exponente = F(D(1.0) / D(indice))                # float exponente = (double)1.0 / indice;
    return F(pow_d(number, D(exponente)))            # return pow(number, exponente);   (double, float) -> pow(double, double) -> float


# ray.cpp:167-211.  draws = iterable of (u1, u2) pairs, one per turn of the do/while; returns (w, turns taken)
def random_unit_vector(v, cos_theta, draws):
    cos_theta = F(cos_theta)
    flag = False
    turns = 0
    for u1, u2 in draws:
        a = D(u1) * 2 * M_PI                         # double a = distribution(generator) * 2 * M_PI;
        r = D(0.5) * np.sqrt(D(u2))                  # double r = 0.5 * sqrt(distribution(generator));
        px = F(r * cos_d(a))                         # px = r * cos(a);    float px
        py = F(r * sin_d(a))
        p = px * px + py * py
        turns += 1
        if p <= F(0.25):                             # while (! (p <= 0.25) );   float p against the double literal: exact either way
            break
    vx, vy, vz = v.x, v.y, v.z
    if np.abs(vx) > np.abs(vy):                      # abs(float): see the module docstring
        vx = vy
        vy = v.x
        flag = True
    b = 1 - vx * vx
    radicando = 1 - cos_theta * cos_theta
    radicando = radicando / (p * b)
    c = np.sqrt(radicando)                           # float sqrt (a double sqrt rounded to float gives the same float)
    px = px * c
    py = py * c
    d = cos_theta - vx * px
    wx = vx * cos_theta - b * px
    wy = vy * d + vz * py
    wz = vz * d - vy * py
    if flag:
        aux = wy
        wy = wx
        wx = aux
    return vec3(wx, wy, wz), turns


# ray.cpp:115-124
def snells_law(ray_direction, surface_normal, incidence_angle, refraction_angle, refr_ratio):
    l, n, c, r = ray_direction, surface_normal, F(incidence_angle), F(refr_ratio)
    return l.scaled(r) + n.scaled(r * c - refraction_angle)


# ray.cpp:126-132
def reflection_intensity(intensity_in, media_1, incidence_angle, media_2, refracted_angle):
    num = media_1 * incidence_angle - media_2 * refracted_angle
    denom = media_1 * incidence_angle + media_2 * refracted_angle
    return F(D(intensity_in) * pow_d(D(num / denom), D(2)))      # float * pow(float, int) -> double product, returned as float


# ray.cpp:154-164 (the btVector3 overload, Eq. 8)
def reflected_intensity(direction, refraction_direction, reflection_direction, colliding_media):
    refraction_angle = direction.dot(refraction_direction)
    refraction_factor = powf(refraction_angle, colliding_media.specularity)
    reflection_angle = direction.dot(reflection_direction)
    reflection_factor = powf(reflection_angle, colliding_media.specularity)

    def std_max(a, b):               # std::max(a, b) = (a < b) ? b : a   -> a NaN first operand is returned
        return b if a < b else a
    return std_max(refraction_factor, F(0.0)) + std_max(reflection_factor, F(0.0))


# ray.cpp:11-97.  u_pc, disc_draws, x: the draws of ray.cpp:220, :178-179, :88 in the order the function makes them
def hit_boundary(r, hit_point, surface_normal, collided_mesh, u_pc, disc_draws, x):
    material_after_vascularities = None
    if r.media_outside is not None:
        if collided_mesh.is_vascular:
            material_after_vascularities = None
            material_after_collision = r.media_outside                       # *r.media_outside
        else:
            material_after_vascularities = (collided_mesh.material_outside if r.media_outside is collided_mesh.material_inside
                                            else collided_mesh.material_inside)
            material_after_collision = r.media
    else:
        if collided_mesh.is_vascular:
            material_after_vascularities = r.media                           # &r.media
            material_after_collision = collided_mesh.material_inside
        else:
            material_after_vascularities = None
            material_after_collision = (collided_mesh.material_outside if r.media is collided_mesh.material_inside
                                        else collided_mesh.material_inside)
    random_angle = power_cosine_variate(int(material_after_collision.shininess), u_pc)      # float -> int parameter: truncation
    random_normal, turns = random_unit_vector(surface_normal, random_angle, disc_draws)
    incidence_angle = r.direction.dot(-random_normal)
    if incidence_angle < 0:
        incidence_angle = r.direction.dot(random_normal)
    refr_ratio = r.media.impedance / material_after_collision.impedance
    refraction_angle = 1 - refr_ratio * refr_ratio * (1 - incidence_angle * incidence_angle)
    total_internal_reflection = bool(refraction_angle < 0)
    refraction_angle = np.sqrt(refraction_angle)
    refraction_direction = snells_law(r.direction, random_normal, incidence_angle, refraction_angle, refr_ratio)
    refraction_direction = refraction_direction.normalized()
    reflection_direction = r.direction + random_normal.scaled(2 * incidence_angle)
    reflection_direction = reflection_direction.normalized()
    intensity_refl = (r.intensity if total_internal_reflection else
                      reflection_intensity(r.intensity, r.media.impedance, incidence_angle, material_after_collision.impedance, refraction_angle))
    intensity_refr = r.intensity - intensity_refl
    back_to_transducer_intensity = reflected_intensity(r.direction, refraction_direction, reflection_direction, material_after_collision) * random_angle
    x = F(D(x))                                                               # float x = distribution2(generator2);
    reflection_probabilily = intensity_refl / r.intensity
    if reflection_probabilily > x:
        returned = ray(hit_point, reflection_direction, r.media, r.media_outside,
                       intensity_refl if intensity_refl > INTENSITY_EPSILON else F(0.0), r.frequency, r.distance_traveled)
        chose_reflection = True
    else:
        returned = ray(hit_point, refraction_direction, material_after_collision, material_after_vascularities,
                       intensity_refr if intensity_refr > INTENSITY_EPSILON else F(0.0), r.frequency, r.distance_traveled)
        chose_reflection = False
    extras = dict(random_angle=random_angle, random_normal=random_normal, incidence=incidence_angle, refr_ratio=refr_ratio,
                  refraction_angle=refraction_angle, refr_dir=refraction_direction, refl_dir=reflection_direction,
                  intensity_refl=intensity_refl, intensity_refr=intensity_refr, tir=total_internal_reflection,
                  chose_reflection=chose_reflection, turns=turns)
    return F(back_to_transducer_intensity), returned, extras


# ray.cpp:99-103
def travel(r, mm):
    mm = D(mm)
    r.distance_traveled = r.distance_traveled + mm
    r.intensity = r.intensity * expf(-r.media.attenuation * (F(mm) * F(0.01)) * r.frequency)


# ray.cpp:110-113
def max_ray_length(r):
    return F(10.0) * logf(INTENSITY_EPSILON / r.intensity) / -r.media.attenuation * r.frequency


# scene.cpp:281-290 (inside `using namespace std`: abs / pow / sqrt are the std:: overloads; pow(float, int) is double)
def distance_in_mm(spacing, v1, v2):
    x_dist = np.abs(v1.x - v2.x) * F(spacing[0])
    y_dist = np.abs(v1.y - v2.y) * F(spacing[1])
    z_dist = np.abs(v1.z - v2.z) * F(spacing[2])
    return np.sqrt(pow_d(D(x_dist), D(2)) + pow_d(D(y_dist), D(2)) + pow_d(D(z_dist), D(2))) * 10


# scene.cpp:292-298
def enlarge(spacing, versor, mm):
    return vec3(F(spacing[0]) * versor.x, F(spacing[1]) * versor.y, F(spacing[2]) * versor.z).scaled(F(mm) / F(100.0))


# scene.cpp:112-117: the segment handed to rayTest
def ray_test_segment(spacing, r):
    r_length = max_ray_length(r)
    to = r.origin + enlarge(spacing, r.direction, r_length)
    return r.origin + r.direction.scaled(F(0.1)), to, r_length


# volume.h:46-61 on a texture array [n][n][n][2] = (texture_noise, scattering_probability)
def get_scattering(tex, resolution, scattering_density, scattering_mu, scattering_sigma, x_millis, y_millis, z_millis):
    size = tex.shape[0]

    def index(q):                    # static_cast<unsigned int>(float): x86-64 cvttss2si to 64 bits, low 32 kept
        q = float(q)
        i = -(1 << 63) if not (abs(q) < 9.2233720368547758e18) else int(q)
        return (i & 0xffffffff) % size
    x, y, z = index(F(x_millis) / resolution), index(F(y_millis) / resolution), index(F(z_millis) / resolution)
    noise, prob = tex[x, y, z]
    return noise * scattering_sigma + scattering_mu if prob >= scattering_density else F(0.0)


# main.cpp:106-144 for one segment into one RF line (rfimage.h:33-40 add_echo).  consts: the unit-typed constants, which are pinned
# to the reference-compiled units.h by tests/golden/ref_probe.json (axial_resolution as float and double, time_step, row dt, max
# travel time); seg: dict(origin, to, direction: vec3; reflected_intensity, initial_intensity, attenuation: float; distance_traveled:
# double; media: material)
def accumulate_segment(rf, consts, tex, tex_resolution, seg, samples_te, sos, transducer_frequency):
    axial_resolution_f, axial_resolution_mm = F(consts["axial_res_f"]), D(consts["axial_res_mm"])
    time_step, row_dt, max_travel_time = D(consts["time_step_us"]), D(consts["row_dt_us"]), D(consts["max_travel_us"])
    max_rows = len(rf)

    def add_echo(echo, micros_from_source):
        row = micros_from_source / row_dt
        if row < max_rows:
            rf[int(row)] += F(echo)

    starting_micros = D(seg["distance_traveled"]) * 1000.0 / D(sos)                       # mm -> um (x1000), / (um/us)
    distance = D(seg["origin"].distance(seg["to"]) * F(10.0))                             # scene::distance scene.cpp:342-346
    q = float(distance / axial_resolution_mm)
    steps = 0 if not (abs(q) < 9.2233720368547758e18) else (int(q) & 0xffffffff)           # (unsigned int)(double)
    delta_step = seg["direction"].scaled(axial_resolution_f)
    point = seg["origin"]
    time_elapsed = starting_micros
    intensity = F(seg["initial_intensity"])
    m = seg["media"]
    step = 0
    while step < steps and time_elapsed < max_travel_time:
        scattering = get_scattering(tex, tex_resolution, m.mu1, m.mu0, m.sigma, point.x, point.y, point.z)
        add_echo(intensity * scattering, time_elapsed)
        point = point + delta_step
        time_elapsed = time_elapsed + time_step
        k = F(1.0)
        intensity = intensity * expf(-F(seg["attenuation"]) * axial_resolution_f * F(0.01) * F(transducer_frequency) * k)
        step += 1
    add_echo(F(seg["reflected_intensity"]) / F(samples_te), starting_micros + time_step * D((steps - 1) & 0xffffffff))
    return step
