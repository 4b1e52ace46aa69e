"""Pins for the oracle's CORE -- the part of the hot path that cannot be compiled from the reference here (ray.cpp needs Bullet's
btVector3.h): SURVEY 8(c)'s analytic known-answer tests, and an independent second reading compared bit for bit.

  (i)   the material transition of ray.cpp:14-47, full truth table, quirks 1-2 included
  (ii)  Fresnel / Snell identities of ray.cpp:53-76,115-132: normal incidence, energy conservation, Snell's vector form, TIR
  (iii) random_unit_vector ray.cpp:167-211: w.v = cos(theta), |w| = 1, both swap branches
  (iv)  power_cosine_variate ray.cpp:213-224: the int parameter truncates the shininess; float exponent
  (v)   C1 first-hit distances against the analytic box face and the analytic sphere within the icosphere's tessellation bound
  (vi)  tests/ref_reading.py (written from the reference text, not from the oracle) == the oracle, bit for bit, on 10^5 random states
        of hit_boundary and on travel / max_ray_length / enlarge / accumulation, both under libm (orc.set_math_mode(1))
  (vii) MEASURED ZEROS: how often the contract's three additions act on the BASELINE frames (padded-bounds rule, |echo| >= 1024
        guard, random_unit_vector's 8-attempt cap)

All CPU; the oracle entry points used are the orc_debug_* exports, which run the same static functions the trace path runs."""
import json
import math
import os
import numpy as np
import pytest

import ref_reading as rr

F = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))

# name: impedance, attenuation, mu0, mu1, sigma, specularity, shininess, thickness
MATS = np.array([
    [1.99, 1e-8, 0.0, 0.0, 0.0, 1.0, 1000000, 0.0],       # 0 GEL
    [1.65, 0.7, 0.19, 1.0, 0.24, 1.0, 1000000, 0.0],      # 1 LIVER
    [7.8, 5.0, 0.78, 0.56, 0.1, 1.0, 1000000, 0.0],       # 2 BONE
    [1.61, 0.18, 0.001, 0.0, 0.01, 1.0, 1000000, 0.0],    # 3 BLOOD
    [1.38, 0.63, 0.5, 0.5, 0.0, 1.0, 1000000, 0.0],       # 4 FAT
    [1.62, 1.0, 0.4, 0.6, 0.3, 2.5, 2.9, 0.0],            # 5 rough: specularity 2.5, shininess 2.9 (-> int 2)
    [3.0, 1.0, 0.4, 0.6, 0.3, 1.0, 2000000000, 0.0],      # 6 mirror: shininess 2e9 -> cos(theta_r) rounds to exactly 1.0f
], np.float32)
# meshes: (material inside, material outside, vascular)
MESHES = [(1, 0, 0), (3, 1, 1), (2, 1, 0), (5, 4, 0), (6, 0, 0), (4, 1, 1)]
ONE_TRI = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32)


@pytest.fixture(scope="module")
def osc(orc):
    return orc.OracleScene(ONE_TRI, np.zeros(1, np.uint32), MESHES, MATS, 0)


def _unit(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v)).astype(np.float32)


def _reading_world():
    """the same materials and meshes as objects with identities (the reference's std::unordered_map elements and mesh references)"""
    mats = [rr.material(*row) for row in MATS]
    meshes = [rr.mesh(v, mats[i], mats[o]) for i, o, v in MESHES]
    return mats, meshes


# ------------------------------------------------------------------------------------------------------------------ (i)
def test_material_transition_truth_table(orc, osc):
    """ray.cpp:14-47.  Rows written from the reference lines:
       media_outside == nullptr (:32-46):  vascular mesh  -> after_vasc = &r.media (:38), material = mesh.inside (:39)
                                           regular mesh   -> after_vasc = nullptr (:43), material = mesh.inside: `&r.media == &mesh.inside`
                                                             compares the address of the ray's by-value copy (ray.h:17) with a map element,
                                                             never equal (:44, quirk 1) -- even when the ray IS in mesh.inside
       media_outside != nullptr (:17-31):  vascular mesh  -> after_vasc = nullptr (:21), material = *media_outside (:22); when media_outside
                                                             aliases the ray's own media (quirk 2) that is the ray's media
                                           regular mesh   -> material = r.media (:30); after_vasc = mesh.outside if media_outside ==
                                                             &mesh.inside else mesh.inside (:27); an aliasing pointer never equals a map element
       The refracted ray takes (material, after_vasc) (:94); the reflected ray keeps (r.media, r.media_outside) (:92)."""
    p = orc.default_params()
    NONE, SELF = orc.OUT_NONE, orc.OUT_SELF
    # (ray media, ray outside, mesh) -> (material after collision, media_outside after)
    table = [
        (0, NONE, 1, 3, SELF),      # GEL ray meets the BLOOD-in-LIVER vessel
        (0, NONE, 0, 1, NONE),      # GEL ray meets LIVER-in-GEL organ
        (1, NONE, 0, 1, NONE),      # quirk 1: a ray already in LIVER leaving the LIVER organ "enters" LIVER again, not GEL
        (3, SELF, 1, 3, NONE),      # quirk 2: leaving the vessel through a vascular mesh keeps BLOOD (the alias points at the ray itself)
        (3, SELF, 5, 3, NONE),      # ... any vascular mesh
        (3, 1, 1, 1, NONE),         # a stored map element IS honoured when leaving a vessel
        (3, SELF, 0, 3, 1),         # in a vessel, crossing a regular organ: stay BLOOD; alias != &LIVER -> surrounding := mesh.inside (LIVER)
        (3, 1, 0, 3, 0),            # stored surrounding == mesh.inside (LIVER) -> surrounding := mesh.outside (GEL)
        (3, 4, 0, 3, 1),            # stored surrounding FAT != mesh.inside -> surrounding := mesh.inside (LIVER)
        (3, 1, 2, 3, 2),            # BONE-in-LIVER organ, stored LIVER != BONE -> surrounding := BONE
        (1, NONE, 2, 2, NONE),      # LIVER ray meets BONE (a strong reflector: both branches of the choice occur)
        (2, NONE, 1, 3, SELF),      # BONE ray meets a vessel
    ]
    mats, meshes = _reading_world()
    seen = set()
    for media, outside, mesh, want_mat, want_out in table:
        for bounce in range(10):                                     # several draws: both the reflected and the refracted branch occur
            rng = (0x5EED, 7, 3, bounce, bounce)
            r = orc.ray_state((0, 0, 1), _unit((0.3, 0.1, -1)), media, outside, intensity=0.2)
            d = osc.hit_boundary(p, r, (0.2, 0.2, 0), (0, 0, 1), mesh, rng)
            assert (d.mat_after, d.after_vasc) == (want_mat, want_out), (media, outside, mesh)
            if d.chose_reflection:
                assert (d.returned.media, d.returned.outside) == (media, outside)
            else:
                assert (d.returned.media, d.returned.outside) == (want_mat, want_out)
            seen.add(bool(d.chose_reflection))
            # the second reading gets the same answer out of object identities
            own = rr.ray(rr.vec3(0, 0, 1), rr.vec3(*_unit((0.3, 0.1, -1))), mats[media], None, 0.2, 4.5, 0.0)
            own.media_outside = None if outside == NONE else (own.media if outside == SELF else mats[outside])
            u_pc, u_x = orc.rng_block(rng, 1)
            with np.errstate(all="ignore"):
                _, ret, ex = rr.hit_boundary(own, rr.vec3(0.2, 0.2, 0), rr.vec3(0, 0, 1), meshes[mesh], u_pc,
                                             [orc.rng_block(rng, 2 + k) for k in range(8)], u_x)
            assert ex["chose_reflection"] == bool(d.chose_reflection)
            got_media = [i for i, m in enumerate(mats) if m.values() == ret.media.values()]
            assert d.returned.media in got_media
            mo = ret.media_outside
            got_out = NONE if mo is None else (SELF if (mo is own.media or mo is ret.media) else [i for i, m in enumerate(mats) if m is mo][0])
            # (a refracted ray's alias points at the CALLER's slot, which scene.cpp:154 then overwrites with the returned ray: SELF)
            assert got_out == d.returned.outside
    assert seen == {True, False}


# ------------------------------------------------------------------------------------------------------------------ (ii)
def test_fresnel_normal_incidence_and_energy(orc, osc):
    """ray.cpp:59-76,126-132 at normal incidence: I_refl = I ((Z1 - Z2) / (Z1 + Z2))^2, I_refl + I_refr = I.  Mesh 4's material has
    shininess 2e9, so cos(theta_r) = u^(1/(2e9+1)) rounds to exactly 1.0f and random_unit_vector returns the surface normal itself."""
    p = orc.default_params()
    for media, I in [(0, 0.2), (1, 1.0), (2, 0.037), (4, 3e-6)]:
        for k in range(8):
            rng = (1, 2, k, 5, 1)
            r = orc.ray_state((0, 0, 1), (0, 0, -1), media, intensity=I)
            d = osc.hit_boundary(p, r, (0.2, 0.2, 0), (0, 0, 1), 4, rng)
            assert d.random_angle == 1.0 and tuple(d.random_normal) == (0.0, 0.0, 1.0) and d.incidence == 1.0 and not d.tir
            z1, z2 = float(MATS[media, 0]), float(MATS[6, 0])
            want = float(F(I)) * ((z1 - z2) / (z1 + z2)) ** 2
            assert d.intensity_refl == pytest.approx(want, rel=3e-7)
            assert F(d.intensity_refl) + F(d.intensity_refr) == pytest.approx(float(F(I)), rel=1.2e-7)      # one float rounding
            assert d.intensity_refr == F(I) - F(d.intensity_refl)
            # straight through / straight back
            assert tuple(d.refr_dir) == (0.0, 0.0, -1.0) and tuple(d.refl_dir) == (0.0, 0.0, 1.0)
            # Eq. 8 ray.cpp:154-164: max(pow(d.refr, spec), 0) + max(pow(d.refl, spec), 0), times cos(theta_r): 1 + max(-1, 0) = 1
            assert d.reflected_intensity == 1.0
            # the choice: reflect iff I_refl / I > x (ray.cpp:89-94)
            assert bool(d.chose_reflection) == (F(d.intensity_refl) / F(I) > F(d.u_x))
            assert d.returned.intensity == (d.intensity_refl if d.chose_reflection else d.intensity_refr)


def test_snell_vector_form_oblique(orc, osc):
    """ray.cpp:53-69,115-124 away from the normal, checked in double precision against the textbook: with n' the perturbed normal
    facing the ray, c1 = -d.n', rho = Z1/Z2 (the reference uses the impedance ratio as the refraction ratio): refraction direction is
    unit, coplanar with (d, n'), tangential part = rho * tangential part of d; reflection direction = d + 2 c1 n' (the mirror image);
    Fresnel (:126-132) with c2 = sqrt(1 - rho^2 (1 - c1^2))."""
    p = orc.default_params()
    rnd = np.random.default_rng(5)
    n_checked = 0
    for k in range(400):
        rng = (9, 9, k, 0, 3)
        d_in = _unit((rnd.normal(), rnd.normal(), -abs(rnd.normal()) - 0.2))
        media, mesh = [(0, 0), (1, 2), (4, 0), (2, 0)][k % 4]                       # rho > 1 and < 1
        I = float(rnd.uniform(1e-3, 1.0))
        d = osc.hit_boundary(p, orc.ray_state((0, 0, 1), d_in, media, intensity=I), (0.2, 0.2, 0), (0, 0, 1), mesh, rng)
        if d.tir:
            continue
        n1 = np.array(d.random_normal, np.float64); dd = d_in.astype(np.float64)
        assert abs(np.linalg.norm(n1) - 1) < 2e-6 and n1[2] > 0.99                  # shininess 1e6: within a fraction of a degree of the surface normal
        c1 = -dd @ n1
        assert c1 > 0 and d.incidence == pytest.approx(c1, abs=2e-7)
        z1, z2 = float(MATS[media, 0]), float(MATS[MESHES[mesh][0], 0])
        rho = z1 / z2
        c2 = math.sqrt(1 - rho * rho * (1 - c1 * c1))
        t = np.array(d.refr_dir, np.float64); m = np.array(d.refl_dir, np.float64)
        assert abs(np.linalg.norm(t) - 1) < 3e-7 and abs(np.linalg.norm(m) - 1) < 3e-7
        tang_in = dd + c1 * n1                                                       # d minus its normal component
        want_t = rho * tang_in - c2 * n1
        # Snell's vector form is unit only when |d| = 1 and |n'| = 1; the reference normalises afterwards (:66)
        assert np.allclose(t, want_t / np.linalg.norm(want_t), atol=2e-6)
        assert np.allclose(m, dd + 2 * c1 * n1, atol=2e-6)
        assert abs(-t @ n1 - c2) < 3e-6                                              # cos(theta_2)
        assert abs(np.linalg.norm(np.cross(t, n1)) - rho * np.linalg.norm(np.cross(dd, n1))) < 3e-6                # sin(theta_2) = rho sin(theta_1)
        R = ((z1 * c1 - z2 * c2) / (z1 * c1 + z2 * c2)) ** 2
        assert d.intensity_refl == pytest.approx(float(F(I)) * R, rel=2e-5, abs=1e-12)
        assert d.intensity_refr == F(I) - F(d.intensity_refl)
        n_checked += 1
    assert n_checked > 200                                                           # (the rest met total internal reflection: rho = 4.7 out of bone)


def test_total_internal_reflection(orc, osc):
    """ray.cpp:61-63,71-72,154-164: rho^2 (1 - c1^2) > 1 -> I_refl = I (the ray always reflects: I_refl / I = 1 > x), sqrt(negative) = NaN
    runs through snells_law into pow -> std::max(NaN, 0.f) returns its FIRST argument -> a NaN echo, unless params.sanitize_tir."""
    p = orc.default_params()
    d_in = _unit((1.0, 0.0, -0.2))                                                   # grazing, BONE (7.8) -> LIVER (1.65): rho = 4.7
    r = orc.ray_state((0, 0, 1), d_in, 2, intensity=0.5)
    d = osc.hit_boundary(p, r, (0.2, 0.2, 0), (0, 0, 1), 0, (1, 1, 1, 1, 1))
    assert d.tir and d.intensity_refl == F(0.5) and d.intensity_refr == 0.0 and d.chose_reflection
    assert math.isnan(d.refraction_angle) and all(math.isnan(x) for x in d.refr_dir) and math.isnan(d.reflected_intensity)
    assert d.returned.media == 2 and d.returned.intensity == F(0.5)
    assert np.allclose(np.array(d.returned.dir), np.array(d.refl_dir)) and abs(np.linalg.norm(d.refl_dir) - 1) < 3e-7
    p2 = orc.default_params(sanitize_tir=1)
    d2 = osc.hit_boundary(p2, r, (0.2, 0.2, 0), (0, 0, 1), 0, (1, 1, 1, 1, 1))
    assert d2.tir and math.isfinite(d2.reflected_intensity) and d2.reflected_intensity >= 0


# ------------------------------------------------------------------------------------------------------------------ (iii)
def test_random_unit_vector_polar_angle_and_length(orc):
    """ray.cpp:167-211.  SURVEY 8(c) asked for "w.v = cos(theta), |w| = 1 for unit v".  Writing the test showed that the REFERENCE's
    formula does not have that property in general: with e1 = (b, -vx vy, -vx vz), e2 = (0, vz, -vy) (b = 1 - vx^2) the vector at polar
    angle theta is  cos(theta) v - px e1 + py e2,  whose y and z components are  vy (cos + vx px) + vz py,  vz (cos + vx px) - vy py;
    ray.cpp:200 computes  d = cos_theta - vx * px  (minus).  Algebraically the reference's w therefore satisfies
        w.v = cos(theta) - 2 b vx px,        w = w_exact - 2 vx px (0, vy, vz)
    (vx = the smaller of the two first components after the swap of :188-193, px the scaled disc coordinate of :198).  It IS the
    documented identity when that component is zero, and within 2|vx| sqrt(b) sin(theta) <= sin(theta) of it otherwise -- a fraction
    of a degree at the scenes' shininess 1e6, which is why it goes unnoticed.  The oracle (and the GPU) reproduce the reference's
    formula, not the textbook's (bug-compatible).  Asserted here:
      A. smaller component exactly 0, both swap branches: w.v = cos(theta), |w| = 1 to 1e-6;
      B. any unit v: the oracle's w equals the reference's formula evaluated in double from the same draws to 2e-6, and
         w.v - cos(theta) = -2 b vx px to 1e-6;
      C. cos(theta) = 1 returns v itself; the azimuth covers all quadrants."""
    rnd = np.random.default_rng(11)
    # A
    worst = 0.0
    for k in range(2000):
        a, c = rnd.normal(size=2)
        v = _unit((0.0, a, c)) if k % 2 else _unit((a, 0.0, c))                       # k odd: no swap (|vx| = 0 <= |vy|); even: swap
        ct = float(F(rnd.uniform(0.05, 1.0)))
        w, attempts = orc.random_unit_vector(v, ct, (3, 4, k, 1, 2))
        assert attempts == 1                                                           # p = r^2 <= 0.25 by construction (:179-183)
        v64, w64 = v.astype(np.float64), w.astype(np.float64)
        v64 /= np.linalg.norm(v64)
        worst = max(worst, abs(w64 @ v64 - ct), abs(np.linalg.norm(w64) - 1))
    assert worst < 1e-6, worst
    # B
    branches = {True: 0, False: 0}
    worst_w = worst_id = worst_dev = 0.0
    for k in range(4000):
        v = _unit(rnd.normal(size=3))
        ct = float(F(rnd.uniform(0.05, 1.0))) if k % 4 else float(F(1 - 10 ** rnd.uniform(-7, -2)))       # incl. the near-specular range the scenes use
        rng = (3, 4, k, 1, 2)
        w, attempts = orc.random_unit_vector(v, ct, rng)
        assert attempts == 1
        u1, u2 = orc.rng_block(rng, 2)
        ang, r = u1 * 2 * math.pi, 0.5 * math.sqrt(u2)
        px, py = float(F(r * math.cos(ang))), float(F(r * math.sin(ang)))              # the float roundings of :181-182 kept, the rest in double
        vx, vy, vz = (float(x) for x in v)
        swap = abs(vx) > abs(vy)
        if swap:
            vx, vy = vy, vx
        bb = 1 - vx * vx
        cc = math.sqrt((1 - ct * ct) / ((px * px + py * py) * bb))
        px *= cc; py *= cc
        dd = ct - vx * px
        want = [vx * ct - bb * px, vy * dd + vz * py, vz * dd - vy * py]
        if swap:
            want[0], want[1] = want[1], want[0]
        scale = max(1.0, cc * 0.5)                                                     # c amplifies the float error of px, py when the disc point is near the centre
        worst_w = max(worst_w, float(np.abs(w.astype(np.float64) - np.array(want)).max()) / scale)
        v64 = v.astype(np.float64)
        worst_id = max(worst_id, abs((w.astype(np.float64) @ v64 - ct * (v64 @ v64)) - (-2 * bb * vx * px)) / scale)
        sin_t = math.sqrt(max(0.0, 1 - ct * ct))
        dev = abs(w.astype(np.float64) @ v64 - ct)
        assert dev <= 2 * abs(vx) * math.sqrt(bb) * sin_t + 2e-6 * scale
        if ct > 1 - 1e-5:
            worst_dev = max(worst_dev, dev)
        branches[swap] += 1
    assert min(branches.values()) > 1000
    assert worst_w < 2e-6 and worst_id < 2e-6, (worst_w, worst_id)
    assert worst_dev < 5e-3                                                            # the regime of every shipped scene (shininess 1e6)
    # C: cos(theta) = 1: the vector itself, bit for bit (c = sqrt(0 / ..) = 0)
    v = _unit((0.3, -0.5, 0.8))
    assert np.array_equal(orc.random_unit_vector(v, 1.0, (1, 1, 1, 1, 1))[0], v)
    # the azimuth really turns with the disc draw: over many draws the perpendicular part covers all quadrants
    v = np.array([0, 0, 1], np.float32)
    q = {(bool(w[0] > 0), bool(w[1] > 0)) for w in (orc.random_unit_vector(v, 0.5, (1, 1, k, 0, 0))[0] for k in range(64))}
    assert len(q) == 4


# ------------------------------------------------------------------------------------------------------------------ (iv)
def test_power_cosine_variate_int_parameter(orc, osc):
    """ray.cpp:213-224: `power_cosine_variate(int v)` called with the FLOAT shininess (:49) truncates it; exponent = float(1.0 / (v + 1))"""
    for u in (0.0, 1e-9, 0.25, 0.5, 0.999999, 1 - 2.0 ** -53):
        for shin, v in ((2.9, 2), (0.99, 0), (1000000.0, 1000000), (7.0, 7)):
            want = F(math.pow(u, float(F(1.0 / (v + 1)))))
            got = F(orc.power_cosine(int(F(shin)), u))
            assert abs(float(got) - float(want)) <= float(np.spacing(want)), (u, shin)
    # through hit_boundary: mesh 3's inside material has shininess 2.9 -> exponent 1/3, not 1/3.9
    p = orc.default_params()
    rng = (4, 4, 4, 4, 4)
    d = osc.hit_boundary(p, orc.ray_state((0, 0, 1), (0, 0, -1), 4, intensity=0.3), (0.2, 0.2, 0), (0, 0, 1), 3, rng)
    u_pc, _ = orc.rng_block(rng, 1)
    assert d.u_pc == u_pc
    assert d.random_angle == pytest.approx(u_pc ** float(F(1 / 3.0)), rel=2e-7)
    assert abs(d.random_angle - u_pc ** (1 / 3.9)) > 1e-3
    # ... and specularity 2.5 enters Eq. 8 as a float power (ray.cpp:157-160)
    a = float(np.dot(np.array(d.refr_dir, np.float32), np.array([0, 0, -1], np.float32)))
    b = float(np.dot(np.array(d.refl_dir, np.float32), np.array([0, 0, -1], np.float32)))
    want = (max(a, 0.0) ** 2.5 + (max(b, 0.0) ** 2.5 if b > 0 else 0.0)) * d.random_angle
    assert d.reflected_intensity == pytest.approx(want, rel=1e-5)


# ------------------------------------------------------------------------------------------------------------------ (v)
def test_c1_first_hits_against_analytic_box_and_sphere(mcrt, orc, sphere):
    """C1 (sphere scene, 32 scan-lines x 64 rays, brute force over all triangles): segment 0 ends on the box face x = -6, at the analytic
    ray / plane distance; a segment-1 that ends on the sphere ends between the analytic sphere of radius 2 (the icosphere's vertices) and
    the inscribed sphere of radius 2 cos(alpha), alpha = the largest angular circumradius of an icosphere triangle."""
    cfg, sd = sphere
    E, S = 32, 64
    tr = mcrt.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    p = orc.default_params(n_elements=E, n_samples=S, tex_n=16)
    o = osc.trace_frame(p, tr.pos, tr.dir, orc.texture(16), frame_id=1, use_bvh=0, n_threads=8, want_segs=True, want_ref=False, want_fix=False)
    segs, hits = o["segs"], o["hits"]
    sphere_tris = sd.tri[sd.tri_mesh == 1].reshape(-1, 3, 3).astype(np.float64)
    assert np.allclose(np.linalg.norm(sphere_tris, axis=2), 2.0, atol=1e-6)
    cent = sphere_tris.mean(1); cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    cos_alpha = min(float(np.min(np.einsum("tvk,tk->tv", sphere_tris / 2.0, cent))), 1.0)          # farthest vertex from its triangle's centre direction
    r_in = 2.0 * cos_alpha
    assert 2.0 - r_in < 1e-3
    n_box = n_sph = 0
    for e in range(E):
        pos, d = tr.pos[e].astype(np.float64), tr.dir[e].astype(np.float64)
        t_box = (-6.0 - pos[0]) / d[0]
        centre = abs(e - (E - 1) / 2) < 1                                           # the two central scan-lines: 7.5 cm from the probe centre
        for s in range(S):
            s0 = segs[e, s, 0]
            assert hits[e, s, 0] >= 0 and sd.tri_mesh[hits[e, s, 0]] == 0
            L0 = np.linalg.norm(s0["to"].astype(np.float64) - s0["from"].astype(np.float64))
            assert abs(L0 - t_box) < 2e-5 and abs(float(s0["to"][0]) + 6.0) < 2e-6
            if centre:
                assert abs(L0 + 3.0 - 7.5) < 2e-3                                   # probe radius 3 cm + 4.5 cm of gel
            n_box += 1
            if hits[e, s, 1] >= 0 and sd.tri_mesh[hits[e, s, 1]] == 1:
                s1 = segs[e, s, 1]
                o1, d1 = s1["from"].astype(np.float64), s1["dir"].astype(np.float64)
                d1 /= np.linalg.norm(d1)
                L1 = np.linalg.norm(s1["to"].astype(np.float64) - o1)

                def t_sphere(R):
                    b = o1 @ d1; disc = b * b - (o1 @ o1 - R * R)
                    return -b - math.sqrt(disc) if disc > 0 else math.inf
                assert t_sphere(2.0) - 3e-5 <= L1 <= t_sphere(r_in) + 3e-5, (e, s, L1, t_sphere(2.0), t_sphere(r_in))
                n_sph += 1
    assert n_box == E * S and n_sph > 100


# ------------------------------------------------------------------------------------------------------------------ (vi)
def _random_state(rnd, mats):
    media = int(rnd.integers(len(MATS)))
    k = int(rnd.integers(4))
    outside = [-1, -2, int(rnd.integers(len(MATS))), -1][k]
    d = _unit(rnd.normal(size=3))
    n = _unit(rnd.normal(size=3))
    if rnd.random() < 0.3:                                       # near-grazing and near-normal incidences
        n = _unit(-d.astype(np.float64) + rnd.normal(size=3) * 10 ** rnd.uniform(-4, 0))
    I = float(F(10 ** rnd.uniform(-10.5, 0)))
    return media, outside, d, n, I


def test_second_reading_equals_oracle_bit_for_bit(orc, osc):
    """10^5 random (ray, normal, mesh, draws) states through hit_boundary: every output of the oracle equals the independent reading's,
    bit for bit, with both on libm.  MCRT_READING_STATES overrides the count."""
    n_states = int(os.environ.get("MCRT_READING_STATES", "100000"))
    p = orc.default_params()
    mats, meshes = _reading_world()
    rnd = np.random.default_rng(2026)
    orc.set_math_mode(1)
    n_tir = n_refl = n_swap = 0
    try:
        with np.errstate(all="ignore"):
            for k in range(n_states):
                media, outside, d, n, I = _random_state(rnd, mats)
                mesh = int(rnd.integers(len(MESHES)))
                rng = (0x5EED, k >> 16, k & 0xffff, k % 7, k % 10)
                hp = rnd.uniform(-5, 5, 3).astype(np.float32)
                dist = float(rnd.uniform(0, 150))
                got = osc.hit_boundary(p, orc.ray_state((0, 0, 0), d, media, outside, intensity=I, dist_mm=dist), hp, n, mesh, rng)
                own = rr.ray(rr.vec3(0, 0, 0), rr.vec3(*d), mats[media], None, I, 4.5, dist)
                own.media_outside = None if outside == -1 else (own.media if outside == -2 else mats[outside])
                u_pc, u_x = orc.rng_block(rng, 1)
                back, ret, ex = rr.hit_boundary(own, rr.vec3(*hp), rr.vec3(*n), meshes[mesh], u_pc,
                                                (orc.rng_block(rng, 2 + a) for a in range(8)), u_x)

                def same(a, b):
                    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))
                ctx = (k, media, outside, mesh)
                assert same(got.random_angle, ex["random_angle"]), ctx
                assert same(list(got.random_normal), ex["random_normal"].tuple()), ctx
                assert same(got.incidence, ex["incidence"]) and same(got.refr_ratio, ex["refr_ratio"]), ctx
                assert same(got.refraction_angle, ex["refraction_angle"]) or (math.isnan(got.refraction_angle) and np.isnan(ex["refraction_angle"])), ctx
                for a, b in ((got.refr_dir, ex["refr_dir"]), (got.refl_dir, ex["refl_dir"])):
                    a = np.array(list(a), np.float32); b = np.array(b.tuple(), np.float32)
                    assert np.array_equal(np.isnan(a), np.isnan(b)) and same(a[~np.isnan(a)], b[~np.isnan(b)]), ctx
                assert same(got.intensity_refl, ex["intensity_refl"]) and same(got.intensity_refr, ex["intensity_refr"]), ctx
                assert bool(got.tir) == ex["tir"] and bool(got.chose_reflection) == ex["chose_reflection"] and got.ruv_attempts == ex["turns"], ctx
                assert same(got.reflected_intensity, back) or (math.isnan(got.reflected_intensity) and np.isnan(back)), ctx
                assert same(got.returned.intensity, ret.intensity) and got.returned.dist_mm == float(ret.distance_traveled), ctx
                assert same(list(got.returned.origin), ret.origin.tuple()), ctx
                a = np.array(list(got.returned.dir), np.float32); b = np.array(ret.direction.tuple(), np.float32)
                assert np.array_equal(np.isnan(a), np.isnan(b)) and same(a[~np.isnan(a)], b[~np.isnan(b)]), ctx
                assert tuple(MATS[got.returned.media]) == tuple(float(x) for x in ret.media.values()), ctx
                mo = ret.media_outside
                want_out = -1 if mo is None else (-2 if (mo is own.media or mo is ret.media) else [i for i, m in enumerate(mats) if m is mo][0])
                assert got.returned.outside == want_out, ctx
                n_tir += ex["tir"]; n_refl += ex["chose_reflection"]; n_swap += bool(abs(n[0]) > abs(n[1]))
    finally:
        orc.set_math_mode(0)
    # the sample exercised every branch
    assert n_tir > n_states // 100 and n_states // 20 < n_refl < n_states and n_states // 4 < n_swap < 3 * n_states // 4


def test_second_reading_travel_and_ray_segment(orc, osc):
    """travel (ray.cpp:99-103) over distance_in_mm (scene.cpp:281-290); max_ray_length (ray.cpp:110-113) + enlarge (scene.cpp:292-298) +
    the 0.1 offset (scene.cpp:115): oracle == second reading bit for bit under libm, 20 000 states, anisotropic spacing included."""
    rnd = np.random.default_rng(7)
    mats, _ = _reading_world()
    orc.set_math_mode(1)
    try:
        for spacing in ((1.0, 1.0, 1.0), (0.5, 1.25, 2.0)):
            sc = orc.OracleScene(ONE_TRI, np.zeros(1, np.uint32), MESHES, MATS, 0, spacing=spacing)
            p = orc.default_params()
            for k in range(10000):
                media = int(rnd.integers(len(MATS)))
                o = rnd.uniform(-15, 15, 3).astype(np.float32); d = _unit(rnd.normal(size=3))
                I = float(F(10 ** rnd.uniform(-9.9, 0))); dist = float(rnd.uniform(0, 150))
                to = (o + d * F(rnd.uniform(0.1, 20))).astype(np.float32)
                st = orc.ray_state(o, d, media, intensity=I, dist_mm=dist)
                own = rr.ray(rr.vec3(*o), rr.vec3(*d), mats[media], None, I, 4.5, dist)
                L, f_off, seg_to = sc.ray_segment(p, st)
                w_from, w_to, w_L = rr.ray_test_segment(spacing, own)
                assert F(L).view(np.uint32) == F(w_L).view(np.uint32)
                assert np.array_equal(f_off.view(np.uint32), np.array(w_from.tuple(), np.float32).view(np.uint32))
                assert np.array_equal(seg_to.view(np.uint32), np.array(w_to.tuple(), np.float32).view(np.uint32))
                mm = sc.travel(st, to)
                w_mm = rr.distance_in_mm(spacing, own.origin, rr.vec3(*to))
                rr.travel(own, w_mm)
                assert mm == float(w_mm) and st.dist_mm == float(own.distance_traveled)
                assert F(st.intensity).view(np.uint32) == own.intensity.view(np.uint32)
    finally:
        orc.set_math_mode(0)


def test_second_reading_accumulation(orc, osc, golden):
    """main.cpp:106-144 + rfimage.h:33-40 + volume.h:46-61 for single segments: the RF line the oracle accumulates equals the second
    reading's, bit for bit under libm -- steps from an (unsigned) cast, the float position walk, the double time axis against the
    INTEGER-micrometre row pitch, the exp attenuation, the boundary echo at steps - 1 (unsigned wrap at steps == 0), negative
    coordinates into the texture index."""
    rnd = np.random.default_rng(3)
    tex = orc.texture(16)
    mats, _ = _reading_world()
    p = orc.default_params(tex_n=16)
    c = orc.constants()
    consts = {k: getattr(c, k) for k in ("axial_res_f", "axial_res_mm", "time_step_us", "row_dt_us", "max_travel_us")}
    # the unit-typed constants themselves are pinned to the reference-compiled units.h
    assert consts["time_step_us"] == golden["time_step_us"] and consts["row_dt_us"] == golden["row_dt_us"] and consts["max_travel_us"] == golden["max_travel_time_us"]
    orc.set_math_mode(1)
    n_wrapped = n_cut = 0
    try:
        for k in range(300):
            media = int(rnd.integers(len(MATS)))
            o = rnd.uniform(-12, 12, 3).astype(np.float32); d = _unit(rnd.normal(size=3))
            length = [0.0, 0.01, float(rnd.uniform(0.05, 4)), float(rnd.uniform(4, 16))][k % 4]      # 0 steps (the wrap), a few, many, past 100 us
            to = (o + d * F(length)).astype(np.float32)
            seg = np.zeros(1, orc.SEGMENT_DTYPE)
            seg["from"] = o; seg["to"] = to; seg["dir"] = d
            seg["reflected_intensity"] = F(rnd.uniform(0, 2)); seg["initial_intensity"] = F(10 ** rnd.uniform(-6, 0))
            seg["attenuation"] = MATS[media, 1]; seg["distance_traveled"] = float(rnd.uniform(0, 160)); seg["media"] = media; seg["tri"] = -1
            rf, n = osc.accumulate_segment(p, tex, seg)
            want = np.zeros(p.n_rows, np.float32)
            steps = rr.accumulate_segment(want, consts, tex, F(p.tex_res), dict(
                origin=rr.vec3(*o), to=rr.vec3(*to), direction=rr.vec3(*d), reflected_intensity=seg["reflected_intensity"][0],
                initial_intensity=seg["initial_intensity"][0], attenuation=seg["attenuation"][0], distance_traveled=seg["distance_traveled"][0],
                media=mats[media]), p.n_samples, p.sos, p.frequency)
            assert n == steps
            assert np.array_equal(rf.view(np.uint32), want.view(np.uint32)), k
            n_wrapped += (length < 0.02 and steps == 0); n_cut += (length > 4 and steps < int(length * 10 / consts["axial_res_mm"]))
    finally:
        orc.set_math_mode(0)
    assert n_wrapped > 20 and n_cut > 5


def test_overload_resolution_fixture():
    """ray.cpp:188,197 call abs / sqrt unqualified on floats.  Compiled against the reference's own units.h + mesh.h (oracle/
    ref_overload_probe.cpp), g++ resolves them to the INT abs and the double sqrt unless a header includes <math.h> directly -- which
    Bullet's btScalar.h does [upstream-memory].  The oracle and the second reading take the float overloads; this fixture records
    what the claim rests on (and that the double sqrt variant gives the same float: sqrt is exactly rounded at both widths)."""
    with open(os.path.join(HERE, "golden", "overloads.json")) as f:
        ov = json.load(f)
    assert ov["btscalar"] == {"abs_returns_float": 1, "abs_of_minus_0p7": pytest.approx(0.7, rel=1e-6), "sqrt_returns_float": 1}
    assert ov["none"]["abs_returns_float"] == 0 and ov["none"]["abs_of_minus_0p7"] == 0
    x = np.random.default_rng(1).uniform(0, 1e6, 200000).astype(np.float32)
    assert np.array_equal(np.sqrt(x), np.sqrt(x.astype(np.float64)).astype(np.float32))


# ------------------------------------------------------------------------------------------------------------------ (vii)
def _counted_frame(mcrt, orc, cfg, sd, E, S, tex, frame, threads=8, e_range=None):
    tr = mcrt.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    nodes, btri, n4, _ = mcrt.host_build_bvh4(sd.tri, sd.tri_mesh)
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri))
    osc.set_bvh4(n4)
    p = orc.default_params(n_elements=E, n_samples=S)
    osc.counting(True)
    try:
        e0, e1 = e_range or (0, E)
        o = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=frame, e_begin=e0, e_end=e1, use_bvh=2, n_threads=threads, want_hits=False, want_ref=False)
        c = orc.counters()
    finally:
        osc.counting(False)
    c.update(queries=o["stats"]["queries"], hits=o["stats"]["hits"], rf_steps=o["stats"]["rf_steps"], nan_bins=int(o["rf_flags"].sum()))
    return c


def test_measured_zeros_on_the_baseline_frames(mcrt, orc, tex256):
    """How often do the contract's additions to the reference's behaviour ACT on the frames BASELINE.json names?  Counted in the oracle
    (the product is bit-identical to it on these frames, tests/test_gpu_baseline_configs.py):
      * the padded-bounds rule turning away a candidate the bare processTriangle tests accept, hit point inside the scene bounds
      * an echo refused by the |e| < 1024 guard of the fixed-point bins
      * random_unit_vector going round its loop again (p > 0.25) / giving up after 8 attempts
    headline = 1 M triangles 128 x 1024; C3 = liver scene 128 x 4096; C5 = liver scene 512 x 16384 (the whole frame: 8.4 M paths,
    38 M closest-hit queries, ~1 minute of 8 cores).  The numbers are quoted in DESIGN.md 3."""
    out = {}
    cfg, meshes = mcrt.synth.random_scene(1_000_000, 8, 12345)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    out["headline"] = _counted_frame(mcrt, orc, cfg, sd, 128, 1024, tex256, frame=0)
    del sd, meshes
    cfg, meshes = mcrt.synth.liver_scene(5)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    out["C3"] = _counted_frame(mcrt, orc, cfg, sd, 128, 4096, tex256, frame=0)
    out["C5"] = _counted_frame(mcrt, orc, cfg, sd, 512, 16384, tex256, frame=0)
    print("\nmeasured zeros:", json.dumps(out))
    for name, c in out.items():
        assert c["queries"] > 100000 and c["hits"] > 0, name
        assert c["pad_rule_rejects_in_bounds"] == 0, (name, c)
        assert c["echo_guard_trips"] == 0 and c["ruv_retries"] == 0 and c["ruv_giveups"] == 0, (name, c)
    # TIR does happen on these scenes (NaN echoes are the reference's own behaviour, quirk 5) -- the count is reported, not asserted zero
