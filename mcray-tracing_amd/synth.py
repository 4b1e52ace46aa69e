"""Synthetic geometry and scene descriptions for tests and bench.py.

The reference ships scene JSON files but NOT the meshes they name (SPHERE.obj, BOX.obj, liver.obj ...
are absent, .gitignore:48-50), and the ircad11 dataset cannot be fetched.  Everything here is generated
and clearly synthetic; material values and poses restate the facts in the reference's example scenes
(examples/sphere/sphere.scene, examples/ircad11/santi-liver.scene)."""
import numpy as np

# name: impedance, attenuation, mu0, mu1, sigma, specularity, shininess, thickness  (sphere.scene:5-127)
_MATERIALS = [
    ("GEL", 1.99, 1e-8, 0.0, 0.0, 0.0, 1.0, 1000000, 0.0),
    ("AIR", 0.0004, 1.64, 0.78, 0.56, 0.1, 1.0, 1000000, 0.0),
    ("FAT", 1.38, 0.63, 0.5, 0.5, 0.0, 1.0, 1000000, 0.0),
    ("LIVER", 1.65, 0.7, 0.19, 1.0, 0.24, 1.0, 1000000, 0.0),
    ("BONE", 7.8, 5.0, 0.78, 0.56, 0.1, 1.0, 1000000, 0.0),
    ("BLOOD", 1.61, 0.18, 0.001, 0.0, 0.01, 1.0, 1000000, 0.0),
    ("VESSEL", 1.99, 1.09, 0.2, 0.1, 0.2, 1.0, 1000000, 0.0),
    ("KIDNEY", 1.62, 1.0, 0.4, 0.6, 0.3, 1.0, 1000000, 0.0),
    ("SUPRARRENAL", 1.62, 1.0, 0.4, 0.6, 0.3, 1.0, 1000000, 0.0),
    ("GALLBLADDER", 1.62, 1.0, 0.4, 0.6, 0.3, 1.0, 1000000, 0.0),
    ("SKIN", 1.99, 1.0, 0.4, 0.6, 0.3, 1.0, 1000000, 0.0),
]
_FIELDS = ("impedance", "attenuation", "mu0", "mu1", "sigma", "specularity", "shininess", "thickness")


def materials(overrides=None):
    out = []
    for row in _MATERIALS:
        m = {"name": row[0]}
        m.update({k: v for k, v in zip(_FIELDS, row[1:])})
        if overrides and row[0] in overrides:
            m.update(overrides[row[0]])
        out.append(m)
    return out


def _mesh(file, material, outside, vascular=False, deltas=(0.0, 0.0, 0.0)):
    return {"file": file, "rigid": True, "vascular": vascular, "deltas": list(deltas), "material": material,
            "outsideMaterial": outside, "outsideNormals": True}


# ---------------------------------------------------------------- meshes
def icosphere(subdiv=5, radius=2.0, center=(0.0, 0.0, 0.0)):
    """20 * 4^subdiv triangles; outward winding."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    V = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], np.float64)
    V /= np.linalg.norm(V, axis=1, keepdims=True)
    F = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], np.int64)
    for _ in range(subdiv):
        a, b, c = F[:, 0], F[:, 1], F[:, 2]
        edges = np.sort(np.concatenate([np.stack([a, b], 1), np.stack([b, c], 1), np.stack([c, a], 1)]), axis=1)
        uniq, inv = np.unique(edges, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        mid = V[uniq[:, 0]] + V[uniq[:, 1]]
        mid /= np.linalg.norm(mid, axis=1, keepdims=True)
        n0 = V.shape[0]
        V = np.concatenate([V, mid])
        nF = F.shape[0]
        ab, bc, ca = n0 + inv[:nF], n0 + inv[nF:2 * nF], n0 + inv[2 * nF:]
        F = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1), np.stack([ab, bc, ca], 1)])
    Vf = (V * radius + np.asarray(center, np.float64)).astype(np.float32)
    return Vf, F.astype(np.int32)


def box(half=(6.0, 6.0, 6.0), center=(0.0, 0.0, 0.0)):
    """axis-aligned box, 12 triangles, outward winding"""
    hx, hy, hz = half
    V = np.array([[-hx, -hy, -hz], [hx, -hy, -hz], [hx, hy, -hz], [-hx, hy, -hz],
                  [-hx, -hy, hz], [hx, -hy, hz], [hx, hy, hz], [-hx, hy, hz]], np.float64) + np.asarray(center, np.float64)
    F = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5],
                  [2, 3, 7], [2, 7, 6], [3, 0, 4], [3, 4, 7]], np.int32)
    return V.astype(np.float32), F


def blob(subdiv, radius, center, seed, roughness=0.25, stretch=(1.0, 1.0, 1.0)):
    """closed organ-like surface: icosphere displaced by a few low-frequency lobes (deterministic)."""
    V, F = icosphere(subdiv, 1.0)
    rng = np.random.Generator(np.random.PCG64(seed))
    d = V.astype(np.float64)
    disp = np.zeros(d.shape[0])
    for _ in range(6):
        k = rng.normal(size=3); k /= np.linalg.norm(k)
        disp += rng.uniform(0.3, 1.0) * np.cos(rng.uniform(1.0, 3.0) * (d @ k) * np.pi + rng.uniform(0, 6.28))
    disp = 1.0 + roughness * disp / 6.0 * 2.0
    P = d * disp[:, None] * radius * np.asarray(stretch) + np.asarray(center, np.float64)
    return P.astype(np.float32), F


def random_triangles(n, seed=12345, lo=(-10.0, -8.0, -8.0), hi=(5.0, 8.0, 8.0), edge=0.1):
    """centroids uniform in the box, edge vectors uniform in [-edge, edge]^3 (PCG64)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    c = rng.uniform(lo, hi, size=(n, 3))
    e1 = rng.uniform(-edge, edge, size=(n, 3)); e2 = rng.uniform(-edge, edge, size=(n, 3))
    v0 = c - (e1 + e2) / 3.0
    V = np.stack([v0, v0 + e1, v0 + e2], 1).reshape(-1, 3).astype(np.float32)
    F = np.arange(3 * n, dtype=np.int32).reshape(-1, 3)
    return V, F


# ---------------------------------------------------------------- scenes (reference JSON schema)
def sphere_scene(sphere_subdiv=5, overrides=None):
    """examples/sphere/sphere.scene: BOX (LIVER in GEL) + SPHERE (BONE in LIVER), probe at (-13.5,0,0) angles (0,0,-90)."""
    cfg = {"transducerPosition": [-13.5, 0.0, 0.0], "transducerAngles": [0.0, 0.0, -90.0], "materials": materials(overrides),
           "meshes": [_mesh("BOX.obj", "LIVER", "GEL"), _mesh("SPHERE.obj", "BONE", "LIVER")],
           "origin": [0.0, 0.0, 0.0], "spacing": [1.0, 1.0, 1.0], "scaling": 1.0, "startingMaterial": "GEL"}
    meshes = {"BOX.obj": box((6.0, 6.0, 6.0)), "SPHERE.obj": icosphere(sphere_subdiv, 2.0)}
    return cfg, meshes


def liver_scene(subdiv=5, seed=11):
    """Stand-in for examples/ircad11/santi-liver.scene: same materials (BONE thickness 0.3), pose, origin, scaling 0.1,
    mesh names / material pairs / vascular flags; the organ meshes themselves are procedural blobs (SYNTHETIC)."""
    mats = materials({"GEL": {"impedance": 1.38}, "BONE": {"thickness": 0.3}})
    names = [("aorta.obj", "BLOOD", "FAT", True), ("bones.obj", "BONE", "FAT", False), ("liver.obj", "LIVER", "FAT", False),
             ("cava.obj", "BLOOD", "FAT", True), ("right_kidney.obj", "KIDNEY", "SKIN", False), ("left_kidney.obj", "KIDNEY", "SKIN", False),
             ("right_suprarrenal.obj", "SUPRARRENAL", "FAT", False), ("left_suprarrenal.obj", "SUPRARRENAL", "FAT", False),
             ("gallbladder.obj", "GALLBLADDER", "FAT", False), ("skin.obj", "FAT", "GEL", False), ("porta.obj", "BLOOD", "FAT", True)]
    # world-space layout in front of the probe (world = local*0.1 + deltas*0.01 + origin); the probe sits at
    # (-17.5,1,5) looking along +x after the (120,0,-90) deg rotation.
    cfg = {"transducerPosition": [-17.5, 1.0, 5.0], "transducerAngles": [120.0, 0.0, -90.0], "materials": mats, "meshes": [],
           "origin": [-18.0, -22.0, -5.0], "spacing": [1.0, 1.0, 1.0], "scaling": 0.1, "startingMaterial": "GEL"}
    rng = np.random.Generator(np.random.PCG64(seed))
    meshes = {}
    # centres in WORLD cm; local coordinates are (world - origin)/0.1 with deltas = 0
    layout = {"skin.obj": ((-6.0, 1.0, 5.0), 9.0, (1.0, 1.0, 1.0)), "liver.obj": ((-8.0, 1.5, 5.5), 4.0, (1.0, 1.2, 1.0)),
              "bones.obj": ((-2.0, -1.0, 4.0), 1.6, (1.0, 2.0, 1.0)), "aorta.obj": ((-7.5, 2.0, 5.0), 0.8, (0.6, 2.5, 0.6)),
              "cava.obj": ((-9.0, 0.0, 6.0), 0.7, (0.6, 2.5, 0.6)), "porta.obj": ((-8.5, 3.0, 6.5), 0.5, (2.0, 0.6, 0.6)),
              "right_kidney.obj": ((-4.5, 4.5, 3.0), 1.4, (1.0, 1.5, 1.0)), "left_kidney.obj": ((-4.5, -3.5, 7.0), 1.4, (1.0, 1.5, 1.0)),
              "right_suprarrenal.obj": ((-5.5, 5.5, 5.0), 0.6, (1.0, 1.0, 1.0)), "left_suprarrenal.obj": ((-5.5, -4.5, 5.0), 0.6, (1.0, 1.0, 1.0)),
              "gallbladder.obj": ((-10.0, 3.0, 4.0), 0.8, (1.0, 1.0, 1.5))}
    origin = np.asarray(cfg["origin"], np.float64)
    for i, (f, mat, out, vasc) in enumerate(names):
        c, r, st = layout[f]
        V, F = blob(subdiv, r, c, seed=int(rng.integers(1 << 30)) + i, stretch=st)
        Vl = ((V.astype(np.float64) - origin) / 0.1).astype(np.float32)
        meshes[f] = (Vl, F)
        cfg["meshes"].append(_mesh(f, mat, out, vasc))
    return cfg, meshes


def random_scene(n_tri=1_000_000, n_mesh=8, seed=12345, edge=0.1):
    """n_mesh meshes of n_tri/n_mesh random triangles in front of the probe at (-13.5,0,0).
    edge: half-range of the edge vectors in cm (0.1 = the 1 M-triangle scene of SURVEY 8(d); the 16 M-triangle streaming scene uses
    0.025, the same total area, so a ray crosses as much tissue before it meets a triangle and walks a tree two levels deeper)."""
    cyc = [("LIVER", "GEL", False), ("FAT", "LIVER", False), ("KIDNEY", "FAT", False), ("BONE", "LIVER", False),
           ("BLOOD", "LIVER", True), ("GALLBLADDER", "FAT", False), ("SKIN", "GEL", False), ("VESSEL", "LIVER", True)]
    cfg = {"transducerPosition": [-13.5, 0.0, 0.0], "transducerAngles": [0.0, 0.0, -90.0], "materials": materials(), "meshes": [],
           "origin": [0.0, 0.0, 0.0], "spacing": [1.0, 1.0, 1.0], "scaling": 1.0, "startingMaterial": "GEL"}
    per = n_tri // n_mesh
    meshes = {}
    for i in range(n_mesh):
        n = per if i < n_mesh - 1 else n_tri - per * (n_mesh - 1)
        f = "random_%d.obj" % i
        meshes[f] = random_triangles(n, seed + i, edge=edge)
        mat, out, vasc = cyc[i % len(cyc)]
        cfg["meshes"].append(_mesh(f, mat, out, vasc))
    return cfg, meshes
