"""Scene JSON (the reference's on-disk format, scene.cpp:185-247 + main.cpp:65-69) and a minimal
Wavefront OBJ reader (positions + faces, fan triangulation, negative indices: the subset of
tiny_obj_loader the hot path depends on -- objloader.h:28-139 uses positions only)."""
import json
import os
import numpy as np

MATERIAL_FIELDS = ("impedance", "attenuation", "mu0", "mu1", "sigma", "specularity", "shininess", "thickness")
REQUIRED_KEYS = ("transducerPosition", "transducerAngles", "materials", "meshes", "origin", "spacing", "scaling", "startingMaterial")


class SceneError(RuntimeError):
    pass


def _atof(tok):
    """(float)atof(token), tiny_obj_loader.cpp:122-128: "nan" / "inf" are values, anything unreadable is 0"""
    try:
        return float(tok)
    except ValueError:
        return 0.0


def _atoi(tok):
    """atoi of a face corner up to its first '/': optional sign + leading digits, 0 when there are none"""
    head = tok.split("/")[0]
    n = 1 if head[:1] in "+-" else 0
    while n < len(head) and head[n].isdigit():
        n += 1
    try:
        return int(head[:n])
    except ValueError:
        return 0


def load_obj(path):
    """-> (V [n,3] float32, F [m,3] int32) in file order; polygons become fans (v0, v[k], v[k+1]).  Indices as the reference's
    loader fixes them (tiny_obj_loader.cpp:97-109): > 0 counts from one, < 0 from the end, 0 is the first vertex.  The reference then
    indexes unchecked; a corner outside the vertices read so far is a SceneError here."""
    verts, faces = [], []
    with open(path, "r", errors="replace") as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                c = [_atof(x) for x in t[1:4]] + [0.0, 0.0, 0.0]
                verts.append((c[0], c[1], c[2]))
            elif t[0] == "f":
                idx = []
                for tok in t[1:]:
                    i = _atoi(tok)
                    idx.append(i - 1 if i > 0 else (0 if i == 0 else len(verts) + i))
                for k in range(1, len(idx) - 1):
                    for i in (idx[0], idx[k], idx[k + 1]):
                        if not 0 <= i < len(verts):
                            raise SceneError("face index out of range in '%s'" % path)
                    faces.append((idx[0], idx[k], idx[k + 1]))
    return np.asarray(verts, np.float32).reshape(-1, 3), np.asarray(faces, np.int32).reshape(-1, 3)


def save_obj(path, V, F):
    with open(path, "w") as f:
        for v in np.asarray(V, np.float32):
            f.write("v %.9g %.9g %.9g\n" % (v[0], v[1], v[2]))
        for t in np.asarray(F):
            f.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))


class SceneData:
    """Flattened scene: what mcrt_upload_scene takes."""

    def __init__(self, tri, tri_mesh, meshes, materials, material_names, start_mat, spacing, config):
        self.tri = np.ascontiguousarray(tri, np.float32).reshape(-1, 9)
        self.tri_mesh = np.ascontiguousarray(tri_mesh, np.uint32)
        self.meshes = [tuple(int(x) for x in m) for m in meshes]       # (mat_inside, mat_outside, vascular)
        self.materials = np.ascontiguousarray(materials, np.float32).reshape(-1, 8)
        self.material_names = list(material_names)
        self.start_mat = int(start_mat)
        self.spacing = tuple(float(x) for x in spacing)
        self.config = config

    @property
    def n_tri(self):
        return self.tri.shape[0]


def parse_config(cfg):
    """scene.cpp:185-247: every key is mandatory except workingDirectory (json.at throws)."""
    try:
        for k in REQUIRED_KEYS:
            if k not in cfg:
                raise KeyError(f"key '{k}' not found")
        if not isinstance(cfg["materials"], list):
            raise SceneError("materials must be an array")
        if not isinstance(cfg["meshes"], list):
            raise SceneError("meshes must be an array")
        names, table = [], {}
        for m in cfg["materials"]:
            for k in ("name",) + MATERIAL_FIELDS:
                if k not in m:
                    raise KeyError(f"key '{k}' not found")
            if m["name"] not in table:
                names.append(m["name"])
            table[m["name"]] = [float(m[k]) for k in MATERIAL_FIELDS]
        mats = np.asarray([table[n] for n in names], np.float32).reshape(-1, 8)
        meshes = []
        for me in cfg["meshes"]:
            for k in ("file", "rigid", "vascular", "deltas", "material", "outsideMaterial", "outsideNormals"):
                if k not in me:
                    raise KeyError(f"key '{k}' not found")
            if me["material"] not in table or me["outsideMaterial"] not in table:
                raise KeyError("unknown material '%s'/'%s'" % (me["material"], me["outsideMaterial"]))
            meshes.append(dict(file=me["file"], vascular=bool(me["vascular"]), deltas=[float(x) for x in me["deltas"]],
                               mat_inside=names.index(me["material"]), mat_outside=names.index(me["outsideMaterial"])))
        if cfg["startingMaterial"] not in table:
            raise KeyError("unknown startingMaterial '%s'" % cfg["startingMaterial"])
    except (KeyError, TypeError, ValueError) as ex:
        raise SceneError("Error while loading scene: " + str(ex).strip('"')) from ex
    return names, mats, meshes, names.index(cfg["startingMaterial"])


def place_vertices(V, scaling, deltas, origin):
    """scene.cpp:313-324 in float32: v*scaling + (deltas*scaling*scaling + origin)."""
    s = np.float32(scaling)
    pos = (np.asarray(deltas, np.float32) * s * s + np.asarray(origin, np.float32)).astype(np.float32)
    return (np.asarray(V, np.float32) * s + pos).astype(np.float32)


def build_scene(cfg, mesh_provider=None):
    """cfg: scene dict (reference schema).  mesh_provider: {file name: (V, F)}; otherwise OBJ files are read
    from workingDirectory + file (scene.cpp:40)."""
    names, mats, meshes, start = parse_config(cfg)
    tris, tri_mesh, recs = [], [], []
    wd = cfg.get("workingDirectory", "")
    for mi, me in enumerate(meshes):
        if mesh_provider is not None and me["file"] in mesh_provider:
            V, F = mesh_provider[me["file"]]
        else:
            path = os.path.join(wd, me["file"]) if wd else me["file"]
            if not os.path.exists(path):
                raise SceneError("Error while loading scene: cannot read mesh '%s'" % path)
            V, F = load_obj(path)
        Vw = place_vertices(V, cfg["scaling"], me["deltas"], cfg["origin"])
        T = Vw[np.asarray(F, np.int64)].reshape(-1, 9)
        tris.append(T)
        tri_mesh.append(np.full(T.shape[0], mi, np.uint32))
        recs.append((me["mat_inside"], me["mat_outside"], int(me["vascular"])))
    tri = np.concatenate(tris) if tris else np.zeros((0, 9), np.float32)
    tm = np.concatenate(tri_mesh) if tri_mesh else np.zeros((0,), np.uint32)
    return SceneData(tri, tm, recs, mats, names, start, cfg["spacing"], cfg)


def load_scene_file(path, mesh_provider=None):
    with open(path) as f:
        cfg = json.load(f)
    return build_scene(cfg, mesh_provider)
