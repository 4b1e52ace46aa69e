"""Scene JSON / OBJ ingestion (the reference's on-disk format) -- host logic, no GPU."""
import json
import numpy as np
import pytest


def test_scene_schema_errors_mirror_reference(mcrt):
    """scene.cpp:185-247 uses json.at(): any missing key throws, wrapped as 'Error while loading scene: ...' (scene.cpp:19-26).
    examples/ircad11/ircad11.scene lacks shininess/thickness and fails exactly like that in the reference."""
    cfg, meshes = mcrt.synth.sphere_scene(1)
    for m in cfg["materials"]:
        del m["thickness"]
    with pytest.raises(mcrt.scene_io.SceneError) as e:
        mcrt.scene_io.build_scene(cfg, meshes)
    assert str(e.value) == "Error while loading scene: key 'thickness' not found"
    cfg, meshes = mcrt.synth.sphere_scene(1)
    del cfg["spacing"]
    with pytest.raises(mcrt.scene_io.SceneError):
        mcrt.scene_io.build_scene(cfg, meshes)
    cfg, meshes = mcrt.synth.sphere_scene(1)
    cfg["materials"] = {"not": "an array"}
    with pytest.raises(mcrt.scene_io.SceneError):
        mcrt.scene_io.build_scene(cfg, meshes)
    cfg, meshes = mcrt.synth.sphere_scene(1)
    cfg["meshes"][0]["material"] = "UNOBTAINIUM"
    with pytest.raises(mcrt.scene_io.SceneError):
        mcrt.scene_io.build_scene(cfg, meshes)


def test_obj_reader_and_placement(mcrt, orc, tmp_path):
    p = tmp_path / "t.obj"
    p.write_text("# quad + triangle with negative indices\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1 4/4/1\nf -4 -3 -2\n")
    V, F = mcrt.scene_io.load_obj(str(p))
    assert V.shape == (4, 3) and F.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2]]          # fan triangulation, file order
    Vr, Fr = mcrt.synth.icosphere(2, 1.5, (0.5, -1, 2))
    mcrt.scene_io.save_obj(str(tmp_path / "s.obj"), Vr, Fr)
    V2, F2 = mcrt.scene_io.load_obj(str(tmp_path / "s.obj"))
    assert np.array_equal(V2, Vr) and np.array_equal(F2, Fr)
    # placement scene.cpp:313-324: v*scaling + (deltas*scaling^2 + origin), float32, identical to the oracle's restatement
    deltas, origin, s = (152.533512115, 174.472991943, 105.106495678), (-18.0, -22.0, -5.0), 0.1
    a = mcrt.scene_io.place_vertices(Vr, s, deltas, origin)
    b = orc.place_vertices(Vr, s, deltas, origin).reshape(-1, 3)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_scene_tables(mcrt):
    cfg, meshes = mcrt.synth.liver_scene(1)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    assert len(sd.meshes) == 11 and sd.material_names[sd.start_mat] == "GEL"
    names = sd.material_names
    assert sum(v for _, _, v in sd.meshes) == 3                                          # aorta, cava, porta are vascular
    assert sd.materials[names.index("BONE")][7] == np.float32(0.3)                      # thickness (santi-liver.scene:59)
    assert sd.tri.shape[0] == sd.tri_mesh.shape[0] == 11 * 80
    assert sd.tri_mesh.max() == 10 and np.all(np.diff(sd.tri_mesh.astype(int)) >= 0)   # meshes in scene order
