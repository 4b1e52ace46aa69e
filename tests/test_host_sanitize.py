"""The product's host-side C++ under AddressSanitizer + UBSan (CPU only; GPU sanitizers are not available on the pool): the SAH builder and
BVH4 collapse, the row / texture / PSF / transducer / scan-map tables (csrc/mcrt_host.cpp, compiled by itself) and the JSON + OBJ readers of
the host shim (host/mcrt_host.hpp) on the fixtures and on a malformed corpus -- truncated and damaged scene files, deeply nested JSON, \\u
escapes, `f` corners as v/vt/vn with negative, zero and out-of-range indices, NaN coordinates, missing keys with the reference's
"Error while loading scene: ..." wrapping (scene.cpp:19-26,185-247; tiny_obj_loader.cpp:97-187,504-717).  tests/host/host_sanitize_driver.cpp
is the driver; the unsanitized library must give the same digests."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mcray-tracing_amd")


def fnv(b, h=1469598103934665603):
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def run_driver(tmp_path, mcrt):
    exe = str(tmp_path / "host_sanitize_driver")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "host"), "-o", exe,
                           os.path.join(ROOT, "tests", "host", "host_sanitize_driver.cpp"), os.path.join(PKG, "csrc", "mcrt_host.cpp")])
    cfg, _ = mcrt.synth.liver_scene(1)
    cfg = dict(cfg, workingDirectory=str(tmp_path) + "/")
    scene = tmp_path / "liver.scene"
    scene.write_text(json.dumps(cfg, indent=1))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "tricky.obj"), str(scene)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.rstrip().endswith("DONE"), r.stdout[-3000:] + r.stderr[-6000:]
    assert "runtime error" not in r.stderr and "Sanitizer" not in r.stderr, r.stderr[-6000:]
    out = {}
    for line in r.stdout.splitlines()[:-1]:
        k, _, v = line.partition(": ")
        out[k] = v
    return out, cfg


def test_host_code_runs_clean_under_asan_ubsan(mcrt, tmp_path):
    out, cfg = run_driver(tmp_path, mcrt)
    err = lambda k: out[k].startswith("error: ")
    # ---- OBJ: the fixture the reference's own loader is pinned on (tests/golden/obj_soup.json), through the sanitized reader
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "obj_soup.json")))
    V, F = mcrt.scene_io.load_obj(os.path.join(ROOT, "tests", "golden", "tricky.obj"))
    soup = V[F].astype(np.float32).reshape(-1, 9)
    assert out["obj.tricky"] == "ok %d triangles, fnv %016x" % (len(soup), fnv(soup.tobytes()))
    n_gold = gold.get("n_triangles", gold.get("triangles") if isinstance(gold.get("triangles"), int) else None)
    if n_gold is not None:
        assert len(soup) == n_gold
    assert out["obj.missing_file"] == "error: cannot read mesh '/nonexistent/mesh.obj'"
    assert out["obj.empty"].startswith("ok 0 triangles") and out["obj.forms"].startswith("ok 4 triangles") and out["obj.negative"].startswith("ok 2 triangles")
    assert out["obj.index_zero_is_first_vertex"].startswith("ok 1 triangles")
    assert err("obj.index_past_end") and err("obj.index_before_start") and err("obj.index_huge") and "out of range" in out["obj.index_past_end"]
    assert out["obj.nan_and_inf"].startswith("ok 2 triangles, 1 NaN")
    assert out["obj.short_lines"] == "error: face index out of range in '<case>'"          # ("f 1 2 3 4" names vertices that were never read)
    assert out["obj.crlf_tabs_polygon"].startswith("ok 3 triangles")
    assert out["obj.garbage"].startswith(("ok", "error: face index out of range")) and out["obj.random_bytes"].startswith(("ok", "error: face index out of range"))
    # the Python reader follows the same rules
    for name, text in (("zero", "v 5 6 7\nv 1 0 0\nv 0 1 0\nf 0 2 3\n"), ("nan", "v nan 0 0\nv 1 inf 0\nv 0 1 -inf\nv 1 1 1\nf 1 2 3\nf 2 3 4\n"),
                       ("forms", "v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\nf 1//1 2//1 4//1\nf 1/1 3/1 4/1\nf 2 3 4\n")):
        p = tmp_path / (name + ".obj"); p.write_text(text)
        Vp, Fp = mcrt.scene_io.load_obj(str(p))
        key = {"zero": "obj.index_zero_is_first_vertex", "nan": "obj.nan_and_inf", "forms": "obj.forms"}[name]
        assert out[key].endswith("fnv %016x" % fnv(Vp[Fp].astype(np.float32).tobytes())), (name, out[key])
    p = tmp_path / "past.obj"; p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n")
    try:
        mcrt.scene_io.load_obj(str(p)); raise AssertionError("no error")
    except mcrt.scene_io.SceneError as ex:
        assert "out of range" in str(ex)

    # ---- JSON + scene.cpp's error wrapping
    assert out["json.scene"] == "ok %d materials, %d meshes, start %d" % (len(cfg["materials"]), len(cfg["meshes"]), [m_["name"] for m_ in cfg["materials"]].index(cfg["startingMaterial"]))
    n_err, rest = out["json.truncated_everywhere"].split(" errors, ")
    assert rest.startswith("0 parsed") and int(n_err) > 1000                       # every proper prefix of a scene file is an error, none a crash
    assert int(out["json.one_byte_damaged_everywhere"].split(" errors")[0]) > 1000
    assert out["json.nested_arrays_1e6"].endswith("nested too deeply") and out["json.nested_objects_1e5"].endswith("nested too deeply") and out["json.nested_at_limit"] == "ok"
    want = "aé€\U0001F600\n\t\\/\"\b\f\r".encode("utf-8")
    assert out["json.escapes"] == "ok fnv %d bytes %d" % (fnv(want), len(want))
    for k, v in out.items():
        if k.startswith("json.bad "): assert v.startswith("error: parse error"), (k, v)
        if k.startswith("json.good "): assert v == "ok", (k, v)
    for key in ("transducerPosition", "origin", "spacing", "startingMaterial", "scaling", "materials", "meshes"):
        assert out["scene.missing " + key] == "error: Error while loading scene: key '%s' not found" % key
    assert out["scene.without_workingDirectory"] == "ok"
    assert out["scene.materials_not_array"] == "error: Error while loading scene: materials must be an array"
    assert out["scene.meshes_not_array"] == "error: Error while loading scene: meshes must be an array"
    assert out["scene.scaling_string"] == "error: Error while loading scene: type must be number"
    assert out["scene.origin_short"].startswith("error: Error while loading scene: ") and out["scene.origin_number"] == "error: Error while loading scene: type must be array"
    assert out["scene.unknown_starting_material"] == "error: Error while loading scene: key 'UNOBTAINIUM' not found"
    assert out["scene.material_without_shininess"] == "error: Error while loading scene: key 'shininess' not found"      # (what examples/ircad11/ircad11.scene does in the reference)
    assert out["scene.mesh_unknown_material"] == "error: Error while loading scene: key 'NOPE' not found"
    assert out["scene.mesh_file_missing"] == "error: cannot read mesh '%s/nonexistent/a.obj'" % cfg["workingDirectory"]      # workingDirectory + file, scene.cpp:40

    # ---- builders and tables: structurally sound under the sanitizers, and the same digests as the shipped (unsanitized) library
    for k in ("bvh.tricky", "bvh.one_triangle", "bvh.5000_identical", "bvh.300_points_at_origin", "bvh.random_20000", "bvh.random_20000_times_1e30", "bvh.random_20000_degenerate"):
        assert out[k].startswith("ok bvh2 "), (k, out[k])
    assert out["bvh.random_20000_with_nan_inf"].startswith(("ok bvh2 ", "error: ")), out["bvh.random_20000_with_nan_inf"]      # refused or built: never a wild write
    assert err("bvh.zero_triangles") and err("bvh.null_arguments") and err("bvh4.null_arguments")
    assert err("tables.row_thresholds_bad") and err("tables.texture_null") and err("tables.transducer_zero") and err("tables.scan_maps_bad")
    L = mcrt._lib.lib() if hasattr(mcrt._lib, "lib") else mcrt.load_library()
    thr = np.zeros(466, np.float64)
    L.mcrt_row_thresholds.argtypes = [C.c_double, C.c_uint32, C.c_void_p]
    assert L.mcrt_row_thresholds(145.0 / 1500.0 * 0.001 * 1000.0, 465, thr.ctypes.data) == 0
    assert out["tables.row_thresholds"] == "ok fnv %d" % fnv(thr.tobytes())
    vox = np.zeros(2 * 512, np.float32)
    L.mcrt_generate_texture.argtypes = [C.c_void_p, C.c_uint32]
    assert L.mcrt_generate_texture(vox.ctypes.data, 8) == 0
    assert out["tables.texture_8"] == "ok fnv %d" % fnv(vox.tobytes())
    ax, lat = np.zeros(7, np.float32), np.zeros(13, np.float32)
    L.mcrt_psf_kernels.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    assert L.mcrt_psf_kernels(4.5, 0.05, 0.2, 145, ax.ctypes.data, 7, lat.ctypes.data, 13) == 0
    assert out["tables.psf"] == "ok fnv %d" % fnv(ax.tobytes(), fnv(lat.tobytes()))
    mr, mc = np.zeros(2000, np.float32), np.zeros(2000, np.float32)
    L.mcrt_scan_maps.argtypes = [C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    assert L.mcrt_scan_maps(128, 465, 30.0, 1.0471975511965976, 100, 1500, 40, 50, mr.ctypes.data, mc.ctypes.data) == 0
    assert out["tables.scan_maps"] == "ok fnv %d" % fnv(mr.tobytes(), fnv(mc.tobytes()))
    # (the transducer digest is not compared across builds: sinf / cosf of the sanitized -O1 build and of the hipcc build may differ in the last place)
    assert out["tables.transducer_512"].startswith("ok fnv ")
