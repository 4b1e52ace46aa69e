#!/usr/bin/env python3
"""Regenerates tests/golden/ref_probe.json and tests/golden/obj_soup.json from the REFERENCE's own sources.

Runs oracle/_ref/ref_probe (built by `make -C oracle ref` from /root/reference/src/psf.h,
src/volume.h and include/units/units.h, compiled where they lie).  Only works in the
container that mounts /root/reference; the JSON it writes is the committed fixture that
pins the oracle (tests/test_oracle_golden.py) everywhere else.
"""
import json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
subprocess.check_call(["make", "-C", here, "ref"])
probe = os.path.join(here, "_ref", "ref_probe")
if not os.path.exists(probe):
    sys.exit("reference not mounted; cannot regenerate goldens")
ref = os.environ.get("MCRT_REFERENCE", "/root/reference")
scenes = []
for sub in ("sphere", "ircad11"):
    d = os.path.join(ref, "examples", sub)
    for f in sorted(os.listdir(d)):
        if f.endswith(".scene"):
            scenes.append("%s=%s" % (f, os.path.join(d, f)))
data = json.loads(subprocess.check_output([probe] + scenes))
data["_generated_by"] = "oracle/gen_golden.py (oracle/ref_probe.cpp compiled against /root/reference headers)"
out = os.path.join(here, "..", "tests", "golden", "ref_probe.json")
with open(out, "w") as f:
    json.dump(data, f, separators=(",", ":"))
    f.write("\n")
print("wrote", os.path.normpath(out), os.path.getsize(out), "bytes")

# the reference's OBJ path (tiny_obj_loader.cpp + the conversion loop of objloader.h) on tests/golden/tricky.obj
obj = os.path.join(here, "..", "tests", "golden", "tricky.obj")
soup = json.loads(subprocess.check_output([os.path.join(here, "_ref", "ref_obj_probe"), obj]))
soup["_generated_by"] = "oracle/gen_golden.py (oracle/ref_obj_probe.cpp + the reference's src/wavefront/tiny_obj_loader.cpp) on tests/golden/tricky.obj"
out = os.path.join(here, "..", "tests", "golden", "obj_soup.json")
with open(out, "w") as f:
    json.dump(soup, f, separators=(",", ":"))
    f.write("\n")
print("wrote", os.path.normpath(out), os.path.getsize(out), "bytes")

# overload resolution of ray.cpp's unqualified abs(float) / sqrt(float) under the reference's own headers, with and without the
# direct <math.h> / <stdlib.h> includes Bullet's btScalar.h contributes (oracle/ref_overload_probe.cpp)
ov = {k: json.loads(subprocess.check_output([os.path.join(here, "_ref", "ref_overload_" + k)])) for k in ("none", "stdlib", "btscalar")}
ov["_generated_by"] = "oracle/gen_golden.py (oracle/ref_overload_probe.cpp against /root/reference/include/units/units.h + src/mesh.h; g++ " + \
    subprocess.check_output(["g++", "-dumpfullversion"]).decode().strip() + ")"
out = os.path.join(here, "..", "tests", "golden", "overloads.json")
with open(out, "w") as f:
    json.dump(ov, f, separators=(",", ":"))
    f.write("\n")
print("wrote", os.path.normpath(out), os.path.getsize(out), "bytes")
