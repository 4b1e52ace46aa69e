import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import mcray_tracing_amd as m
from oracle import orc
orc.build()
cfg, meshes = m.synth.sphere_scene(5)
sd = m.scene_io.build_scene(cfg, meshes)
tex = orc.texture(256)
for (E, S) in ((32, 64), (16, 128), (8, 64)):
    tr = m.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    p = orc.default_params(n_elements=E, n_samples=S)
    o = osc.trace_frame(p, tr.pos, tr.dir, tex, use_bvh=False, n_threads=8, want_segs=True)
    for rep in range(4):
        sim = m.Simulator(sd, tr, n_samples=S, texture=tex)
        for it in range(3):
            hits, segs, cnt = sim.ctx.trace_frame_debug(0, sim.rf_dev, want_segs=True)
            bad = hits != o["hits"]
            print(E, S, "rep", rep, "it", it, "mismatching hits", int(bad.sum()), "per bounce", bad.sum(axis=(0, 1)).tolist(), "cnt equal", bool(np.array_equal(cnt, o["seg_count"])),
                  "live per bounce gpu", (hits != -2).sum(axis=(0, 1)).tolist(), "oracle", (o["hits"] != -2).sum(axis=(0, 1)).tolist(), flush=True)
        sim.close()
