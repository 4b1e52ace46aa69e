"""The oracle's task cut (scan-line x sample block, used when a host has more cores than the sample has scan-lines) does
not change its answers: hits, counts and the fixed-point image equal the one-thread run bit for bit."""
import numpy as np


def test_sample_block_tasks_equal_the_sequential_run(mcrt, orc):
    cfg, meshes = mcrt.synth.sphere_scene(2)
    sd = mcrt.scene_io.build_scene(cfg, meshes)
    E, S = 3, 37
    tr = mcrt.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
    tex = orc.texture(16)
    osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing)
    p = orc.default_params(n_elements=E, n_samples=S, tex_n=16)
    one = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=4, n_threads=1, want_ref=True)
    cut = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=4, n_threads=8, want_ref=False)       # 8 threads > E/2: sample blocks
    assert np.array_equal(one["hits"], cut["hits"])
    assert np.array_equal(one["rf_fix"], cut["rf_fix"]) and np.array_equal(one["rf_flags"], cut["rf_flags"])
    assert one["stats"] == cut["stats"]
    assert np.abs(one["rf_fix"]).sum() > 0
    # with the reference-order float image requested the cut is per scan-line (its sum order is the reference's)
    ref8 = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=4, n_threads=8, want_ref=True)
    assert np.array_equal(ref8["rf_ref"].view(np.uint32), one["rf_ref"].view(np.uint32))
