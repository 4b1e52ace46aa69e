#!/usr/bin/env python3
"""k_march's texture gathers COUNTED on the CPU for several device layouts of the tissue texture, before any is built (VERDICT r5, next #1a).

The reference indexes its 256^3 volume (x*256 + y)*256 + z (volume.h:46-61) while the probe of every scene looks along +x: consecutive RF steps of
a ray are 2.2 cells apart in the SLOWEST index.  The device copy's layout is the product's to choose (mcrt_upload_texture copies; values are
unchanged), so this script replays the accumulation of one frame of the headline workload as k_march schedules it -- a workgroup per scan-line, a
wavefront per 256 sample slots, the slots of a tile sorted by segment length (longest first), 32 lane pairs each 8 steps per iteration, four gather
instructions per iteration -- and counts, per layout: distinct 128-byte lines per gather instruction and per iteration, distinct 4-KiB regions per
iteration, and the footprint (distinct lines) of a wavefront, of a scan-line's workgroup and of the 16 scan-lines an XCD owns, per bounce.

    python tools/march_layout_count.py [workload=random1m|liver|sphere] [rays=1024] [scan-lines=128]   -> JSON (profiles/round6/march_layout_count_*.json)
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcray_tracing_amd as m
from oracle import orc

workload = sys.argv[1] if len(sys.argv) > 1 else "random1m"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
E = int(sys.argv[3]) if len(sys.argv) > 3 else 128
cfg, meshes = {"random1m": lambda: m.synth.random_scene(1_000_000, 8, 12345), "liver": lambda: m.synth.liver_scene(5), "sphere": lambda: m.synth.sphere_scene(5)}[workload]()
sd = m.scene_io.build_scene(cfg, meshes)
tr = m.Transducer(E, position=cfg["transducerPosition"], angles_deg=cfg["transducerAngles"])
nodes, btri, n4, _ = m.host_build_bvh4(sd.tri, sd.tri_mesh)
osc = orc.OracleScene(sd.tri, sd.tri_mesh, sd.meshes, sd.materials, sd.start_mat, sd.spacing, bvh=(nodes, btri)); osc.set_bvh4(n4)
tex = orc.texture(256)
p = orc.default_params(n_elements=E, n_samples=S)
k = orc.constants(p.frequency, p.sos, p.depth_cm)
t0 = time.time()
o = osc.trace_frame(p, tr.pos, tr.dir, tex, frame_id=0, use_bvh=2, n_threads=os.cpu_count(), want_segs=True, want_ref=False, want_fix=False)
segs, cnt = o["segs"], o["seg_count"]
mats = np.asarray(sd.materials, np.float32).reshape(-1, 8)
axial_f, axial_mm, res = np.float32(k.axial_res_f), float(k.axial_res_mm), np.float32(p.tex_res)


def part1by2(v):                      # spread the 8 bits of v to every third bit
    v = v.astype(np.uint32) & 0xff
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


LAYOUTS = {
    "linear x,y,z (the reference's; z fastest)": lambda x, y, z: (x << 16) | (y << 8) | z,
    "x fastest (y,z,x)": lambda x, y, z: (y << 16) | (z << 8) | x,
    "line bricks 4x2x2, bricks linear": lambda x, y, z: ((((x >> 2) << 14) | ((y >> 1) << 7) | (z >> 1)) << 4) | ((x & 3) << 2) | ((y & 1) << 1) | (z & 1),
    "line bricks 4x2x2 in 8x8x8 (4 KiB) blocks, blocks linear": lambda x, y, z: ((((x >> 3) << 10) | ((y >> 3) << 5) | (z >> 3)) << 9) | (((x >> 2) & 1) << 8) | (((y >> 1) & 3) << 6) | (((z >> 1) & 3) << 4) | ((x & 3) << 2) | ((y & 1) << 1) | (z & 1),
    "line bricks 16x1x1 in 16x4x4 (2 KiB) blocks, blocks linear": lambda x, y, z: ((((x >> 4) << 12) | ((y >> 2) << 6) | (z >> 2)) << 8) | ((y & 3) << 6) | ((z & 3) << 4) | (x & 15),
    "line tiles 4x1x4 (x,z), order y | x>>2 | z>>2 (5 bit operations from the packed cell)": lambda x, y, z: (y << 16) | ((x >> 2) << 10) | ((z >> 2) << 4) | ((x & 3) << 2) | (z & 3),
    "line tiles 4x4x1 (x,y), order z | x>>2 | y>>2": lambda x, y, z: (z << 16) | ((x >> 2) << 10) | ((y >> 2) << 4) | ((x & 3) << 2) | (y & 3),
    "Morton (x,y,z bit-interleaved)": lambda x, y, z: (part1by2(x) << 2) | (part1by2(y) << 1) | part1by2(z),
}


def distinct(keys):
    return int(np.unique(keys).size)


rows = []
for b in range(p.max_depth):
    live = cnt > b
    if not live.any(): break
    sg = segs[:, :, b]
    frm = sg["from"].astype(np.float64); to = sg["to"].astype(np.float64); d = sg["dir"].astype(np.float64)
    dist_f = (np.sqrt(((to - frm).astype(np.float32) ** 2).sum(-1, dtype=np.float32)) * np.float32(10.0)).astype(np.float64)
    steps = np.where(live, np.floor(dist_f / axial_mm), 0).astype(np.int64)
    med = sg["media"]
    silent = (mats[med, 2] == 0) & (mats[med, 4] == 0)          # mu0 == sigma == 0 (GEL): every echo is +0, k_march skips the segment
    steps[silent] = 0
    its = (steps + 7) // 8
    # k_march's schedule: wavefront w of a line owns slots [256 w, 256 w + 256); the tile's live slots sorted longest first; 32 pairs run together
    NW = (S + 255) // 256
    seg_e, seg_s = np.nonzero(its > 0)
    if seg_e.size == 0: continue
    wave = seg_e * NW + seg_s // 256
    order = np.lexsort((seg_s, -its[seg_e, seg_s], wave))
    seg_e, seg_s, wave = seg_e[order], seg_s[order], wave[order]
    first = np.concatenate([[True], wave[1:] != wave[:-1]])
    rank = np.arange(len(wave)) - np.maximum.accumulate(np.where(first, np.arange(len(wave)), 0))
    group = wave * 8 + rank // 32                                  # 32 lane pairs of a wavefront start (and, sorted, finish) together
    n_it = its[seg_e, seg_s]
    # every (segment, iteration) issues 8 gathers -- valid step or not (the lean path gathers unconditionally)
    rep = np.repeat(np.arange(len(seg_e)), n_it * 8)
    within = np.arange(rep.size) - np.repeat(np.cumsum(n_it * 8) - n_it * 8, n_it * 8)       # step index 0 .. 8*its-1
    pt = frm[seg_e[rep], seg_s[rep]] + within[:, None] * (d[seg_e[rep], seg_s[rep]] * float(axial_f))
    q = np.floor(pt / float(res)).astype(np.int64) & 255                                       # (int) cast then modulo 256: floor for the positive coordinates of these scenes
    if (pt < 0).any(): q = (np.trunc(pt / float(res)).astype(np.int64)) & 255
    x, y, z = q[:, 0].astype(np.uint32), q[:, 1].astype(np.uint32), q[:, 2].astype(np.uint32)
    itr = within // 8
    ins = (group[rep].astype(np.int64) * 64 + itr) * 4 + (within % 8) // 2                     # the gather instruction (lane j of a pair owns steps j, j+2, ...)
    r = {"bounce": b, "segments": int(len(seg_e)), "gathers": int(rep.size), "gather_instructions": distinct(ins), "layouts": {}}
    for name, f in LAYOUTS.items():
        cell = f(x, y, z).astype(np.int64)
        line = cell >> 4; page = cell >> 9
        n_ins = r["gather_instructions"]; n_iter = distinct(ins >> 2)
        r["layouts"][name] = {
            "lines_per_gather_instruction": distinct(ins * (1 << 20) + line) / n_ins,
            "lines_per_iteration": distinct((ins >> 2) * (1 << 20) + line) / n_iter,
            "regions_4k_per_iteration": distinct((ins >> 2) * (1 << 15) + page) / n_iter,
            "lines_per_wavefront": distinct(wave[rep].astype(np.int64) * (1 << 20) + line) / distinct(wave),
            "lines_per_scan_line": distinct(seg_e[rep].astype(np.int64) * (1 << 20) + line) / distinct(seg_e),
            "lines_per_xcd_MB": distinct((seg_e[rep] // max(E // 8, 1)).astype(np.int64) * (1 << 20) + line) / min(8, E) * 128 / 1e6,
            "lines_frame_MB": distinct(line) * 128 / 1e6}
    rows.append(r); sys.stderr.write(json.dumps(r) + "\n")

tot = {}
for name in LAYOUTS:
    g = sum(r["gather_instructions"] for r in rows)
    tot[name] = {"lines_per_gather_instruction": sum(r["layouts"][name]["lines_per_gather_instruction"] * r["gather_instructions"] for r in rows) / g,
                 "lines_per_iteration": sum(r["layouts"][name]["lines_per_iteration"] * r["gather_instructions"] for r in rows) / g,
                 "regions_4k_per_iteration": sum(r["layouts"][name]["regions_4k_per_iteration"] * r["gather_instructions"] for r in rows) / g}
print(json.dumps({"workload": workload, "scan_lines": E, "rays": S, "seconds": round(time.time() - t0, 1), "gathers_per_frame": sum(r["gathers"] for r in rows),
                  "all_bounces": tot, "per_bounce": rows}, indent=1))
