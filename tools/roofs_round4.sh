#!/bin/bash
# the round-4 roof calibrations on the GPU box (outputs under gpurun_out/roofs/, packaged into profiles/round4/ by hand):
#   valu_roof --quick        issue ceiling of the walk's CURRENT node step (WALK_MIX_LANE) at 4 / 5 wavefronts per SIMD
#   fetch_roof_same + pmc    what TCP_TOTAL_CACHE_ACCESSES counts for uniform / scattered quads (the second roof's cost model)
#   fetch_calib + pmc        what FETCH_SIZE reports for 64-byte nodes, 96-byte records, 8-byte cells against a known byte count
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/roofs; mkdir -p $O
$R/build/valu_roof --quick > $O/valu_roof.json 2> $O/valu_roof.err
$R/build/fetch_roof_same > $O/fetch_roof_same.json 2> $O/fetch_roof_same.err
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VMEM_RD SQ_BUSY_CU_CYCLES --output-format csv -d $O/frs_pmc -- $R/build/fetch_roof_same > $O/frs_pmc.log 2>&1
$R/build/fetch_calib > $O/fetch_calib.json 2> $O/fetch_calib.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fc_fetch -- $R/build/fetch_calib > $O/fc_fetch.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/fc_rdreq -- $R/build/fetch_calib > $O/fc_rdreq.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/fc_hit -- $R/build/fetch_calib > $O/fc_hit.log 2>&1
python3 - <<'PY'
import csv, glob, os, json
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/roofs"
out = {}
for d in ("frs_pmc", "fc_fetch", "fc_rdreq", "fc_hit"):
    per = {}
    for f in glob.glob(os.path.join(O, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            per.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    out[d] = {k: {c: v for c, v in cs.items()} for k, cs in per.items()}
json.dump(out, open(os.path.join(O, "pmc_summary.json"), "w"), indent=1)
print(json.dumps({d: {k: {c: v[-1] for c, v in cs.items()} for k, cs in ks.items()} for d, ks in out.items()}, indent=1)[:6000])
PY
