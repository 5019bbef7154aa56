#!/usr/bin/env python3
"""profiles/sweep_traffic.json from a PMC summary written by tools/collect_profiles.sh:
    python tools/traffic_json.py gpurun_out/<tag>/pmc_summary.txt profiles/<name>_sweep_pmc.txt
HBM-side bytes of one sweep launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 B (gfx950: FETCH_SIZE counts 64-B units
of the 128-B reads as one, MI355X_MICROARCH.md "HBM bytes from rocprofv3"); bench.py quotes it as roofline.traffic
only while the library version and the workload match."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mvsdet_amd import _lib  # noqa: E402

summary, committed_as = sys.argv[1], sys.argv[2]
out = {"lib_version": _lib.load().mvsdet_version(),
       "source": f"{committed_as} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_sweep.py; traffic = "
                 "(2*FETCH_SIZE + WRITE_SIZE) * 1024 B: gfx950 FETCH_SIZE under-reports wide reads by 2x, MI355X_MICROARCH.md)",
       "workloads": {}}
wl, vals = None, {}
for line in open(summary):
    if line.startswith("## "):
        wl = line[3:].strip()
        vals[wl] = {}
        continue
    m = re.search(r"plane_sweep_variance_kernel.*?\s(\w+)\s+dispatches=\d+ mean=([0-9.e+]+)", line)
    if m and wl:
        vals[wl][m.group(1)] = float(m.group(2))
for wl, v in vals.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    w = bench.WORKLOADS[wl]
    alg = bench.sweep_bytes_per_cv(w) * w["N"]
    fetch, write = 2 * v["FETCH_SIZE"] * 1024, v["WRITE_SIZE"] * 1024
    ent = {"traffic_bytes": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "algorithmic_bytes": alg,
           "ratio": round((fetch + write) / alg, 4)}
    if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v:
        ent["wait_any_frac"] = round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3)
    if "SQ_ACTIVE_INST_VALU" in v and "SQ_WAVE_CYCLES" in v:
        ent["valu_active_frac_of_wave_cycles"] = round(v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"], 3)
    if "SQ_INSTS_VALU" in v:
        ent["valu_insts"] = v["SQ_INSTS_VALU"]
    if "SQ_LDS_BANK_CONFLICT" in v and "SQ_LDS_IDX_ACTIVE" in v:
        ent["lds_conflict_frac"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 3)
    out["workloads"][wl] = ent
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "sweep_traffic.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
