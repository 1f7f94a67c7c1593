#!/usr/bin/env python3
"""profiles/traffic_rNN.json (read by bench.py for roofline.traffic) from a tools/pmc_summary.py table:
    python tools/traffic_json.py profiles/r03_binned_hbm_traffic.txt profiles/traffic_r03.json"""
import json
import re
import sys

SLOT_KERNELS = {
    "geometry": ["k_geometry"],
    "raster_tris": ["k_raster_tris"], "raster_large": ["k_raster_large"], "raster_resolve": ["k_raster_resolve", "k_raster_bins"],
    "shade_fwd": ["k_active_segments"], "aa_fwd": ["k_aa_fwd_list"], "loss_pix": ["k_loss_pix"],
    "aa_bwd": ["k_aa_bwd_items"], "shade_bwd": ["k_face_bwd"], "regularisers": ["k_mesh_tail_a", "k_mesh_tail_b"],
}
src = sys.argv[1]
rows = {}
for line in open(src):
    # round 5 table: kernel, launches (FETCH pass), launches (WRITE pass), launches per step, FETCH_KB, WRITE_KB; rounds 1-4: kernel,
    # launches, FETCH_KB, WRITE_KB (one launch per step for every kernel of the step)
    m = re.match(r"^(\S.*?)\s+(\d+)(?:\s+(\d+)\s+([\d.]+))?\s+(\d+)\s+(\d+)\s*$", line)
    if m:
        per = max(1.0, round(float(m.group(4)))) if m.group(4) else 1.0
        rows[m.group(1)] = (float(m.group(5)) * per, float(m.group(6)) * per)
out = {"_source": f"{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, metric config, 1 GPU)",
       "_note": "HBM bytes per launch = (FETCH_SIZE * 2 + WRITE_SIZE) * 1024: gfx950 counts 64 B per 128-B request on coalesced streams, "
                "so this is the upper estimate; the lower one (FETCH_SIZE * 1) is step_total_low"}
lo = hi = 0.0
for slot, ks in SLOT_KERNELS.items():
    f = sum(v[0] for k, v in rows.items() if any(x in k for x in ks))
    w = sum(v[1] for k, v in rows.items() if any(x in k for x in ks))
    out[slot] = int((2 * f + w) * 1024)
    lo += (f + w) * 1024
    hi += (2 * f + w) * 1024
out["step_total_low"], out["step_total_high"] = int(lo), int(hi)
out["_binned"] = any("k_raster_bins" in k for k in rows)
json.dump(out, open(sys.argv[2], "w"), indent=2)
print(json.dumps(out, indent=2))
