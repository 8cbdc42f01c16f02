#!/usr/bin/env python3
"""Turn a tools_profile.sh PMC result (pmc_traffic.json: FETCH_SIZE / WRITE_SIZE totals per kernel) into
profiles/hbm_traffic.json, the per-block HBM traffic bench.py reports as roofline.traffic.
usage: tools_traffic.py <pmc_traffic.json> <blocks profiled> <source tag>"""
import json
import re
import sys

src, blocks, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
raw = json.load(open(src))
if blocks <= 0:   # derive from the K1 dispatch count: one launch per 152-block pass (the default batch holds 256 blocks)
    k1 = [v for k, v in raw.items() if "k1_" in k][0]
    blocks = k1["dispatches"] * 152
out = {"note": "HBM traffic per filterbank block from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes). "
               "FETCH_SIZE is doubled for kernels that stream with 16-B-per-lane loads (gfx950 reports half, "
               "MI355X_MICROARCH.md 'HBM'); the 4-byte-per-lane gather of an unstaged K1 is left uncorrected (uncalibrated width).",
       "source": tag, "blocks_profiled": blocks, "kernels": {}}
for name, rec in raw.items():
    m = re.search(r"(frbch_[a-z0-9_]+)(<[^>]*>)?", name)
    short = (m.group(1) + (m.group(2) or "")).replace(" ", "")
    short = short.replace(",true>", ">").replace(",false>", ">")   # run-time name of the kernel slot (bench.py) has no staging / statistics flag
    staged_k1 = ("k1_wave" in name) and ("true" in name)        # reads its rows from the corner-turned buffer with 16-byte loads
    wide = ("k2_" in short) or ("quantise" in short) or ("stats_partial" in short) or ("k0_stage" in short) or staged_k1
    out["kernels"][short] = {"fetch_kb_per_block": rec.get("FETCH_SIZE_KB_total", 0.0) / blocks,
                             "fetch_correction": 2.0 if wide else 1.0,
                             "write_kb_per_block": rec.get("WRITE_SIZE_KB_total", 0.0) / blocks}
json.dump(out, open("profiles/hbm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
