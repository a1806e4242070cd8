#!/usr/bin/env python3
"""Turns the per-dispatch counter means of tools/pmc_cmd.sh into the HBM-traffic record bench.py quotes.

usage: traffic_json.py <pmc.json> <kernel label> <rows> <cols> <build rev>
FETCH_SIZE / WRITE_SIZE are KiB per dispatch, collected in passes of their own; on gfx950 FETCH_SIZE counts
128-byte requests of 16-byte-per-lane streaming loads as 64 bytes and is doubled (MI355X_MICROARCH.md, HBM)."""
import json
import sys

pmc = json.load(open(sys.argv[1]))["per_dispatch_mean"]
label, rows, cols, rev = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
fetch_raw = 1024.0 * pmc.get("FETCH_SIZE", 0.0)
write = 1024.0 * pmc.get("WRITE_SIZE", 0.0)
rec = {
    "source": "tools/pmc_cmd.sh (rocprofv3 --pmc, one pass per counter group) on MI355X; build " + rev,
    "units": "FETCH_SIZE / WRITE_SIZE in KiB per dispatch; reads doubled per the gfx950 correction",
    "shape": {"rows": rows, "cols": cols, "elements": rows * cols},
    "kernels": {label: {
        "FETCH_SIZE_KiB": pmc.get("FETCH_SIZE"), "WRITE_SIZE_KiB": pmc.get("WRITE_SIZE"),
        "hbm_read_bytes": 2.0 * fetch_raw, "hbm_write_bytes": write,
        "hbm_bytes_per_launch": 2.0 * fetch_raw + write,
        "hbm_bytes_per_element": (2.0 * fetch_raw + write) / (rows * cols),
        "hbm_bytes_per_row": (2.0 * fetch_raw + write) / rows}},
    "sq_counters_per_launch": {k: v for k, v in pmc.items() if k.startswith("SQ_") or k.startswith("GRBM")},
}
c = rec["sq_counters_per_launch"]
if c.get("SQ_LDS_IDX_ACTIVE"):
    rec["lds_conflict_cycle_share"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
if c.get("SQ_WAVES") and c.get("SQ_INSTS_VALU"):
    rec["valu_instructions_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
print(json.dumps(rec, indent=1))
