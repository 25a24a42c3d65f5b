"""Digest gpurun_out/<tag>_dnpmc_* (tools/profile_depthnet_pmc.sh) into
profiles/<tag>_depthnet_pmc.json: per (kernel, grid) the mean counters and what they say --
MFMA busy share of the SIMDs, MFMA instructions against VALU, LDS bank-conflict share."""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
SIMDS = 1024
agg = collections.defaultdict(lambda: collections.defaultdict(list))
import os

files = []
for d in glob.glob(f"gpurun_out/{tag}_dnpmc_*/"):          # gpurun merges every call into the same directories:
    cand = glob.glob(d + "*/*_counter_collection.csv")      # only the latest run of each pass counts
    if cand:
        files.append(max(cand, key=os.path.getmtime))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "gsr::dn" not in name and "gsr2dn" not in name:
            continue
        key = f'{name} grid={r.get("Grid_Size", "?")}'
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    d = {"counters": m, "launches": max(len(v) for v in cs.values())}
    gui = m.get("GRBM_GUI_ACTIVE")          # summed over the 8 XCDs
    if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        # busy cycles are summed over SIMDs; GRBM_GUI_ACTIVE over XCDs -> per-XCD cycles = gui / 8
        d["mfma_busy_frac_of_simd_time"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * SIMDS)
    if "SQ_INSTS_MFMA" in m and "SQ_INSTS_VALU" in m:
        d["valu_non_mfma_per_mfma"] = (m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / max(m["SQ_INSTS_MFMA"], 1)
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    if "SQ_WAIT_ANY" in m and m.get("SQ_WAVE_CYCLES"):
        d["wave_cycles_parked_frac"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
        d["wave_cycles_issue_stall_frac"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
    out[k] = d
json.dump(out, open(f"profiles/{tag}_depthnet_pmc.json", "w"), indent=1, sort_keys=True)
for k, d in out.items():
    print(k[:70], {x: round(y, 3) for x, y in d.items() if isinstance(y, float)})
