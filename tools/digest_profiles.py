"""Digest gpurun_out/<tag>_* (written by tools/profile_round.sh) into profiles/:
kernel stats CSV, per-kernel PMC means (JSON) and profiles/pmc_traffic.json
(HBM bytes per launch of each gsr kernel, gfx950 correction applied: FETCH_SIZE is
reported in KiB and counts half of the bytes of wide coalesced reads ->
traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024; MI355X_MICROARCH.md section HBM)."""
import collections
import csv
import glob
import json
import shutil
import sys

import os


def newest(pattern):
    """gpurun merges every call's files into the same directory: only the latest run counts."""
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
st = newest(f"gpurun_out/{tag}_stats/*/*_kernel_stats.csv")
if st:
    shutil.copy(st[0], f"profiles/{tag}_kernel_stats.csv")
dn = newest(f"gpurun_out/{tag}_dn_stats/*/*_kernel_stats.csv")
if dn:
    shutil.copy(dn[0], f"profiles/{tag}_depthnet_vitl_kernel_stats.csv")
pmc = {}
for d in glob.glob(f"gpurun_out/{tag}_pmc_*/"):
    for f in newest(d + "*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "gsr" in k:
                pmc.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
json.dump(pmc, open(f"profiles/{tag}_pmc.json", "w"), indent=1, sort_keys=True)
traffic = {}
# keyed on the FULL templated kernel name: project_bwd_kernel<true> (Adam fused in: the headline step's kernel)
# and <false> (gradients written) are different kernels with different traffic; first match wins
names = [("raster_bwd_kernel", "gsr_rasterize_bwd"), ("raster_fwd_kernel", "gsr_rasterize_fwd"),
         ("project_fwd_kernel", "gsr_project_fwd"), ("project_bwd_adam1_kernel", "gsr_project_bwd_adam"),
         ("project_bwd_kernel<true>", "gsr_project_bwd_adam_generic"),
         ("project_bwd_kernel<false>", "gsr_project_bwd"), ("bucket_count_kernel", "gsr_bucket_count"),
         ("bucket_emit_kernel", "gsr_bucket_emit"), ("bucket_sort_kernel", "gsr_bucket_sort"),
         ("tile_order_kernel", "gsr_tile_order"), ("l1_fwd_kernel", "gsr_l1_fwd"),
         ("ssim_fwd", "gsr_ssim_fwd"), ("ssim_bwd", "gsr_ssim_bwd"),
         ("isect_emit_kernel", "gsr_isect_emit"), ("tile_sort_small_kernel", "gsr_tile_sort"),
         ("adam_kernel", "gsr_adam_step")]


def entry_for(kernel_name):
    for frag, entry in names:
        if frag in kernel_name:
            return entry
    return None


for k, cs in pmc.items():
    entry = entry_for(k)
    if entry and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        traffic[entry] = int((2 * cs["FETCH_SIZE"] + cs["WRITE_SIZE"]) * 1024)
valu = {}
for k, cs in pmc.items():
    entry = entry_for(k)
    if entry and "SQ_INSTS_VALU" in cs:
        valu[entry] = int(cs["SQ_INSTS_VALU"])
# one file bench.py reads, with provenance: which digest, taken at which commit
import subprocess
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    commit = None
import hashlib
from pathlib import Path
csrc = Path(__file__).resolve().parents[1] / "3dgs_monocular_depth_init_amd" / "csrc"
hashes = {f.name: hashlib.sha256(f.read_bytes()).hexdigest()[:16]
          for f in sorted(csrc.glob("*")) if f.suffix in (".hip", ".h")}
# (source_hashes: the csrc files as they were when the counters were digested -- run the digest at the tree the
# profile was taken from; bench.py marks the counters stale when the compositing sources have changed since)
dominant = {"source": f"profiles/{tag}_pmc.json", "commit": commit, "source_hashes": hashes, "kernels": {}}
for k, cs in pmc.items():
    entry = entry_for(k)
    if entry:
        dominant["kernels"][entry] = dict({c: cs[c] for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE",
                                                            "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                                                            "SQ_BUSY_CYCLES", "TCC_HIT_sum", "TCC_MISS_sum",
                                                            "TCC_EA0_ATOMIC_sum") if c in cs}, kernel=k)
if dominant["kernels"]:
    json.dump(dominant, open("profiles/pmc_dominant.json", "w"), indent=1, sort_keys=True)
if traffic:        # a digest of a counter set without FETCH/WRITE must not wipe the file
    json.dump(traffic, open("profiles/pmc_traffic.json", "w"), indent=1, sort_keys=True)
if valu:
    json.dump(valu, open("profiles/pmc_valu.json", "w"), indent=1, sort_keys=True)
print(json.dumps({"traffic": traffic, "valu_insts": valu}, indent=1))
