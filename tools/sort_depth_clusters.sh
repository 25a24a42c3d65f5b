#!/bin/bash
# bash tools/sort_depth_clusters.sh   -> one line per depth distribution: tile-list kernel averages (us)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for c in uniform shells shells_outliers; do
  rm -rf /tmp/prof_sd
  rocprofv3 --kernel-trace --stats -d /tmp/prof_sd -o sd --output-format csv -- python3 "$root/tools/sort_depth_clusters.py" $c > /tmp/sd_$c.log 2>&1 || { tail -5 /tmp/sd_$c.log; exit 1; }
  grep -a '"case"' /tmp/sd_$c.log
  python3 - /tmp/prof_sd/sd_kernel_stats.csv <<'PY'
import csv, sys
rows = {r["Name"].split("(")[0].replace("void ", "").replace("gsr::", ""): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[1]))}
print("   ", " ".join("%s=%.1f" % (k, v) for k, v in rows.items() if k.startswith(("bucket_", "tile_order", "raster_fwd"))), flush=True)
PY
done
