#!/bin/bash
# PMC probe of one scene (GPU box): tools/pmc_probe.sh <tag> <scene> <spp> "<counters pass 1>" "<counters pass 2>" ...
TAG=$1; SC=$2; SPP=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/probe_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/prof_one.py $SC $SPP > $OUT/p$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 $OUT/p$i.log)"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float)
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("hrt_wgstream") or r["Kernel_Name"].startswith("hrt_trace"): acc[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(acc): print(f"{k:36s} {acc[k]:.6g}")
PY
