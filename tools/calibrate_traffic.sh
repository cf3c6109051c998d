#!/bin/bash
# Calibration of FETCH_SIZE / WRITE_SIZE on the path pool's own access pattern (GPU box, through gpurun):
#   bash tools/calibrate_traffic.sh <tag>      -> gpurun_out/calib_<tag>/calibration.json  (copy to profiles/<tag>_traffic_calibration.json)
# tools/micro/record_pattern.hip performs a known number of "hit visits" (6 x 16-byte group loads + 5 group stores per lane
# from one 128-byte record, every lane its own record) over a pool that is (a) past L2 but inside the Infinity Cache and (b)
# past the Infinity Cache; each counter group is collected in its own rocprofv3 run (no trace options with --pmc).
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/calib_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
BIN=$ROOT/tools/micro/record_pattern
[ -x $BIN ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $BIN tools/micro/record_pattern.hip || exit 1
for mib in 96 2048; do
  $BIN $mib 64 > $OUT/plain_$mib.json || exit 1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p${mib}_$i -- $BIN $mib 64 > $OUT/p${mib}_$i.log 2>&1 || echo "pass $i ($set) at $mib MiB failed: $(tail -2 $OUT/p${mib}_$i.log)"
  done
done
python3 - <<PY
import collections, csv, glob, json
out = {"pattern": "per lane-visit: global_load_dwordx4 x 6 (groups 0 1 2 5 6 7 of one 128-byte record) + global_store_dwordx4 x 5 (groups 0 1 2 6 7), "
                  "64 lanes on 64 pseudo-random records, 256 workgroups x 1024 threads x 64 visits (tools/micro/record_pattern.hip)", "pools": {}}
for mib in (96, 2048):
    known = json.load(open("$OUT/plain_%d.json" % mib))
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/p%d_*/**/*counter_collection.csv" % mib, recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("record_visits"): per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if per:
            last = per[max(per, key=lambda k: int(k))]   # the second, measured launch
            for k, v in last.items(): acc[k].append(v)
    c = {k: sum(v) / len(v) for k, v in acc.items()}
    lv = known["lane_visits"]
    d = {"known": known, "counters_of_the_measured_launch": c}
    if "FETCH_SIZE" in c: d["FETCH_SIZE_bytes_per_lane_visit"] = round(c["FETCH_SIZE"] * 1024.0 / lv, 2)
    if "WRITE_SIZE" in c: d["WRITE_SIZE_bytes_per_lane_visit"] = round(c["WRITE_SIZE"] * 1024.0 / lv, 2)
    if "TCC_EA0_RDREQ_sum" in c:
        n32, n64, n128 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_64B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        d["read_requests_per_lane_visit"] = {"all": round(c["TCC_EA0_RDREQ_sum"] / lv, 3), "32B": round(n32 / lv, 3), "64B": round(n64 / lv, 3), "128B": round(n128 / lv, 3)}
        other = c["TCC_EA0_RDREQ_sum"] - n32 - n64 - n128
        d["read_bytes_by_request_size_per_lane_visit"] = round((32 * n32 + 64 * n64 + 128 * n128 + 64 * max(other, 0.0)) / lv, 2)
    if "TCC_EA0_WRREQ_sum" in c:
        n64 = c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        d["write_requests_per_lane_visit"] = {"all": round(c["TCC_EA0_WRREQ_sum"] / lv, 3), "64B": round(n64 / lv, 3)}
        d["write_bytes_by_request_size_per_lane_visit"] = round((64 * n64 + 32 * (c["TCC_EA0_WRREQ_sum"] - n64)) / lv, 2)
    if "TCC_HIT_sum" in c: d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if "FETCH_SIZE_bytes_per_lane_visit" in d:
        # what one visit must move across the fabric when nothing is cached: both 64-byte halves of the record's line are read
        d["fetch_factor_vs_line_bytes"] = round(128.0 / d["FETCH_SIZE_bytes_per_lane_visit"], 3)
        d["fetch_factor_vs_requested_bytes"] = round(96.0 / d["FETCH_SIZE_bytes_per_lane_visit"], 3)
    if "WRITE_SIZE_bytes_per_lane_visit" in d:
        d["write_factor_vs_stored_bytes"] = round(80.0 / d["WRITE_SIZE_bytes_per_lane_visit"], 3)
    out["pools"]["%d MiB" % mib] = d
json.dump(out, open("$OUT/calibration.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
