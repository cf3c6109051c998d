#!/bin/bash
# Fabric-side bytes of tools/micro/record_coop's two kernels (GPU box, through gpurun): each counter group in its own rocprofv3 run.
#   bash tools/record_coop_pmc.sh <tag>   -> gpurun_out/coop_<tag>/summary.json  (copy to profiles/<tag>_record_coop.json)
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/coop_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
BIN=$ROOT/tools/micro/record_coop
[ -x $BIN ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $BIN tools/micro/record_coop.hip || exit 1
$BIN 128 256 > $OUT/plain.jsonl || exit 1
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p_$i -- $BIN 128 256 > $OUT/p_$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 $OUT/p_$i.log)"
done
python3 - <<PY
import collections, csv, glob, json
rows = [json.loads(l) for l in open("$OUT/plain.jsonl")]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p_*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "visits_kernel" in r["Kernel_Name"]: per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, d), c in per.items():
        for n, v in c.items(): acc["coop" if "true" in k or "1" in k.split("<")[-1] else "gather"][n].append(v)
out = {"what": "tools/micro/record_coop.hip, 128 MiB pool, 256 visits per wave, 256 x 1024 threads: the same 5 + 5 sixteen-byte pieces of 64 records per wave and visit, "
               "as 64 lines per instruction (gather: the streaming kernel's hit visit) or 13 (coop); the LARGEST dispatch of each kernel (the 256-visit one)", "timing": rows, "counters": {}}
for k, c in acc.items():
    big = {n: max(v) for n, v in c.items()}
    rd = 32 * big.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * big.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * big.get("TCC_EA0_RDREQ_128B_sum", 0)
    w64 = big.get("TCC_EA0_WRREQ_64B_sum", 0)
    wr = 64 * w64 + 32 * max(big.get("TCC_EA0_WRREQ_sum", 0) - w64, 0)
    ms = min(r["ms"] for r in rows if r["pattern"].startswith(k))
    out["counters"][k] = dict(big, fabric_read_bytes=rd, fabric_write_bytes=wr, ms=ms, fabric_tbps=round((rd + wr) / (ms * 1e-3) / 1e12, 3),
                              fabric_requests_per_ns=round((big.get("TCC_EA0_RDREQ_sum", 0) + big.get("TCC_EA0_WRREQ_sum", 0)) / (ms * 1e6), 2))
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out["counters"], indent=1))
PY
