#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats of bench.py, then PMC passes (each its own run).
# Summaries land in gpurun_out/prof_<tag>/ ; copy the ones to be judged into profiles/.
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_under_trace.log 2>&1 || echo "trace run failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.log 2>&1 || echo "pmc fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_write.log 2>&1 || echo "pmc write failed"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_sq.log 2>&1 || echo "pmc sq failed"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_THREAD_CYCLES_VALU TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_sq2.log 2>&1 || echo "pmc sq2 failed"
find $OUT -name "*.csv" | head -40
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
python3 - <<PY
import csv, glob, collections
for d in ["pmc_fetch","pmc_write","pmc_sq","pmc_sq2"]:
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name","?")[:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, cs in acc.items():
            print(d, k, {c: (v, n[(k,c)]) for c, v in cs.items()})
PY
python3 tools/pmc_summary.py $OUT $TAG
