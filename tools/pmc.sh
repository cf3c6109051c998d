#!/bin/bash
# PMC counters of the trace kernel for a list of scenes (GPU box). Usage: tools/pmc.sh tag scene:spp ...
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
for spec in "$@"; do
  sc=${spec%%:*}; spp=${spec#*:}
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/$sc -- python3 tools/prof_one.py $sc $spp > $OUT/$sc.log 2>&1
  rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/${sc}_b -- python3 tools/prof_one.py $sc $spp > $OUT/${sc}_b.log 2>&1
  tail -1 $OUT/$sc.log
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("hrt_trace_kernel"): acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split("/pmc_$TAG/")[1].split("/")[0], {k: "%.4g" % v for k, v in acc.items()})
PY
