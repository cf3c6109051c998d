#!/bin/bash
# Memory-pipeline counters (TA / TCP / TCC) of the trace kernel. Usage: tools/pmc_mem.sh tag scene:spp ...
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmcmem_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd $ROOT
rocprofv3 -L > $OUT/counters.txt 2>&1
for spec in "$@"; do
  sc=${spec%%:*}; spp=${spec#*:}
  i=0
  for set in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
             "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" \
             "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --pmc $set --output-format csv -d $OUT/${sc}_$i -- python3 tools/prof_one.py $sc $spp > $OUT/${sc}_$i.log 2>&1 || echo "pass $i ($set) failed"
    echo "pass $i done"
  done
  tail -1 $OUT/${sc}_1.log
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("hrt_trace"): acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(f.split("/pmcmem_$TAG/")[1].split("/")[0], {k: "%.4g" % v for k, v in acc.items()})
PY
