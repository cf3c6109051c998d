#!/bin/bash
# Runs on the GPU box (through gpurun): for every BASELINE configuration (and the metric line) the kernel-trace stats of the
# trace kernel, then the PMC passes, each counter group in its own run (no trace options together with --pmc).
#   bash tools/profile.sh <tag> [cfg ...]     cfg: cfg2_256 (the metric line) cfg1 cfg2 cfg3 cfg4 cfg5
# Summaries: gpurun_out/prof_<tag>/<tag>_<cfg>_pmc.json + <tag>_<cfg>_kernel_stats.csv -- copy into profiles/.
set -o pipefail
TAG=${1:-r03}; shift
CFGS=${@:-cfg2_256 cfg1 cfg2 cfg3 cfg4 cfg5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
declare -A SPEC=( [cfg2_256]="cornell_mesh 256 1920 1080" [cfg1]="cornell_box 4 256 256" [cfg2]="cornell_mesh 64 1920 1080" [cfg3]="random_spheres 256 1920 1080"
                  [cfg4]="mesh_in_box 512 3840 2160" [cfg5]="backrooms_pool 1024 3840 2160" )
PMCSETS=( "FETCH_SIZE" "WRITE_SIZE"
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
         "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_THREAD_CYCLES_VALU" )
for cfg in $CFGS; do
  spec=${SPEC[$cfg]}
  [ -n "$spec" ] || { echo "unknown configuration $cfg"; continue; }
  D=$OUT/$cfg; mkdir -p $D
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 tools/prof_one.py $spec 3 > $D/trace.log 2>&1 || echo "$cfg: trace run failed: $(tail -2 $D/trace.log)"
  i=0
  for set in "${PMCSETS[@]}"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $D/pmc_$i -- python3 tools/prof_one.py $spec > $D/pmc_$i.log 2>&1 || echo "$cfg: pmc pass $i ($set) failed: $(tail -2 $D/pmc_$i.log)"
  done
  python3 tools/pmc_summary.py $D $TAG $cfg $spec || echo "$cfg: summary failed"
  echo "== $cfg done"
done
