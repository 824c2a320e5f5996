#!/bin/bash
# Profiles of one round on the GPU box: rocprofv3 kernel stats of the bench command + PMC passes (separate runs, counters
# only with --kernel-trace, as the microarch guide prescribes).  usage: tools/gpu_profile.sh <tag>   -> gpurun_out/<tag>/
TAG=${1:-r02p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
export SSLREC_SPARSE_GRAD=0      # the profiled command issues the headline's dense launches only (bench.py times the hinted variant separately)
CMD="python $ROOTDIR/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-configs"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/$OUT/prof -o bench -- $CMD > $ROOTDIR/$OUT/prof_bench.log 2>&1; echo "== rocprof stats exit $?")
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/bench_kernel_stats.csv && head -8 $OUT/bench_kernel_stats.csv
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $ROOTDIR/$OUT/pmc_$i -o p -- python $ROOTDIR/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-configs > $ROOTDIR/$OUT/pmc_$i.log 2>&1; echo "== pmc [$pmc] exit $?")
done
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('$OUT/pmc_*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: {'launches': len(v), 'mean': sum(v) / len(v)} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open('$OUT/pmc_summary.json', 'w'), indent=1)
for k, cs in out.items():
    if 'spmm' in k:
        print(k, {c: round(v['mean'], 1) for c, v in cs.items()})
PY
