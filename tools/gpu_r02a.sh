#!/bin/bash
# r02 call A: XCD-split experiment + gather ceiling + PMC passes
OUT=gpurun_out/r02a
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
timeout 600 python tools/spmm_xcd.py > $OUT/spmm_xcd.log 2>&1; echo "== spmm_xcd exit $?"; cat $OUT/spmm_xcd.log | tail -8
timeout 600 python tools/micro/run_gather_ceiling.py $OUT/gather_ceiling.json > $OUT/gather_ceiling.log 2>&1; echo "== gather exit $?"; tail -40 $OUT/gather_ceiling.log
for split in 0 1; do
for pmc in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=s${split}_$(echo $pmc | tr ' ' '_')
  (cd /tmp && timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $ROOTDIR/$OUT/pmc_$tag -o p -- python $ROOTDIR/tools/spmm_xcd.py --only amazon-book --split $split --reps 10 > $ROOTDIR/$OUT/pmc_$tag.log 2>&1; echo "== pmc $tag exit $?")
done
done
python - <<'PY'
import csv, glob, collections, json
for f in sorted(glob.glob('gpurun_out/r02a/pmc_*/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'spmm_swept_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[2], {k: (len(v), sum(v) / len(v)) for k, v in acc.items()})
PY
