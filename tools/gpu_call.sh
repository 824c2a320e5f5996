OUT=gpurun_out/r02w; mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
for lib in base nocand; do
if [ $lib = nocand ]; then export SSLREC_HIP_LIBRARY=$ROOTDIR/build_variants/lib_nocand.so; else unset SSLREC_HIP_LIBRARY; fi
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $ROOTDIR/$OUT/pmc_${lib}_$i -o p -- python $ROOTDIR/tools/eval_pmc.py > $ROOTDIR/$OUT/pmc_${lib}_$i.log 2>&1; echo "== $lib pmc [$pmc] exit $?")
done
done
python - <<PY
import csv, glob, collections, json
out = {}
for lib in ('base', 'nocand'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob('$OUT/pmc_%s_*/*counter_collection.csv' % lib)):
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
            acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
    out[lib] = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items() if 'eval' in k}
json.dump(out, open('$OUT/eval_pmc.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
