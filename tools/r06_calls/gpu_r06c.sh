#!/bin/bash
# round 6, call c: the folded InfoNCE launches (fused preparation, one-launch forward finish, normalization backward in the all-gradient
# role's epilogue, scatter registration in finish_bwd), h3 at the ends of the temperature range; kernel stats + issue-stall counters
O=gpurun_out/r06c; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "round6 or infonce or contrastive or simgcl or sgl or lightgcl" > $O/pytest_infonce.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_infonce.log | cut -c1-300
for F in 1 0; do
  SSLREC_INFONCE_FOLD=$F INFONCE_MODES=h3,x6 timeout 300 python tools/infonce_modes.py $O/infonce_modes_fold$F.json > $O/modes_fold$F.log 2>&1; echo "modes fold=$F rc $?"; grep fwd_w $O/modes_fold$F.log | cut -c1-400
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o infonce -- python $R/tools/infonce_profile.py 20 > $R/$O/prof.log 2>&1; echo "rocprof rc $?"
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp "$f" $R/$O/infonce_kernel_stats.csv; head -14 "$f" | cut -c1-150; fi; rm -rf $R/$O/prof
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$O/pmc_$i -o p -- python $R/tools/infonce_profile.py 10 > $R/$O/pmc_$i.log 2>&1; echo "pmc [$pmc] rc $?"
done
cd $R && python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('$O/pmc_*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: {'launches': len(v), 'mean': sum(v) / len(v)} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open('$O/pmc_summary.json', 'w'), indent=1)
for k, cs in out.items():
    if 'lds_kernel' in k or 'prep' in k:
        print(k, {c: round(v['mean'], 1) for c, v in cs.items()})
PY
rm -rf $O/pmc_*/
