#!/bin/bash
O=gpurun_out/r03h; mkdir -p $O
ROOTDIR=$(pwd); export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/$O/prof -o cfg4 -- python $ROOTDIR/bench.py --config cfg4 --steps 30 > $ROOTDIR/$O/prof.log 2>&1; echo "rocprof exit $?")
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/cfg4_kernel_stats.csv && head -12 $O/cfg4_kernel_stats.csv | cut -c1-200
t=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
sw=[r for r in rows if 'spmm_swept' in r['Kernel_Name']]
print('swept launches', len(sw)); 
import statistics
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in sw]
print('dur us: min %.1f median %.1f mean %.1f max %.1f'%(min(d), statistics.median(d), statistics.mean(d), max(d)))
print('fields', list(rows[0].keys()))
r=sw[len(sw)//2]; print({k:r[k] for k in r if k in ('Kernel_Name','Workgroup_Size','Grid_Size','LDS_Block_Size','Scratch_Size','VGPR_Count','SGPR_Count','Accum_VGPR_Count')})
# histogram by grid size
c=collections.Counter((r['Grid_Size'], r['LDS_Block_Size']) for r in sw); print(c)
PY
rm -rf $O/prof
