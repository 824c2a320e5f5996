mkdir -p gpurun_out/r02k
for q in 0 1; do
SSLREC_SWEPT_DEGREE_SWEEP=$q timeout 600 python tools/spmm_xcd.py --split 1 > gpurun_out/r02k/spmm_deg$q.log 2>&1; echo "deg $q exit $?"; tail -2 gpurun_out/r02k/spmm_deg$q.log | cut -c 100-420
done
