mkdir -p gpurun_out/r02c
for b in 530 500 480 460 440; do
SSLREC_XCD_BALANCE=$b timeout 600 python tools/spmm_xcd.py --split 1 --only amazon-book > gpurun_out/r02c/spmm_bal$b.log 2>&1; echo "bal $b exit $?"; tail -1 gpurun_out/r02c/spmm_bal$b.log | cut -c 150-400
done
