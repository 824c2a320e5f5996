#!/bin/bash
# round 6, call b: the InfoNCE hot loop -- peeled ragged tile, no packed-f32 VALU, v_fma_mix residuals (both loops) and the
# software-pipelined loop (SSLREC_INFONCE_PIPE=1, default) against round 5's order of work (=0): errors + times, parity tests, kernel stats
O=gpurun_out/r06b; mkdir -p $O
for P in 1 0; do
  SSLREC_INFONCE_PIPE=$P INFONCE_MODES=h3,x6 python tools/infonce_modes.py $O/infonce_modes_pipe$P.json > $O/modes_pipe$P.log 2>&1; echo "modes pipe=$P rc $?"; cat $O/modes_pipe$P.log | cut -c1-400
done
python -m pytest tests -x -q -m gpu -k "infonce or contrastive or simgcl or sgl or lightgcl" > $O/pytest_infonce.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_infonce.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for P in 1 0; do
  SSLREC_INFONCE_PIPE=$P rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_pipe$P -o infonce -- python $GRAFT_REPO_ROOT/tools/infonce_run.py 20 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/$O/prof_pipe$P -name "*kernel_stats.csv" | head -1); echo "pipe=$P"; head -6 $f | cut -c1-160
  cp $f $GRAFT_REPO_ROOT/$O/infonce_kernel_stats_pipe$P.csv; rm -rf $GRAFT_REPO_ROOT/$O/prof_pipe$P
done
