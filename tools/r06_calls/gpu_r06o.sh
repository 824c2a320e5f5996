#!/bin/bash
# round 6, call o: InfoNCE row sums by MFMA against a ones operand (SSLREC_INFONCE_ZMFMA=1) vs the scalar adds: errors + times, alternating;
# + the evaluation tests on the default (register) form after the staged form became opt-in
O=gpurun_out/r06o; mkdir -p $O
for Z in 0 1 0 1; do
  SSLREC_INFONCE_ZMFMA=$Z INFONCE_MODES=h3 timeout 300 python tools/infonce_modes.py $O/modes_zm${Z}.json > $O/modes_zm$Z.log 2>&1; echo "zm=$Z rc $?"; grep fwd_w $O/modes_zm$Z.log | sed 's/.*loss_rel_err/loss_rel_err/' | cut -c1-400
done
SSLREC_INFONCE_ZMFMA=1 timeout 600 python -m pytest tests -x -q -m gpu -k "infonce_normalized or infonce_gathered or tuner or forward_that_keeps" > $O/pytest_zm.log 2>&1; echo "pytest zm rc $?"; tail -3 $O/pytest_zm.log | cut -c1-200
timeout 600 python -m pytest tests -x -q -m gpu -k "eval or topk or metric" > $O/pytest_eval.log 2>&1; echo "pytest eval rc $?"; tail -2 $O/pytest_eval.log | cut -c1-200
