#!/bin/bash
# round 6, call p: the anchor-gradient role with two score accumulators per resident tile (SSLREC_INFONCE_SC2=1: four chains), alternating
O=gpurun_out/r06p; mkdir -p $O
for Z in 0 1 0 1; do
  SSLREC_INFONCE_SC2=$Z INFONCE_MODES=h3 timeout 300 python tools/infonce_modes.py $O/modes_sc2_${Z}.json > $O/modes_sc2_$Z.log 2>&1; echo "sc2=$Z rc $?"; grep fwd_w $O/modes_sc2_$Z.log | sed 's/.*fwd_nograd_ms/fwd_nograd_ms/' | cut -c1-200
done
SSLREC_INFONCE_SC2=1 timeout 600 python -m pytest tests -x -q -m gpu -k "infonce_normalized or infonce_gathered or tuner or forward_that_keeps" > $O/pytest_sc2.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_sc2.log | cut -c1-200
