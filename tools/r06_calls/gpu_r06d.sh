#!/bin/bash
# round 6, call d: the whole GPU suite on the round's InfoNCE / loader / alias changes, the bench line with its new blocks (roofline.ceiling,
# extras.epoch), the all-gradient role with two score accumulators
O=gpurun_out/r06d; mkdir -p $O
INFONCE_MODES=h3 timeout 300 python tools/infonce_modes.py $O/infonce_modes.json > $O/modes.log 2>&1; echo "modes rc $?"; grep fwd_w $O/modes.log | cut -c1-420
timeout 1200 python -m pytest tests -x -q -m gpu --durations=5 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-300 $O/bench.json; tail -3 $O/bench.err | cut -c1-300
