#!/bin/bash
# round 6, call a: the GPU suite on the round's first changes (native sampler / exact loader / alias guard) + the bench line as the round's baseline
O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-600 $O/bench.json
