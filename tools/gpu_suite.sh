#!/bin/bash
# the GPU suite + smoke, as the driver runs them at the end of a round.   usage: bash tools/gpu_suite.sh [tag]
T=${1:-suite}; O=gpurun_out/$T; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
