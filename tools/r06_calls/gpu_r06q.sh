#!/bin/bash
# round 6, call q: the driver's checks on the final library (experiments reverted, staged evaluation opt-in): full suite + smoke
O=gpurun_out/r06q; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log | cut -c1-200; tail -6 $O/pytest.log > $O/pytest_gpu_tail.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
