#!/bin/bash
# round 4, call a: the GPU suite (FWD_W InfoNCE default, RCCL world-1 tests) + InfoNCE A/B of the two forward forms
O=gpurun_out/r04a; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
INFONCE_MODES=fp32,x6 python tools/infonce_modes.py $O/infonce_modes.json 2>&1 | tail -6
