#!/bin/bash
# round 6, call f: the whole GPU suite + the bench line with rccl_check, per-position launch accounting of cfg 3 / cfg 4, the epoch block
# after the Trainer's one-synchronisation epoch and the device-side permutation
O=gpurun_out/r06f; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log | cut -c1-200
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-200 $O/bench.json; tail -3 $O/bench.err | cut -c1-300
