#!/bin/bash
O=gpurun_out/r03r; mkdir -p $O
timeout 600 python -m pytest tests -q -x -m gpu -k "replay or polynomial or generator or hip_graph" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
timeout 300 python tools/mt_replay_bench.py > $O/mt_replay.json 2> $O/mt_replay.err; cat $O/mt_replay.json; tail -3 $O/mt_replay.err
