#!/bin/bash
# round 6, call l: the driver's sequence on the round's last code: full suite, smoke, the default bench line
O=gpurun_out/r06l; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -12 $O/pytest.log | cut -c1-200; tail -14 $O/pytest.log > $O/pytest_gpu_tail.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
/usr/bin/time -v timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-300 $O/bench_line.json; grep -E "Elapsed" $O/bench.err
