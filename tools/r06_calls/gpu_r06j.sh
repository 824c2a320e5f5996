#!/bin/bash
# round 6, call j: the round's evidence on its last code -- rocprofv3 kernel stats + PMC passes of the bench command (tools/gpu_profile.sh),
# kernel stats of the cfg 3 / cfg 4 / cfg 1 lines, of the InfoNCE call, the whole GPU suite, smoke, and the default bench line as the driver runs it
O=gpurun_out/r06j; mkdir -p $O
bash tools/gpu_profile.sh r06j 2>&1 | tail -25 | cut -c1-300
python tools/pmc_summary.py gpurun_out/r06j $O/spmm_pmc_summary.json r06 final > $O/pmc_summary.log 2>&1; tail -2 $O/pmc_summary.log | cut -c1-300; cp profiles/spmm_traffic.json $O/spmm_traffic.json
bash tools/gpu_prof_configs.sh r06j 2>&1 | grep -E "== rocprof|infonce_bwd|spmm_swept" | cut -c1-200
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_inf -o infonce -- python $R/tools/infonce_profile.py 20 > $R/$O/prof_inf.log 2>&1; echo "rocprof infonce rc $?"
f=$(find $R/$O/prof_inf -name "*kernel_stats.csv" | head -1); if [ -n "$f" ]; then cp "$f" $R/$O/infonce_kernel_stats.csv; head -12 "$f" | cut -c1-140; fi
cd $R; rm -rf $O/prof $O/prof_* $O/pmc_*/
timeout 1500 python -m pytest tests -q -m gpu --durations=5 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log | cut -c1-200; tail -12 $O/pytest.log > $O/pytest_gpu_tail.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-400 $O/bench_line.json
