mkdir -p gpurun_out/r02f
for mode in all_gather pipelined; do
SSLREC_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --shard-mode $mode > gpurun_out/r02f/bench2_$mode.log 2> gpurun_out/r02f/bench2_$mode.err; echo "bench2 $mode exit $?"; tail -c 1200 gpurun_out/r02f/bench2_$mode.log; tail -3 gpurun_out/r02f/bench2_$mode.err
done
