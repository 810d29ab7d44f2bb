mkdir -p gpurun_out/r02
export FFQ_BENCH_DRY_MULTI=1
for wl in single-1g wrapped-64m decode-64m; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --workload $wl > gpurun_out/r02/dry2_$wl.json 2> gpurun_out/r02/dry2_$wl.err
echo "rc=$?"; tail -c 600 gpurun_out/r02/dry2_$wl.json; tail -3 gpurun_out/r02/dry2_$wl.err
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --steps 4 --warmup 1 --workload single-64m > gpurun_out/r02/dry4.json 2> gpurun_out/r02/dry4.err
echo "rc=$?"; tail -c 400 gpurun_out/r02/dry4.json; tail -3 gpurun_out/r02/dry4.err
