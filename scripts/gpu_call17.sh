cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c17
export SPHX_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 12 --warmup 11 --particles 4e6 > gpurun_out/c17/bench_2ranks.json 2> gpurun_out/c17/bench_2ranks.err
echo "rc=$?" >> gpurun_out/c17/bench_2ranks.err
unset SPHX_BENCH_BACKEND
python bench.py --steps 12 --warmup 11 --particles 4e6 --no-cpu-baseline > gpurun_out/c17/bench_1rank.json 2> gpurun_out/c17/bench_1rank.err
