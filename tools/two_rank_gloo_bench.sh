# debug aid: 2 ranks on ONE GPU (gloo rendezvous) through bench.py's N>1 code path, both database placements
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for mode in replicate shard; do
SYLPH_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 3 --warmup 1 --workload c2 --db-mode $mode --no-cpu-baseline > gpurun_out/two_rank_$mode.out 2>gpurun_out/two_rank_$mode.err
echo "rc=$? mode=$mode"; tail -c 600 gpurun_out/two_rank_$mode.out; grep -v "amdgpu.ids\|socket.cpp\|OMP_NUM\|\*\*\*\*" gpurun_out/two_rank_$mode.err | tail -15
done
