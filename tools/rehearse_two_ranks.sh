# The N > 1 code paths of bench.py with two ranks sharing the box's one GPU (gloo instead of RCCL; kernels unchanged):
# gpurun -- 'bash tools/rehearse_two_ranks.sh > gpurun_out/two_ranks.txt 2>&1'
R=$GRAFT_REPO_ROOT
cd $R
run() { timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 --backend gloo --one-device "${@:2}" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('metric','value','n_gpus','steps','ms_per_step','scaling','mapped_reads_last_step','counts_checksum','job_s','wall_s')}, d.get('config',{}).get('parallelism'))"; }
echo "default (weak, read-sharded)"; run 29511 --steps 3 --warmup 1 --cpu-sample 0 --reads 50000
echo "strong"; run 29512 --steps 3 --warmup 1 --cpu-sample 0 --scaling strong --total-reads 100000
echo "config3 (1 M reads as one job)"; run 29513 --mode config3 --total-reads 1000000 --reads 100000
echo "shard (config 4 shape, 64 genomes in 2 parts)"; run 29514 --mode shard --genomes 64 --parts 2 --reads 100000 --block 50000 --steps 2 --warmup 1
