# Round 4, the other workloads on the round's last build (gpurun -- 'bash tools/prof_r04_extra.sh r04z')
R=$GRAFT_REPO_ROOT; TAG=${1:-r04z}; O=$R/gpurun_out
cd $R
python tools/tier_fit.py 30000 > $O/${TAG}_tier_fit.txt 2>&1
bash tools/prof_r04_tails.sh > $O/${TAG}_tails.txt 2>&1
cd $R
python bench.py --mode shard --genomes 500 --parts 8 --reads 1000000 --block 100000 --steps 1 --warmup 1 > $O/${TAG}_config4_block100000.json 2> $O/${TAG}_config4.err
python bench.py --mode shard --genomes 500 --parts 8 --reads 1000000 --block 500000 --steps 1 --warmup 1 > $O/${TAG}_config4_block500000.json 2>> $O/${TAG}_config4.err
python bench.py --mode config3 --steps 1 --warmup 1 > $O/${TAG}_config3.json 2> $O/${TAG}_config3.err
python tools/len_profile.py > $O/${TAG}_len_profile.txt 2>&1
python tools/parity_sweep.py 20000 > $O/${TAG}_parity_sweep.txt 2>&1
bash tools/files_sweep.sh > $O/${TAG}_files_sweep.txt 2>&1
python bench.py --gpus 2 --one-device --backend gloo --steps 3 --warmup 1 --cpu-sample 0 > $O/${TAG}_two_ranks_one_gpu.json 2> $O/${TAG}_two_ranks.err
tail -n 3 $O/${TAG}_tier_fit.txt; tail -c 300 $O/${TAG}_config4_block100000.json; tail -c 300 $O/${TAG}_config3.json
