# Round 5, the other workloads on the round's last build (gpurun -- 'bash tools/prof_r05_extra.sh r05z')
R=$GRAFT_REPO_ROOT; TAG=${1:-r05z}; O=$R/gpurun_out
cd $R
bash tools/prof_r05_tails.sh > $O/${TAG}_tails.txt 2>&1
cd $R
python bench.py --mode shard --genomes 500 --parts 8 --reads 1000000 --block 100000 --steps 1 --warmup 1 > $O/${TAG}_config4_block100000.json 2> $O/${TAG}_config4.err
python bench.py --mode shard --genomes 500 --parts 8 --reads 1000000 --block 500000 --steps 1 --warmup 1 > $O/${TAG}_config4_block500000.json 2>> $O/${TAG}_config4.err
python bench.py --mode config3 --steps 1 --warmup 1 > $O/${TAG}_config3.json 2> $O/${TAG}_config3.err
python tools/len_profile.py > $O/${TAG}_len_profile.txt 2>&1
bash tools/files_sweep.sh > $O/${TAG}_files_sweep.txt 2>&1
python bench.py --gpus 2 --one-device --backend gloo --steps 3 --warmup 1 --cpu-sample 0 > $O/${TAG}_two_ranks_one_gpu.json 2> $O/${TAG}_two_ranks.err
python bench.py --mode stream --stream-seconds 600 > $O/${TAG}_stream600.json 2> /dev/null
tail -c 300 $O/${TAG}_config4_block100000.json; tail -c 300 $O/${TAG}_config3.json
