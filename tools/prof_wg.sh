cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/wg_probe.py 1500 64
python3 $R/tools/wg_probe.py 1500 512
python3 $R/tools/wg_probe.py 3000 2048 0,8,1,0x80
