# two engines on resident batches (tools/two_engines.py) with experimental builds of the library: do the chain-level stages of one
# engine's batch run under the other's alignment window when the persistent kernels leave wave slots free?
cd $GRAFT_REPO_ROOT
for v in base "$@"; do
  if [ $v = base ]; then unset MONICA_AMD_LIB; else export MONICA_AMD_LIB=$GRAFT_REPO_ROOT/variants/lib_$v.so; fi
  echo "== $v"
  python tools/two_engines.py 2 6 100000 2>&1 | tail -3
  python tools/two_engines.py 2 8 50000 2>&1 | tail -3
done
