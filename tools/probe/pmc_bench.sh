cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $set --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmcb_$set -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --gmmn-steps 0 --no-roofline > /dev/null 2>&1
  ls -la $GRAFT_REPO_ROOT/gpurun_out/pmcb_$set | tail -3
done
