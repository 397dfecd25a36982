#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 200 python tools/probe/bench_flags.py $1 -- --steps 20 --warmup 5 --no-cpu-baseline --gmmn-steps 0 --no-roofline --shard-steps 0 --ddp-steps 0 --bf16-steps 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2; do
  echo "[maxk 512 ] $(run ops.PW_MAXK=512)"
  echo "[maxk 1024] $(run ops.PW_MAXK=1024)"
  echo "[maxk 2048] $(run ops.PW_MAXK=2048)"
done
