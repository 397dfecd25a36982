#!/bin/bash
# round 5: weight-gradient side streams at a lower HIP priority than the main stream; pool sizes re-checked
cd "$GRAFT_REPO_ROOT"
python -c "
import ctypes, sys; sys.path.insert(0, '.')
from zs3_amd._lib import lib
lo, hi = ctypes.c_int(9), ctypes.c_int(9); print(lib().zs3_stream_priority_range(ctypes.byref(lo), ctypes.byref(hi)), 'hip range least', lo.value, 'greatest', hi.value)
import torch
print('priority range', torch.cuda.Stream.priority_range())
for p in (-1,0,1,2): print(p, torch.cuda.Stream(priority=p).priority)"
F="--no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --no-roofline"
run() { timeout 300 python tools/probe/bench_flags.py $1 -- $F --steps 20 --warmup 5 $2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms loss %.6f' % (d['ms_per_step'], d['last_loss']))"; }
for rep in 1 2; do
  for pr in 0 1 2; do echo "[prio $pr] $(run functional.WGRAD_STREAM_PRIORITY=$pr)"; done


done
echo "[prio 1, roofline events on] $(timeout 300 python tools/probe/bench_flags.py functional.WGRAD_STREAM_PRIORITY=1 -- --no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])")"
