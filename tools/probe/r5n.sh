#!/bin/bash
# round 5: where the one-rank N>1 path's +4.9 ms goes: GradSync alone, + SyncBN, and the kernel trace of the latter
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r5n; mkdir -p $O
F="--no-cpu-baseline --bf16-steps 0 --shard-steps 0 --ddp-steps 0 --script-steps 0 --gmmn-steps 0 --no-roofline --steps 20 --warmup 5"
timeout 300 python bench.py $F > $O/plain.json 2> $O/plain.err
timeout 300 python bench.py $F --ddp-selftest --sync-bn 0 > $O/ddp_nosyncbn.json 2> $O/ddp_nosyncbn.err
timeout 300 python bench.py $F --ddp-selftest --sync-bn 1 > $O/ddp_syncbn.json 2> $O/ddp_syncbn.err
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o ddp -- python bench.py $F --steps 5 --warmup 2 --ddp-selftest --sync-bn 1 > $O/prof.log 2>&1
python - <<'PY'
import json,glob,csv,os
O='gpurun_out/r5n'
for n in ('plain','ddp_nosyncbn','ddp_syncbn'):
    try:
        d=json.loads(open(f'{O}/{n}.json').read().strip().splitlines()[-1]); print(n, d['ms_per_step'])
    except Exception as e: print(n,'ERR',e, open(f'{O}/{n}.err').read()[-500:])
for f in glob.glob(O+'/prof/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print(f, 'total ms', tot/1e6)
    for r in rows[:45]:
        print(r['Name'][:70], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2), round(float(r['AverageNs'])/1e3,1))
PY
