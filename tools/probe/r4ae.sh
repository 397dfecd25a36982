#!/bin/bash
# last call of the round: the GMMN kernel trace after the six-launch update, then the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt_gmmn
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt_gmmn -o p -- python $R/bench.py --no-cpu-baseline --bf16-steps 0 --workload gmmn --steps 4 --warmup 2 --no-roofline > $O/kt_gmmn.log 2>&1
python $R/tools/rocprof_summary.py $(find $O/kt_gmmn -name '*.db' | head -1) 45 > $O/kt_gmmn.md
find $O/kt_gmmn -type f ! -name '*.md' -delete
grep '^{' $O/kt_gmmn.log | tail -1 > $O/bench_gmmn.json
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_final2.json 2> gpurun_out/r4_bench_final2.err
python -c "
import json; d=json.loads(open('gpurun_out/r4_bench_final2.json').read().strip().splitlines()[-1])
print('main', round(d['ms_per_step'],2), round(d['value'],1), d['last_loss'], d['roofline']['kernel'], round(d['roofline']['achieved'],1), round(d['roofline']['frac'],4))
b=d['bf16']; print('bf16', round(b['ms_per_step'],2), round(b['value'],1), b['last_loss'], b['roofline']['kernel'], round(b['roofline']['achieved'],1), round(b['roofline']['frac'],4))
g=d['gmmn']; print('gmmn', round(g['ms_per_step'],2), round(g['value'],1), g['breakdown'], g['roofline']['generator_update']['launches_per_update'])"
