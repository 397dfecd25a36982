#!/bin/bash
# round-2 GPU run F: tests after the binding change, then same-box A/Bs: roofline overhead, stream-K, hardware queues
mkdir -p gpurun_out/r2f
timeout 420 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r2f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2f/pytest.log
B="python bench.py --no-cpu-baseline --gmmn-steps 0 --steps 10 --warmup 3"
run() { name=$1; shift; env "$@" timeout 120 $B $EXTRA > gpurun_out/r2f/$name.json 2> gpurun_out/r2f/$name.err; }
EXTRA="--no-roofline" run noroof_a ZS3_STREAMK=1
EXTRA="" run roof_a ZS3_STREAMK=1
EXTRA="--no-roofline" run sk0_a ZS3_STREAMK=0
EXTRA="--no-roofline" run q8_a ZS3_STREAMK=1 GPU_MAX_HW_QUEUES=8
EXTRA="--no-roofline" run noroof_b ZS3_STREAMK=1
EXTRA="--no-roofline" run sk0_b ZS3_STREAMK=0
EXTRA="--no-roofline" run q8_b ZS3_STREAMK=1 GPU_MAX_HW_QUEUES=8
EXTRA="--no-roofline" run q8s3 ZS3_STREAMK=1 GPU_MAX_HW_QUEUES=8 ZS3_WGRAD_STREAMS=3 ZS3_WGRAD_CUS=96
tail -4 gpurun_out/r2f/pytest.log
for f in gpurun_out/r2f/*.json; do echo $f $(grep -o '"ms_per_step": [0-9.]*' $f); done
