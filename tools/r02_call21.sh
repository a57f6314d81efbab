#!/bin/bash
O=gpurun_out/r02w5; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_parity_gpu.py tests/test_training_gpu.py -x -q -k "c4 or C4 or grad or training or adam or ddp or loss" > $O/tests.log 2>&1; grep -E "passed|failed|rel|err" $O/tests.log | tail -8
timeout 300 python bench.py --workload c4 --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err; tail -2 $O/bench_c4.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('c4', d['ms_per_step']); print(d['kernel_ms_per_step'])"
