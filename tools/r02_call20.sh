#!/bin/bash
O=gpurun_out/r02w4; mkdir -p $O
timeout 600 python -m pytest tests/test_backward_kernels_gpu.py -x -q -k "groupnorm or resample" > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2
timeout 300 python bench.py --workload c4 --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err; tail -2 $O/bench_c4.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('c4', d['ms_per_step']); print(d['kernel_ms_per_step'])"
