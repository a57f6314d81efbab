#!/bin/bash
O=gpurun_out/r02w3; mkdir -p $O
timeout 900 python -m pytest tests/test_backward_kernels_gpu.py tests/test_training_gpu.py -x -q 2>&1 | tail -4
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or layernorm or geglu" 2>&1 | tail -2
timeout 300 python bench.py --workload c4 --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err; tail -2 $O/bench_c4.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('c4', d['ms_per_step']); print(d['kernel_ms_per_step'])"
