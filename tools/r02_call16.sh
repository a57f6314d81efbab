#!/bin/bash
# Winograd-domain weight gradient: parity at the training shapes, then the micro-benchmark against the direct kernel
O=gpurun_out/r02w; mkdir -p $O
timeout 600 python -m pytest tests/test_backward_kernels_gpu.py -x -q -k "winograd_wgrad or gemm_tn" -s 2>&1 | grep -E "rel err|passed|failed|Error|error" | tail -30 > $O/wgrad_tests.txt
cat $O/wgrad_tests.txt
timeout 300 python tools/wgrad_bench.py > $O/wgrad_bench.txt 2>&1; cat $O/wgrad_bench.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "winograd" 2>&1 | tail -3
timeout 300 python bench.py --workload c4 --no-cpu > $O/bench_c4.json 2> $O/bench_c4.err; python -c "
import json; d=json.load(open('$O/bench_c4.json')); print('c4', d['ms_per_step'])"
BBDM_WINOGRAD_WGRAD=0 timeout 300 python bench.py --workload c4 --no-cpu > $O/bench_c4_direct.json 2> /dev/null; python -c "
import json; d=json.load(open('$O/bench_c4_direct.json')); print('c4 direct wgrad', d['ms_per_step'])"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu > $O/bench_c2.json 2> $O/bench_c2.err; python -c "
import json; d=json.load(open('$O/bench_c2.json')); print('c2', d['ms_per_step'], d['kernel_ms_per_step'])"
