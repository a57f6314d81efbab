set -u
O=gpurun_out/r03j
mkdir -p $O
timeout 1500 python -m pytest tests/test_first_stage_gpu.py tests/test_egress_gpu.py tests/test_optim_gpu.py tests/test_dist_gpu.py -q -x -s > $O/tests1.log 2>&1; echo "tests rc=$?" >> $O/tests1.log; grep -E "golden|rel err|passed|failed|rc=" $O/tests1.log | tail -8
timeout 1500 python -m pytest tests/test_fullsize_parity_gpu.py -q -x -s -k "benchmarked_plan or c4_full" > $O/tests2.log 2>&1; echo "tests rc=$?" >> $O/tests2.log; grep -E "benchmarked plan|sign flips|passed|failed|rc=|Error" $O/tests2.log | tail -14
timeout 1500 python -m pytest tests/test_training_gpu.py tests/test_latent_gpu.py tests/test_model_gpu.py -q -x > $O/tests3.log 2>&1; echo "tests rc=$?" >> $O/tests3.log; tail -3 $O/tests3.log
