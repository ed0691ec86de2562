mkdir -p gpurun_out
python -m pytest tests/test_bf16_gpu.py tests/test_runner_gpu.py tests/test_getup_gpu.py tests/test_agent_parity2_gpu.py tests/test_return_parity_gpu.py -m gpu -q -x 2>&1 | tail -40 > gpurun_out/tests_a.log
tools/gemm_bench --bf16 --only upd > gpurun_out/gemm_bf16.txt 2>&1
python tools/return_parity.py --side device --iters 1000 --out gpurun_out/r02_return_parity_device.json > gpurun_out/return_parity_device.log 2>&1
tail -30 gpurun_out/tests_a.log; grep bf16 gpurun_out/gemm_bf16.txt; tail -12 gpurun_out/return_parity_device.log
