python tools/kbench.py --only attn --dtype f16 --variants 0,1,2 > gpurun_out/kb3_attn_f16.txt 2>&1
python tools/kbench.py --only attn --dtype bf16 --variants 0,1,2 > gpurun_out/kb3_attn_bf16.txt 2>&1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_parity_fullsize.py -q > gpurun_out/pytest_k3.log 2>&1
cat gpurun_out/kb3_attn_f16.txt gpurun_out/kb3_attn_bf16.txt; tail -n 25 gpurun_out/pytest_k3.log
