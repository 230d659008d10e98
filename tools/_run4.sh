python tools/kbench.py --only attn --dtype f16 --variants 0,1,2 > gpurun_out/kb4_attn_f16.txt 2>&1
python tools/kbench.py --only attn --dtype bf16 --variants 0,1,2 > gpurun_out/kb4_attn_bf16.txt 2>&1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_anny_model.py tests/test_anny_hph.py tests/test_gpu_fullsize.py -q -x > gpurun_out/pytest_k4.log 2>&1
cat gpurun_out/kb4_attn_f16.txt gpurun_out/kb4_attn_bf16.txt; tail -n 25 gpurun_out/pytest_k4.log
