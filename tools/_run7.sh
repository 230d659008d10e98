python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -q -x -k "lbs" > gpurun_out/pytest_lbs7.log 2>&1
tail -n 12 gpurun_out/pytest_lbs7.log
for P in 160 20 1 300; do python tools/lbs_bench.py $P; done 2>&1 | grep -v amdgpu
