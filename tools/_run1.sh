set -x
mkdir -p gpurun_out
./tools/ubench/valu_rates > gpurun_out/ubench_valu.txt 2>&1
./tools/ubench/mfma_valu_overlap > gpurun_out/ubench_overlap.txt 2>&1
python tools/kbench.py --only attn --dtype f16 --variants 0,1,3,4,5 > gpurun_out/kb_attn_f16.txt 2>&1
python tools/kbench.py --only attn --dtype bf16 --variants 0,1,3,4,5 > gpurun_out/kb_attn_bf16.txt 2>&1
python -m pytest tests/test_gpu_kernels.py -x -q -k "attention or gemm" > gpurun_out/pytest_k1.log 2>&1
python -m pytest tests/test_gpu_parity_fullsize.py -q -k f16 > gpurun_out/pytest_p1.log 2>&1
tail -3 gpurun_out/pytest_k1.log gpurun_out/pytest_p1.log; cat gpurun_out/kb_attn_f16.txt gpurun_out/kb_attn_bf16.txt
