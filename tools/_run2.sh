set -x
python tools/kbench.py --only attn --dtype f16 --variants 0,5,6,7,8,9,10 > gpurun_out/kb2_attn_f16.txt 2>&1
python tools/kbench.py --only attn --dtype bf16 --variants 0,9,10 > gpurun_out/kb2_attn_bf16.txt 2>&1
python tools/debug_vits.py dinov2_vits14 672 1,2,12 1,2 > gpurun_out/dbg_vits_a.txt 2>&1
MHMR_GEMM128=1 python tools/debug_vits.py dinov2_vits14 672 1,12 1,2 > gpurun_out/dbg_vits_b.txt 2>&1
python tools/debug_vits.py dinov2_vits14 448 2 1,2,3 >> gpurun_out/dbg_vits_a.txt 2>&1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -q > gpurun_out/pytest_k2.log 2>&1
cat gpurun_out/kb2_attn_f16.txt gpurun_out/kb2_attn_bf16.txt gpurun_out/dbg_vits_a.txt gpurun_out/dbg_vits_b.txt; tail -n 15 gpurun_out/pytest_k2.log
