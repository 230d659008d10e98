python -m pytest tests/ -m gpu -q -x > gpurun_out/pytest_all6.log 2>&1
tail -n 30 gpurun_out/pytest_all6.log
python bench.py --steps 20 --warmup 5 > gpurun_out/bench6.json 2> gpurun_out/bench6.err
tail -c 6000 gpurun_out/bench6.json; tail -n 5 gpurun_out/bench6.err
