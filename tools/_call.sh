python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03e_tests.txt 2>&1; tail -3 gpurun_out/r03e_tests.txt
python3 tools/split_sweep.py c3 default 8192,-1,-1,-1,3 32768,-1,-1,-1,3 > gpurun_out/r03e_sweep_c3.txt 2>&1; cat gpurun_out/r03e_sweep_c3.txt
python3 tools/split_sweep.py c5 default -1,-1,-1,-1,3,6,2048,2 > gpurun_out/r03e_sweep_c5.txt 2>&1; cat gpurun_out/r03e_sweep_c5.txt
python3 bench.py --workload c2 --no-pmc --no-cpu-baseline > gpurun_out/r03e_bench_c2.json 2>gpurun_out/r03e_bench_c2.err; python3 tools/jl.py gpurun_out/r03e_bench_c2.json 2>/dev/null | head -5; tail -c 600 gpurun_out/r03e_bench_c2.json
