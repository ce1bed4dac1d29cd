cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r04_gpu_tests_full.txt 2>&1; tail -3 gpurun_out/r04_gpu_tests_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r04_f_bench.json 2> gpurun_out/r04_f_bench.err; tail -c 600 gpurun_out/r04_f_bench.json; tail -3 gpurun_out/r04_f_bench.err
