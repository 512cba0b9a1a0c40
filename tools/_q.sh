python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r06j_gputest.log; tail -3 gpurun_out/r06j_gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8
python bench.py > gpurun_out/r06j_bench_default.json 2> gpurun_out/r06j_bench_default.err; tail -c 600 gpurun_out/r06j_bench_default.json
