set -e
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > gpurun_out/t_full.log 2>&1 || { tail -30 gpurun_out/t_full.log; exit 1; }
tail -2 gpurun_out/t_full.log
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_v3.json 2> gpurun_out/bench_v3.err
tail -1 gpurun_out/bench_v3.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v3 -o v3 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_v3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/pmc_w.log 2>&1
find gpurun_out/prof_v3 gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" | head -20
