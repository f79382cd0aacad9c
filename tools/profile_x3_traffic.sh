# HBM traffic counters of the f16x3 plan (run ON the GPU box from the repo root); outputs in gpurun_out/r5x3pmc/.
set -e
export TMPDIR=/tmp
O=gpurun_out/r5x3pmc
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python bench.py --steps 2 --warmup 1 --dtype f16x3 --no-cpu-baseline --no-roofline --no-extras --pipeline 1 > $O/pmc_f.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python bench.py --steps 2 --warmup 1 --dtype f16x3 --no-cpu-baseline --no-roofline --no-extras --pipeline 1 > $O/pmc_w.log 2>&1
echo write done
