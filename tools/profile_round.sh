# Round profile: run from the repo root ON the GPU box (gpurun -- 'bash tools/profile_round.sh').
# Full GPU test suite, the bench line (default command), rocprofv3 kernel stats, and counter passes (each --pmc set in its own run,
# never combined with a trace: MI355X_MICROARCH.md / gpurun rules).  Outputs land in gpurun_out/.
set -e
export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > gpurun_out/t_full.log 2>&1 || { tail -30 gpurun_out/t_full.log; exit 1; }
tail -2 gpurun_out/t_full.log
python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err
tail -1 gpurun_out/bench.json | cut -c1-400
# kernel stats twice: with ONE step in flight (per-kernel durations comparable with bench.py's live HIP-event numbers, which are
# taken launch by launch) and with the default three (durations inflated wherever the two streams' kernels share the GPU)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --pipeline 1 > gpurun_out/prof_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats_p2 -o s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_stats_p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --pipeline 1 > gpurun_out/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --pipeline 1 > gpurun_out/pmc_w.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq1 -o q -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --pipeline 1 > gpurun_out/pmc_q1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_sq2 -o q -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --pipeline 1 > gpurun_out/pmc_q2.log 2>&1
# (counter passes run ONE step in flight: with several, kernels of different steps share the GPU and per-dispatch cycle counts stretch)
find gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 -name "*.csv" | head -20
