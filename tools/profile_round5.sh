# Round-5 profile: run from the repo root ON the GPU box (gpurun -- 'bash tools/profile_round5.sh').  As tools/profile_round.sh without
# the test suite: the bench line (default command), rocprofv3 kernel stats (one and three steps in flight) and the counter passes
# (each --pmc set in its own run, never combined with a trace).  Outputs land in gpurun_out/r5prof/.
set -e
export TMPDIR=/tmp
O=gpurun_out/r5prof
mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o s -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --pipeline 1 > $O/prof_stats.log 2>&1
echo stats1 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_p3 -o s -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_stats_p3.log 2>&1
echo stats3 done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --pipeline 1 > $O/pmc_f.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --pipeline 1 > $O/pmc_w.log 2>&1
echo write done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq1 -o q -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --pipeline 1 > $O/pmc_q1.log 2>&1
echo sq1 done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq2 -o q -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --pipeline 1 > $O/pmc_q2.log 2>&1
echo sq2 done
# keep the merged-back payload small: the per-dispatch traces are not needed, the stats and counter tables are
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.csv" | head -30
du -sh $O
# the parity plan on the fp16 matrix cores (f16x3): kernel stats of one step in flight
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_x3 -o s -- python bench.py --steps 10 --warmup 2 --dtype f16x3 --no-cpu-baseline --no-extras --no-roofline --pipeline 1 > $O/prof_stats_x3.log 2>&1
echo stats_x3 done
find $O -name "*kernel_trace.csv" -size +8M -delete
du -sh $O
