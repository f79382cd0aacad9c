# SQ counter passes of the f16x3 plan (run ON the GPU box from the repo root); outputs in gpurun_out/r5x3pmc/.
set -e
export TMPDIR=/tmp
O=gpurun_out/r5x3pmc
mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq1 -o q -- python bench.py --steps 2 --warmup 1 --dtype f16x3 --no-cpu-baseline --no-roofline --no-extras --pipeline 1 > $O/pmc_q1.log 2>&1
echo sq1 done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq2 -o q -- python bench.py --steps 2 --warmup 1 --dtype f16x3 --no-cpu-baseline --no-roofline --no-extras --pipeline 1 > $O/pmc_q2.log 2>&1
echo sq2 done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_x3 -o s -- python bench.py --steps 10 --warmup 2 --dtype f16x3 --no-cpu-baseline --no-extras --no-roofline --pipeline 1 > $O/prof_stats_x3.log 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
du -sh $O
