#!/bin/bash
# rocprofv3 passes for profiles/ (round 6): kernel trace + stats, then PMC passes in their own runs (the pool refuses --pmc
# combined with other trace domains).  The bench runs its default workload (f16x2s K1) with the fp32-MFMA side line.
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r6
rm -rf $OUT; mkdir -p $OUT
python3 -c "import bench; print(bench.csrc_sha())" > $OUT/csrc_sha.txt      # identity of the sources these passes run on
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline --per-call-utts 0 --no-recipe-beam-line"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $ARGS > /dev/null 2> $OUT/pmc_mfma.log
# K1 issue-level counters (summarised by tools/summarize_k1_pmc.py)
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/b -- python3 $ARGS > /dev/null 2> $OUT/b.log
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES --output-format csv -d $OUT/c -- python3 $ARGS > /dev/null 2> $OUT/c.log
rm -rf $OUT/a; mkdir -p $OUT/a; cp -r $OUT/pmc_mfma/* $OUT/a/
python3 tools/summarize_k1_pmc.py $OUT > $OUT/k1_issue_pmc.json
ls -R $OUT | head -40
tail -2 $OUT/bench_trace.json | cut -c1-400
