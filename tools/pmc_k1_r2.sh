#!/bin/bash
# K1 issue-level counters at the bench's full launch size; summarised by tools/summarize_k1_pmc.py
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_k1_r2; rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/a -- python3 $ARGS > /dev/null 2> $OUT/a.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/b -- python3 $ARGS > /dev/null 2> $OUT/b.log
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/c -- python3 $ARGS > /dev/null 2> $OUT/c.log
python3 tools/summarize_k1_pmc.py $OUT > $OUT/summary.json; cat $OUT/summary.json
