cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_k1p; rm -rf $OUT; mkdir -p $OUT
ARGS="bench.py --utts 30000 --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/a -- python3 $ARGS > /dev/null 2> $OUT/a.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d $OUT/b -- python3 $ARGS > /dev/null 2> $OUT/b.log
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $OUT/c -- python3 $ARGS > /dev/null 2> $OUT/c.log
python3 - <<'PY'
import csv,glob
for sub in 'abc':
    fs=glob.glob(f'gpurun_out/pmc_k1p/{sub}/*/*_counter_collection.csv')
    if not fs: print(sub,'no file'); continue
    d={}
    for r in csv.DictReader(open(fs[0])):
        if 'k1p_loglikes' in r['Kernel_Name'] or 'k1_loglikes' in r['Kernel_Name']:
            d.setdefault(r['Counter_Name'],{}).setdefault(r['Dispatch_Id'],0.0)
            d[r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
    for k,v in d.items():
        vals=list(v.values()); print(sub,k,sum(vals)/len(vals))
PY
