#!/bin/bash
# cycle-level MFMA utilisation and effective clock of the bf16x3 K1 on fixed-length sets (tools/k1b_probe.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_k1b_probe; rm -rf $OUT; mkdir -p $OUT
ARGS="tools/k1b_probe.py ${T:-512}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $ARGS > $OUT/t.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, glob
dur = {}
for f in glob.glob('gpurun_out/pmc_k1b_probe/t/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'k1b_loglikes' in r['Kernel_Name']:
            dur.setdefault('k1b', []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
print('kernel-trace ms', dur)
for sub in 'ab':
    for f in glob.glob(f'gpurun_out/pmc_k1b_probe/{sub}/*/*_counter_collection.csv'):
        d = {}
        for r in csv.DictReader(open(f)):
            if 'k1b_loglikes' in r['Kernel_Name']:
                d.setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
                d[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
        for k, v in d.items():
            vals = list(v.values()); print(sub, k, sum(vals) / len(vals))
PY
