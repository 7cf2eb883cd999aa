cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
OUT=gpurun_out/pmc_k1; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 bench.py --utts 25000 --steps 2 --warmup 1 --batches 2 --no-cpu-baseline > $OUT/bench.json 2> $OUT/log.txt
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/pmc_k1/f/*/*_counter_collection.csv')[0]
d={}
for r in csv.DictReader(open(f)):
    if 'k1_loglikes' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
        d.setdefault(r['Dispatch_Id'],0.0); d[r['Dispatch_Id']]+=float(r['Counter_Value'])
v=list(d.values()); print('K1 FETCH_SIZE KB per launch', sum(v)/len(v), 'x2 GB', sum(v)/len(v)*2048/1e9)
PY
cut -c1-900 $OUT/bench.json | tail -1
