cd "$GRAFT_REPO_ROOT"
for c in "mono100x8 1000" "tri2000x32 20000" "stress10000x128 2000"; do
  set -- $c
  python bench.py --config $1 --utts $2 --no-cpu-baseline --per-call-utts 0 --no-recipe-beam-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); k=d['kernel_ms_per_step']
print('$1', 'value %.1f M  step %.2f ms' % (d['value']/1e6, d['ms_per_step']), {a: round(b,2) for a,b in k.items()}, 'fp32 line %.1f M' % (d['fp32_mfma_line']['value']/1e6), 'frac', round(d['roofline']['frac'],3))"
done
python bench.py --k1 f16x2 --no-cpu-baseline --no-fp32-line --per-call-utts 0 --no-recipe-beam-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); k=d['kernel_ms_per_step']
print('tri5000x64 with --k1 f16x2', 'value %.1f M  step %.2f ms' % (d['value']/1e6, d['ms_per_step']), {a: round(b,2) for a,b in k.items()})"
python bench.py --utts 12500 --no-cpu-baseline --no-fp32-line --per-call-utts 0 --no-recipe-beam-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); k=d['kernel_ms_per_step']
print('8-GPU shard size (12500 utts)', 'value %.1f M  step %.2f ms' % (d['value']/1e6, d['ms_per_step']), {a: round(b,2) for a,b in k.items()})"
python bench.py --transcripts zipf --no-cpu-baseline --no-fp32-line --per-call-utts 0 --no-recipe-beam-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().splitlines()[-1]); k=d['kernel_ms_per_step']
print('tri5000x64, Zipf-lexicon transcripts', 'value %.1f M  step %.2f ms' % (d['value']/1e6, d['ms_per_step']), {a: round(b,2) for a,b in k.items()})"
