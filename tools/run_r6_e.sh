cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
F="--no-cpu-baseline --no-alt --no-extra --no-real-data --steps 10 --warmup 3"
for sk in 1 0 1 0; do echo "== CDETR_RCDA_SKIP_SAVE=$sk"; CDETR_RCDA_SKIP_SAVE=$sk python bench.py $F 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('train', round(r['ms_per_step'],3), 'inference', [(x.get('image'), x.get('images_per_launch', x.get('batch')), round(x['value'],1)) for x in r['inference']['shapes']] if 'shapes' in r['inference'] else r['inference'])
"; done > $O/ab_skip_save.txt 2>&1
cat $O/ab_skip_save.txt | cut -c1-600
