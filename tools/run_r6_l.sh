cd $GRAFT_REPO_ROOT
O=gpurun_out/r6l; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q > $O/t_kern.log 2>&1; echo "kernels rc=$?"; tail -2 $O/t_kern.log
python tools/rcda_time.py 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_model_gpu.py tests/test_timed_path_gpu.py -m gpu -x -q > $O/t_full.log 2>&1; echo "full rc=$?"; tail -2 $O/t_full.log
python bench.py --no-cpu-baseline --no-alt --no-extra --no-real-data > $O/bench.log 2>$O/bench.err; python - <<'PY'
import json
r=json.loads([l for l in open('gpurun_out/r6l/bench.log') if l.startswith('{')][-1])
print('train', r['ms_per_step'], r['value'], r['step_ms'])
print({k:(round(v['tflops'],1), round(v['ms_per_step'],3)) for k,v in r['roofline']['families'].items()})
print([(s['image'], s['images_per_gpu'], round(s['graph']['value'],1)) for s in r['inference']['shapes']])
PY
