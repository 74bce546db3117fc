R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 400 python -m pytest $R/tests/test_gpu_fused.py -q -m gpu -x 2>&1 | tail -4
python $R/scripts/stamp_passes.py 2>/dev/null
timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['roofline']['kernels'].items()})"
