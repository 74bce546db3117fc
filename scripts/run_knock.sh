R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-secondary --pipeline off --graph off --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('standalone eager', d['value'], d['ms_per_step'], {k:v['avg_us'] for k,v in d['roofline']['kernels'].items()})"
timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-secondary --pipeline off --steps 2000 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('graph, no pipeline', d['value'], d['ms_per_step'])"
