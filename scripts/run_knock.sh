R=$GRAFT_REPO_ROOT; cd /tmp
for spg in 8 16 20 25 40; do
  timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-secondary --steps 2000 --steps-per-graph $spg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('spg $spg', d['value'], d['ms_per_step'], k['fps']['avg_us'], k['sa_prep_stats']['avg_us'], k['sa_bwd_point_grads']['avg_us'])"
done
