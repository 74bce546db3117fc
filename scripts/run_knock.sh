R=$GRAFT_REPO_ROOT; cd /tmp
timeout -k 10 300 python -m pytest $R/tests/test_gpu_fused.py -q -m gpu -x -k "tile_map or backward_matches" 2>&1 | tail -5
for t in 0 1; do
  APN_TMAP_BWD=$t timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('tmap_bwd $t', d['value'], d['ms_per_step'], d['roofline']['kernels']['sa_bwd_main']['avg_us'])"
done
