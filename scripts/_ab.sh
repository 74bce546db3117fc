for opt in "--index-overlap off" "--index-overlap on" "--index-overlap off" "--index-overlap on"; do
python bench.py --no-cpu-baseline $opt > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$opt', d['value'], d['ms_per_step'])"
done
