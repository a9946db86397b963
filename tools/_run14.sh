cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
for i in 1 2 3; do python3 bench.py > gpurun_out/r05/bench_default_rep$i.json 2> gpurun_out/r05/bench_default_rep$i.err; python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r05/bench_default_rep$i.json').read().strip().splitlines()[-1]); print('rep$i', d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline'].get('traffic'))"; done
