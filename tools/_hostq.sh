for wl in cfg2 cfg3; do python bench.py --workload $wl --no-cpu-baseline --no-cold --no-e2e --steps 50 > gpurun_out/s36_$wl.json 2>gpurun_out/s36_$wl.err; python -c "
import json;d=json.load(open('gpurun_out/s36_$wl.json'));print('$wl',d['ms_per_step'],d['ms_per_step_index_ready'],d['host'])"; done
