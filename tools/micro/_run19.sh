mkdir -p gpurun_out/r5
timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r5/pytest3.log 2>&1; tail -4 gpurun_out/r5/pytest3.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r5/bench_default.json 2> gpurun_out/r5/bench_default.err; python -c "
import json;d=json.load(open('gpurun_out/r5/bench_default.json'));print(d['value'],d['ms_per_step'],d['config']['stage_s_per_step']);c=d['config']['cli'];print(c['reads_per_s'],c['wall_s'],c['ratio_to_hot_path'],c['stage_s']);print(d['roofline']['traffic_source'], d['roofline']['valu_issue']['frac'], d['roofline']['valu_issue'].get('pipe_busy_by_counters'), d['cpu_baseline']['all_cores']['value'])"
