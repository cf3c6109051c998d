# scratch batch for one gpurun call (edited per experiment): the full GPU suite, smoke, default bench
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_default.log 2>&1; python3 -c "
import json; d=json.loads(open('gpurun_out/bench_default.log').read().strip().split('\n')[-1]); print(d['value'], d['kernel_ms_per_launch'], d['roofline']['frac'], d['valu']['frac'], d['cpu_baseline']['value'])"
