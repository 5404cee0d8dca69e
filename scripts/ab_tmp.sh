timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "several_pairs or parity_cases or c2_subset or randomised or all_align or alphabet" > gpurun_out/t32.log 2>&1; tail -n 2 gpurun_out/t32.log
show() { python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('$1',d['ms_per_step'],r['kernel_ms'])"; }
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | show C2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | show C2
SR_PROFILE_TICKS=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);k=d['kernels'];tp=k['ticks_pair']
for n in ('ticks_breakpoint','ticks_base','tk_pass','tk_barrier','tk_control','tk_phase2','tk_recompute'): print(n, round(k[n]/tp,4))
print('per pair ms', tp/4096/1e5)"
