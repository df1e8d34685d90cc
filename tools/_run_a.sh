mkdir -p gpurun_out/r5d
for cfg in "X=1" "FDX_NO_STAGED_COPIES=1" "X=2" "FDX_NO_STAGED_COPIES=1"; do
env $cfg FDX_TRACE_HOST=1 python bench.py --family sparse --no-cpu-baseline > gpurun_out/r5d/sp.json 2> gpurun_out/r5d/sp.err
python -c "
import json
d=json.loads(open('gpurun_out/r5d/sp.json').read().strip().splitlines()[-1]); s=d.get('sparse_csr', d)
print('$cfg sparse', round(s['ms_per_step'],2), 'cold', s.get('cold_ms'))"
grep "python: entry" gpurun_out/r5d/sp.err | cut -c1-130
done
