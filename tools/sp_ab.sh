for v in ${SP_DBG:-0}; do
FDX_CSR_DBG=$v python bench.py --family sparse --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/sp_$v.json 2> gpurun_out/sp_$v.err
python - <<PY
import json
d=json.loads(open("gpurun_out/sp_$v.json").read().strip().splitlines()[-1])
s=d.get("sparse_csr", d)
print("dbg $v sketch_ms", s["stage_ms"]["sketch_ms"], "step", round(s["ms_per_step"],3))
PY
done
