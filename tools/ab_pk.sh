for r in 1 2; do
for v in a b; do
LAVIE_HIP_LIB=$PWD/ab_libs/$v.so python bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-extra-legs > gpurun_out/ab_pk_$v$r.json 2> gpurun_out/ab_pk_$v$r.err || exit 1
python - <<PY
import json
d=json.loads(open("gpurun_out/ab_pk_$v$r.json").read().strip().splitlines()[-1])
print("$v$r", d["value"], d["ms_per_step"], [(c["name"][:14], c["ms"]) for c in d["kernel_breakdown"]["classes"] if c["ms"]])
PY
done; done
