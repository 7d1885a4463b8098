set -u
cd "$GRAFT_REPO_ROOT"
run() { name="$1"; shift; timeout -k 10 400 python bench.py "$@" > gpurun_out/cfg_$name.log 2>&1; python3 - "$name" <<'PY'
import json,sys
name=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/cfg_{name}.log").read().strip().splitlines()[-1])
    p=d.get("parity",{})
    print(name, round(d["value"],1), round(d.get("reference_precision",{}).get("value",0),1), round(d.get("production_f32",{}).get("value",0),1), p.get("mixed_vs_f64_rmse"), p.get("f32_vs_f64_rmse"))
except Exception as e:
    print(name, "failed", e)
PY
}
run c1 --tris 100000 --width 1024 --height 1024 --spp 64 --alt-steps 1 --no-cpu-baseline
run c3 --config 3 --spp 64 --alt-steps 0 --no-cpu-baseline
run mats --materials mixed --spp 64 --alt-steps 1 --no-cpu-baseline
run soup10m --tris 10000000 --spp 64 --alt-steps 0 --no-cpu-baseline --builder auto
run inst --instanced 1000x10000 --spp 64 --alt-steps 1 --no-cpu-baseline
run instflat --instanced 1000x10000 --flatten --builder auto --precision f32 --spp 64 --alt-steps 0 --no-cpu-baseline
