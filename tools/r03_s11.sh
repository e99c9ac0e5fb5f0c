cd /root/repo; mkdir -p gpurun_out/r03k
timeout 300 python bench.py --no-cpu-baseline --no-matrix > gpurun_out/r03k/stagger.json 2> gpurun_out/r03k/stagger.err
timeout 300 python bench.py --no-cpu-baseline --no-matrix --no-start-stagger --steps 20 --warmup 6 > gpurun_out/r03k/r02proto.json 2> gpurun_out/r03k/r02proto.err
python - <<PY
import json
for f in ("stagger","r02proto"):
    try:
        d=json.load(open(f"gpurun_out/r03k/{f}.json")); print(f, round(d["value"]), round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["kernel_ms_per_step"].items()}, round(d["games_per_sec"]), round(d["sp_phases"]["states_per_step"]), d["sp_phases"]["share"])
    except Exception as e: print(f, "no result", e); print(open(f"gpurun_out/r03k/{f}.err").read()[-500:])
PY
