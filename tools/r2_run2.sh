python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r2c_n1.json 2> gpurun_out/r2c_n1.err
PA_FUSED2=0 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r2c_n1_old.json 2> gpurun_out/r2c_n1_old.err
for n in 2 4 8; do python bench.py --steps 20 --warmup 5 --sim-of $n > gpurun_out/r2c_sim$n.json 2> gpurun_out/r2c_sim$n.err; done
python - <<'PY'
import json
for f in ("r2c_n1","r2c_n1_old","r2c_sim2","r2c_sim4","r2c_sim8"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        bd=d.get("breakdown_ms_per_step",{})
        print(f, "ms/step %.3f"%d["ms_per_step"], "value %.0f"%d["value"], "sweep %.3f frac %.3f"%(d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), {k:round(v,3) for k,v in bd.items()})
    except Exception as e:
        print(f, "ERR", e, open(f"gpurun_out/{f}.err").read()[-500:])
PY
