python -m pytest tests/test_gpu_gradcurv.py tests/test_gpu_dist.py tests/test_gpu_random.py -x -q -m gpu --durations=5 > gpurun_out/r2_t4.log 2>&1; echo rc=$? >> gpurun_out/r2_t4.log; tail -12 gpurun_out/r2_t4.log
python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r2e_n1.json 2> gpurun_out/r2e_n1.err
for n in 2 4 8; do python bench.py --steps 20 --warmup 5 --sim-of $n > gpurun_out/r2e_sim$n.json 2> gpurun_out/r2e_sim$n.err; done
python - <<'PY'
import json
for f in ("r2e_n1","r2e_sim2","r2e_sim4","r2e_sim8"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        bd=d.get("breakdown_ms_per_step",{})
        print(f, "ms/step %.3f"%d["ms_per_step"], "value %.0f"%d["value"], "sweep %.3f frac %.3f"%(d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), {k:round(v,3) for k,v in bd.items()})
    except Exception as e:
        print(f, "ERR", e, open(f"gpurun_out/{f}.err").read()[-800:])
PY
