python -m pytest tests/test_gpu_gradcurv.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r2_t7.log 2>&1; echo rc=$? >> gpurun_out/r2_t7.log; tail -12 gpurun_out/r2_t7.log
python bench.py --steps 5 --warmup 2 --no-cpu --base 256 --nlev 4 --box 64 --ncomp 55 > gpurun_out/r2g_c5.json 2> gpurun_out/r2g_c5.err
python bench.py --steps 5 --warmup 2 --no-cpu --nlev 1 --ncomp 10 > gpurun_out/r2g_c2.json 2> gpurun_out/r2g_c2.err
python bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/r2g_n1.json 2> gpurun_out/r2g_n1.err
python - <<'PY'
import json
for f in ("r2g_c5","r2g_c2","r2g_n1"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        bd=d.get("breakdown_ms_per_step",{})
        print(f, "ms/step %.3f"%d["ms_per_step"], "value %.0f"%d["value"], "sweep %.3f frac %.3f"%(d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), {k:round(v,3) for k,v in bd.items()})
    except Exception as e:
        print(f, "ERR", e, open(f"gpurun_out/{f}.err").read()[-800:])
PY
