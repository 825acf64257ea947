python -m pytest tests/test_gpu_gradcurv.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r2_t3.log 2>&1; echo rc=$? >> gpurun_out/r2_t3.log; tail -3 gpurun_out/r2_t3.log
for ov in 1 0 1 0; do PA_OVERLAP2=$ov python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r2d_ov$ov.json 2> gpurun_out/r2d_ov$ov.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r2d_ov$ov.json").read().strip().splitlines()[-1])
bd=d.get("breakdown_ms_per_step",{})
print("ov$ov", "ms/step %.3f"%d["ms_per_step"], "value %.0f"%d["value"], "sweep %.3f frac %.3f"%(d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), {k:round(v,3) for k,v in bd.items()})
PY
done
