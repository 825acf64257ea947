python -m pytest tests/test_gpu_gradcurv.py tests/test_gpu_random.py -x -q -m gpu > gpurun_out/r2_t9.log 2>&1; echo rc=$? >> gpurun_out/r2_t9.log; tail -3 gpurun_out/r2_t9.log
for n in 1 2 4 8; do
if [ $n = 1 ]; then A=""; else A="--sim-of $n"; fi
python bench.py --steps 20 --warmup 5 --no-cpu $A > gpurun_out/r2i_sim$n.json 2> gpurun_out/r2i_sim$n.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r2i_sim$n.json").read().strip().splitlines()[-1])
bd=d.get("breakdown_ms_per_step",{})
print("N=$n", "ms/step %.3f"%d["ms_per_step"], "sweep %.4f frac %.3f"%(d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]), {k:round(v,3) for k,v in bd.items()})
PY
done
python bench.py --steps 5 --warmup 2 --no-cpu --base 256 --nlev 4 --box 64 --ncomp 55 > gpurun_out/r2i_c5.json 2> gpurun_out/r2i_c5.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r2i_c5.json").read().strip().splitlines()[-1])
print("C5", "ms/step %.3f"%d["ms_per_step"], "value %.0f"%d["value"], "sweep %.4f frac %.3f"%(d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
PY
