for k in 16 22 26 32 43 64; do
PA_KSEG=$k python bench.py --steps 20 --warmup 5 --sim-of 8 > gpurun_out/r2h_sim8_k$k.json 2> gpurun_out/r2h_sim8_k$k.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r2h_sim8_k$k.json").read().strip().splitlines()[-1])
print("kseg $k", "ms/step %.3f"%d["ms_per_step"], "sweep %.4f frac %.3f"%(d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
PY
done
for k in 16 32 43 64; do
PA_KSEG=$k python bench.py --steps 20 --warmup 5 --sim-of 4 > gpurun_out/r2h_sim4_k$k.json 2> gpurun_out/r2h_sim4_k$k.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r2h_sim4_k$k.json").read().strip().splitlines()[-1])
print("N=4 kseg $k", "ms/step %.3f"%d["ms_per_step"], "sweep %.4f frac %.3f"%(d["roofline"]["avg_launch_ms"], d["roofline"]["frac"]))
PY
done
