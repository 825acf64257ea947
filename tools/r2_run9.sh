python -m pytest tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r2_t10.log 2>&1; echo rc=$? >> gpurun_out/r2_t10.log; tail -5 gpurun_out/r2_t10.log
for ov in 1 0; do
PA_XOVERLAP=$ov python bench.py --steps 20 --warmup 5 --sim-of 8 > gpurun_out/r2j_sim8_ov$ov.json 2> gpurun_out/r2j_sim8_ov$ov.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r2j_sim8_ov$ov.json").read().strip().splitlines()[-1])
bd=d.get("breakdown_ms_per_step",{})
print("N=8 xoverlap=$ov", "ms/step %.3f"%d["ms_per_step"], "sweep %.4f"%d["roofline"]["avg_launch_ms"], {k:round(v,3) for k,v in bd.items()})
PY
done
PA_BENCH_REHEARSE=1 python bench.py --gpus 4 --steps 5 --warmup 2 > gpurun_out/r2j_reh4.json 2> gpurun_out/r2j_reh4.err; tail -c 600 gpurun_out/r2j_reh4.json; tail -2 gpurun_out/r2j_reh4.err
