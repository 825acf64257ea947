set -x
python -m pytest tests -x -q -m gpu > gpurun_out/r2_gpu_all.log 2>&1; echo rc=$? >> gpurun_out/r2_gpu_all.log; tail -5 gpurun_out/r2_gpu_all.log
python bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/r2_b_n1.json 2> gpurun_out/r2_b_n1.err; tail -c 1500 gpurun_out/r2_b_n1.json
for n in 2 4 8; do python bench.py --steps 10 --warmup 3 --sim-of $n > gpurun_out/r2_b_sim$n.json 2> gpurun_out/r2_b_sim$n.err; tail -c 1200 gpurun_out/r2_b_sim$n.json; done
PA_BENCH_REHEARSE=1 python bench.py --gpus 2 --steps 5 --warmup 2 --base 256 --box 64 > gpurun_out/r2_b_reh2.json 2> gpurun_out/r2_b_reh2.err; tail -c 1200 gpurun_out/r2_b_reh2.json; tail -5 gpurun_out/r2_b_reh2.err
PA_BENCH_REHEARSE=1 python bench.py --gpus 4 --steps 5 --warmup 2 --base 256 --box 64 > gpurun_out/r2_b_reh4.json 2> gpurun_out/r2_b_reh4.err; tail -c 1200 gpurun_out/r2_b_reh4.json; tail -5 gpurun_out/r2_b_reh4.err
