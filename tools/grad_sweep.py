import os, subprocess, json, sys
for ty in (4, 8, 16):
    for tz in (16, 32, 64, 128):
        env = dict(os.environ, PA_GRAD_TY=str(ty), PA_GRAD_TZ=str(tz))
        out = subprocess.run([sys.executable, "tools/kernel_bench.py", "512", "128", "gradonly"], env=env, capture_output=True, text=True).stdout
        try:
            d = json.loads(out)["kernels"]["k_grad"]
            print(ty, tz, round(d["ms"], 4), round(d["frac_hbm"], 4), flush=True)
        except Exception as e:
            print(ty, tz, "failed", out[-300:])
