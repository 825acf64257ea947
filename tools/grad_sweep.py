"""A/B of the gradient kernels on a 512^3 level of 128^3 boxes (one MI355X): the tiled k_grad against
k_grad_march over tile rows and z-segment length."""
import os, subprocess, json, sys
def run(env):
    out = subprocess.run([sys.executable, "tools/kernel_bench.py", "512", "128", "gradonly"], env=dict(os.environ, **env), capture_output=True, text=True).stdout
    try:
        d = json.loads(out)["kernels"]["k_grad"]
        return round(d["ms"], 4), round(d["frac_hbm"], 4)
    except Exception:
        return "failed", out[-300:]
box = sys.argv[1] if len(sys.argv) > 1 else "128"
def runb(env):
    out = subprocess.run([sys.executable, "tools/kernel_bench.py", "512", box, "gradonly"], env=dict(os.environ, **env), capture_output=True, text=True).stdout
    try:
        d = json.loads(out)["kernels"]["k_grad"]
        return round(d["ms"], 4), round(d["frac_hbm"], 4)
    except Exception:
        return "failed", out[-300:]
print("box", box, "tiled k_grad", runb({"PA_GRAD_MARCH": "0"}), flush=True)
for mty in (13, 8):
    for kseg in (8, 16, 22, 32, 43, 64):
        print("march", mty, kseg, runb({"PA_GRAD_MTY": str(mty), "PA_GRAD_KSEG": str(kseg)}), flush=True)
