# FETCH_SIZE / WRITE_SIZE of the options kernels (k_curvopts) for a list of "YQ TY KZ" settings; bench.py's options entry under rocprofv3 --pmc
export TMPDIR=/tmp
for cfg in "${@:-1 4 64}"; do set -- $cfg
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pp; PA_OPT_YQ=$1 PA_OPT_TY=$2 PA_OPT_KZ=$3 timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pp -- python3 bench.py --secondary-only f1_curvature_options_headline > /tmp/pp.log 2>&1
    python3 - "$cfg" $C <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for p in glob.glob("/tmp/pp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] == sys.argv[2] and "k_curvopts" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(sys.argv[1], sys.argv[2], k, "calls", len(v), "avg GB", (2 if sys.argv[2] == "FETCH_SIZE" else 1) * sum(v) / len(v) * 1024 / 1e9)
PY
  done
done
