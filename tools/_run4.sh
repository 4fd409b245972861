cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in 1 0; do
rm -rf $R/gpurun_out/kt_fold
ASR_AMD_FOLD_LN=$f rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_fold -- python3 $R/bench.py --no-cpu-baseline --steps 10 --graph 0 > /dev/null 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/kt_fold/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        n = r["Name"]
        if "ffn_bwd" in n or "add_layernorm_bwd" in n:
            print("FOLD=$f %-60s calls %s avg %.1f us total %.2f ms" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
