cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/finprobe
for sh in "6144 6144 1" "16384 1001 1" "512 512 9" "2048 512 1" "4096 4096 1"; do
  n=$(echo $sh | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/finprobe/$n -- python3 tools/shape_case.py $sh --ops k2 --iters 30 > /dev/null 2>&1 || exit 1
  f=$(find gpurun_out/finprobe/$n -name "*kernel_stats.csv" | head -1)
  echo "== $sh"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "lq::" in r["Name"] and "selftest" not in r["Name"]:
        print(f'  {r["Name"].split("(")[0][:70]:70s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:8.2f}')
PY
done
find gpurun_out -name "*kernel_trace.csv" -delete
