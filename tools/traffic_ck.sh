R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-12}   # pattern length: traffic_ck.sh [m]
O=$R/gpurun_out/r06_o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in ck=0 ck=1 none; do
  export SMH_WM_TUNE=$t
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$t -- python3 $R/tools/wmbench.py $M 100000 4096 256 > $O/wmbench_$t.log 2>&1
  tail -2 $O/wmbench_$t.log
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for t in ("ck=0","ck=1","none"):
    for f in glob.glob("gpurun_out/r06_o/fetch_%s/**/*counter_collection.csv" % t, recursive=True):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "wm_gram_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE":
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            big=[x for x in v if x>1e6]
            print(t, k, len(big), "launches, FETCH_SIZE x2 = %.3f GiB" % (2*1024*sum(big)/len(big)/2**30))
PY
find $O -name "*counter_collection.csv" -delete
