R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tsdf_pmc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for c in "SQ_WAVES SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
 n=$(echo $c | tr ' ' '_')
 timeout -s KILL 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$n -o p -- python $R/scripts/tsdf_integrate_probe.py 16 > $O/$n.log 2>&1
 echo "$c rc=$?"
 f=$(find $O/p_$n -name "*counter_collection.csv" | head -1)
 [ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "tsdf_integrate" in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in d.items(): print("   ",k, "per launch", sum(v)/len(v), "n",len(v))
PY
 rm -rf $O/p_$n
done
