cd /tmp; export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --steps 50 --warmup 5"
O=$GRAFT_REPO_ROOT/gpurun_out/pmc1
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES --output-format csv -d $O/a -- $B > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/b -- $B > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/c -- $B > $O/c.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --output-format csv -d $O/d -- $B > $O/d.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc1/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "mppi_rollout_kernel" not in r["Kernel_Name"]: continue
        k = r["Counter_Name"]; agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    print(f.split("/")[2])
    for k, (v, n) in sorted(agg.items()):
        print("   %-34s per launch %14.0f   (n=%d)" % (k, v / n, n))
PY
tail -3 $O/*.log | grep -i -E "error|invalid|not" | head
