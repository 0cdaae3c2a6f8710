cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq_mfcc; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/p1 -- python3 $R/tools/bench_chains.py --only mfcc --iters 2 > $O/p1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/p2 -- python3 $R/tools/bench_chains.py --only mfcc --iters 2 > $O/p2.log 2>&1
python3 - $O <<'PY'
import csv,glob,sys,collections
O=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mfcc" in r["Kernel_Name"] and r["Grid_Size"]==str(32768*64):
            agg[r["Kernel_Name"][:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()): print("   %-24s %14.0f  (n=%d)"%(c,sum(vals)/len(vals),len(vals)))
PY
