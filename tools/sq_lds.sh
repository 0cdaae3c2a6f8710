#!/bin/bash
# One counter pass over the chain benchmarks aimed at the LDS pipe: how much of each kernel's wave time is
# spent unable to issue an LDS instruction, and how busy the LDS is (GPU box; counters only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq_lds; rm -rf $O; mkdir -p $O
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU \
    --output-format csv -d $O/p -- python3 $R/tools/bench_chains.py --iters 2 > $O/p.log 2>&1
python3 - $O <<'PY'
import csv,glob,sys,collections
O=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob(O+"/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "jdsp::" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Counter_Name"]=="SQ_WAVE_CYCLES":
            n[k]+=1; agg[k]["dur"]+=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
print("%-36s %8s %7s %7s %7s %7s %9s"%("kernel","us","wait%","stall%","ldsstl%","valu%","LDSbusy%"))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1]["dur"]):
    wc=v["SQ_WAVE_CYCLES"]
    if wc<1e6: continue
    dur=v["dur"]/n[k]
    # LDS busy: IDX_ACTIVE cycles per CU (256 CUs) over the kernel duration at 2.4 GHz
    busy=v["SQ_LDS_IDX_ACTIVE"]/n[k]/256/2.4e3/(dur/1e3)*100
    print("%-36s %8.1f %7.1f %7.1f %7.1f %7.1f %9.1f"%(k[6:42],dur/1e3,100*v["SQ_WAIT_ANY"]/wc,100*v["SQ_WAIT_INST_ANY"]/wc,100*v["SQ_WAIT_INST_LDS"]/wc,100*v["SQ_ACTIVE_INST_VALU"]/wc,busy))
PY
