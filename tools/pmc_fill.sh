#!/bin/bash
# usage: tools/pmc_fill.sh "<K list>" [bench flags]  -> SQ activity counters of the fill kernel per K (serial schedule)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for K in $1; do
 for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
            "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU" \
            "SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS"; do
  rm -rf gpurun_out/pmcx
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmcx -- python3 bench.py --steps 4 --warmup 2 --spinup-steps 0 --cpu-sample 0 --no-check --serial --input-sets 2 --cols-per-lane $K $2 > gpurun_out/pmcx.log 2>&1 || { tail -5 gpurun_out/pmcx.log; continue; }
  python3 - "$K" <<'PY'
import csv,glob,sys,collections
f=glob.glob('gpurun_out/pmcx/*/*_counter_collection.csv')
agg=collections.defaultdict(list); dur={}
for r in csv.DictReader(open(f[0])):
    if 'fill_kernel' not in r['Kernel_Name']: continue
    agg[r['Counter_Name']].append(float(r['Counter_Value']))
    dur[r['Dispatch_Id']]=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
wall=sum(dur.values())/len(dur)/1e3
print(f"K={sys.argv[1]} fill wall {wall:.0f}us " + " ".join(f"{c}={sum(v)/len(v):.4g}" for c,v in sorted(agg.items())))
PY
 done
done
