#!/bin/bash
# usage: tools/pmc_variants.sh "<lib:K list>"  -> per kernel: wall (us), cycles, clock, waits
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for item in $1; do L=${item%%:*}; K=${item##*:}
  rm -rf gpurun_out/pmcx
  CTCFA_LIB=$PWD/$L rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/pmcx -- python3 bench.py --steps 4 --warmup 2 --spinup-steps 0 --cpu-sample 0 --no-check --cols-per-lane $K > gpurun_out/pmcx.log 2>&1
  python3 - "$L" "$K" <<'PY'
import csv,glob,sys,collections
f=glob.glob('gpurun_out/pmcx/*/*_counter_collection.csv')
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for r in rows:
    k='fill' if 'fill_kernel' in r['Kernel_Name'] else ('bt' if 'backtrack' in r['Kernel_Name'] else None)
    if not k: continue
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[(k,r['Dispatch_Id'])]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k in ('fill','bt'):
    v={c: sum(x)/len(x) for c,x in agg[k].items()}
    ds=[d for (kk,_),d in dur.items() if kk==k]; wall=sum(ds)/len(ds)/1e3
    clk=v['GRBM_GUI_ACTIVE']/8/wall/1e3
    wc=v['SQ_WAVE_CYCLES']
    print(f"{sys.argv[1]} K={sys.argv[2]} {k}: wall {wall:.0f}us clk {clk:.2f}GHz gui_cyc/8 {v['GRBM_GUI_ACTIVE']/8/1e3:.0f}K  waves {v['SQ_WAVES']:.0f} wavecyc/wave {wc*4/v['SQ_WAVES']/1e3:.0f}K  active {v['SQ_ACTIVE_INST_ANY']/wc:.2f} wait_any {v['SQ_WAIT_ANY']/wc:.2f} wait_inst {v['SQ_WAIT_INST_ANY']/wc:.2f} valu_insts {v['SQ_INSTS_VALU']/1e6:.0f}M")
PY
done
