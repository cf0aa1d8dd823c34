# rocprofv3 --pmc pass over tools/bench_sizes.py (one block per lane forced) at 64 / 128 / 1024 block rows: per-wave SQ wait counters
# (run on the GPU box from the repo root: bash tools/pmc_sizes.sh -> gpurun_out/r04y/pmc_sizes.txt)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04y; mkdir -p $O
cd /tmp
for rows in 64 128 1024; do
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM --output-format csv -d /tmp/pmc_$rows -o pmc -- env QS_HIP_DP=0 python $R/tools/bench_sizes.py --rows $rows > /dev/null 2>&1
  f=$(find /tmp/pmc_$rows -name '*counter_collection.csv' | head -1)
  python - "$f" $rows <<'PY' >> $O/pmc_sizes.txt
import csv,sys
from collections import defaultdict
acc=defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if 'qs_smooth_plane_kernel' in row['Kernel_Name']:
        acc[row['Counter_Name']].append(float(row['Counter_Value']))
m={k:sum(v)/len(v) for k,v in acc.items()}
w=m.get('SQ_WAVES',1)
print(f"rows {sys.argv[2]}: waves {w:.0f}  per wave: cycles {m['SQ_WAVE_CYCLES']/w:.0f}  wait_any {m['SQ_WAIT_ANY']/w:.0f} ({100*m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.1f} %)  wait_inst_any {m['SQ_WAIT_INST_ANY']/w:.0f} ({100*m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES']:.1f} %)  valu {m['SQ_INSTS_VALU']/w:.0f}  active_valu {m['SQ_ACTIVE_INST_VALU']/w:.0f}  smem {m['SQ_INSTS_SMEM']/w:.0f}  smem_cycles {m.get('SQ_INST_CYCLES_SMEM',0)/w:.0f}")
PY
done
cat $O/pmc_sizes.txt
