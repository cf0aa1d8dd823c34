#!/bin/bash
# rocprofv3 --pmc passes over tools/bench_sizes.py (one block per lane forced): per-wave SQ / scalar-cache counters of the
# recovery kernel at a few plane heights.  Run on the GPU box from the repo root:
#   [TAG=r04y] [PASS_LIST="0 1 2"] [ROWS="64 128 1024"] [LIBS="build/variants/libjpegqs_hip_x.so ..."] bash tools/pmc_sizes.sh
# -> gpurun_out/$TAG/pmc_sizes.txt   (counters only, no trace domains: see the gpurun rule on --pmc)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${TAG:-r04y}; mkdir -p $O
ROWS=${ROWS:-"64 128 1024"}; LIBS=${LIBS:-default}
PASSES=("SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM"
        "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
        "SQ_WAVES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ SQ_INSTS_SALU SQ_ACTIVE_INST_SCA")
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_INST_LEVEL_[A-Z_]+|SQC_DCACHE_[A-Z_]+|SQ_WAIT_[A-Z_]+|SQC_TC_[A-Z_]+)\b" | sort -u | tr '\n' ' ' > $O/pmc_available.txt
for lib in $LIBS; do for rows in $ROWS; do
  [ $lib = default ] && libarg="" || libarg=$R/$lib
  echo "== $lib rows $rows" >> $O/pmc_sizes.txt
  for p in "${!PASSES[@]}"; do
    case " ${PASS_LIST:-0 1 2} " in *" $p "*) ;; *) continue ;; esac
    d=/tmp/pmc_${rows}_${p}_$RANDOM
    rocprofv3 --pmc ${PASSES[$p]} --output-format csv -d $d -o pmc -- env QS_HIP_DP=0 python $R/tools/bench_sizes.py --rows $rows $libarg > $d.log 2>&1
    f=$(find $d -name '*counter_collection.csv' 2>/dev/null | head -1)
    [ -z "$f" ] && { echo "  pass $p: no counters (see log)"; tail -3 $d.log; } >> $O/pmc_sizes.txt && continue
    python - "$f" <<'PY' >> $O/pmc_sizes.txt
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if 'qs_smooth_plane_kernel' in row['Kernel_Name']:
        acc[row['Counter_Name']].append(float(row['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m.get('SQ_WAVES', 1) or 1
print("  per wave (waves %d): " % w + "  ".join(f"{k[3:] if k.startswith('SQ_') else k} {v / w:.0f}" for k, v in sorted(m.items()) if k != 'SQ_WAVES'))
PY
  done
done; done
cat $O/pmc_available.txt; echo; cat $O/pmc_sizes.txt
