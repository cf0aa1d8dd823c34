#!/bin/bash
# rocprofv3 passes for the round's profile evidence (run on the GPU box via gpurun).
#   tools/profile.sh <tag> [bench args]
# Writes small CSV summaries under gpurun_out/prof_<tag>/ (copy to profiles/ to commit).
set -u
TAG=${1:-r01}; shift || true
ARGS="${@:---steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-extras}"
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
W=/tmp/prof_$TAG; rm -rf $W; mkdir -p $W
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $W/stats -o stats -- python $REPO/bench.py $ARGS > $OUT/stats.log 2>&1
pmc() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $W/$name -o pmc -- python $REPO/bench.py $ARGS > $OUT/$name.log 2>&1; }
pmc pmc_sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES
pmc pmc_sq2 SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SMEM GRBM_GUI_ACTIVE
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE
find $W -name "*.csv" | while read f; do sz=$(stat -c %s "$f"); echo "$sz $f"; done
for f in $(find $W -name "*kernel_stats.csv" -o -name "*counter_collection.csv"); do
  d=$(echo $f | sed -e "s#^$W/##" -e "s#/.*##")
  # keep the header and the rows of our kernels only (torch's input-generation kernels are noise)
  (head -1 $f; grep -E "qs_[a-z_]+kernel" $f) > $OUT/${d}_$(basename $f); done
python $REPO/tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
