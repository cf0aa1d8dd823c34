#!/bin/bash
# ONE command for the first hour on a multi-GPU node (Next 3).  Nothing in this repository has run on
# more than one GPU; this script is the order in which to find out what works, cheapest and most basic first:
#   (a) tools/first_contact_p2p      peer-access matrix, hipMemcpyPeerAsync ring with the cross-device event protocol of
#                                    csrc/qs_shard.cpp, every byte checked; halo latency per round, xGMI GB/s per hop
#   (b) tools/first_contact_shard.py qs_hip_do_quantsmooth_sharded over devices 0..N-1: 8192^2 q3, 16384^2 q3,
#                                    8192^2 4:2:0 q6 n5 -- EVERY block against the compiled reference (libqsref_none.so);
#                                    then again with QS_HIP_SHARD_SCHEDULE=deep (no halo exchange): the schedules' A/B
#   (c) bench.py --gpus 2/4/8        RCCL band driver: q3 with 12 planes per step (`value`) and with one (`--batch 1`:
#                                    single-image strong scaling), q6; `--backend nccl` (the default) ENDS the run if RCCL
#                                    does not come up -- a host-staged number never looks like a result
#   (d) pytest -m gpu -k multigpu    the same three steps as tests, plus qs_hip_do_quantsmooth_band (one thread per device,
#                                    halo rows through RCCL behind the C ABI) -- tests/test_multigpu.py; they skip below 2 devices
# Everything lands in ONE folder, gpurun_out/first_contact/ (or $1), with a PASS / FAIL line per step in SUMMARY.txt.
#   bash tools/first_contact.sh [outdir] [N ...]         N defaults to every power of two up to the visible device count
# On a one-GPU box every step runs in its degenerate form (N = 1).
set -u
export TMPDIR=${TMPDIR:-/tmp} HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
O=${1:-$R/gpurun_out/first_contact}; shift || true
mkdir -p "$O"; : > "$O/SUMMARY.txt"
NDEV=$(python -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 0)
[ "$NDEV" -ge 1 ] || { echo "FAIL: no HIP device visible" | tee -a "$O/SUMMARY.txt"; exit 2; }
NS="$*"
if [ -z "$NS" ]; then n=2; NS=""; while [ $n -le "$NDEV" ]; do NS="$NS $n"; n=$((n * 2)); done; [ -z "$NS" ] && NS=1; fi
say() { echo "$1" | tee -a "$O/SUMMARY.txt"; }
step() {   # step <name> <timeout s> <command...>: run, log, one summary line
  local name=$1 tmo=$2; shift 2
  local t0=$(date +%s)
  timeout "$tmo" "$@" > "$O/$name.log" 2>&1; local rc=$?
  say "$( [ $rc -eq 0 ] && echo PASS || echo "FAIL(rc=$rc)" )  $name  ($(( $(date +%s) - t0 )) s)  -> $O/$name.log"
  return $rc
}
say "first_contact: $NDEV visible device(s); N = $NS; $(date -u +%FT%TZ)"
rocm-smi --showtopo > "$O/topology.txt" 2>&1 || true
FAILS=0

# (a) the transport alone
[ -x tools/first_contact_p2p ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/first_contact_p2p.hip -o tools/first_contact_p2p
for n in $NS; do step "a_p2p_n$n" 120 tools/first_contact_p2p "$n" 64 || FAILS=$((FAILS + 1)); done

# (b) the product's route, every block against the reference
for n in $NS; do
  devs=$(seq -s, 0 $((n - 1)))
  step "b_shard_n$n" 1500 python tools/first_contact_shard.py --devices "$devs" || FAILS=$((FAILS + 1))
  # the same three configurations on the COMMUNICATION-AVOIDING schedule (niter extra block rows per cut side, no halo
  # exchange; qs_hip_set_shard_schedule(1)): the A/B of the two schedules on real links is the `ms` column of the two logs
  [ "$n" -ge 2 ] && { step "b_shard_deep_n$n" 1500 env QS_HIP_SHARD_SCHEDULE=deep python tools/first_contact_shard.py --devices "$devs" || FAILS=$((FAILS + 1)); }
done

# (c) the RCCL band driver: q3 (12 planes per step, then one), q6
for n in $NS; do
  [ "$n" -ge 2 ] || continue
  step "c_bench_q3_n$n" 900 python bench.py --gpus "$n" --steps 10 --warmup 3 --no-cpu-baseline --edge-first || FAILS=$((FAILS + 1))
  step "c_bench_q3_batch1_n$n" 900 python bench.py --gpus "$n" --steps 20 --warmup 3 --batch 1 --no-cpu-baseline --no-extras || FAILS=$((FAILS + 1))
  step "c_bench_q6_n$n" 900 python bench.py --gpus "$n" --quality 6 --steps 5 --warmup 2 --no-cpu-baseline || FAILS=$((FAILS + 1))
  for f in c_bench_q3_n$n c_bench_q3_batch1_n$n c_bench_q6_n$n; do
    python - "$O/$f.log" <<'PY' | tee -a "$O/SUMMARY.txt"
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
if not lines:
    print("      (no JSON line)")
else:
    d = json.loads(lines[-1])
    print(f"      value {d['value'] / 1e6:.1f} M blocks/s, ms/step {d['ms_per_step']:.2f}, n_gpus {d['n_gpus']}, rccl_ranks {d['config'].get('rccl_ranks')}, "
          f"verify_ok {d.get('verify_ok')}, band edges {d.get('verify_band_edges_ok')}, value_batch1 {d.get('value_batch1')}, "
          f"product_route {(d.get('product_route') or {}).get('ms_per_image')}, "
          f"deep-halo schedule batch1 {(d.get('deep_halo_schedule') or {}).get('value_batch1')} (equal: {(d.get('deep_halo_schedule') or {}).get('equals_exchange_schedule')}), "
          f"edge-first schedule batch1 {(d.get('edge_first_schedule') or {}).get('value_batch1')} (equal: {(d.get('edge_first_schedule') or {}).get('equals_exchange_schedule')})")
PY
  done
done
[ "$NDEV" -eq 1 ] && step "c_bench_q3_n1" 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline

# (d) the same as tests
step "d_pytest_multigpu" 1700 python -m pytest tests/test_multigpu.py -x -q -m gpu -rs || FAILS=$((FAILS + 1))

say "first_contact: $( [ $FAILS -eq 0 ] && echo "PASS (every step)" || echo "FAIL ($FAILS step(s))" )"
exit $FAILS
