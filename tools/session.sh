#!/bin/bash
# One parametrised GPU session (run through gpurun from the repo root):
#   tools/session.sh <tag> <step> [<step> ...]
# Every step writes under gpurun_out/<tag>/ (copy what should be judged into profiles/).
# Steps:
#   host            CPU model, cgroup quota, affinity of the GPU box
#   pytest[:expr]   the GPU suite (-m gpu), optionally restricted with -k expr
#   bench:<name>:<args...>   python bench.py <args> -> bench_<name>.json (args separated by ',')
#   benchlib:<name>:<variant>:<args...>  the same on build/variants/libjpegqs_hip_<variant>.so
#   prof:<name>:<args...>    tools/profile.sh (kernel-trace + PMC passes) of bench.py <args>
#   variants:<size> tools/bench_variants.py over build/variants/*.so
#   cold            tools/cold_phases (fresh-process phase times) for 1080p and 8192^2, three runs each
#   cli[:size]      tools/bench_cli.py (drop-in CLI against the reference CLIs on libjpeg-encoded files)
#   sizes:<args>    tools/bench_sizes.py <args>
#   py:<name>:<script>:<args...>  any tools/*.py script;  pyenv:<name>:<VAR=value>:<script>:<args...> with one environment variable
#   exe:<name>:<path>       a prebuilt measurement binary
set -u
export TMPDIR=/tmp
R=$PWD
TAG=${1:?tag}; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" = "$step" ] && rest=""
  echo "== $step ($(date +%T))"
  case $kind in
    host)
      { nproc; lscpu | head -25; cat /sys/fs/cgroup/cpu.max; python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; } > $O/host.txt 2>&1 ;;
    pytest)
      if [ -n "$rest" ]; then ( time timeout 1700 python -m pytest tests -x -q -m gpu -s -k "$rest" ) > $O/pytest_${rest// /_}.log 2>&1; tail -4 $O/pytest_${rest// /_}.log
      else ( time timeout 1700 python -m pytest tests -x -q -m gpu -s --durations=15 ) > $O/pytest.log 2>&1; tail -25 $O/pytest.log; fi ;;
    bench)
      name=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""
      timeout 900 python bench.py ${args//,/ } > $O/bench_$name.json 2> $O/bench_$name.err; tail -c 1500 $O/bench_$name.json ;;
    benchlib)   # benchlib:<name>:<variant>:<args>  -- bench.py on build/variants/libjpegqs_hip_<variant>.so
      name=${rest%%:*}; r2=${rest#*:}; var=${r2%%:*}; args=${r2#*:}; [ "$args" = "$r2" ] && args=""
      QS_HIP_LIB=$R/build/variants/libjpegqs_hip_$var.so timeout 900 python bench.py ${args//,/ } > $O/bench_$name.json 2> $O/bench_$name.err; tail -c 700 $O/bench_$name.json ;;
    prof)
      name=${rest%%:*}; args=${rest#*:}; [ "$args" = "$rest" ] && args=""
      timeout 1200 bash tools/profile.sh $name ${args//,/ } > $O/prof_$name.log 2>&1; tail -12 $O/prof_$name.log ;;
    variants)
      timeout 900 python tools/bench_variants.py ${rest:-8192} > $O/variants_${rest:-8192}.txt 2>&1; cat $O/variants_${rest:-8192}.txt ;;
    cold)
      for i in 1 2 3; do ./tools/cold_phases 1920 1080 3; done > $O/cold_1080p.txt 2>&1
      for i in 1 2 3; do COLD_COPY_PROBE=1 QS_HIP_TRACE=1 ./tools/cold_phases 8192 8192 1; done > $O/cold_8192.txt 2>&1
      for i in 1 2; do COLD_SKIP_RUNTIME=1 QS_HIP_TRACE=1 ./tools/cold_phases 1920 1080 3; done > $O/cold_1080p_libfirst.txt 2>&1
      for i in 1 2; do COLD_SKIP_RUNTIME=1 QS_HIP_TRACE=1 ./tools/cold_phases 8192 8192 1; done > $O/cold_8192_libfirst.txt 2>&1
      for ms in 0 20 150; do COLD_SKIP_RUNTIME=1 COLD_PREWARM=$ms QS_HIP_TRACE=1 ./tools/cold_phases 1920 1080 3; done > $O/cold_1080p_prewarm.txt 2>&1
      for ms in 0 150 400; do COLD_SKIP_RUNTIME=1 COLD_PREWARM=$ms QS_HIP_TRACE=1 ./tools/cold_phases 8192 8192 1; done > $O/cold_8192_prewarm.txt 2>&1
      head -45 $O/cold_1080p.txt; head -12 $O/cold_8192_libfirst.txt; cat $O/cold_1080p_prewarm.txt $O/cold_8192_prewarm.txt ;;
    cli)
      timeout 1200 python tools/bench_cli.py ${rest:-8192} > $O/bench_cli.txt 2>&1; cat $O/bench_cli.txt ;;
    sizes)
      timeout 900 python tools/bench_sizes.py ${rest//,/ } > $O/sizes.txt 2>&1; tail -30 $O/sizes.txt ;;
    py)
      name=${rest%%:*}; r2=${rest#*:}; script=${r2%%:*}; args=${r2#*:}; [ "$args" = "$r2" ] && args=""
      timeout 1200 python tools/$script ${args//,/ } > $O/$name.txt 2>&1; tail -30 $O/$name.txt ;;
    pyenv)    # pyenv:<name>:<VAR=value>:<script>:<args...>
      name=${rest%%:*}; r2=${rest#*:}; var=${r2%%:*}; r3=${r2#*:}; script=${r3%%:*}; args=${r3#*:}; [ "$args" = "$r3" ] && args=""
      env "$var" timeout 1200 python tools/$script ${args//,/ } > $O/$name.txt 2>&1; tail -30 $O/$name.txt ;;
    exe)      # exe:<name>:<path>  -- a prebuilt measurement binary (build/ubench_clock ...)
      name=${rest%%:*}; path=${rest#*:}
      timeout 600 $path > $O/$name.txt 2>&1; tail -40 $O/$name.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
ls -la $O
