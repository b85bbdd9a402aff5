#!/bin/bash
# Rebuild the whole library on the GPU box with each given option set (options that change the table as well as the
# kernels) and run the headline bench: tools/define_sweep_all.sh "-DA=1" "-DA=0" ...
# These builds replace the product objects while the sweep runs; the default build is restored on exit (also on ^C).
cd ${GRAFT_REPO_ROOT:-$(pwd)}/cuclark_amd/csrc
GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(cd ../.. && pwd)}
trap 'rm -f obj/mic_kernels.o obj/mic_build.o obj/mic_engine.o obj/mic_synth.o obj/mic_dbbuild.o; make -j8 all > /dev/null 2>&1' EXIT
for d in "$@"; do
  rm -f obj/mic_kernels.o obj/mic_build.o obj/mic_engine.o obj/mic_synth.o obj/mic_dbbuild.o
  make -j8 all EXTRA="$d" 2>&1 | grep -E "error" -A3
  for i in 1 2; do
    python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-pipeline --no-e2e --steps 10 --warmup 2 2>/dev/null | tail -1 | TAG="$d" python -c "import os,sys,json; d=json.loads(sys.stdin.read()); t=d['config']['table']; print(os.environ['TAG'], d['value'], d['ms_per_step'], 'overflow', t['overflow_slots'], 'largest', t['largest_minimizer_bucket'], 'GB', t['hbm_GB'], 'ok', d['known_answer']['label_and_count_ok'])"
  done
done
