#!/bin/bash
# Sweep of the CLI's streaming parameters on an existing /tmp/e2e (tools/e2e_cli_timing.sh leaves it)
# usage: CFGS="n:slots:nd:nw:mb ..." tools/e2e_sweep.sh
D=/tmp/e2e
for cfg in ${CFGS:-16:24:4:2:32}; do IFS=: read n sl nd nw mb <<< "$cfg"; for rep in 1 2 3; do
  rm -f $D/sw.csv
  r=$(MIC_INGEST_SLOTS=$sl MIC_INGEST_ND=$nd MIC_INGEST_NW=$nw MIC_INGEST_MB=$mb MIC_CLI_TIMING=1 ./exe/cuCLARK-l -T $D/targets.txt -D $D/DB/ -O $D/reads.fq -R $D/sw -n $n 2>&1 | grep -E "threads:" | sed 's/.*threads: //; s/thread-seconds: //; s/input.*ms since start: //')
  echo "-n $n slots $sl nd $nd nw $nw ${mb}MB: $r"
done; done
