#!/bin/sh
# classify_metagenome.sh — launcher with the calling convention of CuCLARK's script of the same name:
# it reads the targets definition and database directory from ./.settings (written by set_targets.sh as
# "-T <targets file>" and "-D <database directory>/"), appends the user's options, and runs exe/cuCLARK, or
# exe/cuCLARK-l when --light is given.  --gzipped is accepted and needs no temporary copy: the binary inflates
# gzip input itself.
DIR=$(dirname "$0")
if [ $# -lt 1 ]; then
  echo "Usage: $0 -O <objects> | -P <mate1> <mate2>  -R <results> [-k n] [-n threads] [-b batches] [-d gpus] [--light] [--gzipped] [--extended] ..."
  exit 0
fi
if [ ! -f ./.settings ]; then
  echo "Please run set_targets.sh first: ./.settings (targets definition and database directory) is missing."
  exit 1
fi
SETTINGS=$(tr '\n' ' ' < ./.settings)
EXE="$DIR/exe/cuCLARK"
ARGS=""
for a in "$@"; do
  case "$a" in
    --light) EXE="$DIR/exe/cuCLARK-l" ;;
    --gzipped) ;;
    -T|-D) echo "The targets and the database directory are set by set_targets.sh and cannot be changed here."; exit 1 ;;
    *) ARGS="$ARGS $a" ;;
  esac
done
exec $EXE $SETTINGS $ARGS
