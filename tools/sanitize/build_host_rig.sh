#!/bin/bash
# The library's host code compiled host-only under ASan + UBSan against tools/sanitize/hip_mock.cpp, and the rig that drives it
# on the CPU (no GPU): tools/sanitize/build_host_rig.sh <out dir>          (SAN=thread: the same under ThreadSanitizer)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${1:-$R/build/host_rig}
mkdir -p $OUT
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
if [ "${SAN:-address}" = thread ]; then SANF="-fsanitize=thread"; else SANF="-fsanitize=address,undefined -fno-sanitize-recover=undefined"; fi
FLAGS="--offload-arch=gfx950 --offload-host-only -O1 -g -std=c++17 -fPIC $SANF -fno-omit-frame-pointer -Wno-unused-value -I$R/include -I$R/cuclark_amd/csrc"
pids=()
for f in mic_engine mic_kernels mic_build mic_synth mic_dbbuild mic_ingest mic_gz; do
  $HIPCC -x hip $FLAGS -c $R/cuclark_amd/csrc/$f.hip -o $OUT/$f.o & pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
$HIPCC -x hip $FLAGS -c $R/cuclark_amd/csrc/mic_host.cpp -o $OUT/mic_host.o
$HIPCC -x hip $FLAGS -c $R/tools/sanitize/hip_mock.cpp -o $OUT/hip_mock.o
# the fat binaries the host-only objects refer to (there is no device code in this build)
nm --undefined-only $OUT/mic_*.o | awk '/__hip_fatbin_/ {print $2}' | sort -u | awk '{print "char " $1 "[8];"}' > $OUT/fatbins.c
gcc -c $OUT/fatbins.c -o $OUT/fatbins.o
$HIPCC -x hip $FLAGS -c $R/tools/sanitize/host_rig.cpp -o $OUT/host_rig.o
$HIPCC $SANF -o $OUT/host_rig $OUT/host_rig.o $OUT/mic_*.o $OUT/hip_mock.o $OUT/fatbins.o -lpthread -lz 2>&1 | grep -v "^$" || true
# the command line itself (classifier*.cpp, cli_main.cpp) on the same objects: exe/cuCLARK's multi-device paths on MOCK_HIP_DEVICES devices
for f in classifier classifier_stream classifier_batch cli_main; do
  $HIPCC -x c++ -O1 -g -std=c++17 -fopenmp $SANF -fno-omit-frame-pointer -I$R/include -I$R/cuclark_amd/csrc -c $R/cuclark_amd/csrc/$f.cpp -o $OUT/cli_$f.o
done
$HIPCC $SANF -fopenmp -o $OUT/cuCLARK_mock $OUT/cli_classifier.o $OUT/cli_classifier_stream.o $OUT/cli_classifier_batch.o $OUT/cli_cli_main.o $OUT/mic_*.o $OUT/hip_mock.o $OUT/fatbins.o -lpthread -lz
ln -sf cuCLARK_mock $OUT/cuCLARK_mock-l
# the same objects as a shared library for tools/sanitize/py_rig.py (references bound inside: torch brings the real runtime along)
if [ "${SAN:-address}" != thread ]; then $HIPCC -shared -fsanitize=address -shared-libasan -Wl,-Bsymbolic -o $OUT/libmi_clark_mock.so $OUT/mic_*.o $OUT/hip_mock.o $OUT/fatbins.o -lpthread; fi
ls -la $OUT/host_rig $OUT/cuCLARK_mock
