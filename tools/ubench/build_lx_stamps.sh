#!/bin/bash
# Diagnostic twin of libnbasr_hip.so with the stamped lstm_xcd_kernel (run in the build container; the .so travels with gpurun).
set -e
cd "$(dirname "$0")/../.."
python -m nb_asr_amd.build > /dev/null
B=nb_asr_amd/csrc/build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DNBASR_LX_STAMPS=1 -Iinclude -Inb_asr_amd/csrc -x hip -c nb_asr_amd/csrc/lstm_xcd.hip -o $B/lstm_xcd_stamps.o
OBJS=$(python -c "from nb_asr_amd import build; print(' '.join('$B/' + s.rsplit('.', 1)[0] + '.o' for s in build.SOURCES if s != 'lstm_xcd.hip'))")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o nb_asr_amd/lib/libnbasr_hip_stamps.so $OBJS $B/lstm_xcd_stamps.o
echo built nb_asr_amd/lib/libnbasr_hip_stamps.so
