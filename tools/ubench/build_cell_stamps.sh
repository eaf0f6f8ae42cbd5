#!/bin/bash
# Diagnostic twin of libnbasr_hip.so with the phase stamps of the fused cell compiled in (tools/gpu/cell_stamps.py --lib ...).
set -e
cd "$(dirname "$0")/../.."
python -m nb_asr_amd.build > /dev/null
B=nb_asr_amd/csrc/build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DNBASR_CELL_STAMPS=1 -Iinclude -Inb_asr_amd/csrc -x hip -c nb_asr_amd/csrc/grouped_cell.hip -o $B/grouped_cell_stamps.o
OBJS=$(python -c "from nb_asr_amd import build; print(' '.join('$B/' + s.rsplit('.', 1)[0] + '.o' for s in build.SOURCES if s != 'grouped_cell.hip'))")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o nb_asr_amd/lib/libnbasr_hip_cstamps.so $OBJS $B/grouped_cell_stamps.o
echo built nb_asr_amd/lib/libnbasr_hip_cstamps.so
