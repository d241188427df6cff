#!/bin/bash
# Development: one build of kernels_panel_x.hip per owner-step segment (-DLSX_PX_SEG=k), linked into
# linalg_solver_amd/liblsx_seg<k>.so; tools/seg_panel.py then runs each on the GPU box.
set -e
cd "$(dirname "$0")/../linalg_solver_amd/csrc"
OBJS=$(ls _obj/*.o | grep -v kernels_panel_x.o)
for k in 0 1 2 3 4 5 6 7; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DLSX_PX_SEG=$k -c kernels_panel_x.hip -o /tmp/px_seg$k.o &
done
wait
for k in 0 1 2 3 4 5 6 7; do
  /opt/rocm/lib/llvm/bin/clang++ -shared -fPIC -o ../liblsx_seg$k.so $OBJS /tmp/px_seg$k.o
done
