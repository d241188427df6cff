#!/bin/bash
# Development: variants of kernels_panel_y.hip (switches PY_*), one library each: linalg_solver_amd/liblsx_v<name>.so;
# tools/var_panel.py times them on the GPU box.   usage: tools/var_panel.sh name "flags" [name "flags" ...]
set -e
cd "$(dirname "$0")/../linalg_solver_amd/csrc"
OBJS=$(ls _obj/*.o | grep -v kernels_panel_y.o)
names=()
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2; names+=($n)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast $f -c kernels_panel_y.hip -o /tmp/py_v$n.o 2>/tmp/var_err_$n.log &
done
wait
for n in "${names[@]}"; do
  /opt/rocm/lib/llvm/bin/clang++ -shared -fPIC -o ../liblsx_v$n.so $OBJS /tmp/py_v$n.o
done
