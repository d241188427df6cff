#!/bin/bash
# Development: build liblsx_ts.so with -DLSX_TSTAMP (device-clock stamps at kernel starts / ends, common.h) for
# tools/ts_lu.py, which prints the panel-to-panel timeline of the look-ahead LU without a profiler attached.
set -e
cd "$(dirname "$0")/../linalg_solver_amd/csrc"
OBJS=$(ls _obj/*.o | grep -v -e kernels_panel_x.o -e kernels_misc.o -e kernels_gemm.o -e kernels_panel_coop.o -e kernels_panel_blk.o)
for f in kernels_panel_x kernels_misc kernels_gemm; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DLSX_TSTAMP -c $f.hip -o /tmp/ts_$f.o &
done
wait
/opt/rocm/lib/llvm/bin/clang++ -shared -fPIC -o ../liblsx_ts.so $OBJS /tmp/ts_kernels_panel_x.o /tmp/ts_kernels_misc.o /tmp/ts_kernels_gemm.o
ls -la ../liblsx_ts.so
