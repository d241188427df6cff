set -e
# Round-3 evidence: bench lines, kernel-level micro-benchmarks, rocprofv3 kernel statistics of the bench command,
# PMC passes (separate runs per counter, as MI355X_MICROARCH.md prescribes) for the update and the panel kernels,
# the device-stamp timeline of the look-ahead LU and the kernel trace / owner stamps of the single-RHS solve.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev3
mkdir -p $O
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1
tail -1 $O/bench.log > $O/r03_bench.json
timeout -k 10 300 python bench.py --dtype f32 --steps 10 --warmup 3 --no-cpu --no-extras > $O/bench32.log 2>&1
tail -1 $O/bench32.log > $O/r03_bench_f32.json
timeout -k 10 300 python tools/kbench.py mfma gemm gemmq xchg panelx lu4 hyb > $O/r03_kbench.log 2>&1
timeout -k 10 120 python tools/lu_tall.py > $O/r03_lu_tall.log 2>&1
timeout -k 10 120 python tools/lu_ab.py > $O/r03_lu_chain_fused.log 2>&1
LSX_LIB_OVERRIDE=$R/linalg_solver_amd/liblsx_ts.so timeout -k 5 120 python tools/ts_lu.py 8192 30 2 > $O/r03_ts_lu_8192.log 2>&1
LSX_LIB_OVERRIDE=$R/linalg_solver_amd/liblsx_ts.so timeout -k 5 120 python tools/ts_lu.py 8192 3 2 > $O/r03_ts_lu_8192_early.log 2>&1
LSX_LIB_OVERRIDE=$R/linalg_solver_amd/liblsx_ts.so timeout -k 5 120 python tools/ts_lu.py 4096 10 2 > $O/r03_ts_lu_4096.log 2>&1
timeout -k 5 120 python tools/solve_stamps.py > $O/r03_solve_stamps.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extras > $O/prof.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_solve -- python3 $R/tools/solve_trace.py > $O/prof_solve.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 $R/tools/kbench.py pmc > $O/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 $R/tools/kbench.py pmc > $O/pmc_w.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_h -- python3 $R/tools/kbench.py pmc > $O/pmc_h.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace --output-format csv -d $O/pmc_d -- python3 $R/tools/kbench.py pmc > $O/pmc_d.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_m1 -- python3 $R/tools/kbench.py pmc > $O/pmc_m1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $O/pmc_m2 -- python3 $R/tools/kbench.py pmc > $O/pmc_m2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcp_f -- python3 $R/tools/kbench.py pmcpanel > $O/pmcp_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcp_w -- python3 $R/tools/kbench.py pmcpanel > $O/pmcp_w.log 2>&1
cd $R
python tools/pmc_mfma_summary.py $O/pmc_m1 $O/pmc_m2 $O/r03_pmc_gemm_mfma.json || true
find gpurun_out/ev3 -name "*.csv" | head -40
