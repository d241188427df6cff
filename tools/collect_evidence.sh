set -e
# Round-2 evidence: bench lines, kernel-level micro-benchmarks, rocprofv3 kernel statistics of the bench command,
# PMC passes (separate runs per counter, as MI355X_MICROARCH.md prescribes) for the update and the panel kernels.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ev2
mkdir -p $O
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1
tail -1 $O/bench.log > $O/r02_bench.json
timeout -k 10 300 python bench.py --dtype f32 --steps 10 --warmup 3 --no-cpu --no-extras > $O/bench32.log 2>&1
tail -1 $O/bench32.log > $O/r02_bench_f32.json
timeout -k 10 300 python tools/kbench.py mfma gemm gemmq xchg panelx lu4 hyb > $O/r02_kbench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-extras > $O/prof.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 $R/tools/kbench.py pmc > $O/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 $R/tools/kbench.py pmc > $O/pmc_w.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcp_f -- python3 $R/tools/kbench.py pmcpanel > $O/pmcp_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcp_w -- python3 $R/tools/kbench.py pmcpanel > $O/pmcp_w.log 2>&1
cd $R
find gpurun_out/ev2 -name "*.csv" | head -40
