set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ev
timeout -k 10 400 python bench.py --steps 5 --warmup 2 > $R/gpurun_out/ev/bench.log 2>&1
tail -1 $R/gpurun_out/ev/bench.log > $R/gpurun_out/ev/r01_bench.json
timeout -k 10 300 python bench.py --dtype f32 --steps 5 --warmup 2 --no-cpu > $R/gpurun_out/ev/bench32.log 2>&1
tail -1 $R/gpurun_out/ev/bench32.log > $R/gpurun_out/ev/r01_bench_f32.json
timeout -k 10 300 python tools/kbench.py mfma gemm stamps panel3 > $R/gpurun_out/ev/kbench.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ev/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $R/gpurun_out/ev/prof.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ev/pmc_f -- python3 $R/tools/kbench.py pmc > $R/gpurun_out/ev/pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/ev/pmc_w -- python3 $R/tools/kbench.py pmc > $R/gpurun_out/ev/pmc_w.log 2>&1
cd $R
find gpurun_out/ev -name "*.csv" | head -20
