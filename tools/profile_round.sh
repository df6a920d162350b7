set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python bench.py --steps 5 --warmup 1 > gpurun_out/v6_bench.json 2> gpurun_out/v6_bench.err
tail -c 1500 gpurun_out/v6_bench.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v6 -o run -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/v6_bench_under_rocprof.json 2> $R/gpurun_out/v6_rocprof.err
cp /tmp/prof_v6/run_kernel_stats.csv $R/gpurun_out/v6_kernel_stats.csv
head -6 $R/gpurun_out/v6_kernel_stats.csv | cut -c1-200
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o run -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/v6_pmc_f.err
python $R/tools/pmc_summary.py /tmp/pmc_f > $R/gpurun_out/v6_pmc_fetch_summary.txt
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o run -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/v6_pmc_w.err
python $R/tools/pmc_summary.py /tmp/pmc_w > $R/gpurun_out/v6_pmc_write_summary.txt
head -5 $R/gpurun_out/v6_pmc_fetch_summary.txt $R/gpurun_out/v6_pmc_write_summary.txt
