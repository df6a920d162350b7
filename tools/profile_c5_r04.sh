# Round-4 profile of BASELINE configs[4] (10M x 1M) on one GPU: rocprofv3 kernel stats and the two
# PMC passes (separate runs) of a whole iteration of the wide pass.
#   bash tools/profile_c5_r04.sh
set -e
export TMPDIR=/tmp
export SPFM_BENCH_N=10000000 SPFM_BENCH_D=1000000
R=$GRAFT_REPO_ROOT
TAG=r04_c5
cd /tmp
rm -rf /tmp/prof_$TAG /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o run -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.err
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
head -5 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f_$TAG -o run -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/${TAG}_pmc_f.err
python $R/tools/pmc_summary.py /tmp/pmc_f_$TAG > $R/gpurun_out/${TAG}_pmc_fetch_summary.txt
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w_$TAG -o run -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/${TAG}_pmc_w.err
python $R/tools/pmc_summary.py /tmp/pmc_w_$TAG > $R/gpurun_out/${TAG}_pmc_write_summary.txt
head -4 $R/gpurun_out/${TAG}_pmc_fetch_summary.txt $R/gpurun_out/${TAG}_pmc_write_summary.txt | cut -c1-200
