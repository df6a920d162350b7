# Round-4 profile collection on the GPU box: bench line, rocprofv3 kernel stats and the two PMC
# passes (separate runs, per MI355X_MICROARCH.md "HBM") for one BASELINE config.
#   bash tools/profile_r04.sh <config 2|3|4> <tag>
set -e
export TMPDIR=/tmp
CFG=$1
TAG=$2
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python bench.py --config $CFG --steps 5 --warmup 1 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench.json; echo
cd /tmp
rm -rf /tmp/prof_$TAG /tmp/pmc_f_$TAG /tmp/pmc_w_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o run -- python $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_rocprof.err
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_kernel_stats.csv
head -8 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-220
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f_$TAG -o run -- python $R/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/${TAG}_pmc_f.err
python $R/tools/pmc_summary.py /tmp/pmc_f_$TAG > $R/gpurun_out/${TAG}_pmc_fetch_summary.txt
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w_$TAG -o run -- python $R/bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-extras > /dev/null 2> $R/gpurun_out/${TAG}_pmc_w.err
python $R/tools/pmc_summary.py /tmp/pmc_w_$TAG > $R/gpurun_out/${TAG}_pmc_write_summary.txt
head -6 $R/gpurun_out/${TAG}_pmc_fetch_summary.txt $R/gpurun_out/${TAG}_pmc_write_summary.txt | cut -c1-200
