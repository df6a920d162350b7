# Round-3 extras on the GPU box: counter calibration on the persistent pass's entry stream, and
# the 2- and 4-rank weak-scaling rehearsals (ranks sharing the box's one GPU) through bench.py's
# own launcher.
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf /tmp/pmc_cal
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_cal -o run -- python $R/tools/fetch_calibration.py > $R/gpurun_out/r03_fetch_calibration.json 2> $R/gpurun_out/r03_fetch_calibration.err
python $R/tools/pmc_summary.py /tmp/pmc_cal > $R/gpurun_out/r03_fetch_calibration_pmc.txt
cat $R/gpurun_out/r03_fetch_calibration.json; head -4 $R/gpurun_out/r03_fetch_calibration_pmc.txt | cut -c1-200
cd $R
timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r03_bench_2rank_weak_onegpu.json 2> gpurun_out/r03_bench_2rank.err
tail -c 400 gpurun_out/r03_bench_2rank_weak_onegpu.json; echo
timeout -k 10 500 python bench.py --gpus 4 --steps 3 --warmup 1 > gpurun_out/r03_bench_4rank_weak_onegpu.json 2> gpurun_out/r03_bench_4rank.err
tail -c 400 gpurun_out/r03_bench_4rank_weak_onegpu.json; echo
