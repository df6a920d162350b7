# Round-4 profile collection for the configs given (default 2 3 4), then profiles/r04_traffic.json
# from the PMC summaries with the build tag of the library that produced them, and the WRITE_SIZE
# calibration on the scatter pattern of the passes with rows in global memory.
#   bash tools/profile_all_r04.sh [configs...]
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for c in ${@:-2 3 4}; do
  bash tools/profile_r04.sh $c r04_c$c
  cp gpurun_out/r04_c${c}_pmc_fetch_summary.txt gpurun_out/r04_c${c}_pmc_write_summary.txt gpurun_out/r04_c${c}_kernel_stats.csv gpurun_out/r04_c${c}_bench.json gpurun_out/r04_c${c}_bench_under_rocprof.json profiles/
done
python tools/make_traffic_json.py "" r04 > /dev/null
cp profiles/r04_traffic.json gpurun_out/
cat profiles/r04_traffic.json
cd /tmp
rm -rf /tmp/pmc_wcal
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_wcal -o run -- python3 $R/tools/write_calibration.py > $R/gpurun_out/r04_write_calibration.json 2> $R/gpurun_out/r04_write_calibration.err
python3 $R/tools/pmc_summary.py /tmp/pmc_wcal > $R/gpurun_out/r04_write_calibration_pmc.txt
cat $R/gpurun_out/r04_write_calibration.json; cat $R/gpurun_out/r04_write_calibration_pmc.txt | cut -c1-200
