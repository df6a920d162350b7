# Round-3 profile collection for the configs given (default 2 3 4), then profiles/r03_traffic.json
# from the PMC summaries with the build tag of the library that produced them.
#   bash tools/profile_all_r03.sh [configs...]
set -e
R=$GRAFT_REPO_ROOT
cd $R
for c in ${@:-2 3 4}; do
  bash tools/profile_r03.sh $c r03_c$c
  cp gpurun_out/r03_c${c}_pmc_fetch_summary.txt gpurun_out/r03_c${c}_pmc_write_summary.txt profiles/
done
python tools/make_traffic_json.py > /dev/null
cp profiles/r03_traffic.json gpurun_out/
cat profiles/r03_traffic.json
