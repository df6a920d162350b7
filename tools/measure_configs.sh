# Secondary measurements quoted in DESIGN.md section 5 / 7 (run on the GPU box from the repo root)
set -e
timeout -k 10 600 python tools/bench_configs.py c3 c4 as_pcd as_pbcd > gpurun_out/v6_configs.jsonl 2> gpurun_out/v6_configs.err
timeout -k 10 300 python tools/bench_skew.py > gpurun_out/v6_skew.log 2>&1
GS=64 timeout -k 10 300 python tools/prb_stamp_probe.py > gpurun_out/v6_stamps.log 2>&1
SPFM_CPU=0 timeout -k 10 300 python tools/bench_psgd.py > gpurun_out/v6_psgd.jsonl 2> gpurun_out/v6_psgd.err
cut -c1-330 gpurun_out/v6_configs.jsonl
tail -4 gpurun_out/v6_skew.log
tail -19 gpurun_out/v6_stamps.log
cut -c1-200 gpurun_out/v6_psgd.jsonl
