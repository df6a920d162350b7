# Round 3: independent fits side by side on the one GPU (tools/concurrent_fits.py), all three
# single-GPU BASELINE configurations, plus the hardware-queue diagnostics DESIGN.md 3g quotes.
#   bash tools/profile_concurrent_r03.sh   ->  gpurun_out/r03_concurrent_fits.jsonl, ..._queues.txt
set -e
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r03_concurrent_fits.jsonl
: > $O
ITERS=3 timeout -k 10 600 python tools/concurrent_fits.py 1 2 4 2>/dev/null >> $O          # with the bitwise check
NO_CHECK=1 ITERS=2 timeout -k 10 300 python tools/concurrent_fits.py 8 2>/dev/null >> $O   # 32 row blocks each: rows leave LDS
NO_CHECK=1 ITERS=2 CONFIG=3 timeout -k 10 300 python tools/concurrent_fits.py 1 4 2>/dev/null >> $O
NO_CHECK=1 ITERS=2 CONFIG=4 timeout -k 10 300 python tools/concurrent_fits.py 1 2 4 2>/dev/null >> $O
cat $O
Q=gpurun_out/r03_concurrent_fits_queues.txt
echo "# four fits after a torch kernel on the null stream; the runtime's default of 4 hardware queues, then this package's default (8)" > $Q
NO_CHECK=1 ITERS=2 WITH_TORCH=1 GPU_MAX_HW_QUEUES=4 timeout -k 10 300 python tools/concurrent_fits.py 4 2>/dev/null >> $Q
NO_CHECK=1 ITERS=2 WITH_TORCH=1 timeout -k 10 300 python tools/concurrent_fits.py 4 2>/dev/null >> $Q
cat $Q
