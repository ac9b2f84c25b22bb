# rocprofv3 kernel trace of bench.py (R regions per GPU); usage: bash tools/prof_bench.sh <R> <tag>
cd /tmp && export TMPDIR=/tmp
R=${1:-8}; TAG=${2:-bench}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$TAG -o r -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --regions-per-gpu $R --no-cpu > /tmp/pb_$TAG.log 2>&1
tail -1 /tmp/pb_$TAG.log | cut -c1-400
f=$(find /tmp/pb_$TAG -name '*kernel_stats.csv' | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/prof
cp $f $GRAFT_REPO_ROOT/gpurun_out/prof/${TAG}_kernel_stats.csv; python3 $GRAFT_REPO_ROOT/tools/concur.py $(find /tmp/pb_$TAG -name "*kernel_trace.csv" | head -1) 0.5
head -12 $f | cut -d, -f1-8
