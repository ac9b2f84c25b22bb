# rocprofv3 kernel stats of an arbitrary python script: bash tools/prof_cmd.sh <tag> <script> [args...]
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$TAG -o r -- python3 $GRAFT_REPO_ROOT/"$@" > /tmp/pc_$TAG.log 2>&1
tail -3 /tmp/pc_$TAG.log
f=$(find /tmp/pc_$TAG -name '*kernel_stats.csv' | head -1)
head -${TOPN:-12} $f | sed "s/([^)]*)/()/g" | cut -d, -f1-6
