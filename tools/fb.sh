cd /tmp && export TMPDIR=/tmp
for w in 150 75 30; do
  export PS_RW=$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fb_$w -o r -- python3 $GRAFT_REPO_ROOT/tools/gpu_fillbench.py 2>&1 | grep RW=
  f=$(find /tmp/fb_$w -name '*kernel_stats.csv' | head -1)
  grep -E "k_recur|k_emis|k_steps|k_backtrace" $f | cut -d, -f1-5
done
