cd /tmp && cp $GRAFT_REPO_ROOT/tools/swprobe.hip . 
for v in ${VARIANTS:-0 1 2 3 4 5 6}; do hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -DVARIANT=$v -o swprobe$v swprobe.hip 2>/dev/null && ./swprobe$v; done
