#!/bin/bash
# A/B of kernel variants on ONE box without rebuilding there: time a command under several prebuilt libraries.
#   here (CPU container): build each variant of the changed object with its -D flags, link it with the other objects of
#   poreseq_amd/csrc/build/ into tools/_libs/lib_<name>.so (built .so files travel with the snapshot; gpurun_out/ does not), e.g.
#     hipcc $FLAGS -DPS_X=1 -c ps_kernels.hip -o /tmp/pk_x1.o && hipcc --offload-arch=gfx950 -shared -fPIC -pthread -Wl,-Bsymbolic \
#         -o tools/_libs/lib_x1.so /tmp/pk_x1.o build/ps_sweep.o ... build/ps_api.o
#   on the GPU box:  bash tools/ab_variants.sh "<name> <name> ..." <command ...>     (each name's library is swapped in, the command's
#   first output line is printed; the snapshot's own library is restored at the end).  ~30 s of box time for eight runs of
#   tools/gpu_scorebench.py: how the k_score loop cuts of round 5 were chosen (two of five candidates were slower and left out).
# Remove tools/_libs/ before committing.
set -u
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it)}"
cd "$GRAFT_REPO_ROOT"
names=$1; shift
lib=poreseq_amd/csrc/libporeseq_hip.so
cp $lib /tmp/lib_keep.so || exit 1
trap 'cp /tmp/lib_keep.so "$lib"' EXIT      # an interrupt or a timeout must not leave a variant library installed
for v in $names; do
  cp tools/_libs/lib_$v.so $lib || continue
  echo "variant $v: $("$@" 2>&1 | head -1)"
done
