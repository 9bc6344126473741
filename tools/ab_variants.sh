# Same-box A/B of library variants under build_ablate/ (tools/k1_time.py: BP stage alone, or K1_STAGES=2 for BP + OSD; checksums must
# agree): tools/ab_variants.sh <tag> <variant names...>     (through gpurun; K1_ARGS="name shots max_iter" picks another window)
set -u
TAG=$1; shift
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do
  for v in tree "$@"; do
    if [ $v = tree ]; then unset QUITS_AMD_LIB; else export QUITS_AMD_LIB=$PWD/build_ablate/lib_$v.so; fi
    python tools/k1_time.py ${K1_ARGS:-} 2>/dev/null | sed "s|^[^ ]* |$v |"
  done
done 2>&1 | tee -a $O/ab.txt
unset QUITS_AMD_LIB
