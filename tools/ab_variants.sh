# Same-box A/B of library variants under build_ablate/ (BP stage alone on the headline window, tools/k1_time.py; checksums must agree):
# tools/ab_variants.sh <tag> <variant names...>     (through gpurun)
set -u
TAG=$1; shift
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do
  for v in tree "$@"; do
    if [ $v = tree ]; then unset QUITS_AMD_LIB; else export QUITS_AMD_LIB=$PWD/build_ablate/lib_$v.so; fi
    echo -n "$v: "; K1_STAGES=1 python tools/k1_time.py 2>/dev/null | tail -1
  done
done 2>&1 | tee $O/ab.txt
unset QUITS_AMD_LIB
