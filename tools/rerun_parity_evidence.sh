# Re-measure the statistical parity evidence on the GPU box after a change of the device's float arithmetic:
#   the published-result anchor (profiles/r03_published_anchor.json) and the paired product-sum LER study (profiles/r03_ler_productsum.json)
cd $GRAFT_REPO_ROOT
make -C oracle -s 2>&1 | tail -2
mkdir -p gpurun_out/evidence
python tools/published_anchor.py --shots 131072 --oracle-shots 2048 --out gpurun_out/evidence/published_anchor.json > gpurun_out/evidence/published_anchor.log 2>&1
tail -3 gpurun_out/evidence/published_anchor.log | cut -c1-300
python tools/ler_productsum.py gpu tests/golden/ler/bb144_ps_serial_osdcs1_seed1_part*.npz gpurun_out/evidence/ler_productsum.json > gpurun_out/evidence/ler_productsum.log 2>&1
tail -5 gpurun_out/evidence/ler_productsum.log | cut -c1-400
