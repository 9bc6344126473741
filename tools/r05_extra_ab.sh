#!/bin/bash
# Round-5 side experiments on one GPU box: tools/r05_extra_ab.sh <tag>
#  (1) bench.py's N = 2 branch with both ranks on the box's single GPU (QD_BENCH_SHARE_GPU=1, reductions over gloo): a plumbing check
#      of the WORLD_SIZE > 1 code path on hardware, NOT a scaling measurement;
#  (2) the pipelined driver for plans whose BP runs in the per-edge kernel (QD_PIPELINE_EDGE=1), A/B on the reference's settings.
set -u
TAG=${1:-r05x}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
QD_BENCH_SHARE_GPU=1 QD_BENCH_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $O/two_ranks_one_gpu_gloo.json 2> $O/two_ranks_one_gpu_gloo.err
tail -c 400 $O/two_ranks_one_gpu_gloo.json; echo
QD_BENCH_SHARE_GPU=1 NCCL_DEBUG=WARN timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 > $O/two_ranks_one_gpu_rccl.json 2> $O/two_ranks_one_gpu_rccl.err
echo "rccl on a shared GPU: rc $?"; tail -c 300 $O/two_ranks_one_gpu_rccl.json; tail -5 $O/two_ranks_one_gpu_rccl.err
REF="--bp-method product_sum --schedule serial --max-iter 10 --osd-method osd_cs --osd-order 1 --no-cpu --no-api --no-other-configs"
for rep in 1 2; do
for pe in 0 1; do
    QD_PIPELINE_EDGE=$pe python bench.py --window 5 3 --shots 163840 --steps 3 --warmup 1 $REF > $O/ref_w5f3_pipe${pe}_$rep.json 2> $O/ref_w5f3_pipe${pe}_$rep.err
    QD_PIPELINE_EDGE=$pe python bench.py --code hgp225 --window 3 1 --shots 163840 --steps 3 --warmup 1 $REF > $O/hgp_w3f1_pipe${pe}_$rep.json 2> $O/hgp_w3f1_pipe${pe}_$rep.err
    QD_PIPELINE_EDGE=$pe python bench.py --shots 163840 --steps 3 --warmup 1 $REF > $O/ref_single_pipe${pe}_$rep.json 2> $O/ref_single_pipe${pe}_$rep.err
done
done
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/*pipe*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-32s %10.0f shots/s  LER %.6f  ms/step %.2f" % (os.path.basename(f), d["value"], d["logical_error_rate"], d["ms_per_step"]))
    except Exception as e:
        print(os.path.basename(f), "FAILED", e)
PY
