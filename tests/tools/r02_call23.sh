R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r02w
mkdir -p $O
DD3D_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_1gpu.json 2> $O/bench_2ranks.err; echo "rc=$?"; tail -c 1500 $O/bench_2ranks_1gpu.json; tail -3 $O/bench_2ranks.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20steps.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_20steps.json')); print('20 steps:', d['value'], d['blocks']['median_images_per_s'], d['roofline']['traffic'], d['roofline']['traffic_source'])"
timeout 300 python tests/gpu_rccl_check.py 2>&1 | tail -2
