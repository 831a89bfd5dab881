#!/bin/bash
# round 5, GPU visit g: ME parity (new cases), randomised soak inside the margins, two ranks sharing the GPU through bench.py's N > 1 path (gloo dry run), a quick bench
OUT=gpurun_out/r5g; mkdir -p $OUT
python -m pytest tests/test_gpu_me.py -m gpu -q > $OUT/me_tests.log 2>&1; tail -3 $OUT/me_tests.log
python tools/r5_me_minima_soak.py --seconds 90 --seed 2 > $OUT/soak.txt 2>&1; tail -2 $OUT/soak.txt
X265HIP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 --width 1920 --height 1080 > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err; echo "2-rank rc=$?"; tail -1 $OUT/bench_2ranks_gloo.json | cut -c1-1500
timeout 300 python bench.py --no-encoder --steps 100 > $OUT/bench_quick.json 2> $OUT/bench_quick.err; tail -1 $OUT/bench_quick.json | cut -c1-1200
