#!/bin/bash
# round 6, session a: the product as 2 and 8 real processes on the one MI355X (host-staged gloo), then the N = 1 line of this box
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r6a; mkdir -p $O
timeout 1500 python -m pytest tests/test_multibrick_gpu.py -m gpu -x -q -k "as_processes" > $O/pytest_processes.txt 2>&1; tail -5 $O/pytest_processes.txt
for n in 2 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2960$n bench.py --gpus $n \
     --backend gloo --ranks-per-gpu $n --ncell 128 --steps 10 --warmup 3 --dry-comm > $O/dry_comm_${n}ranks.json 2> $O/dry_comm_${n}ranks.err
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n \
     --backend gloo --ranks-per-gpu $n --ncell 128 --steps 10 --warmup 3 > $O/bench_${n}ranks.json 2> $O/bench_${n}ranks.err
  tail -c 600 $O/bench_${n}ranks.json; tail -3 $O/bench_${n}ranks.err
done
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 1500 $O/bench_n1.json
