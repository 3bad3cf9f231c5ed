set -e
python3 -c "import __graft_entry__ as g; g.smoke()"
for fl in 0 8; do for pr in 4 16; do
 r=$(timeout -k 10 120 python3 bench.py --steps 10 --warmup 2 --no-cpu --workload cfg2_1080p_w9_d64 --pairs $pr --flags $fl 2>/dev/null)
 echo "cfg2 flags=$fl pairs=$pr $(echo $r | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["kernel_ms"], d["config"]["kernel"])')"
done; done
timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu 2>&1 | tail -2
timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu --split bands --workload cfg4_8k_w21_d256 2>&1 | tail -2
