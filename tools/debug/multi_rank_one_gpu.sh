export OVVC_BENCH_DEBUG_GLOO=1
for deal in gop picture; do
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-isolated-survey --dealing $deal --both-dealings > gpurun_out/mr_$deal.json 2> gpurun_out/mr_$deal.err; echo rc=$?
python -c "
import json; d=json.loads([l for l in open('gpurun_out/mr_$deal.json') if l.startswith('{')][-1]); print('$deal', d['value'], d['n_gpus'], d['scaling'], d['config'].get('transfers_this_rank'), d['config']['check'].get('differ') if d['config'].get('check') else None)"
tail -2 gpurun_out/mr_$deal.err
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --no-isolated-survey --scaling strong > gpurun_out/mr_strong.json 2> gpurun_out/mr_strong.err; echo rc=$?
python -c "
import json; d=json.loads([l for l in open('gpurun_out/mr_strong.json') if l.startswith('{')][-1]); print('strong', d['value'], d['scaling'], d['config'].get('transfers_this_rank'))"
