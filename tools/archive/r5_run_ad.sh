B="--steps 16 --warmup 4 --no-cpu-baseline --alt-steps 0 --dropin-steps 0 --kernel-steps 0 --no-kernel-timing --other-presets="
line() { python -c "import sys,json; l=json.loads([x for x in sys.stdin.read().splitlines() if x.startswith('{')][-1]); print('$1', l['value'], l['ms_per_step'], l['ms_d_call_median'], l['ms_g_call_median'], l['config'].get('streams','')[:40])"; }
mkdir -p gpurun_out/r5ad
python bench.py $B 2>/dev/null | line "two streams" | tee -a gpurun_out/r5ad/streams.txt
SAE_TWO_STREAMS=0 python bench.py $B 2>/dev/null | line "one stream" | tee -a gpurun_out/r5ad/streams.txt
SAE_TWO_STREAMS=0 python bench.py $B --force-allreduce 2>/dev/null | line "one stream, force_allreduce" | tee -a gpurun_out/r5ad/streams.txt
python bench.py $B --force-allreduce 2>/dev/null | line "two streams, force_allreduce" | tee -a gpurun_out/r5ad/streams.txt
