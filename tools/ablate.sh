# ablation of the bench step through the DEODR_HIP_DEBUG switches (see KParams::debug); args: the masks to try
cd $GRAFT_REPO_ROOT
for d in ${@:-0 1 2 3 4 16 32 128 160 176}; do
 echo -n "debug=$d: "; DEODR_HIP_DEBUG=$d python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k[:10]:round(v['avg_ms'],4) for k,v in d['roofline']['per_kernel'].items()})"
done
