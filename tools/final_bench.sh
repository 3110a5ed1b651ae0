set -e
cd $GRAFT_REPO_ROOT
o=gpurun_out/final
mkdir -p $o
timeout -k 10 400 python bench.py > $o/r03_bench_default_with_cpu_baseline.json 2> $o/default.err
timeout -k 10 300 python bench.py --no-cpu-baseline --dtype bf16 > $o/r03_bench_dtype_bf16.json 2> $o/bf16.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-finetune > $o/r03_bench_no_finetune.json 2> $o/nf.err
timeout -k 10 300 python bench.py --no-cpu-baseline --with-tagger > $o/r03_bench_with_tagger.json 2> $o/tg.err
timeout -k 10 300 python bench.py --no-cpu-baseline --decoder-only > $o/r03_bench_decoder_only.json 2> $o/dec.err
timeout -k 10 300 python bench.py --no-cpu-baseline --workload pure_scn > $o/r03_bench_workload_pure_scn.json 2> $o/ps.err
timeout -k 10 300 python bench.py --no-cpu-baseline --workload pure_attention --batch 4 --max-len 20 --no-finetune > $o/r03_bench_workload_pure_attention_batch_4_max_len_20_no_finetune.json 2> $o/pa.err
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 10 > $o/r03_bench_soak_200steps.json 2> $o/soak.err
timeout -k 10 300 python bench.py --no-cpu-baseline --data hdf5-staged > $o/r03_bench_hdf5_staged.json 2> $o/h5.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/final/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print("%-75s %9.2f %s  %.3f ms (median %s)" % (f.split('/')[-1], d["value"], d["unit"], d["ms_per_step"], d.get("ms_per_step_median")))
    except Exception as e: print(f, "ERR", e)
PY
