mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/t_all.txt
bash tools/profile_round.sh r02 cfg2 > gpurun_out/profile_cfg2.log 2>&1
python bench.py > gpurun_out/r02_bench_cfg2.json 2> gpurun_out/bench_cfg2.err
python bench.py --config cfg5 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bench_cfg5.json 2> gpurun_out/bench_cfg5.err
cat gpurun_out/t_all.txt
for f in gpurun_out/r02_bench_cfg2.json gpurun_out/r02_bench_cfg5.json; do cut -c1-200 $f; done
