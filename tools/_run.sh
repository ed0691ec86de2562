mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -8 > gpurun_out/t_all.txt
for c in cfg2 cfg5 cfg3; do bash tools/profile_round.sh r02 $c > gpurun_out/profile_$c.log 2>&1; done
python bench.py > gpurun_out/r02_bench_cfg2.json 2> gpurun_out/bench_cfg2.err
PULSE_GEMM_F32=mfma32 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_cfg2_mfma32.json 2> gpurun_out/bench_cfg2m.err
python bench.py --config cfg5 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_cfg5.json 2> gpurun_out/bench_cfg5.err
python bench.py --config cfg5_f32 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_cfg5_f32.json 2> gpurun_out/bench_cfg5f.err
python bench.py --config cfg3 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_cfg3.json 2> gpurun_out/bench_cfg3.err
cat gpurun_out/t_all.txt
for f in gpurun_out/r02_bench_*.json; do echo $f; cut -c1-400 $f; done
