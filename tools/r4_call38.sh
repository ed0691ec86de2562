#!/bin/bash
mkdir -p gpurun_out/r4 gpurun_out/r4p
timeout 1200 python -m pytest tests/test_bf16_gpu.py tests/test_amp_agent_gpu.py tests/test_disc_gpu.py tests/test_full_size_parity_gpu.py tests/test_agent_parity2_gpu.py -q > gpurun_out/r4/t_c38.log 2>&1; tail -3 gpurun_out/r4/t_c38.log
for i in 1 2; do
timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r04_bench_cfg5.json 2> gpurun_out/r4/bench_cfg5_final.err; python -c "
import json; d=json.loads(open('gpurun_out/r04_bench_cfg5.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg5', round(d['ms_per_step'],2), round(d['value']), 'play', round(d['play_ms_per_step'],2), 'upd', round(d['update_ms_per_step'],2), round(r['achieved'],1), {k:(v['launches'],round(v['avg_us'],1),round(v['tflops'],1)) for k,v in r['by_variant'].items()})"
done
PULSE_DISC_STREAM=0 timeout 300 python bench.py --config cfg5 --no-cpu-baseline --steps 6 --warmup 2 > gpurun_out/r04_bench_cfg5_one_chain.json 2> gpurun_out/r4/bench_cfg5_oc.err; python -c "
import json; d=json.loads(open('gpurun_out/r04_bench_cfg5_one_chain.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cfg5 one chain', round(d['ms_per_step'],2), round(d['value']), 'play', round(d['play_ms_per_step'],2), 'upd', round(d['update_ms_per_step'],2), round(r['achieved'],1), {k:(v['launches'],round(v['avg_us'],1),round(v['tflops'],1)) for k,v in r['by_variant'].items()})"
bash tools/profile_round.sh r04 cfg5 > gpurun_out/r4p/profile_cfg5.log 2>&1
PULSE_DISC_STREAM=0 SKIP_PMC=1 bash tools/profile_round.sh r04onechain cfg5 > gpurun_out/r4p/profile_cfg5_onechain.log 2>&1
tail -1 gpurun_out/r04_bench_kernel_stats_cfg5.md; tail -1 gpurun_out/r04onechain_bench_kernel_stats_cfg5.md
