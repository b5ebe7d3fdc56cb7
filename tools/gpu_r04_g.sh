#!/bin/bash
# call G: the driver's bench command, complete line (fp32 / bf16x3 legs, other configs, config-5 end to end), wall time
cd "$GRAFT_REPO_ROOT"; o=gpurun_out/r04g; mkdir -p $o
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench.json 2> $o/bench.err
echo "bench rc=$? wall=$(( $(date +%s) - t0 ))s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04g/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'seq', d['sequential_clips_per_s'], 'h2d', d['host_frames']['value_including_upload'])
print('roofline', d['roofline']['frac'], d['roofline']['achieved'], d['roofline']['all_conv_kernels'])
print('fp32', d.get('fp32_mode')); print('x3', d.get('bf16x3_mode'))
for k,v in d['other_configs'].items(): print(k, {a:b for a,b in v.items() if a not in ('workload',)})
print([ (r['kernel'], r['frac']) for r in d['roofline_hbm']])
PY
grep -v amdgpu.ids $o/bench.err | tail -5
