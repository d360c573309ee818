# does placing kernel arguments in device memory (HIP_FORCE_DEV_KERNARG) change the dependent-launch cost?
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v python bench.py --no-cpu-baseline > gpurun_out/kernarg_$v.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("gpurun_out/kernarg_$v.json").read().strip().splitlines()[-1])
print("HIP_FORCE_DEV_KERNARG=$v", d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["single_block_kernel"]["avg_launch_us"], d["kernels"]["call_us"])
PY
done
