for cfg in "64 64" "128 64" "256 64" "128 128" "256 128" "256 256"; do
  set -- $cfg
  echo "tail $1 head $2: $(SR_WGS_TAIL=$1 SR_WGS_HEAD=$2 python bench.py --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels"]["call_us"])')"
done
