for w in 16 32 8; do
  echo "wgs_body $w: $(SR_WGS_BODY=$w python bench.py --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["kernels"]["call_us"])')"
done
