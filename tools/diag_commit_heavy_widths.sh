set -e
for cfg in "0 0" "64 2" "128 2" "256 2" "64 1" "128 1"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --workload commit-heavy --no-live-traffic --steps 10 --warmup 2 --threads-per-chain $1 --serial-program $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
