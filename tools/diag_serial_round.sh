# One round of measurements for work on the serial part: parity tests, the Plaza phase profiles of the
# diagnostic build, the Plaza wall times and the commit-heavy and headline rates.
set -e
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
timeout -k 10 120 python tools/phase_profile.py --plaza Plaza1
timeout -k 10 120 python bench.py --workload commit-heavy --no-live-traffic --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('commit-heavy', d['value'], d['ms_per_step'])"
timeout -k 10 240 python bench.py --no-live-traffic --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_step'], d.get('plaza1'), d.get('plaza2'))"
