timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['one_frame_at_a_time']['ms_per_frame'])"
done
git stash -q; python -c "
from cobevt_amd import build; build.build(verbose=False)" > /dev/null 2>&1
