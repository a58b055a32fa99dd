for f in "USE_CONV3_S2=1" "USE_CONV3_S2=0" "USE_CONV3_S2=1" "USE_CONV3_S2=0"; do
COBEVT_FLAGS=$f timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['one_frame_at_a_time']['ms_per_frame'])"
done
