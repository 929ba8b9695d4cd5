#!/bin/bash
# the headline step several times, as the driver runs it (steps 20): spread and host pace
for i in 1 2 3 4 5 6; do
python bench.py --steps ${STEPS:-20} --warmup ${WARM:-5} --no-other-configs --no-cpu-baseline --no-end-to-end 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step %.4f  kernel %.4f  host gaps %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], {k: round(v,3) for k,v in d['extra']['step_host_ms'].items() if k!='note'}))"
done
