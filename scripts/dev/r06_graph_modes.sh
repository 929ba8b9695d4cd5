#!/bin/bash
# eager vs --graph vs graph with the copy node outside vs --serial-host: ms/step of the bench step, same box
F="--steps 200 --no-other-configs --no-cpu-baseline --no-end-to-end"
get() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', d['ms_per_step'], d['roofline']['kernel_ms'] if 'kernel_ms' in d['roofline'] else '')"; }
python bench.py $F 2>/dev/null | get eager
python bench.py $F --graph 2>/dev/null | get graph
BENCH_GRAPH_COPY_OUTSIDE=1 python bench.py $F --graph 2>/dev/null | get graph_copy_outside
python bench.py $F --serial-host 2>/dev/null | get serial_host
python bench.py $F 2>/dev/null | get eager
