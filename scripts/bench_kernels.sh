#!/bin/bash
# prints per-kernel ms/step from bench.py (dev aid)
python bench.py --steps 3 --warmup 2 --no-cpu "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('Msps',d['value'],'ms/step',d['ms_per_step'],d['loop_passes'],'chain_frac',d['roofline']['chain_frac'])
print('  '+'  '.join(f\"{k}={v['total_ms']/d['steps']:.3f}/{v['launches']/d['steps']:.0f}\" for k,v in d['kernels'].items()))
"
