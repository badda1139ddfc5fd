#!/bin/bash
# measurement aid: short bench, prints ms/step and step-kernel average; extra args go to bench.py
python bench.py --steps ${1:-100} --warmup 25 --no-cpu-baseline ${@:2} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step=%.3f  kernel_avg_ms=%.3f  value=%.3e frac=%.3f' % (d['ms_per_step'], d['roofline']['kernel_avg_ms'], d['value'], d['roofline']['frac']))"
