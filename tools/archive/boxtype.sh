#!/bin/bash
# measurement aid: which kind of box is this (see DESIGN.md "Where the output buffers live")?  prints the placement line of a short bench
python bench.py --steps 50 --warmup 25 --no-cpu-baseline --sync-reset 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_avg_ms=%.3f  %s' % (d['roofline']['kernel_avg_ms'], d['config']['output_placement']))"
