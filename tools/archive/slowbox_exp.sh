#!/bin/bash
# measurement aid: on a box without a fast output placement, what limits the step kernel?
# exits at once on the other kind of box.  usage: tools/slowbox_exp.sh   (on the GPU box)
cd "$(dirname "$0")/.."
T=$(python bench.py --steps 25 --warmup 25 --no-cpu-baseline --sync-reset 2>/dev/null | python -c "import sys,json,re; d=json.loads(sys.stdin.read()); print(re.search(r'launch ms ([0-9.]+)', d['config']['output_placement']).group(1))")
echo "best emission-only pair: $T ms"
python -c "import sys; sys.exit(0 if float('$T') > 1.45 else 1)" || { echo "fast box: nothing to do"; exit 0; }
echo "SLOW BOX"
rocm-smi --showclocks --showpower 2>&1 | grep -i "clk\|power"
cd fair_marl_amd/csrc && cp libfmarl.so libfmarl_ship.so && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -DFMARL_MEASURE -shared -fPIC -o libfmarl.so libfmarl.hip && cd ../..
q() { python bench.py --steps 100 --warmup 25 --no-cpu-baseline --sync-reset 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 ms_per_step=%.3f kernel_avg_ms=%.3f' % (d['ms_per_step'], d['roofline']['kernel_avg_ms']))"; }
for m in 0 31 1 8 16 2 4; do FMARL_ABLATE=$m q "ablate=$m"; done
for pad in 0 20000 35000 60000; do FMARL_LDS_PAD=$pad q "lds_pad=$pad"; FMARL_LDS_PAD=$pad FMARL_ABLATE=31 q "lds_pad=$pad emission-only"; done
cp fair_marl_amd/csrc/libfmarl_ship.so fair_marl_amd/csrc/libfmarl.so && rm fair_marl_amd/csrc/libfmarl_ship.so
