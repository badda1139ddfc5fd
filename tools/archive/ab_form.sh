# measurement aid (GPU box): formation_kernel at four / five workgroups per CU (variants built by tools/mkvariant.sh measure -DFMARL_MEASURE
# and tools/mkvariant.sh form5 -DFMARL_MEASURE -DFMARL_FORM_MIN_BLOCKS=5; FMARL_FORM_EPB = envs per workgroup).  usage: bash tools/archive/ab_form.sh
R=$GRAFT_REPO_ROOT; cd $R
run() { echo -n "$1 epb=${2:-24}: "; FMARL_LIB=$R/fair_marl_amd/csrc/variants/libfmarl_$1.so FMARL_FORM_EPB=$2 python bench.py --config cfg4 --steps 300 --warmup 50 --no-cpu-baseline --no-secondary --launch ${3:-step} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms_per_step=%.4f kernel=%.4f' % (d['ms_per_step'], r.get('step_kernels_ms_per_step') or r['kernel_avg_ms']))"; }
for r in 1 2; do run measure ""; run form5 ""; run form5 23; run measure 23; run form5 22; done
echo spans; run measure "" span; run form5 23 span
