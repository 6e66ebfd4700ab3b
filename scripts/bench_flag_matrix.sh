cd $GRAFT_REPO_ROOT
run() { echo "== $*"; timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-legs "$@" 2>/tmp/err.txt | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value %.1f ms/step %.4f frac %.3f sets %s launch %s'%(d['value'],d['ms_per_step'],d['roofline']['frac'],d['config'].get('input_sets'),d['config']['launch'][:12]))
except Exception as e:
    print('   FAILED',e); print(open('/tmp/err.txt').read()[-800:])
"; }
run --graph
run --graph --input-sets 2
run --streams 1
run --streams 1 --input-sets 1
run --force-dist
run --no-fuse
run --gpus 1 --steps 20 --warmup 5
run --mlp-precision split16
run --config cfg3_small_b8 --steps 6 --warmup 2
