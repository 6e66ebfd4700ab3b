cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
DT_HEADS_IN_CONV=$v python bench.py --config cfg5_small_d96 --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('heads_in_conv=$v cfg5_small: %.1f f/s %.4f ms conv %.4f launches %s' % (d['value'], d['ms_per_step'], d['roofline_conv']['avg_ms'], d['roofline_conv']['launches']))"
done
for v in 1 0; do
DT_HEADS_IN_CONV=$v python bench.py --config cfg3_small_b8 --steps 30 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('heads_in_conv=$v cfg3_small_b8: %.1f f/s %.4f ms conv %.4f launches %s' % (d['value'], d['ms_per_step'], d['roofline_conv']['avg_ms'], d['roofline_conv']['launches']))"
done
