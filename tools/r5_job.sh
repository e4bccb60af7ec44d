for w in c2 c3 c4; do for ts in 32 64; do python tools/rank_probe.py $w ranks=1,8 tile_size=$ts 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); r=d.pop('ranks'); print('$w tile', d['tile_size'], 'N', d['nranks'], 't_max', d['t_max_ms'], 'max/mean', d['max_over_mean'], 'rays max/mean', d['rays_max_over_mean'], 'N*mean/t1', d['n_x_mean_over_t1'], 'pred', d['predicted_speedup'], 'resolve/rank', round(sum(x['resolve'] for x in r)/len(r),1))"
done; done
