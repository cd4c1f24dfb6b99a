for L in costas_exact costas_exact_fix agc_exact costas_final clock_overlap; do
XRIT_LIGHT_LIST=$L timeout 300 python bench.py --front-exact 2 --no-exact --no-cpu --no-serial-floor --no-other-configs --steps 40 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels'].get('$L',{})
print('$L', 'ms/step', d['ms_per_step'], 'in-pipeline avg', k.get('avg_launch_ms'), 'launches', k.get('launches'), k.get('measured'))"
done
