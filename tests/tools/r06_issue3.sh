run() { timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --alt-issue "" --e2e-requests 120 --pipeline $2 --compute-streams $2 --microbatch $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d['config'].get('e2e') or {}
print('$1', d['value'], d['blocks']['median_images_per_s'], 'e2e pinned', (e.get('pinned') or {}).get('images_per_s'), 'pageable', (e.get('pageable') or {}).get('images_per_s'), 'parity', d.get('parity',{}).get('pass'), 'work', d.get('work_verified'))
"; }
for rep in 1 2; do
run 5x4 5 4
run 3x10 3 10
run 2x10 2 10
run 4x10 4 10
done
