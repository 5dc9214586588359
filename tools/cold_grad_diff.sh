mkdir -p /tmp/cg
for i in $(seq 14); do S2ST_ATTN_GFUSE=1 S2ST_NO_SIDE_STREAM=1 timeout 120 python tools/cold_grad_dump.py /tmp/cg/$i.json 2>&1 | grep gnorm3; done
python - <<'PY'
import json, glob
runs = [json.load(open(f)) for f in sorted(glob.glob('/tmp/cg/*.json'))]
good = [r for r in runs if abs(r['__gnorm3__'] - 1.3372) < 2e-4]
bad = [r for r in runs if abs(r['__gnorm3__'] - 1.3372) >= 2e-4]
print(len(good), 'good', len(bad), 'bad')
if good and bad:
    ref = good[0]
    for b in bad[:3]:
        print('bad run gnorm3', b['__gnorm3__'])
        for n in ref:
            if n.startswith('__'): continue
            dg = abs(b[n][0] - ref[n][0]) / (abs(ref[n][0]) + 1e-12); dp = abs(b[n][1] - ref[n][1]) / (abs(ref[n][1]) + 1e-12)
            if dg > 1e-4 or dp > 1e-7: print('   grad after update 2 differs: %-60s rel %.2e  (param checksum rel %.1e)' % (n, dg, dp))
        if len(ref[n]) > 2 if False else True:
            print('   losses of step 3: bad', ['%.7f' % x for x in b['__stats3__']], ' good', ['%.7f' % x for x in ref['__stats3__']])
            rows = sorted(((abs(b[n][2] - ref[n][2]) / (abs(ref[n][2]) + 1e-12), n) for n in ref if not n.startswith('__')), reverse=True)
            for d_, n in rows[:14]: print('   step-3 gradient differs: %-62s rel %.2e' % (n, d_))
    # also good vs good
    if len(good) > 1:
        mx = max(abs(good[1][n][0] - ref[n][0]) / (abs(ref[n][0]) + 1e-12) for n in ref if not n.startswith('__'))
        print('good vs good: max per-tensor gradient checksum difference %.1e' % mx)
PY
