# bash tools/cold_grad_diff.sh [runs]: fresh processes of tools/cold_grad_dump.py, then the first update / tensor at which
# a run that ends on an unusual third gradient norm departs from the majority
n=${1:-14}
rm -rf /tmp/cg; mkdir -p /tmp/cg
for i in $(seq $n); do timeout 120 python tools/cold_grad_dump.py /tmp/cg/$i.json 2>&1 | grep gnorm3; done
python - <<'PY'
import json, glob, collections
runs = [json.load(open(f)) for f in sorted(glob.glob('/tmp/cg/*.json'))]
key = lambda r: round(r['steps'][2]['gnorm'], 4)
groups = collections.Counter(key(r) for r in runs)
major = groups.most_common(1)[0][0]
print('third-update gradient norms:', dict(groups))
ref = next(r for r in runs if key(r) == major)
for b in [r for r in runs if key(r) != major][:3]:
    print('run with gnorm3 %.7f:' % b['steps'][2]['gnorm'])
    for u in range(3):
        rows = []
        for n, v in ref['steps'][u]['t'].items():
            w = b['steps'][u]['t'][n]
            dg = abs(w[0] - v[0]) / (v[1] + 1e-30); dp = abs(w[2] - v[2]) / (v[3] + 1e-30)
            rows.append((dg, dp, n))
        worst_g = sorted(rows, reverse=True)[:5]; worst_p = sorted(rows, key=lambda x: -x[1])[:5]
        print('  update %d: gnorm %.7f vs %.7f; losses differ by %.1e' % (u, b['steps'][u]['gnorm'], ref['steps'][u]['gnorm'],
              max(abs(x - y) for x, y in zip(b['steps'][u]['stats'], ref['steps'][u]['stats']))))
        print('     gradients (weighted checksum, rel. to abs-sum):', [(n.replace('transformer_layers', 'L'), '%.1e' % d) for d, _, n in worst_g if d > 1e-6])
        print('     parameters after the update               :', [(n.replace('transformer_layers', 'L'), '%.1e' % d) for _, d, n in worst_p if d > 1e-7])
PY
