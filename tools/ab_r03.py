"""A/B timing of the round-3 schedule options on the GPU box: one bench.py run
(no checks, no extras) per (workload, option set); prints one JSON line each and
a table.  Run from the repo root:  python tools/ab_r03.py [quick]"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE = [sys.executable, os.path.join(REPO, 'bench.py'), '--no-cpu-baseline', '--no-check',
        '--no-extras', '--steps', '10', '--warmup', '3']

CASES = [
    ('cube f64', []),
    ('cube f32', ['--dtype', 'f32']),
    ('dam 0.0087', ['--workload', 'dam_break']),
    ('dam 0.0055', ['--workload', 'dam_break', '--dx', '0.0055']),
    ('tg', ['--workload', 'taylor_green']),
    ('rings f64', ['--workload', 'elastic']),
    ('rings f32', ['--workload', 'elastic', '--dtype', 'f32']),
]
CASES += [(n + ' prefetch_records=1', a + ['--opt', 'prefetch_records=1']) for n, a in list(CASES)]


def main():
    extra = sys.argv[1:]
    rows = []
    for name, argv in CASES:
        if extra and not any(e in name for e in extra):
            continue
        p = subprocess.run(BASE + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith('{')]
        if p.returncode != 0 or not line:
            print(json.dumps({'case': name, 'error': p.stderr[-400:]}), flush=True)
            continue
        d = json.loads(line[-1])
        r = {'case': name, 'ms_per_step': d['ms_per_step'], 'kernels': d['kernel_ms_per_step'],
             'families': d.get('pair_ms_per_family'), 'frac': d['roofline']['frac']}
        rows.append(r)
        print(json.dumps(r), flush=True)
    print()
    for r in rows:
        print('%-36s %8.3f ms/step  pair %7.3f  %s' % (
            r['case'], r['ms_per_step'], r['kernels']['pair'],
            ' '.join('%s=%.3f' % kv for kv in sorted((r['families'] or {}).items()))))


if __name__ == '__main__':
    main()
