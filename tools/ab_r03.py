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
    ('cube f64 eos_fuse=0', ['--opt', 'eos_fuse=0']),
    ('cube f32', ['--dtype', 'f32']),
    ('cube f32 eos_fuse=0', ['--dtype', 'f32', '--opt', 'eos_fuse=0']),
    ('dam 0.0087', ['--workload', 'dam_break']),
    ('dam 0.0087 eos_fuse=0', ['--workload', 'dam_break', '--opt', 'eos_fuse=0']),
    ('tg', ['--workload', 'taylor_green']),
    ('tg nl_reuse=0', ['--workload', 'taylor_green', '--opt', 'nl_reuse=0']),
    ('tg norm_masks=0', ['--workload', 'taylor_green', '--opt', 'norm_masks=0']),
    ('tg nl_reuse=0 norm_masks=0', ['--workload', 'taylor_green', '--opt', 'nl_reuse=0', '--opt', 'norm_masks=0']),
    ('rings f64', ['--workload', 'elastic']),
    ('rings f64 nl_reuse=0', ['--workload', 'elastic', '--opt', 'nl_reuse=0']),
    ('rings f64 nl_reuse=0 norm_masks=0', ['--workload', 'elastic', '--opt', 'nl_reuse=0', '--opt', 'norm_masks=0']),
    ('rings f32', ['--workload', 'elastic', '--dtype', 'f32']),
    ('rings f32 nl_reuse=0', ['--workload', 'elastic', '--dtype', 'f32', '--opt', 'nl_reuse=0']),
    ('block f64', ['--workload', 'elastic_block', '--n1', '126']),
    ('cube.py params', ['--params', 'cube']),
    ('cube.py params norm_masks=0', ['--params', 'cube', '--opt', 'norm_masks=0']),
]


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
