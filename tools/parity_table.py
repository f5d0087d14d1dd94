"""Print the oracle-parity numbers of the BASELINE-size cases (the table of
BASELINE.md section 4 / DESIGN.md section 2); same code path as tests/test_baseline_sizes.py."""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from test_baseline_sizes import _case

CASES = [
    ('cube 159^3 f64 sorted', ['--n1', '159']),
    ('cube 159^3 f64 unsorted', ['--n1', '159', '--no-reorder']),
    ('cube 159^3 f64 h+-15%', ['--n1', '159', '--vary-h', '0.15']),
    ('cube 159^3 f64 cube.py params', ['--n1', '159', '--params', 'cube']),
    ('cube 159^3 f32', ['--n1', '159', '--dtype', 'f32']),
    ('dam break C2 f64', ['--workload', 'dam_break', '--dx', '0.0087']),
    ('taylor-green 159^3 f64', ['--workload', 'taylor_green', '--n1', '159']),
    ('taylor-green 159^3 f32', ['--workload', 'taylor_green', '--n1', '159', '--dtype', 'f32']),
    ('elastic 126^3 f64', ['--workload', 'elastic', '--n1', '126']),
    ('elastic 126^3 f32', ['--workload', 'elastic', '--n1', '126', '--dtype', 'f32']),
]
out = {}
for name, argv in CASES:
    res, n, ordered = _case(argv)
    out[name] = {'particles': n, 'max_rel': res['parity_max_rel'], 'worst': res['parity_worst_field'],
                 'count_mismatches': res.get('parity_neighbour_count_mismatches')}
    print(name, out[name], flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'parity_table.json'), 'w'), indent=1)
