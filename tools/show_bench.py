#!/usr/bin/env python3
"""One line per bench.py JSON file: ms per step, value, CUs per query, roofline fraction.  python tools/show_bench.py FILE..."""
import json, sys
for path in sys.argv[1:]:
    d = json.load(open(path))
    print(f"{path}: {d['ms_per_step']:.2f} ms/step  {d['value']:.0f} {d['unit']}  cus/query {d['config'].get('cus_per_query')}  "
          f"kernel {d['roofline']['kernel_ms']:.2f} ms  achieved {d['roofline']['achieved']:.0f} GB/s  frac {d['roofline']['frac']:.4f}")
