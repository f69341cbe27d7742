"""Regenerates tests/golden/cases.json: outputs of the CPU oracle (oracle/ccref.c) on seeded inputs.

    python tests/golden/make_golden.py

The reference itself (Go) cannot be executed in the build image, so these vectors are oracle outputs, not
reference outputs; the oracle in turn is pinned to the reference's published known answers by
tests/test_oracle_known_answers.py.  They freeze the oracle (a change of any result shows up as a diff of
this file) and let the GPU tests check the HIP path on a box without rebuilding confidence in the oracle.
Inputs are described by (generator, arguments, seed) -- see tests/golden_cases.py -- not stored."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
import ccref_py  # noqa: E402
import golden_cases  # noqa: E402
from cluster_capacity_amd import report as R  # noqa: E402


def main():
    out = {}
    for name in golden_cases.CASES:
        nodes, pod, prof, limit = golden_cases.build(name)
        r = ccref_py.run(prof, nodes, pod, max_limit=limit)
        out[name] = golden_cases.summarize(r, nodes.n, limit)
    with open(os.path.join(HERE, "cases.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
