"""oracle/ -- TEST INFRASTRUCTURE ONLY (never the product path).

A CPU restatement (plain torch fp32/fp64 on CPU + a plain-C LSAP) of the Counting-DETR
2nd-stage hot path named by BASELINE.json:north_star.  Every function cites the reference
file:line it restates (abbreviations as in SURVEY.md: A2/ = src/CountDETR_147_2nd_stage/).

Who may import this package: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg --
and there only as the checker / the timed CPU baseline.  The product package
(counting_detr_amd/) never imports it and fails loudly when its HIP extension is missing.

Parity pinning: the reference ships NO tests / golden vectors for this path, so
the oracle is pinned against outputs of the reference itself, imported in the build container
with torchvision stubs by oracle/gen_golden.py; the resulting vectors are committed under
tests/golden/ and checked by tests/test_oracle_golden.py.  The third-party solver on the path is
scipy.optimize.linear_sum_assignment (scipy 1.15.3 in this image; unpinned in the reference's
requirements) -- oracle/lsap.c restates its published shortest-augmenting-path algorithm and is
checked against scipy itself in tests/test_lsap_oracle.py.
"""
