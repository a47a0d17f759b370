#!/usr/bin/env python3
"""Writes tests/golden/padlock_golden.json: TSV / JSON / exit code of every case of tests/padlock_fixture.py as produced by
the oracle's restated padlock.h + gtf.h driver calling the REFERENCE's own thal.h (oracle/_ref, built in place from
/root/reference).  Paths inside the JSON meta block are stored relative to the scenario directory ("$D")."""
import json, os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O, padlock_fixture as F
d = tempfile.mkdtemp(prefix="padlock_golden_")
sc = F.build(d)
O.build_fm9(sc["text"], sc["fm9"])
orc = O.Index(sc["fm9"])
gold = {}
for case in F.CASES:
    tsv, js, err, rc = F.oracle_run(orc, sc, case, os.path.join(d, "out.tsv"), os.path.join(d, "out.json.gz"))
    gold[case[0]] = {"tsv": tsv.replace(d, "$D"), "json": js.replace(d, "$D"), "rc": rc, "rows": max(0, tsv.count("\n") - 1)}
    print(case[0], "rows", gold[case[0]]["rows"], "rc", rc)
json.dump(gold, open(os.path.join(HERE, "padlock_golden.json"), "w"), indent=0, sort_keys=True)
