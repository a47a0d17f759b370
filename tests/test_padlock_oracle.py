"""`dicey padlock` restatement (oracle/padlock_ref.hpp) against the committed golden outputs (tests/golden/padlock_golden.json,
generated with the reference's own thal.h) and against properties the reference's filters imply."""
import json
import os

import pytest

import oracle_lib as O
import padlock_fixture as F

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "padlock_golden.json")))
needs_ref = pytest.mark.skipif(O.ref_libs() is None, reason="oracle/_ref (reference thal.h built in place) is not available")


@pytest.fixture(scope="module")
def scenario(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("padlock"))
    sc = F.build(d)
    O.build_fm9(sc["text"], sc["fm9"])
    sc["orc"] = O.Index(sc["fm9"])
    return sc


@needs_ref
@pytest.mark.parametrize("case", F.CASES, ids=[c[0] for c in F.CASES])
def test_oracle_reproduces_golden(scenario, case):
    d = scenario["dir"]
    tsv, js, err, rc = F.oracle_run(scenario["orc"], scenario, case, os.path.join(d, "out.tsv"), os.path.join(d, "out.json.gz"))
    g = GOLD[case[0]]
    assert rc == g["rc"]
    assert tsv.replace(d, "$D") == g["tsv"]
    assert js.replace(d, "$D") == g["json"]


def test_golden_rows_respect_the_filters():
    comp = str.maketrans("ACGT", "TGCA")
    for label, g in GOLD.items():
        if g["rc"]:
            assert g["rows"] == 0 and not g["json"].rstrip().endswith("]}}")   # nothing, or a header cut short by the error exit
            continue
        lines = g["tsv"].rstrip("\n").split("\n")
        assert lines[0].startswith("Gene\tSymbol\tCode\tPosition")
        rows = [ln.split("\t") for ln in lines[1:]]
        assert len(rows) == g["rows"] > 0
        data = json.loads(g["json"])["data"]
        assert [list(map(str, r)) for r in data["rows"]] == rows          # the JSON rows carry the same fields as the TSV
        for r in rows:
            arm1, arm2 = r[7].split("-")
            assert len(arm1) == len(arm2)
            padlock = r[12]
            assert padlock.startswith(arm1.translate(comp)[::-1]) and padlock.endswith(arm2.translate(comp)[::-1])  # padlock.h:431
            gc = lambda s: (s.count("C") + s.count("G")) / len(s)
            assert abs(gc(arm1) - float(r[17])) < 1e-5 and abs(gc(arm2) - float(r[18])) < 1e-5
        if label == "genelist":   # non-overlapping probes of one exon are at least one probe length apart (padlock.h:506)
            by_exon = {}
            for r in rows:
                by_exon.setdefault(r[6], []).append(int(r[3].split(":")[1]))
            for v in by_exon.values():
                v.sort()
                assert all(b - a >= 40 for a, b in zip(v, v[1:]))


def test_gtf_flattening_joins_overlapping_and_touching_exons(scenario):
    # ENSG01 has exons 2100-2500 and 2400-2800 (1-based, closed): one joined feature chr1:2100-2800 in the output
    rows = [ln.split("\t") for ln in GOLD["hamming_overlapping"]["tsv"].rstrip("\n").split("\n")[1:]]
    feats = {r[6] for r in rows if r[0] == "ENSG01"}
    assert "chr1:2100-2800" in feats and not any(f.startswith("chr1:2400") for f in feats)
