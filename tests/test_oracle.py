"""CPU tests of the ORACLE: pinned by the known answers SURVEY.md §8(c) measured from the unmodified reference
headers, by the brute-force text-search twin, and regression-pinned by tests/golden/*.json."""
import json
import os
import random

import oracle_lib as O
from conftest import genome_text, make_genome

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_neighbors_known_answers_from_survey():
    # 20-mer TCTCTGCACACACGTTGTAC -> |N| = 1/61/1771 (hamming d=0/1/2), 1/116/6019 (edit d=0/1/2)
    q = "TCTCTGCACACACGTTGTAC"
    assert [len(O.neighbors(q, d, False)) for d in (0, 1, 2)] == [1, 61, 1771]
    assert [len(O.neighbors(q, d, True)) for d in (0, 1)] == [1, 116]


def test_neighbors_d2_known_answer_from_survey():
    assert len(O.neighbors("TCTCTGCACACACGTTGTAC", 2, True)) == 6019


def test_needle_known_answer_from_survey():
    # needle("GACGTTCGTACA","ACGTACGTAC") -> score -1, rows GACGTTCGTACA / -ACGTACGTAC-, trailGap 1
    assert O.needle("GACGTTCGTACA", "ACGTACGTAC") == (-1, "GACGTTCGTACA", "-ACGTACGTAC-", 1)


def _full_language(q, d):
    """Independent enumeration of the <=d-edit language of neighbors.h (no insertion after the last character)."""
    lang = {q}
    frontier = {(q, 0)}  # (string, position)
    res = set()

    def rec(s, pos, left, used):
        if pos >= len(s):
            if used:
                res.add(s)
            return
        if left:
            rec(s[:pos] + s[pos + 1:], pos, left - 1, True)
        rec(s, pos + 1, left, used)
        if left:
            for a in "ACGT":
                if a != s[pos]:
                    rec(s[:pos] + a + s[pos + 1:], pos + 1, left - 1, True)
            for a in "ACGT":
                rec(s[:pos] + a + s[pos:], pos + 1, left - 1, True)

    rec(q, 0, d, False)
    return lang | res


def test_neighbors_equal_substring_minimal_language():
    rng = random.Random(3)
    for q in ["ACGTACGTAC", "AAAAAAAAAAAA", "ACACACACACAC", "".join(rng.choice("ACGT") for _ in range(14)), "ACGTNACGTTGCA"]:
        for d in (1, 2):
            if d == 2 and len(q) > 12:
                continue
            lang = _full_language(q, d)
            minimal = sorted(s for s in lang if not any(t != s and t in s for t in lang))
            assert O.neighbors(q, d, True) == minimal, (q, d)
            ham = sorted(s for s in lang if len(s) == len(q) and sum(a != b for a, b in zip(s, q)) <= d)
            assert O.neighbors(q, d, False) == ham


def test_fm_index_matches_bruteforce(small_genome):
    g = small_genome
    ix = O.Index(g["fm9"])
    text = g["text"]
    assert ix.size == len(text) + 1
    rng = random.Random(9)
    for _ in range(400):
        if rng.random() < 0.7:
            p = rng.randrange(len(text) - 12)
            pat = text[p:p + rng.randint(1, 12)]
        else:
            pat = "".join(rng.choice("ACGTNR") for _ in range(rng.randint(1, 7))).encode()
        want = O.bf_locate(text, pat)
        assert ix.count(pat) == len(want)
        assert sorted(ix.locate(pat)) == want
    full = text + b"\0"
    for _ in range(100):
        b = rng.randrange(len(text))
        e = min(len(text), b + rng.randint(0, 40))
        assert ix.extract(b, e) == full[b:e + 1]


def test_fm9_layout_is_fully_accounted(small_genome, tmp_path):
    # serialise -> parse -> serialise must be the identity on bytes (reader and writer agree on every field)
    raw = open(small_genome["fm9"], "rb").read()
    assert int.from_bytes(raw[8:16], "little") == len(small_genome["text"]) + 1  # wt size right after the hash
    ix = O.Index(small_genome["fm9"])
    for ch in "ACGT":
        assert 1 <= ix.code_len(ch) <= 4


def test_oracle_reproduces_golden_vectors():
    gold = json.load(open(os.path.join(GOLD, "oracle_vectors.json")))
    for case in gold["neighbors"]:
        assert O.neighbors(case["query"], case["distance"], case["indel"]) == case["set"]
    for case in gold["needle"]:
        assert list(O.needle(case["a1"], case["a2"])) == case["out"]
    g = gold["hunt"]
    seqs = make_genome(*g["genome_args"])
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "g.fm9")
        O.build_fm9(genome_text(seqs), path)
        ix = O.Index(path)
        for case in g["cases"]:
            js, _ = ix.hunt([len(s) + 1 for s in seqs], g["names"], case["queries"], qnames=case["qnames"],
                            genome="genome.fa.gz", **case["params"])
            assert js == case["json"]


NASTY = ['chr1', 'chr "quoted"', 'back\\slash', 'tab\there', 'nl\nin', 'ctl\x01\x1f', 'del\x7f', 'ünï©ødé', '漢字', '😀 emoji', '',
         'a/b', "single'quote", 'cr\rlf', '\x08\x0c', 'percent% &amp; <tag>']


def test_own_json_writer_equals_reference_nlohmann(small_genome):
    """Every JSON object `dicey hunt` writes goes through nlohmann::json::dump() in the reference (hunter.h:112-116,
    122-152).  oracle/_ref/libjsonref.so is that library, compiled in place; the oracle's writers use it when it is loaded.
    Here the restated writer (escaping, key order, integers) is held against it on names and sequences chosen to hurt."""
    import pytest
    if not os.path.exists(O.REF_JSON):
        pytest.skip("oracle/_ref/libjsonref.so not built (no /root/reference on this box)")
    g = small_genome
    orc = O.Index(g["fm9"])
    rng = random.Random(4)
    seqs = g["seqs"]
    qs, qn = [], []
    for i, nm in enumerate(NASTY * 2):
        s = seqs[i % 3]
        p = rng.randrange(len(s) - 30)
        qs.append(s[p:p + rng.choice([12, 20, 25])])
        qn.append(nm)
    qs += ["ACGTAC", "acgtnnacgtacgtacgtac"]
    qn += ["short \"one\"", "lower\tcase"]
    names = [NASTY[1], NASTY[7], NASTY[3]]
    out = {}
    try:
        for mode in (True, False):
            assert O.use_ref_json(mode) == mode
            out[mode] = [orc.hunt(g["seqlen"], names, qs, qnames=qn, genome='/data/"genomes"/g\tx.fa.gz', outfile="o\\ut.json.gz", **kw)[0]
                         for kw in (dict(distance=1), dict(distance=0, hamming=True, forward_only=True, max_locations=1),
                                    dict(distance=2, max_neighborhood=50))]
    finally:
        O.use_ref_json(True)
    assert out[True] == out[False]
    assert O.ref_json_in_use()
    assert '"chr":"chr \\"quoted\\""' in out[True][0] and "\\u0001\\u001f" in out[True][0] and "漢字" in out[True][0]


def test_fast_neighbors_equal_the_literal_restatement():
    """bench.py's distance-2 parity sample enumerates neighbourhoods with the hash-set form; it must be the same set"""
    rng = random.Random(6)
    cases = [("".join(rng.choice("ACGT") for _ in range(m)), d, indel, cap)
             for m in (10, 13, 20) for d in (0, 1, 2) for indel in (True, False) for cap in (10000, 60)]
    cases += [("A" * 14, 2, True, 10000), ("ACGTNNACGTACGT", 2, True, 10000), ("ACACACACACACAC", 2, True, 10000)]
    for q, d, indel, cap in cases:
        assert O.neighbors_fast(q, d, indel, cap) == list(O.neighbors(q, d, indel, cap)), (q, d, indel, cap)
